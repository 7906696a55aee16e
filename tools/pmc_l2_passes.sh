#!/bin/bash
# Dev: L2 / fabric counter passes over tools/gemm_pmc.py (variants given as arguments), one rocprofv3 run per 4-counter group.
# usage: ARX_LIB=... tools/pmc_l2_passes.sh <outdir> <variants...>
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/$out/counters_avail.txt 2>&1 || true
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum" \
           "TCC_READ_sum TCC_WRITE_sum TCC_NORMAL_EVICT_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$out/p$i -- python3 $R/tools/gemm_pmc.py "$@" > $R/$out/p$i.log 2>&1 || echo "pass $i failed" >> $R/$out/failed.txt
done

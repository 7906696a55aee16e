#!/bin/bash
# The round's measurement set on the GPU box (usage: tools/final_profile.sh <tag>; writes gpurun_out/<tag>_*):
#   1. FETCH_SIZE / WRITE_SIZE, one counter per pass (HBM-side traffic per kernel) -> profiles/traffic.json for THIS code, then the bench line (which quotes it);
#   2. rocprofv3 --kernel-trace --stats of the ENCODE leg alone and of the 10 M x 768 SEARCH leg alone (one workload per table: a kernel's
#      average duration is then that workload's, and bytes / average reproduces the bench line's roofline figures);
#   3. one SQ pass: matrix-pipe busy cycles and wave-state cycles per kernel (MFMA utilisation).
# PMC passes never carry a trace domain other than the kernel trace; the program is python3 itself (no shell / env hop behind the profiler).
tag=${1:-final}
R=$GRAFT_REPO_ROOT; cd $R
cd /tmp && export TMPDIR=/tmp
OFF="--no-cpu-baseline --no-query-leg --sustained-chunks 0 --search-total-rows 0 --d1024-rows 0 --clustered-rows 0 --bge-steps 0 --e2e-rows 0"
ENC="--steps 5 --warmup 2 $OFF --search-rows 0"
SRCH="--steps 1 --warmup 1 $OFF --search-queries 2048 --search-qbs 64,256"
BOTH="--steps 5 --warmup 2 $OFF --search-queries 2048 --search-qbs 64,256"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc_$c -- python3 $R/bench.py $BOTH > $R/gpurun_out/${tag}_pmc_$c.log 2>&1 || { echo "pmc $c failed"; exit 1; }
done
# the traffic figures the bench line quotes are gated on the hash of csrc/*: derive them from THESE passes first, then take the line
(cd $R && python tools/prof_summary.py pmc gpurun_out/${tag}_pmc.json gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE \
   && python tools/prof_summary.py traffic gpurun_out/${tag}_pmc.json profiles/traffic.json > /dev/null && cp profiles/traffic.json gpurun_out/${tag}_traffic.json) || { echo "traffic.json failed"; exit 1; }
(cd $R && timeout -k 10 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err) || { echo "bench failed"; tail -5 $R/gpurun_out/${tag}_bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_enc -- python3 $R/bench.py $ENC > $R/gpurun_out/${tag}_prof_enc.log 2>&1 || { echo "stats pass (encode) failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_srch -- python3 $R/bench.py $SRCH > $R/gpurun_out/${tag}_prof_srch.log 2>&1 || { echo "stats pass (search) failed"; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_pmc_SQ -- python3 $R/bench.py $BOTH > $R/gpurun_out/${tag}_pmc_SQ.log 2>&1 || { echo "pmc SQ failed"; tail -5 $R/gpurun_out/${tag}_pmc_SQ.log; exit 1; }
cd $R
python tools/prof_summary.py stats gpurun_out/${tag}_prof_enc gpurun_out/${tag}_kernel_stats_encode.md "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $ENC"
python tools/prof_summary.py stats gpurun_out/${tag}_prof_srch gpurun_out/${tag}_kernel_stats_search_10M.md "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $SRCH"
python tools/prof_summary.py pmc gpurun_out/${tag}_pmc_sq.json gpurun_out/${tag}_pmc_SQ
python tools/prof_summary.py mfma gpurun_out/${tag}_pmc_sq.json gpurun_out/${tag}_kernel_stats_encode.md gpurun_out/${tag}_kernel_stats_search_10M.md gpurun_out/${tag}_mfma_utilisation.md
# keep only the summaries (the raw CSVs are tens of MB)
rm -rf gpurun_out/${tag}_prof_enc gpurun_out/${tag}_prof_srch gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_pmc_SQ

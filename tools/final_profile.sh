#!/bin/bash
# The round's measurement set on the GPU box: bench line, rocprofv3 kernel stats of the same program, FETCH_SIZE / WRITE_SIZE passes.
# usage: tools/final_profile.sh <tag>        (writes gpurun_out/<tag>_*)
tag=${1:-final}
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 700 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-query-leg --sustained-chunks 0 --search-queries 2048"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py $ARGS > $R/gpurun_out/${tag}_prof.log 2>&1 || { echo "stats pass failed"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc_$c -- python3 $R/bench.py $ARGS --search-total-rows 0 --d1024-rows 0 > $R/gpurun_out/${tag}_pmc_$c.log 2>&1 || { echo "pmc $c failed"; exit 1; }
done
cd $R
python tools/prof_summary.py stats gpurun_out/${tag}_prof gpurun_out/${tag}_kernel_stats.md "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $ARGS"
python tools/prof_summary.py pmc gpurun_out/${tag}_pmc.json gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE
# keep only the summaries (the raw CSVs are tens of MB)
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE

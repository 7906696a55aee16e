#!/bin/bash
# Other shapes and launch modes of the same build (parity cases, not the bench line): bge-large, MiniLM, S = 128, 1-rank torchrun.
# usage: tools/other_shapes.sh <outdir>
out=${1:-gpurun_out/other}; mkdir -p $out
A="--no-cpu-baseline --sustained-chunks 0 --search-rows 0 --search-total-rows 0 --d1024-rows 0 --no-query-leg"
timeout -k 10 200 python bench.py $A --model BAAI/bge-large-en-v1.5 --batch 512 2>/dev/null | tail -1 > $out/bench_bge_large.json
timeout -k 10 200 python bench.py $A --model all-MiniLM-L6-v2 2>/dev/null | tail -1 > $out/bench_minilm.json
timeout -k 10 200 python bench.py $A --batch 2048 --seq-len 128 2>/dev/null | tail -1 > $out/bench_mpnet_s128.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --sustained-chunks 700000 --search-queries 2048 --d1024-rows 0 2>/dev/null | tail -1 > $out/bench_torchrun_1rank.json
for f in $out/*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done

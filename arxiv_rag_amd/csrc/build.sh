#!/bin/bash
# Build libarx_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=${ARX_OUT:-../libarx_hip.so}          # ARX_OUT=../libarx_dev.so ARX_HIPCC_EXTRA="-DARX_DEV_VARIANTS": a dev build beside the shipped one
BLD=../_build${ARX_OUT:+_$(basename "$ARX_OUT" .so)}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast"
mkdir -p $BLD
pids=()
# per-kernel register / scratch / spill figures of THIS build go to _build/<file>.resources.txt (tests/test_build_resources.py reads them:
# a hot kernel that spills writes its registers to memory once per wave — 0.4 GB per attention launch when it happened, round 2)
for f in runtime encoder search; do
  ( hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $f.hip -o $BLD/$f.o ${ARX_HIPCC_EXTRA} 2> $BLD/$f.resources.txt \
      || { grep -v "kernel-resource-usage" $BLD/$f.resources.txt >&2; exit 1; } ) &
  pids+=($!)
done
# host-only part of the C ABI (WordPiece feeder): plain C++, no device code
g++ -O3 -std=c++17 -fPIC -pthread -c wordpiece.cpp -o $BLD/wordpiece.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p || exit 1; done
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $OUT $BLD/runtime.o $BLD/encoder.o $BLD/search.o $BLD/wordpiece.o
echo "built $(realpath $OUT)"

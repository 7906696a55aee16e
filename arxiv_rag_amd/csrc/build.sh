#!/bin/bash
# Build libarx_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libarx_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast"
mkdir -p ../_build
pids=()
# per-kernel register / scratch / spill figures of THIS build go to _build/<file>.resources.txt (tests/test_build_resources.py reads them:
# a hot kernel that spills writes its registers to memory once per wave — 0.4 GB per attention launch when it happened, round 2)
for f in runtime encoder search; do
  ( hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $f.hip -o ../_build/$f.o ${ARX_HIPCC_EXTRA} 2> ../_build/$f.resources.txt \
      || { grep -v "kernel-resource-usage" ../_build/$f.resources.txt >&2; exit 1; } ) &
  pids+=($!)
done
# host-only part of the C ABI (WordPiece feeder): plain C++, no device code
g++ -O3 -std=c++17 -fPIC -pthread -c wordpiece.cpp -o ../_build/wordpiece.o &
pids+=($!)
for p in "${pids[@]}"; do wait $p || exit 1; done
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $OUT ../_build/runtime.o ../_build/encoder.o ../_build/search.o ../_build/wordpiece.o
echo "built $(realpath $OUT)"

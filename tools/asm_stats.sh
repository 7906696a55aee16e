#!/bin/bash
# Dev helper: device asm of encoder.hip -> /tmp/enc.s, VGPR/scratch per kernel matching $1 (default: gemm_8phase)
cd /root/repo/arxiv_rag_amd/csrc || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -S --cuda-device-only -o /tmp/enc.s encoder.hip ${ARX_HIPCC_EXTRA} 2>&1 | grep -E " error|error:"
pat=${1:-gemm_8phase}
awk -v pat="$pat" '/^_Z[A-Za-z0-9_]*:/ {name=$1} /; NumVgprs:/ {v=$3} /; ScratchSize:/ {s=$3} /; Occupancy:/ { if (name ~ pat) print name, "vgpr", v, "scratch", s, "occ", $3 }' /tmp/enc.s

#!/usr/bin/env python3
"""Dev (dev build, ARX_DEV_BW): the tile walk's band width swept in situ — encode step and per-GEMM times per band width, arms interleaved."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import ab_encode
lib = str((ROOT / "arxiv_rag_amd" / "libarx_dev.so").resolve())
for r in range(2):
    for bw in (0, 2, 3, 4, 6, 12):
        env = dict(os.environ, ARX_LIB=lib)
        if bw: env["ARX_DEV_BW"] = str(bw)
        o = subprocess.run([sys.executable, "-c", ab_encode.CHILD], env=env, capture_output=True, text=True)
        print(f"bw={bw or 'default'}", o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:], flush=True)

"""ORACLE — test infrastructure, not product code: one worker of the reference-style CPU fan-out baseline.

The reference embeds on host cores with `mp.Pool(int(cpu_count*0.75), initializer=init_worker_model)` (GEN:190, GEN:205):
one full model replica per worker process, each running `model.encode` on its share.  bench.py's `cpu_baseline` leg C
starts that many of THIS program as child processes (never a fork of the GPU process), each pinned to ONE torch thread and
running `oracle/tf_reference.py` — the `transformers` modules sentence-transformers chains, fp32 eager + pool + L2.

Protocol (stdin/stdout, line based): load weights from the .npz the parent wrote -> build the model -> one warm-up batch ->
print "READY <g>" (g = 1 if this process ended up with a GPU device node open, which a CPU worker must not: a GPU box admits
only a handful of processes on its card) -> wait for a line on stdin -> encode batches until `budget_s` of wall time has
passed -> print "DONE <chunks> <seconds>".
"""
from __future__ import annotations

import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
for _k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
    os.environ[_k] = ""                       # a CPU worker: no device is visible ...


def _gpu_nodes_open() -> int:
    n = 0
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                t = os.readlink(f"/proc/self/fd/{fd}")
            except OSError:
                continue
            if t.startswith("/dev/kfd") or t.startswith("/dev/dri/"):
                n += 1
    except OSError:
        pass
    return n


def main():
    root, model_name, wpath, S, B, budget_s = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
    sys.path.insert(0, root)
    import numpy as np
    import torch
    torch.cuda.is_available = lambda: False   # ... and nothing may initialise the HIP runtime while probing for one
    torch.cuda.device_count = lambda: 0
    torch.set_num_threads(1)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass
    from arxiv_rag_amd import config as C
    from oracle import tf_reference as TF
    cfg = C.PRESETS[model_name]
    with np.load(wpath, mmap_mode="r") as z:
        sd = {k: z[k] for k in z.files}
    m = TF.build_model(cfg, sd)
    del sd
    rs = np.random.RandomState(1000 + os.getpid() % 1000)
    ids = rs.randint(4, cfg.vocab_size - 1, size=(B, S)).astype(np.int64)
    ids[:, 0] = 0; ids[:, -1] = 2
    lens = np.full(B, S, np.int64)
    TF.encode_tokens(m, cfg, ids[:1], lens[:1])
    print(f"READY {1 if _gpu_nodes_open() else 0}", flush=True)
    if not sys.stdin.readline():
        return
    n, t0 = 0, time.perf_counter()
    while True:
        TF.encode_tokens(m, cfg, ids, lens)
        n += B
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    print(f"DONE {n} {dt:.4f}", flush=True)


if __name__ == "__main__":
    main()

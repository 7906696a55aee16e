#!/usr/bin/env python3
"""bench.py --gpus N --steps K --warmup W  ->  ONE JSON line on rank 0.

Launch: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` each process is one rank; called plainly as
`python bench.py --gpus N` with N > 1 it starts those N ranks itself as child processes (before touching the GPU) and passes their
output through.

Step = one pass of the encode hot path over one batch: 1024 synthetic chunks x exactly 256 token ids
(BASELINE.json configs[1]: all-mpnet-base-v2 shape, bf16 weights/activations, fp32 accumulate), ids already
resident in HBM, output = unit-norm fp16 rows written into the rank's HBM corpus shard.
value = whole-job chunks/s (all ranks; weak scaling: every rank encodes its own batches, no collective on
the encode path).  Extra objects on the same line:
  roofline      dominant kernel (FFN-1 GEMM), algorithmic FLOPs per launch / mean launch time (hipEvents
                recorded by the library on the launch stream during the timed region), vs 2.5 PFLOP/s bf16
  encode        whole-forward MFMA fraction + per-kernel-class time split
  search        QPS@top-10 over a 10 M x 768 fp16 shard per rank (configs[2]) at several query-batch sizes,
                pass-A HBM roofline fraction; with N > 1 the per-shard partials are all-gathered over RCCL;
                search.strong_scaling = configs[3] (ONE 5 M-row corpus cut N ways, merged answer checked against a single index),
                search.shard_625k = its per-rank slice at N = 8, search.d1024 = configs[4]'s per-rank slice (6.25 M x 1024)
  cpu_baseline  the numpy oracle (oracle/encoder_oracle.py) timed on this box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_PEAK_BF16 = 2.5e15       # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def csrc_sha16() -> str:
    """Hash of the kernel sources the library is built from: PMC traffic figures are only quoted for the code they were
    measured on (profiles/traffic.json carries the hash; `tools/prof_summary.py traffic` writes it)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "arxiv_rag_amd" / "csrc").glob("*")):
        if f.suffix in (".h", ".hip", ".cpp"):
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def load_traffic():
    tj = ROOT / "profiles" / "traffic.json"
    try:
        t = json.loads(tj.read_text())
    except Exception:
        return {}
    return t if t.get("csrc_sha16") == csrc_sha16() else {}


def flops_per_chunk(cfg, S):
    H, F, L = cfg.hidden, cfg.ffn, cfg.layers
    return L * (8 * S * H * H + 4 * S * H * F + 4 * S * S * H)


from arxiv_rag_amd.generate_embeddings_parallel import effective_cpus      # noqa: E402  (affinity mask capped by the cgroup quota; numpy only)


def _cpu_fanout(model_name, cfg, sd, S, workers, budget_s, B=4, timeout_s=240.0):
    """Leg C: the reference's own CPU strategy (GEN:190 `num_workers = int(cpu_count*0.75)`, GEN:205 one model replica per pool
    process): `workers` child processes x 1 torch thread, each running oracle/cpu_fanout_worker.py; aggregate chunks / the
    slowest worker's wall time."""
    import select
    import subprocess
    import tempfile
    tmpd = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    fd, wpath = tempfile.mkstemp(suffix=".npz", prefix="arx_cpu_weights_", dir=tmpd)
    os.close(fd)
    procs = []
    try:
        np.savez(wpath, **sd)
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        shim = ROOT / "oracle" / "_build" / "libnogpu.so"       # CPU workers must not hold the GPU device node (oracle/nogpu_shim.c)
        if shim.exists():
            env["LD_PRELOAD"] = (str(shim) + " " + env.get("LD_PRELOAD", "")).strip()
        cmd = [sys.executable, str(ROOT / "oracle" / "cpu_fanout_worker.py"), str(ROOT), model_name, wpath, str(S), str(B), str(budget_s)]
        def spawn():
            procs.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env))
        deadline = time.time() + timeout_s

        def read_line(p):
            left = deadline - time.time()
            if left <= 0 or not select.select([p.stdout], [], [], left)[0]:
                raise TimeoutError("cpu fan-out worker did not answer in time")
            return p.stdout.readline()
        t_start = time.time()
        note = ""
        spawn()                                              # probe: ONE worker first — does a CPU worker end up holding the GPU node?
        ln = read_line(procs[0])
        if not ln.startswith("READY"):
            raise RuntimeError(f"worker said {ln!r}")
        if ln.split()[1:] == ["1"] and workers > 4:          # a GPU box admits 6 processes on its card: never go near that
            note = f" (pool capped from {workers}: CPU workers hold the GPU device node open on this box)"
            workers = 4
        for _ in range(workers - 1):
            spawn()
        for p in procs[1:]:
            ln = read_line(p)
            if not ln.startswith("READY"):
                raise RuntimeError(f"worker said {ln!r}")
        startup = time.time() - t_start
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        n_tot, t_max = 0, 0.0
        for p in procs:
            tok = read_line(p).split()
            if len(tok) != 3 or tok[0] != "DONE":
                raise RuntimeError(f"worker said {tok!r}")
            n_tot += int(tok[1]); t_max = max(t_max, float(tok[2]))
        return {"value": round(n_tot / t_max, 2), "unit": "chunks/s", "processes": workers, "threads_per_process": 1,
                "what": f"GEN:190/205 fan-out: {workers} processes x 1 thread, each a full {cfg.layers}L/{cfg.hidden} fp32 replica "
                        f"(transformers eager + pool + L2, oracle/tf_reference.py), {n_tot} chunks x {S} tokens, batch {B}, "
                        f"{t_max:.1f} s timed after {startup:.0f} s of model loading" + note}
    except Exception as e:                                   # noqa: BLE001
        return {"error": repr(e)[:200], "processes": workers}
    finally:
        for p in procs:
            try:
                p.kill()                                     # exact children, by handle
            except Exception:
                pass
        try:
            os.unlink(wpath)
        except OSError:
            pass


def cpu_baseline(model_name, cfg, sd, S, budget_s=30.0):
    """The reference's CPU encode on this box's host cores, bounded to ~budget_s of CPU work in all.  The reference itself cannot
    run (SyntaxError at GEN:239, sentence-transformers absent), so it is stood in for by the modules it chains:
      A  transformers fp32 eager + pool + L2 (oracle/tf_reference.py), ONE process, torch threads = usable cores;
      C  the reference's own strategy (GEN:190, GEN:205): int(0.75 x cores) processes x 1 thread, one replica each;
      value = the better of A and C (`cores` = the threads that leg used).  The numpy restatement (oracle/encoder_oracle.py) is
      timed beside them as `numpy_oracle`."""
    from oracle import encoder_oracle as EO
    eff = effective_cpus()
    rs = np.random.RandomState(99)
    out = {"value": None, "unit": "chunks/s", "cores": None, "kind": "port", "sample": None,
           "host": {"os_cpu_count": os.cpu_count(), "usable_cpus": eff}}
    # numpy restatement
    B = 8
    ids = rs.randint(4, cfg.vocab_size - 1, size=(B, S)).astype(np.int64)
    ids[:, 0] = 0; ids[:, -1] = 2
    lens = np.full(B, S, np.int64)
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=eff)
    except Exception:
        lim = None
    EO.encode_tokens(sd, cfg, ids[:2], lens[:2])
    n, t = 0, 0.0
    while t < budget_s * 0.2 or n == 0:
        t0 = time.time(); EO.encode_tokens(sd, cfg, ids, lens); t += time.time() - t0; n += B
    if lim is not None:
        lim.restore_original_limits()
    out["numpy_oracle"] = {"value": round(n / t, 3), "unit": "chunks/s", "threads": eff,
                           "what": f"{n} chunks x {S} tokens, {cfg.layers}L/{cfg.hidden} fp32 numpy oracle (oracle/encoder_oracle.py), batch {B}"}
    legs = {}
    try:
        import torch
        from oracle import tf_reference as TF
        torch.set_num_threads(eff)
        m = TF.build_model(cfg, sd)
        B2 = 32
        ids2 = rs.randint(4, cfg.vocab_size - 1, size=(B2, S)).astype(np.int64); ids2[:, 0] = 0; ids2[:, -1] = 2
        lens2 = np.full(B2, S, np.int64)
        TF.encode_tokens(m, cfg, ids2[:4], lens2[:4])
        n2, t2 = 0, 0.0
        while t2 < budget_s * 0.3 or n2 == 0:
            t0 = time.time(); TF.encode_tokens(m, cfg, ids2, lens2); t2 += time.time() - t0; n2 += B2
        del m
        legs["A"] = {"value": round(n2 / t2, 2), "unit": "chunks/s", "processes": 1, "threads_per_process": eff,
                     "what": f"transformers {cfg.layers}L/{cfg.hidden} fp32 eager + pool + L2 (oracle/tf_reference.py), one process, "
                             f"{n2} chunks x {S} tokens, batch {B2}"}
    except Exception as e:                                   # noqa: BLE001
        legs["A"] = {"error": repr(e)[:200]}
    workers = max(1, int(eff * 0.75))                        # GEN:190
    wbytes = sum(v.nbytes for v in sd.values())
    try:                                                     # one replica per process (GEN:205): bound the pool by host memory
        avail = int(next(ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")).split()[1]) * 1024
    except Exception:
        avail = 64 << 30
    by_mem = max(1, int(min(avail, 200 << 30) * 0.6 / (2.2 * wbytes + (500 << 20))))
    capped = workers > by_mem
    workers = min(workers, by_mem)
    legs["C"] = _cpu_fanout(model_name, cfg, sd, S, workers, budget_s * 0.4)
    if capped and "what" in legs["C"]:
        legs["C"]["what"] += f" (pool capped from {int(eff * 0.75)} by host memory)"
    out["legs"] = legs
    best = max((k for k in legs if "value" in legs[k]), key=lambda k: legs[k]["value"], default=None)
    if best is not None:
        L = legs[best]
        out["value"] = L["value"]
        out["cores"] = L["processes"] * L["threads_per_process"]
        out["sample"] = f"leg {best} (best of A/C): " + L["what"] + f"; host: {os.cpu_count()} logical cpus, {eff} usable by this job"
    else:                                                    # both stand-ins failed: fall back to the numpy restatement's number
        out["value"], out["cores"], out["sample"] = out["numpy_oracle"]["value"], eff, out["numpy_oracle"]["what"]
    # the search step on the same cores (BASELINE.md CPU-S): the definitional oracle (fp32 matmul on the fp16-rounded rows + exact
    # top-10), 256 queries over a 100 k-row sample of the 10 M-row workload; the scan is linear in the rows, so the 10 M figure is
    # the measured one / 100 (stated as scaled)
    try:
        from oracle import search_oracle as SO
        try:
            from threadpoolctl import threadpool_limits
            lim2 = threadpool_limits(limits=eff)
        except Exception:
            lim2 = None
        n_s, nq_s, d_s = 100_000, 256, cfg.hidden
        Cs = SO.unit_rows_f16(n_s, d_s, 7); Qs = SO.unit_rows_f16(nq_s, d_s, 11)
        SO.topk_search(Cs[:20000], Qs, 10)
        t0 = time.time(); SO.topk_search(Cs, Qs, 10); ts = time.time() - t0
        if lim2 is not None:
            lim2.restore_original_limits()
        out["search"] = {"qps_at_sample": round(nq_s / ts, 1), "qps_scaled_to_10M_rows": round(nq_s / ts * n_s / 10_000_000, 2),
                         "unit": "queries/s", "threads": eff,
                         "what": f"oracle/search_oracle.py (numpy fp32 Q @ C.T + exact top-10), {nq_s} queries x {n_s} x {d_s} fp16 rows, "
                                 f"{ts:.1f} s; scaled linearly in the row count"}
    except Exception as e:                                   # noqa: BLE001
        out["search"] = {"error": repr(e)[:200]}
    return out


class ClockSampler:
    """Shader clock (and package power) of THIS rank's GPU while a leg runs: a thread reads the card's hwmon files
    (`/sys/class/drm/card*/device/hwmon/*/freq1_input` in Hz, `power1_input` in uW; the card is matched by PCI address) every
    `period` seconds; `rocm-smi` text as the fallback.  The 2.5-PFLOP/s figure of `roofline.peak` assumes 2.4 GHz: on real operands
    these kernels sit on the 1.4-kW package cap and the chip runs 19-24 % below that, so the bench line also
    carries the clock it measured and the peak that clock allows."""

    def __init__(self, props, period=0.25):
        import threading
        self.period, self.samples, self.power, self.source = period, [], [], None
        self._stop = threading.Event()
        self._files = self._find(props)
        self._th = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _find(props):
        import glob
        try:
            want = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        except Exception:
            want = None
        cands = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            devdir = os.path.realpath(os.path.join(os.path.dirname(f), "..", ".."))
            cands.append((os.path.basename(devdir), f, os.path.join(os.path.dirname(f), "power1_input")))
        hit = [c for c in cands if c[0] == want]
        if not hit and len(cands) == 1:
            hit = cands
        return hit[0][1:] if hit else None

    def _read(self):
        if self._files is not None:
            try:
                mhz = int(open(self._files[0]).read()) / 1e6
                try:
                    w = int(open(self._files[1]).read()) / 1e6
                except Exception:
                    w = None
                self.source = "hwmon"
                return mhz, w
            except Exception:
                self._files = None
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out); pw = re.findall(r"Power \(W\): ([\d.]+)", out)
            self.source = "rocm-smi"
            return (float(sclk[0]) if sclk else None), (float(pw[0]) if pw else None)
        except Exception:
            return None, None

    def _run(self):
        while not self._stop.is_set():
            mhz, w = self._read()
            if mhz:
                self.samples.append(mhz)
            if w:
                self.power.append(w)
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set(); self._th.join(timeout=10)

    def summary(self, skip=2):
        xs = sorted(self.samples[skip:] or self.samples)
        ws = sorted(self.power[skip:] or self.power)
        if not xs:
            return None
        med = xs[len(xs) // 2]
        return {"clock_mhz_under_load": round(med, 1), "clock_mhz_min_max": [round(xs[0], 1), round(xs[-1], 1)], "samples": len(xs),
                "package_power_w": round(ws[len(ws) // 2], 1) if ws else None, "source": self.source}


def relaunch_command(n_gpus: int, argv, port: int):
    """The command `python bench.py --gpus N` (no RANK in the environment) re-issues: N ranks of THIS file under torch.distributed.run
    on this node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def needs_self_launch(n_gpus: int, environ) -> bool:
    """True when this process is NOT already a rank of a torch.distributed.run launch and more than one rank is asked for
    (ARX_BENCH_FORCE_LAUNCH=1 forces the child-launch path for N = 1 too: the one-GPU rehearsal of the N-rank start-up)."""
    if "RANK" in environ:
        return False
    return n_gpus > 1 or environ.get("ARX_BENCH_FORCE_LAUNCH") == "1"


def self_launch(n_gpus: int, argv) -> int:
    """Start the N ranks as CHILD processes — before this process has touched torch.cuda or HIP (a process that initialised the GPU must
    never exec or be replaced) — pass their stdout through (rank 0 prints the one JSON line) and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cpus() // max(1, n_gpus))))
    return subprocess.run(relaunch_command(n_gpus, argv, port), env=env).returncode


def main():
    wall, _last = {}, [time.perf_counter()]

    def mark(leg):                                           # where the run's wall time went, leg by leg (printed as `wall_s`)
        now = time.perf_counter(); wall[leg] = round(wall.get(leg, 0.0) + now - _last[0], 1); _last[0] = now
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--seq-len", type=int, default=256)
    ap.add_argument("--model", default="all-mpnet-base-v2")
    ap.add_argument("--search-rows", type=int, default=10_000_000, help="configs[2]: corpus rows PER RANK, weak scaling (0 = skip)")
    ap.add_argument("--search-total-rows", type=int, default=5_000_000,
                    help="configs[3]: rows of ONE corpus cut across the ranks with shard_bounds, strong scaling (0 = skip)")
    ap.add_argument("--d1024-rows", type=int, default=6_250_000,
                    help="configs[4]'s per-rank slice: rows of a dim-1024 shard searched on one GPU (N = 1 only; 0 = skip)")
    ap.add_argument("--search-queries", type=int, default=10_000)
    ap.add_argument("--search-qbs", type=str, default="1,64,256,all",
                    help="query-batch sizes of the configs[2] leg ('all' = --search-queries); a profiling pass narrows it so that a kernel's "
                         "average duration in the rocprofv3 table is ONE workload's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-query-leg", action="store_true",
                    help="skip search.encode_plus_search (its query-sized forwards launch the GEMM kernels on tiny problems: under rocprofv3 "
                         "they would pollute the per-kernel averages the roofline figures are checked against)")
    ap.add_argument("--sustained-chunks", type=int, default=1_000_000,
                    help="untimed-by-driver leg: encode this many chunks back to back (configs[1] = 1 M; 0 = skip)")
    ap.add_argument("--prof-all-in-timed-region", action="store_true",
                    help="A/B: record HIP events around every kernel inside the timed region (the pre-change behaviour)")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--bge-steps", type=int, default=5,
                    help="configs[4]'s encode half in the precision that is feasible: this many steps of 256 x 256 tokens on the bge-large "
                         "shape (24 L / 1024, CLS pool, bf16) after 2 warm-up steps (N = 1 only; 0 = skip)")
    ap.add_argument("--clustered-rows", type=int, default=10_000_000,
                    help="search.clustered: rows of an embedding-LIKE corpus (clusters + hot dimensions) searched on one GPU (N = 1 only; 0 = skip)")
    ap.add_argument("--e2e-rows", type=int, default=625_000,
                    help="e2e_rank_slice: rows of the sustained leg's OUTPUT (rows the encoder wrote) searched where they lie (0 = skip)")
    args = ap.parse_args()
    if needs_self_launch(args.gpus, os.environ):              # `python bench.py --gpus N` as the driver calls it: start the ranks ourselves
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from arxiv_rag_amd import _lib, config as C
    from arxiv_rag_amd.encoder import HipEncoder
    from arxiv_rag_amd.index import ShardIndex, fill_unit_rows, gather_partials, merge_partials, shard_bounds
    from arxiv_rag_amd.weights import seeded_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run (RANK set) the process group is created whatever the world size, so that a 1-rank launch
    # exercises exactly the code an N-rank launch runs: RCCL init, barrier, MAX all-reduce of the time, all-gather of partial top-k
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL printf()s a five-line version banner to STDOUT when the communicator is created (NCCL_DEBUG_FILE does not move it); stdout
        # carries the one JSON line of the contract, so file descriptor 1 points at stderr while the group comes up
        sys.stdout.flush()
        fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
            torch.cuda.set_device(local_rank)
            dist.barrier()                               # the communicator exists (and has said so) before stdout comes back
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(fd1, 1)
            os.close(fd1)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    # First-N-rank-run self-checks (no multi-GPU node has run this yet: fail loudly, not quietly).  Every rank: the group is the size the
    # command line says, this node shows a GPU for every local rank, and the ranks RCCL really connected are 0..world-1, each on its own GPU.
    launch = {"world_size": world, "ranks_seen": [0], "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1")),
              "gpus_visible": torch.cuda.device_count(), "backend": None}
    assert torch.cuda.device_count() >= launch["local_world_size"], \
        f"LOCAL_WORLD_SIZE={launch['local_world_size']} but only {torch.cuda.device_count()} GPU(s) visible on this node"
    if use_dist:
        assert dist.get_world_size() == args.gpus and dist.get_rank() == rank, (dist.get_world_size(), args.gpus, dist.get_rank(), rank)
        launch["backend"] = dist.get_backend()
        pr = torch.cuda.get_device_properties(dev)
        me = torch.tensor([rank, local_rank, getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", -1)], dtype=torch.int64, device=dev)
        seen = torch.empty((world, 4), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(seen, me)
        seen = seen.cpu().tolist()
        launch["ranks_seen"] = [r[0] for r in seen]
        assert launch["ranks_seen"] == list(range(world)), f"RCCL connected ranks {launch['ranks_seen']}, expected 0..{world - 1}"
        if launch["local_world_size"] == world and all(r[2] >= 0 for r in seen):
            assert len({(r[2], r[3]) for r in seen}) == world, f"two ranks share a GPU: {seen}"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    cfg = C.PRESETS[args.model]
    B, S, K, W = args.batch, args.seq_len, args.steps, args.warmup
    sd = seeded_state_dict(cfg, seed=0)                      # N(0, 0.02^2), LN gamma=1 beta=0 (SURVEY §8d cfg 2)
    enc = HipEncoder(cfg, sd, device=dev, max_tokens=B * S, max_seqs=B)

    # synthetic chunks: ids[0]=<s>=0, ids[S-1]=</s>=2, body ~ U{4..vocab-2}; every step its own batch
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    ids = torch.randint(4, cfg.vocab_size - 1, (K + W, B, S), generator=g, device=dev, dtype=torch.int32)
    ids[:, :, 0] = 0; ids[:, :, S - 1] = 2
    lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    shard = torch.empty((K * B, cfg.hidden), dtype=torch.float16, device=dev)   # this rank's corpus rows

    def step(i, slot):
        enc.forward_tokens(ids[i], lens, S, B * S, out=None, out_f16=shard[slot * B:(slot + 1) * B], normalize=True)

    for i in range(W):
        step(i, 0)
    # timed region: only the kernel the roofline reports (FFN-1) records HIP events — an event pair around every one of the ~110
    # launches of a forward puts ~170 extra packets between the kernels being timed; the per-kernel table below comes from a
    # separate, untimed pass over the same K batches
    _lib.prof_reset(); _lib.prof_classes(None if args.prof_all_in_timed_region else ["gemm_fc1"]); _lib.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        step(W + i, i)
    barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    prof_timed = _lib.prof_read()
    _lib.prof_reset(); _lib.prof_classes(None); _lib.prof_enable(True)
    for i in range(K):
        step(W + i, i)
    barrier()
    _lib.prof_enable(False)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    chunks_per_s = world * K * B / dt
    prof = _lib.prof_read()
    norms = shard[:B].float().norm(dim=1)
    assert torch.isfinite(norms).all() and (norms - 1).abs().max() < 2e-3, "encoder output is not unit-norm"

    T = B * S
    gemm_flops = {"gemm_qkv": 2 * T * cfg.hidden * 3 * cfg.hidden, "gemm_oproj": 2 * T * cfg.hidden * cfg.hidden,
                  "gemm_fc1": 2 * T * cfg.hidden * cfg.ffn, "gemm_fc2": 2 * T * cfg.hidden * cfg.ffn,
                  "attention": 4 * B * S * S * cfg.hidden}
    kernels = {}
    for name, (ms, n) in prof.items():
        if n == 0:
            continue
        e = {"launches": n, "ms_per_launch": round(ms / n, 4), "ms_per_step": round(ms / K, 3)}
        if name in gemm_flops:
            e["tflops"] = round(gemm_flops[name] / (ms / n * 1e-3) / 1e12, 1)
        kernels[name] = e
    dom = "gemm_fc1"
    ach = gemm_flops[dom] / (prof_timed[dom][0] / prof_timed[dom][1] * 1e-3)      # from the TIMED region
    tjson = load_traffic()                                   # {} when the kernels changed since the PMC passes were taken
    traffic = tjson.get("gemm_fc1_hbm_bytes_per_launch")
    roofline = {"kernel": "gemm_8phase_persistent_kernel<EPI_LN_BIAS_GELU> (FFN-1: LN1 folded, [T,768]x[3072,768]^T + erf-GELU)", "bound": "mfma",
                "achieved": round(ach / 1e12, 1), "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_BF16, 4), "traffic": traffic,
                "flops_per_launch": gemm_flops[dom]}
    fpc = flops_per_chunk(cfg, S)
    encode = {"flops_per_chunk": fpc, "mfma_frac_whole_forward": round(chunks_per_s / world * fpc / MFMA_PEAK_BF16, 4),
              "kernels": kernels, "sustained": None,
              "kernels_note": "per-kernel table: separate untimed pass over the same batches with every class recording events; "
                              "roofline.achieved: FFN-1 events inside the timed region"}

    mark("start_up+burst")
    # ---- sustained leg: the whole configs[1] job (1 M chunks = 977 batches, ~50 s) back to back, so that a clock droop under
    # sustained load is visible next to the K-step burst `value` is timed on
    sustained = None
    e2e_rows = None
    if args.sustained_chunks > 0:
        total = args.sustained_chunks
        nb = (total + B - 1) // B
        big = torch.empty((total, cfg.hidden), dtype=torch.float16, device=dev)
        marks = sorted({0, min(100, nb), max(nb - 100, 0), nb})
        ev = {m: torch.cuda.Event(enable_timing=True) for m in marks}
        # every batch its OWN chunks (the rows of this leg are searched below as "rows the encoder wrote": a corpus that repeated the
        # burst's 23 batches would be 42 exact copies of 23 552 rows): ids drawn on the device, 64 batches at a time (0.1 ms per draw)
        gs_ = torch.Generator(device=dev); gs_.manual_seed(4321 + rank)
        IDS_BLOCK = 64
        ids_blk = None
        clock = ClockSampler(torch.cuda.get_device_properties(dev))
        barrier()
        with clock:
            t0 = time.perf_counter()
            for j in range(nb):
                if j in ev:
                    ev[j].record()
                if j % IDS_BLOCK == 0:
                    ids_blk = torch.randint(4, cfg.vocab_size - 1, (min(IDS_BLOCK, nb - j), B, S), generator=gs_, device=dev, dtype=torch.int32)
                    ids_blk[:, :, 0] = 0; ids_blk[:, :, S - 1] = 2
                nrow = min(B, total - j * B)
                enc.forward_tokens(ids_blk[j % IDS_BLOCK][:nrow], lens[:nrow], S, nrow * S, out=None, out_f16=big[j * B:j * B + nrow], normalize=True)
            ev[nb].record()
            barrier()
            dts = time.perf_counter() - t0
        clk = clock.summary()
        if use_dist:
            tm = torch.tensor([dts], dtype=torch.float64, device=dev); dist.all_reduce(tm, op=dist.ReduceOp.MAX); dts = float(tm.item())
        first_n, last_n = min(100, nb), nb - max(nb - 100, 0)
        nrm = big[-min(B, total):].float().norm(dim=1)
        assert torch.isfinite(nrm).all() and (nrm - 1).abs().max() < 2e-3
        sustained = {"chunks": total * world, "batches_per_gpu": nb, "seconds": round(dts, 2), "chunks_per_s": round(world * total / dts, 1),
                     "mfma_frac_whole_forward": round(total / dts * flops_per_chunk(cfg, S) / MFMA_PEAK_BF16, 4),
                     "ms_per_step_first_100": round(ev[0].elapsed_time(ev[min(100, nb)]) / max(first_n, 1), 3),
                     "ms_per_step_last_100": round(ev[max(nb - 100, 0)].elapsed_time(ev[nb]) / max(last_n, 1), 3),
                     "clock": clk,
                     "note": "same step as `value`, 977 batches back to back, every batch its own chunks (rank 0's events); the last batch is partial"}
        if clk and clk.get("clock_mhz_under_load"):
            # the MFMA peak at the clock the package power cap allowed during THIS leg, and the fractions against it
            att = MFMA_PEAK_BF16 * clk["clock_mhz_under_load"] / 2400.0
            roofline.update({"clock_mhz_under_load": clk["clock_mhz_under_load"], "package_power_w": clk.get("package_power_w"),
                             "peak_at_measured_clock": round(att / 1e12, 1), "frac_of_peak_at_measured_clock": round(ach / att, 4),
                             "clock_note": "`peak` is the 2.4-GHz spec figure; the clock is the median of hwmon samples over the sustained "
                                           "leg (same kernels, 45 s): the chip sits on its package power cap on real operands"})
            sustained["mfma_frac_at_measured_clock"] = round(total / dts * flops_per_chunk(cfg, S) / att, 4)
        e2e_rows = None
        if args.e2e_rows > 0 and total >= args.e2e_rows:
            e2e_rows = big[:args.e2e_rows].clone()              # this rank's slice, as the encoder left it (0.96 GB at 625 k x 768)
        del big
        torch.cuda.empty_cache()

    mark("encode.sustained")
    # ---- search legs ------------------------------------------------------------------------------------
    lib = _lib.load()

    def allmax(x):
        if use_dist:
            tm = torch.tensor([x], dtype=torch.float64, device=dev); dist.all_reduce(tm, op=dist.ReduceOp.MAX); return float(tm.item())
        return x

    def time_search(ix, queries, qb, n_rows, D, int8_bytes=False, pipelined=False):
        """QPS of `ix.search_distributed` (local top-k [+ all-gather + merge under a process group]) on batches of qb queries, and the
        per-kernel split of the same calls (library events on the launch stream)."""
        nq_all = queries.shape[0]
        qb = min(qb, nq_all)
        reps = max(1, min(20, (2048 // qb) if qb < nq_all else 1))
        ix.search_distributed(queries[:qb], 10)                            # warm
        _lib.prof_reset(); _lib.prof_classes(None); _lib.prof_enable(True)
        barrier(); t0 = time.perf_counter()
        for r in range(reps):
            q0 = (r * qb) % max(1, nq_all - qb + 1)
            ix.search_distributed(queries[q0:q0 + qb], 10)
        barrier(); dts = allmax(time.perf_counter() - t0)
        _lib.prof_enable(False)
        p = _lib.prof_read()
        gms, gn = p["search_groupmax"]
        pass_bytes = n_rows * D * (1 if int8_bytes else 2)                 # the bytes of the pass that RAN (int8 rows or fp16 rows)
        e = {"qps": round(reps * qb / dts, 1), "ms_per_batch": round(dts / reps * 1e3, 3),
             "passA_ms_per_launch": round(gms / gn, 4), "passA_launches_per_batch": gn // reps,
             "passA_hbm_GBps": round(pass_bytes / (gms / gn * 1e-3) / 1e9, 1),
             "passA_hbm_frac": round(pass_bytes / (gms / gn * 1e-3) / HBM_PEAK, 4),
             "passA_tops" if int8_bytes else "passA_tflops": round(2 * min(qb, 1024) * n_rows * D / (gms / gn * 1e-3) / 1e12, 1),
             "select_ms": round(p["search_select"][0] / reps, 3), "rescore_ms": round(p["search_rescore"][0] / reps, 3)}
        if int8_bytes:
            e["passA_bytes_per_launch"] = pass_bytes
        if pipelined:
            # the same batches as a STREAM, two in flight (ShardIndex.search_many): batch b + 1's pass A under batch b's select / rescore
            # tail and exchange
            nb = max(32, reps)
            qs_ = [queries[((r * qb) % max(1, nq_all - qb + 1)):((r * qb) % max(1, nq_all - qb + 1)) + qb] for r in range(nb)]
            try:                                                           # an extra: its failure must not cost the bench line
                ix.search_many(qs_[:2], 10, distributed=use_dist)
                barrier(); t0 = time.perf_counter()
                ix.search_many(qs_, 10, distributed=use_dist)
                barrier(); dtp = allmax(time.perf_counter() - t0)
                e["qps_pipelined"] = round(nb * qb / dtp, 1)
                e["ms_per_batch_pipelined"] = round(dtp / nb * 1e3, 3)
                e["pipeline"] = "ShardIndex.search_many: 2 lanes x (scan stream, tail stream), one workspace per lane"
            except Exception as ex:                                        # noqa: BLE001
                e["qps_pipelined_error"] = repr(ex)[:200]
                barrier()
        return e

    search = None
    if args.search_rows > 0:
        # configs[2] (weak scaling with N > 1: every rank holds its own 10 M rows of ONE world x 10 M-row corpus)
        N, nq_all, D = args.search_rows, args.search_queries, cfg.hidden
        corpus = fill_unit_rows(N, D, seed=7, device=dev, row_base=rank * N)
        queries = fill_unit_rows(nq_all, D, seed=11, device=dev)          # same on every rank
        idx = ShardIndex(corpus, idx_base=rank * N)
        qbs = [nq_all if t == "all" else int(t) for t in args.search_qbs.split(",") if t]
        res = {f"Qb={min(qb, nq_all)}": time_search(idx, queries, qb, N, D) for qb in qbs}
        # the same searches with the int8 pre-filter (ShardIndex(prefilter="int8"): first pass over an int8 copy of the rows that yields
        # upper bounds; identical exact answers).  The library's default policy takes the int8 pass at every Qb
        # (a per-index crossover is `ShardIndex(i8_max_queries=...)`, passed with each call: the library keeps no settings).
        res8 = None
        if D % 128 == 0 and D <= 1024:
            idx8 = ShardIndex(corpus, idx_base=rank * N, prefilter="int8")
            res8 = {}
            for qb in qbs:
                qb = min(qb, nq_all)
                s8, i8 = idx8.search_distributed(queries[:qb], 10)
                s16, i16 = idx.search_distributed(queries[:qb], 10)
                same_rows = float((i8 == i16).all(dim=1).float().mean().item())
                e = time_search(idx8, queries, qb, N, D, int8_bytes=True)
                flagged, extra = idx8.certificate_stats()
                e.update({"candidate_groups_per_query": round(extra / max(1, min(qb, 1024)), 1), "queries_overflowed_in_last_pass": flagged,
                          "rows_identical_to_fp16_pass": same_rows,
                          "speedup_vs_fp16_pass": round(e["qps"] / res[f"Qb={qb}"]["qps"], 3)})
                res8[f"Qb={qb}"] = e
        # the same step from query TEXT lengths (16 synthetic token ids per query): encode on the small-batch schedule
        # (arx_encoder_set_low_latency, <= 256 token rows) into an fp16 device matrix, then the search above
        from_tokens = {}
        rs_q = np.random.RandomState(99)
        for qb in (() if args.no_query_leg else (1, 16)):
            qids = rs_q.randint(4, cfg.vocab_size - 1, size=(qb, 16)).astype(np.int32); qids[:, 0] = 0; qids[:, -1] = 2
            d_q = torch.from_numpy(qids).to(dev); d_l = torch.full((qb,), 16, dtype=torch.int32, device=dev)
            q16 = torch.empty((qb, D), dtype=torch.float16, device=dev)
            lat = {}
            for tag, ll, ix in (("default_schedule", False, idx), ("small_batch_schedule", True, idx),
                                ("small_batch_schedule_int8_prefilter", True, idx8 if res8 is not None else None)):
                if ix is None:
                    continue
                def one():
                    enc.forward_tokens(d_q, d_l, 16, qb * 16, out=None, out_f16=q16, normalize=True, low_latency=ll)
                    return ix.search_distributed(q16, 10)
                for _ in range(3): one()
                torch.cuda.synchronize(dev)
                ts = []
                for _ in range(15):
                    t0 = time.perf_counter(); one(); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t0)
                lat[tag + "_ms"] = round(float(np.median(ts)) * 1e3, 3)
            best = min(v for kk, v in lat.items() if kk.startswith("small_batch"))
            lat["qps_best"] = round(qb / (best * 1e-3), 1)
            from_tokens[f"Qb={qb}"] = lat
        if res8 is not None:
            del idx8
            torch.cuda.empty_cache()
        r64 = res.get("Qb=64") or next(iter(res.values()))
        straffic = tjson.get("search_groupmax64_hbm_bytes_per_launch")
        search = {"workload": f"configs[2]: {N} x {D} fp16 rows per rank (rows [rank*N, (rank+1)*N) of one seeded corpus), {nq_all} queries, k=10, world {world}",
                  "results": res,
                  "int8_prefilter": None if res8 is None else {
                      "note": "same corpus + an int8 copy (dim + 8 bytes per row more): pass A reads the int8 rows and writes rigorous upper "
                              "bounds; select / fp32 rescoring of the fp16 rows / certificate unchanged, answers identical.  Every row of this "
                              "table ran the int8 pass (bytes and rates are that pass's)", "results": res8},
                  "encode_plus_search": {"note": "queries given as 16 token ids each: encoder forward + top-10 search, GPU-synchronised wall time per batch",
                                         **from_tokens},
                  "roofline": {"kernel": "search_groupmax_kernel<64> (pass A at Qb=64)", "bound": "hbm",
                               "achieved": r64["passA_hbm_GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": r64["passA_hbm_frac"], "traffic": straffic if N == 10_000_000 and D == 768 else None,
                               "bytes_per_launch": N * D * 2}}
        del corpus, idx
        torch.cuda.empty_cache()

    mark("search.10M")
    # ---- configs[3]: ONE corpus of --search-total-rows rows cut across the ranks (strong scaling), the query set replicated; every rank
    # scans its own slice, one all-gather of the [Q, k] partials over RCCL, merge.  Rank 0 then searches the WHOLE corpus as a single
    # index and checks the merged answer of the first 64 queries against it, bit for bit (the rows are a function of (seed, global row)).
    if args.search_total_rows > 0:
        NT_, nq_all, D = args.search_total_rows, args.search_queries, cfg.hidden
        lo, hi = shard_bounds(NT_, world, rank)
        if use_dist:                                            # the ranks' row ranges tile [0, N): checked on every rank before anything is timed
            mine = torch.tensor([lo, hi], dtype=torch.int64, device=dev)
            spans = torch.empty((world, 2), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(spans, mine)
            spans = spans.cpu().tolist()
            assert spans[0][0] == 0 and spans[-1][1] == NT_ and all(spans[r][1] == spans[r + 1][0] for r in range(world - 1)), spans
            launch["row_spans"] = spans
        queries = fill_unit_rows(nq_all, D, seed=11, device=dev)
        shard_rows = fill_unit_rows(hi - lo, D, seed=7, device=dev, row_base=lo)
        sidx = ShardIndex(shard_rows, idx_base=lo)
        strong = {}
        for qb in (64, 256, nq_all):
            qb = min(qb, nq_all)
            e = time_search(sidx, queries, qb, hi - lo, D, pipelined=qb < nq_all)
            e["passA_ms_at_hbm_roofline"] = round((hi - lo) * D * 2 / HBM_PEAK * 1e3, 4)
            # the exchange step alone: all-gather of [qb, 10] scores + ids and the merge kernel
            ps_, pi_ = sidx.search(queries[:qb], 10)
            if use_dist:
                gather_partials(ps_, pi_); barrier()
                t0 = time.perf_counter()
                for _ in range(20):
                    a_s, a_i = gather_partials(ps_, pi_)
                    merge_partials(a_s, a_i, 10)
                barrier()
                e["allgather_plus_merge_ms"] = round(allmax(time.perf_counter() - t0) / 20 * 1e3, 4)
            strong[f"Qb={qb}"] = e
        ms_, mi_ = sidx.search_distributed(queries[:64], 10)
        check = None
        if rank == 0:
            try:
                whole = fill_unit_rows(NT_, D, seed=7, device=dev) if world > 1 else shard_rows
                ws_, wi_ = ShardIndex(whole).search(queries[:64], 10)
                check = {"queries": 64, "ids_equal": bool(torch.equal(mi_, wi_)), "scores_equal": bool(torch.equal(ms_, ws_))}
                del whole
            except Exception as e_:                                        # noqa: BLE001
                check = {"error": repr(e_)[:200]}
        if search is None:
            search = {}
        search["strong_scaling"] = {
            "workload": f"configs[3]: ONE corpus of {NT_} x {D} fp16 rows cut {world} way(s) ({hi - lo} rows on rank {rank}), {nq_all} queries "
                        f"replicated, local top-10 -> all-gather of [Q,10] partials -> merge; qps includes the collective", "results": strong,
            "merged_vs_single_index": check, "ranks_seen": launch["ranks_seen"], "launch": launch}
        if world == 1 and NT_ >= 8:
            # configs[3]'s PER-RANK slice on this one GPU: the 625 k-row shard rank 0 of an 8-way cut would hold (launch overheads, select
            # and rescore are as long as the 0.12-ms pass here)
            l8, h8 = shard_bounds(NT_, 8, 0)
            sl = ShardIndex(shard_rows[l8:h8], idx_base=l8)
            r625 = {}
            for qb in (64, 256, nq_all):
                qb = min(qb, nq_all)
                e = time_search(sl, queries, qb, h8 - l8, D, pipelined=qb < nq_all)
                e["passA_ms_at_hbm_roofline"] = round((h8 - l8) * D * 2 / HBM_PEAK * 1e3, 4)
                e["batch_frac_of_hbm_roofline"] = round(((h8 - l8) * D * 2 / HBM_PEAK * 1e3) * max(1, (qb + 1023) // 1024) / e["ms_per_batch"], 4)
                if "ms_per_batch_pipelined" in e:
                    e["batch_frac_of_hbm_roofline_pipelined"] = round(((h8 - l8) * D * 2 / HBM_PEAK * 1e3) / e["ms_per_batch_pipelined"], 4)
                r625[f"Qb={qb}"] = e
            # the same slice behind the int8 first pass (half the bytes; identical exact rows): the batch against the SAME fp16-bytes roofline
            sl8 = ShardIndex(shard_rows[l8:h8], idx_base=l8, prefilter="int8")
            r625_8 = {}
            for qb in (64, 256):
                qb = min(qb, nq_all)
                e = time_search(sl8, queries, qb, h8 - l8, D, int8_bytes=True, pipelined=True)
                e["batch_frac_of_fp16_hbm_roofline"] = round(((h8 - l8) * D * 2 / HBM_PEAK * 1e3) / e["ms_per_batch"], 4)
                if "ms_per_batch_pipelined" in e:
                    e["batch_frac_of_fp16_hbm_roofline_pipelined"] = round(((h8 - l8) * D * 2 / HBM_PEAK * 1e3) / e["ms_per_batch_pipelined"], 4)
                s16, i16 = sl.search(queries[:qb], 10)
                s8, i8 = sl8.search(queries[:qb], 10)
                e["rows_identical_to_fp16_pass"] = float((i16 == i8).all(dim=1).float().mean().item())
                r625_8[f"Qb={qb}"] = e
            search["shard_625k"] = {"workload": f"rows [{l8}, {h8}) of the configs[3] corpus = one rank's slice of an 8-way cut, {D}-d, on one GPU",
                                    "results": r625, "int8_prefilter": r625_8}
            del sl, sl8
        del sidx, shard_rows
        torch.cuda.empty_cache()

    mark("search.strong_scaling+shard_625k")
    # ---- configs[4]'s per-rank slice in the precision that is feasible (fp8 encode: measured infeasible at the 1e-3 bar):
    # a 6.25 M x 1024 fp16 shard (12.8 GB) searched on one GPU
    if args.d1024_rows > 0 and world == 1:
        N4, D4 = args.d1024_rows, 1024
        c4 = fill_unit_rows(N4, D4, seed=7, device=dev)
        q4 = fill_unit_rows(2048, D4, seed=11, device=dev)
        i4 = ShardIndex(c4)
        r4 = {}
        for qb in (1, 64, 256):
            e = time_search(i4, q4, qb, N4, D4)
            e["passA_ms_at_hbm_roofline"] = round(N4 * D4 * 2 / HBM_PEAK * 1e3, 4)
            r4[f"Qb={qb}"] = e
        i48 = ShardIndex(c4, prefilter="int8")
        r48 = {}
        for qb in (1, 64):
            e = time_search(i48, q4, qb, N4, D4, int8_bytes=True)
            r48[f"Qb={qb}"] = e
        if search is None:
            search = {}
        search["d1024"] = {"workload": f"configs[4] per-rank slice: {N4} x {D4} fp16 rows (one of 8 shards of 50 M), 2048 queries, k=10, one GPU; "
                                       f"bge-large encode in bf16: profiles/r03 (fp8 infeasible at the parity bar)",
                           "results": r4, "int8_prefilter": r48}
        del i4, i48, c4, q4
        torch.cuda.empty_cache()

    mark("search.d1024")
    # ---- search.clustered: the same search on EMBEDDING-LIKE rows (every figure above is on iid Gaussian unit rows, the
    # friendliest case for group-max selection and for the int8 bound).  Clusters of ~500 rows around shared centres + 3 hot dimensions;
    # queries are new points of the same mixture.  Reported: QPS per Qb for the fp16 and int8 first passes, how often the certificate's slow
    # path ran, and what an `adaptive` index decides.
    if args.clustered_rows > 0 and world == 1:
        try:                                                   # an extra leg: its failure must not cost the bench line
            from arxiv_rag_amd.index import fill_clustered_rows
            Nc, Dc = args.clustered_rows, cfg.hidden
            ncl = max(16, Nc // 500)
            cc = fill_clustered_rows(Nc, Dc, seed=21, n_clusters=ncl, device=dev)
            qc = fill_clustered_rows(2048, Dc, seed=21, n_clusters=ncl, device=dev, row_base=1 << 40)
            ic = ShardIndex(cc)
            rc16 = {}
            for qb in (1, 64, 256):
                e = time_search(ic, qc, qb, Nc, Dc)
                ic.search(qc[:qb], 10)
                fl, ex = ic.certificate_stats()
                e["certificate"] = {"queries_on_the_slow_path": fl, "extra_groups_rescored": ex, "of_queries": qb}
                rc16[f"Qb={qb}"] = e
            rc8, adaptive = None, None
            if Dc % 128 == 0 and Dc <= 1024:
                ic8 = ShardIndex(cc, prefilter="int8")
                rc8 = {}
                for qb in (1, 64, 256):
                    e = time_search(ic8, qc, qb, Nc, Dc, int8_bytes=True)
                    s8, i8 = ic8.search(qc[:qb], 10)
                    fl, ex = ic8.certificate_stats()
                    s16, i16 = ic.search(qc[:qb], 10)
                    e["certificate"] = {"queries_overflowed_to_the_exhaustive_kernel": fl, "candidate_pairs": ex, "of_queries": qb}
                    e["rows_identical_to_fp16_pass"] = float((i8 == i16).all(dim=1).float().mean().item())
                    e["speedup_vs_fp16_pass"] = round(e["qps"] / rc16[f"Qb={qb}"]["qps"], 3)
                    rc8[f"Qb={qb}"] = e
                ia = ShardIndex(cc, prefilter="int8", adaptive=True)
                for _ in range(4):
                    ia.search(qc[:64], 10)
                adaptive = {"prefilter_switched_off": bool(ia.prefilter_disabled)}
                del ic8, ia
            # the HARD layout: topic order — runs of 500 consecutive rows share a centre (the chunks of one paper), so a query's neighbours fill whole
            # 64-row groups with near-tied rows: every selected group expands, the int8 bound lets whole runs through
            topic = None
            try:
                ct_ = fill_clustered_rows(Nc, Dc, seed=22, n_clusters=-500, device=dev)
                qrows = torch.randint(0, Nc, (2048,), device=dev)
                # a query = a NEW point near a random corpus row's run: that row + isotropic noise (generator with spread 1, no hot dimensions), renormalised
                qt_ = ct_[qrows].clone()
                noise = fill_clustered_rows(2048, Dc, seed=23, n_clusters=1 << 20, device=dev, spread=1.0, hot_dims=0)
                qt_ = torch.nn.functional.normalize(qt_.float() + 0.15 * noise.float(), dim=1).to(torch.float16)
                it_ = ShardIndex(ct_)
                topic = {"fp16_pass": {}, "int8_prefilter": {}}
                for qb in (64, 256):
                    e = time_search(it_, qt_, qb, Nc, Dc)
                    it_.search(qt_[:qb], 10); fl, ex = it_.certificate_stats()
                    e["certificate"] = {"queries_on_the_slow_path": fl, "extra_groups_rescored": ex, "of_queries": qb}
                    topic["fp16_pass"][f"Qb={qb}"] = e
                it8 = ShardIndex(ct_, prefilter="int8")
                for qb in (64, 256):
                    e = time_search(it8, qt_, qb, Nc, Dc, int8_bytes=True)
                    s8, i8 = it8.search(qt_[:qb], 10); fl, ex = it8.certificate_stats()
                    s16, i16 = it_.search(qt_[:qb], 10)
                    e["certificate"] = {"queries_overflowed_to_the_exhaustive_kernel": fl, "candidate_pairs": ex, "of_queries": qb}
                    e["scores_identical_to_fp16_pass"] = bool(torch.equal(s8, s16))
                    e["speedup_vs_fp16_pass"] = round(e["qps"] / topic["fp16_pass"][f"Qb={qb}"]["qps"], 3)
                    topic["int8_prefilter"][f"Qb={qb}"] = e
                ia_ = ShardIndex(ct_, prefilter="int8", adaptive=True)
                for _ in range(4):
                    ia_.search(qt_[:64], 10)
                topic["adaptive_index"] = {"prefilter_switched_off": bool(ia_.prefilter_disabled)}
                s_, i_ = it_.search(qt_[:8], 10)
                full_t = torch.cat([qt_[:8].float() @ ct_[a:a + 1_000_000].float().T for a in range(0, Nc, 1_000_000)], dim=1)
                topic["top10_scores_equal_fp32_reference_on_8_queries"] = bool(((full_t.topk(10, dim=1).values - s_).abs().max() < 1e-5).item())
                topic["workload"] = (f"{Nc} x {Dc} rows in TOPIC ORDER: runs of 500 consecutive rows share a centre (spread 0.35, 3 hot dimensions); a query = a "
                                     f"corpus row + 0.15 x unit noise, renormalised")
                del it_, it8, ia_, ct_, qt_, full_t, noise
                torch.cuda.empty_cache()
            except Exception as ex_:                                   # noqa: BLE001
                topic = {"error": repr(ex_)[:300]}
            # exactness on a subset, against the fp32 scores of every row (device fp32 matmul of the same fp16 values, 8 queries)
            s_, i_ = ic.search(qc[:8], 10)
            full = qc[:8].float() @ cc.float().T if Nc <= 2_000_000 else torch.cat([qc[:8].float() @ cc[a:a + 1_000_000].float().T
                                                                                     for a in range(0, Nc, 1_000_000)], dim=1)
            ref = full.topk(10, dim=1)
            ok = bool(((ref.values - s_).abs().max() < 1e-5).item())
            # ANISOTROPIC rows: the same corpus with a component shared by every row added (c <- normalize(c + 1.5 u): mean pairwise cosine ~0.69, the
            # range of models like bge; the encoder's own seeded-weight rows, 0.977, are the e2e leg's).  The int8 index is centred on the shard's mean
            # and — ShardIndex's own choice from the index's |mean| — the queries on its direction: candidate lists as on iid rows.  Beside it the
            # rows-only centring (centre_query=False) for what the query side is worth.
            aniso = None
            try:
                if Dc % 128 == 0 and Dc <= 1024:
                    gu = torch.Generator(device=dev); gu.manual_seed(31)
                    u_ = torch.nn.functional.normalize(torch.randn(Dc, generator=gu, device=dev), dim=0) * 1.5
                    for t_ in (cc, qc):
                        for a in range(0, t_.shape[0], 1_000_000):
                            t_[a:a + 1_000_000] = torch.nn.functional.normalize(t_[a:a + 1_000_000].float() + u_[None, :], dim=1).to(torch.float16)
                    cos_ = float((cc[:2048].float() @ cc[2048:4096].float().T).mean().item())
                    ia16 = ShardIndex(cc)
                    aniso = {"workload": f"the clustered corpus with a shared component: c <- normalize(c + 1.5 u), queries alike; mean pairwise cosine {cos_:.3f}",
                             "fp16_pass": {}, "int8_prefilter": {}, "int8_rows_centred_only": {}}
                    for qb in (64, 256):
                        aniso["fp16_pass"][f"Qb={qb}"] = time_search(ia16, qc, qb, Nc, Dc)
                    for key, cq in (("int8_prefilter", None), ("int8_rows_centred_only", False)):
                        ia8 = ShardIndex(cc, prefilter="int8", centre_query=cq)
                        for qb in (64, 256):
                            e = time_search(ia8, qc, qb, Nc, Dc, int8_bytes=True)
                            s8, i8 = ia8.search(qc[:qb], 10); fl, ex = ia8.certificate_stats()
                            s16, i16 = ia16.search(qc[:qb], 10)
                            e["certificate"] = {"queries_overflowed_to_the_exhaustive_kernel": fl, "candidate_pairs": ex, "of_queries": qb}
                            e["rows_identical_to_fp16_pass"] = float((i8 == i16).all(dim=1).float().mean().item())
                            e["speedup_vs_fp16_pass"] = round(e["qps"] / aniso["fp16_pass"][f"Qb={qb}"]["qps"], 3)
                            aniso[key][f"Qb={qb}"] = e
                        aniso[key]["index_mean_norm"] = round(ia8.i8_mean_norm, 4); aniso[key]["query_centred_too"] = bool(ia8.centre_query)
                        del ia8
                    iaa = ShardIndex(cc, prefilter="int8", adaptive=True)
                    for _ in range(4):
                        iaa.search(qc[:64], 10)
                    aniso["adaptive_index"] = {"prefilter_switched_off": bool(iaa.prefilter_disabled)}
                    s_, i_ = ia16.search(qc[:8], 10)
                    full_a = torch.cat([qc[:8].float() @ cc[a:a + 1_000_000].float().T for a in range(0, Nc, 1_000_000)], dim=1)
                    aniso["top10_scores_equal_fp32_reference_on_8_queries"] = bool(((full_a.topk(10, dim=1).values - s_).abs().max() < 1e-5).item())
                    del ia16, iaa, full_a
            except Exception as ex_:                                   # noqa: BLE001
                aniso = {"error": repr(ex_)[:300]}
            if search is None:
                search = {}
            search["clustered"] = {"workload": f"{Nc} x {Dc} fp16 rows in {ncl} clusters (spread 0.35) with 3 hot dimensions (gain 6), unit-normalised, "
                                               f"generated in HBM (arx_fill_clustered_rows_f16_at); 2048 queries from the same mixture; k=10",
                                   "fp16_pass": rc16, "int8_prefilter": rc8, "adaptive_index": adaptive, "topic_order": topic, "anisotropic": aniso,
                                   "top10_scores_equal_fp32_reference_on_8_queries": ok}
            del ic, cc, qc, full
            torch.cuda.empty_cache()
        except Exception as ex_:                                   # noqa: BLE001
            if search is None:
                search = {}
            search["clustered"] = {"error": repr(ex_)[:300]}
            torch.cuda.empty_cache()

    mark("search.clustered")
    # ---- e2e_rank_slice (configs[3]'s real flow, one rank's share): the rows the ENCODER wrote during the sustained leg (the first
    # --e2e-rows of them, still in HBM) are searched where they lie, by queries the encoder also wrote, through search_distributed (RCCL
    # all-gather + merge under a process group).  Seeded-weight embeddings are nearly parallel (every pair of rows has a large cosine):
    # the worst case for the certificate; `certificate` says what it costs.
    e2e = None
    if e2e_rows is not None:
        try:                                                   # an extra leg: its failure must not cost the bench line
            n_e, D_e = e2e_rows.shape
            gq = torch.Generator(device=dev); gq.manual_seed(777)
            nq_e = min(args.search_queries, 10_000)
            qe = torch.empty((nq_e, D_e), dtype=torch.float16, device=dev)
            for a in range(0, nq_e, B):
                nrow = min(B, nq_e - a)
                qids = torch.randint(4, cfg.vocab_size - 1, (nrow, S), generator=gq, device=dev, dtype=torch.int32)
                qids[:, 0] = 0; qids[:, S - 1] = 2
                enc.forward_tokens(qids, lens[:nrow], S, nrow * S, out=None, out_f16=qe[a:a + nrow], normalize=True)
            ie = ShardIndex(e2e_rows, idx_base=rank * n_e)
            res_e = {}
            for qb in (64, 256, nq_e):
                qb = min(qb, nq_e)
                e = time_search(ie, qe, qb, n_e, D_e, pipelined=qb < nq_e)
                ie.search(qe[:min(qb, 1024)], 10)
                fl, ex = ie.certificate_stats()
                e["certificate"] = {"queries_on_the_slow_path": fl, "extra_groups_rescored": ex, "of_queries": min(qb, 1024)}
                res_e[f"Qb={qb}"] = e
            pair_cos = float((e2e_rows[:2048].float() @ e2e_rows[2048:4096].float().T).mean().item())
            # the int8 first pass on THESE rows (all within cosine ~0.98 of each other: the bound's slack exceeds their spread, every group is a
            # candidate): what the `adaptive` index (the CLI's and HipCollection's setting) does about it
            i8_adaptive = None
            if D_e % 128 == 0 and D_e <= 1024:
                ia_e = ShardIndex(e2e_rows, idx_base=rank * n_e, prefilter="int8", adaptive=True)
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                s8_, i8_ = ia_e.search(qe[:64], 10)
                torch.cuda.synchronize(dev); first_ms = (time.perf_counter() - t0) * 1e3
                s16_, i16_ = ie.search(qe[:64], 10)
                i8_adaptive = {"prefilter_switched_off_after_first_batch": bool(ia_e.prefilter_disabled), "first_batch_ms": round(first_ms, 3),
                               "first_batch_rows_identical_to_fp16_pass": bool(torch.equal(i8_, i16_)),
                               "index_mean_norm": round(ia_e.i8_mean_norm, 4), "query_centred_too": bool(ia_e.centre_query)}
                if not ia_e.prefilter_disabled:              # the int8 first pass ON these rows: per batch, alone and pipelined, against the fp16 pass above
                    for qb_ in (64, 256):
                        qs_ = [qe[j:j + qb_] for j in range(0, min(len(qe), 40 * qb_), qb_)]
                        for _ in range(2): ia_e.search(qs_[0], 10)
                        torch.cuda.synchronize(dev); t0 = time.perf_counter()
                        for q_ in qs_: ia_e.search(q_, 10)
                        torch.cuda.synchronize(dev); t1 = time.perf_counter()
                        ia_e.search_many(qs_, 10)
                        torch.cuda.synchronize(dev); t2 = time.perf_counter()
                        fl_, pr_ = ia_e.certificate_stats()
                        i8_adaptive[f"Qb={qb_}"] = {"ms_per_batch": round((t1 - t0) / len(qs_) * 1e3, 3), "ms_per_batch_pipelined": round((t2 - t1) / len(qs_) * 1e3, 3),
                                                    "queries_overflowed_in_last_batch": int(fl_), "candidate_groups_per_query": round(pr_ / qb_, 1)}
                del ia_e
            chk = None
            if rank == 0:
                try:                                                 # exactness vs the oracle (CPU, fp32 on the same fp16 values), 8 queries
                    from oracle import search_oracle as SO
                    ms_, mi_ = ie.search(qe[:8], 10)
                    rs_, ri_ = SO.topk_search(e2e_rows.cpu().numpy(), qe[:8].cpu().numpy(), 11, idx_base=rank * n_e)
                    got = mi_.cpu().numpy()
                    bad = 0
                    for qq in range(8):
                        if set(got[qq].tolist()) != set(ri_[qq, :10].tolist()) and not (rs_[qq, 9] - rs_[qq, 10] < 1e-6):
                            bad += 1
                    chk = {"queries": 8, "top10_sets_equal_oracle": bad == 0,
                           "max_abs_score_diff": float(np.abs(ms_.cpu().numpy() - rs_[:, :10]).max())}
                except Exception as ex_:                               # noqa: BLE001
                    chk = {"error": repr(ex_)[:200]}
            e2e = {"workload": f"{n_e} x {D_e} fp16 rows written by the encoder in the sustained leg (rank {rank}'s slice, never left HBM), {nq_e} queries "
                               f"encoded by the same forward ({S} token ids each), top-10 through search_distributed; world {world}",
                   "encode_chunks_per_s": None if sustained is None else sustained["chunks_per_s"], "results": res_e,
                   "mean_cosine_between_random_rows": round(pair_cos, 4), "vs_oracle": chk, "int8_adaptive_index": i8_adaptive}
            del ie, e2e_rows, qe
            torch.cuda.empty_cache()
        except Exception as ex_:                                   # noqa: BLE001
            e2e = {"error": repr(ex_)[:300]}
            torch.cuda.empty_cache()

    mark("e2e_rank_slice")
    # ---- configs[4]'s encode half in the precision that is feasible (fp8: measured infeasible at the 1e-3 bar): bge-large shape
    # (24 L / 1024 / 16 heads / FFN 4096, CLS pool), bf16, 256 chunks x 256 tokens per step
    bge = None
    if args.bge_steps > 0 and world == 1:
        try:
            enc.close()
            cfg_b = C.PRESETS["BAAI/bge-large-en-v1.5"]
            Bb, Sb, Kb, Wb = 256, 256, args.bge_steps, 2
            enc_b = HipEncoder(cfg_b, seeded_state_dict(cfg_b, seed=0), device=dev, max_tokens=Bb * Sb, max_seqs=Bb)
            ids_b = torch.randint(4, cfg_b.vocab_size - 1, (Kb + Wb, Bb, Sb), generator=g, device=dev, dtype=torch.int32)
            ids_b[:, :, 0] = 101; ids_b[:, :, Sb - 1] = 102
            lens_b = torch.full((Bb,), Sb, dtype=torch.int32, device=dev)
            out_b = torch.empty((Bb, cfg_b.hidden), dtype=torch.float16, device=dev)
            for i in range(Wb):
                enc_b.forward_tokens(ids_b[i], lens_b, Sb, Bb * Sb, out=None, out_f16=out_b, normalize=True)
            _lib.prof_reset(); _lib.prof_classes(["gemm_fc1"]); _lib.prof_enable(True)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for i in range(Kb):
                enc_b.forward_tokens(ids_b[Wb + i], lens_b, Sb, Bb * Sb, out=None, out_f16=out_b, normalize=True)
            torch.cuda.synchronize(dev); dtb = time.perf_counter() - t0
            _lib.prof_enable(False)
            pb = _lib.prof_read()["gemm_fc1"]
            fpc_b = flops_per_chunk(cfg_b, Sb)
            nb_ = out_b.float().norm(dim=1)
            assert torch.isfinite(nb_).all() and (nb_ - 1).abs().max() < 2e-3
            bge = {"workload": f"configs[4] encode half, bf16: bge-large shape ({cfg_b.layers} L / {cfg_b.hidden} / FFN {cfg_b.ffn}, CLS pool), "
                               f"{Bb} chunks x {Sb} token ids per step, {Kb} steps after {Wb} warm-up; fp8 ruled out at the 1e-3 bar (DESIGN §4b)",
                   "chunks_per_s": round(Kb * Bb / dtb, 1), "ms_per_step": round(dtb / Kb * 1e3, 3), "flops_per_chunk": fpc_b,
                   "mfma_frac_whole_forward": round(Kb * Bb / dtb * fpc_b / MFMA_PEAK_BF16, 4),
                   "ffn1_tflops": round(2 * Bb * Sb * cfg_b.hidden * cfg_b.ffn / (pb[0] / max(pb[1], 1) * 1e-3) / 1e12, 1) if pb[1] else None}
            enc_b.close()
            del enc_b, ids_b, out_b
            torch.cuda.empty_cache()
        except Exception as ex_:                                   # noqa: BLE001  (an extra leg: its failure must not cost the bench line)
            bge = {"error": repr(ex_)[:300]}

    mark("encode.bge_large")
    encode["sustained"] = sustained
    encode["bge_large"] = bge
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, cfg, sd, S, args.cpu_budget)

    mark("cpu_baseline")
    if rank == 0:
        out = {
            "metric": "chunks embedded/sec + QPS@top-10, all-mpnet-base-v2 768-d, 1/2/4/8 MI355X",
            "value": round(chunks_per_s, 1), "unit": "chunks/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs[1]: {args.model} shape, {B} chunks x {S} token ids per step per GPU, "
                                   f"encode-only (embed+{cfg.layers} layers+{'CLS' if cfg.pool == C.POOL_CLS else 'mean'}-pool+L2 -> fp16 corpus rows)",
                       "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}",
                       "weights": "seeded N(0,0.02^2), seed 0"},
            "roofline": roofline, "cpu_baseline": cpu, "encode": encode, "search": search, "e2e_rank_slice": e2e, "launch": launch,
            "wall_s": dict(wall, total=round(sum(wall.values()), 1)),
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

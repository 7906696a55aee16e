#!/usr/bin/env python3
"""Dev: which step of the 1-rank RCCL pipelined search dies (faulthandler + a line per step)."""
import faulthandler, os, sys
faulthandler.enable()
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch, torch.distributed as dist
from arxiv_rag_amd.index import ShardIndex, gather_partials, merge_partials
from oracle import search_oracle as SO
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
def say(*a): print(*a, flush=True)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29544", rank=0, world_size=1, device_id=torch.device("cuda:0"))
Cm = SO.unit_rows_f16(5000, 128, 1); Q = SO.unit_rows_f16(33, 128, 2)
idx = ShardIndex(torch.from_numpy(Cm).cuda(), idx_base=7)
qd = torch.from_numpy(Q).cuda()
s0, i0 = idx.search(qd, 10); say("search ok")
s1, i1 = idx.search_distributed(qd, 10); say("search_distributed ok")
mode = sys.argv[1] if len(sys.argv) > 1 else "many"
if mode == "ext":
    pipe = idx._pipeline(32); say("pipeline ok")
    with torch.cuda.stream(pipe["tail"]):
        say("in stream ctx")
        a = gather_partials(s0, i0); say("gather on external masked stream ok")
        m = merge_partials(*a, 10); say("merge ok")
    torch.cuda.synchronize(); say("sync ok")
else:
    for tail in (0, 32):
        got = idx.search_many([qd[:16], qd[16:32], qd[32:]], 10, distributed=True, tail_cus=tail); say("search_many", tail, "ok")
        torch.cuda.synchronize()
        assert torch.equal(torch.cat([i for _, i in got]), i1)
dist.destroy_process_group(); say("done")

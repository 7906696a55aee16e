"""Which import leaves a GPU device node open in a CPU-only child?  (bench.py cpu_baseline leg C sizing)"""
import os, sys
def gpu_fds():
    out = []
    for fd in os.listdir("/proc/self/fd"):
        try:
            t = os.readlink(f"/proc/self/fd/{fd}")
        except OSError:
            continue
        if "kfd" in t or "/dev/dri" in t:
            out.append(t)
    return out
for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
    os.environ[k] = ""
print("start", gpu_fds())
import numpy; print("numpy", gpu_fds())
import torch; print("torch", gpu_fds())
torch.cuda.is_available = lambda: False
torch.cuda.device_count = lambda: 0
import transformers; print("transformers", gpu_fds())
from transformers import MPNetModel, MPNetConfig
m = MPNetModel(MPNetConfig(num_hidden_layers=1), add_pooling_layer=False); print("model", gpu_fds())

"""dev tool: cycle stamps of ncfp_fwd_kernel / ncfp_bwd_kernel (wave 0 of workgroup 7).  Needs the -DCTR_STAMPS build of
ncf_proj.hip: `bash dev/build_stamps.sh`, then CTRHIP_LIB=dev/timing/libctrhip_stamps.so python dev/ncfp_stamps.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeplearningrecommendationsystem_amd import _lib

dev = torch.device("cuda:0")
model, inputs, y, _ = bench.build_workload("neuralcf", dev, 0)
loss_fn = torch.nn.BCELoss()
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
for _ in range(6):
    if mode == "eval":
        with torch.no_grad():
            model(*inputs)
    else:
        model.zero_grad(set_to_none=True)
        loss_fn(model(*inputs), y).backward()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib = _lib.load()
lib.ctr_ncfp_debug_stamps.argtypes = [ctypes.c_void_p]
assert lib.ctr_ncfp_debug_stamps(buf) == 0
for k, name in ((0, "fwd"), (1, "bwd")):
    st = [buf[k * 64 + i] for i in range(64)]
    if not any(st):
        continue
    n = max(i for i in range(64) if st[i]) + 1
    print(name, "deltas (cycles):", [st[i] - st[i - 1] for i in range(1, n)], "total", st[n - 1] - st[0])

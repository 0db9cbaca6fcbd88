"""dev tool: cycle stamps of ncf16_bwd_kernel (needs the -DCTR_STAMPS build: CTRHIP_LIB=dev/timing/libctrhip_stamps.so)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeplearningrecommendationsystem_amd import _lib

dev = torch.device("cuda:0")
model, inputs, y, _ = bench.build_workload("neuralcf", dev, 0)
loss_fn = torch.nn.BCELoss()
for _ in range(4):
    model.zero_grad(set_to_none=True)
    loss = loss_fn(model(*inputs), y)
    loss.backward()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = _lib.load()
lib.ctr_ncf16_debug_stamps.argtypes = [ctypes.c_void_p]
rc = lib.ctr_ncf16_debug_stamps(buf)
names = {0: "start", 12: "w requested", 13: "operands requested", 14: "w in LDS", 1: "staged", 2: "zeroed", 3: "g0", 4: "g1", 5: "g2", 6: "g3", 8: "loop end", 9: "round0", 10: "round1", 11: "out"}
for w in range(4):
    st = [buf[w * 16 + i] for i in range(16)]
    print(f"wave {w}: " + "  ".join(f"{n}={st[i] - st[0]}" for i, n in names.items() if i == 0 or st[i]))

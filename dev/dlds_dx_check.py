"""dev tool: direct-to-LDS input-gradient GEMM vs fp64"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for m, n, k, act, accum in [(4096, 128, 64, 1, False), (5000, 112, 36, 0, True), (65536, 256, 512, 1, False), (4099, 208, 160, 2, True), (70001, 16, 32, 1, False), (65536, 512, 96, 1, False), (8191, 96, 260, 1, True), (77, 64, 128, 1, False), (65536, 64, 16, 1, False), (65536, 641, 641, 1, False), (4097, 161, 256, 1, True), (4097, 256, 161, 2, False), (1000, 33, 17, 0, False), (300, 70, 103, 1, True), (513, 19, 30, 1, False)]:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / n ** 0.5
    z = torch.randn(m, n, device=dev)
    y = z.relu() if act == 1 else z.sigmoid() if act == 2 else z
    gy = torch.randn(m, n, device=dev)
    gx = torch.ones(m, k, device=dev)
    ops.linear_bwd(x, w, y, gy, act, gx, None, None, accumulate_gx=accum)
    yd = y.double()
    gz = gy.double() * ((yd > 0).double() if act == 1 else yd * (1 - yd) if act == 2 else 1.0)
    ref = gz @ w.double() + (1.0 if accum else 0.0)
    e = ((gx.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{m}x{n}x{k} act{act} accum{accum}: rel err gx {e:.3e}", flush=True)
    assert e < 1e-5
print("ok")

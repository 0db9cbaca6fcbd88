"""dev tool: direct-to-LDS forward GEMM vs the tile kernel (same process cannot toggle; run twice) and vs fp64"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for m, n, k, act in [(1000, 128, 64, 1), (129, 33, 16, 0), (65536, 64, 128, 1), (4097, 200, 160, 2), (70000, 8, 16, 1), (5, 300, 48, 1), (65536, 256, 512, 1), (65536, 641, 641, 1), (4097, 161, 256, 1), (4097, 256, 161, 2), (1000, 33, 17, 0), (300, 70, 103, 1), (513, 100, 30, 1)]:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    y = ops.linear_fwd(x, w, b, act)
    z = x.double() @ w.double().t() + b.double()
    ref = (z.relu() if act == 1 else z.sigmoid() if act == 2 else z)
    err = (y.double() - ref).abs().max().item()
    print(f"{m}x{n}x{k} act{act}: max err {err:.3e}", flush=True)
    assert err < 2e-5, err
print("ok")

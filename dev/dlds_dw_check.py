"""dev tool: direct-to-LDS weight-gradient GEMM vs fp64"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for m, n, k, act in [(4096, 128, 64, 1), (5000, 100, 36, 0), (65536, 256, 512, 1), (4099, 200, 160, 2), (70001, 128, 32, 1), (65536, 512, 96, 1), (8191, 96, 260, 1), (70001, 64, 128, 1), (9000, 32, 100, 2), (65536, 64, 256, 0), (65536, 641, 641, 1), (4097, 161, 256, 1), (4097, 256, 161, 2), (5000, 33, 101, 0), (4100, 97, 35, 1), (4100, 130, 131, 1)]:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5
    z = torch.randn(m, n, device=dev)
    y = z.relu() if act == 1 else z.sigmoid() if act == 2 else z
    gy = torch.randn(m, n, device=dev)
    gw = torch.ones(n, k, device=dev); gb = torch.ones(n, device=dev)
    ops.linear_bwd(x, w, y, gy, act, None, gw, gb)
    yd = y.double()
    gz = gy.double() * ((yd > 0).double() if act == 1 else yd * (1 - yd) if act == 2 else 1.0)
    rw = 1.0 + gz.t() @ x.double(); rb = 1.0 + gz.sum(0)
    ew = ((gw.double() - rw).abs().max() / rw.abs().max()).item(); eb = ((gb.double() - rb).abs().max() / rb.abs().max()).item()
    print(f"{m}x{n}x{k} act{act}: rel err gw {ew:.3e} gb {eb:.3e}", flush=True)
    assert ew < 1e-5 and eb < 1e-5
print("ok")

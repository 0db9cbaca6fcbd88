"""dev tool: a few launches of one GEMM shape (ours: fwd / dx / dw; hipBLASLt: fwd) for rocprofv3 runs"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops

m, n, k = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "65536x256x512").split("x"))
dev = "cuda:0"
x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
y = torch.empty(m, n, device=dev); gy = torch.randn(m, n, device=dev)
gx = torch.empty(m, k, device=dev); gw = torch.zeros(n, k, device=dev); gb = torch.zeros(n, device=dev)
for _ in range(6):
    ops.linear_fwd(x, w, b, 1, out=y)
    ops.linear_bwd(x, w, y, gy, 1, gx, None, None)
    ops.linear_bwd(x, w, y, gy, 1, None, gw, gb)
    torch.nn.functional.linear(x, w, b)
torch.cuda.synchronize()

"""dev tool: do 16-byte direct-to-LDS loads work from rows that are only 4-byte aligned?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for m, n, k, ld in [(65536, 640, 640, 641), (65536, 256, 160, 161), (65536, 512, 640, 641)]:
    xb = torch.randn(m, ld, device=dev); wb = torch.randn(n, ld, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    x, w = xb[:, 1:1 + k] if ld > k else xb, wb[:, :k]
    y = torch.empty(m, n, device=dev)
    ops.linear_fwd(x, w, b, 1, out=y)
    ref = (x.double() @ w.double().t() + b.double()).relu()
    err = (y.double() - ref).abs().max().item()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.linear_fwd(x, w, b, 1, out=y)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e3
    print(f"{m}x{n}x{k} ld{ld}: err {err:.2e}  {t:.1f} us  {2.0*m*n*k/t/1e6:.1f} TF", flush=True)

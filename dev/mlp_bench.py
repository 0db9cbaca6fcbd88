"""dev tool: fused MLP fwd/bwd time vs. batch (slope = per-tile cost, intercept = prolog/epilog)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops, _lib

dims = [128, 64, 32, 16, 8, 64]
acts = [1, 1, 1, 1, 0]
if len(sys.argv) > 1:
    dims = [int(v) for v in sys.argv[1].split(",")]
    acts = [int(v) for v in sys.argv[2].split(",")]
dev = "cuda:0"
layers = [ops.Layer(torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, device=dev) * 0.1, a)
          for k, n, a in zip(dims[:-1], dims[1:], acts)]


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for m in (8192, 32768, 65536, 131072, 262144, 524288):
    x = torch.randn(m, dims[0], device=dev)
    ys = ops.mlp_fwd(x, layers)
    gy = torch.randn(m, dims[-1], device=dev)
    gx = torch.empty(m, dims[0], device=dev)
    t_f = timeit(lambda: ops.mlp_fwd(x, layers))
    t_b = timeit(lambda: ops.mlp_bwd(ys, layers, gy, gx))
    print(f"m={m:7d}: fwd {t_f:8.1f} us  bwd {t_b:8.1f} us", flush=True)

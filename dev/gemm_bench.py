"""dev tool: time the linear kernels on given shapes (GPU box)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops

shapes = [(65536, 256, 512), (65536, 512, 96), (65536, 128, 256), (65536, 64, 128), (65536, 8, 16), (65536, 256, 164), (3276800, 128, 192), (3276800, 64, 128), (3276800, 64, 48), (3276800, 32, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
dev = "cuda:0"
for m, n, k in shapes:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    y = torch.empty(m, n, device=dev); gy = torch.randn(m, n, device=dev)
    gx = torch.empty(m, k, device=dev); gw = torch.zeros(n, k, device=dev); gb = torch.zeros(n, device=dev)
    def timeit(fn, reps=10):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e3
    fl = 2.0 * m * n * k
    t_f = timeit(lambda: ops.linear_fwd(x, w, b, 1, out=y))
    t_dx = timeit(lambda: ops.linear_bwd(x, w, y, gy, 1, gx, None, None))
    t_dw = timeit(lambda: ops.linear_bwd(x, w, y, gy, 1, None, gw, gb))
    t_ref = timeit(lambda: torch.nn.functional.linear(x, w, b))
    print(f"{m}x{n}x{k}: fwd {t_f:8.1f}us {fl/t_f/1e6:6.1f}TF | dx {t_dx:8.1f}us {fl/t_dx/1e6:6.1f}TF | dw {t_dw:8.1f}us {fl/t_dw/1e6:6.1f}TF | hipblaslt fwd {t_ref:8.1f}us {fl/t_ref/1e6:6.1f}TF", flush=True)

"""dev tool: cycle stamps of the fused MLP backward (needs the -DCTR_MLP_TIMING build: CTRHIP_LIB=dev/timing/libctrhip_timing.so)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops

dims = [128, 64, 32, 16, 8]; acts = [1, 1, 1, 1]   # the pinned NeuralCF tower (NcfTowerShape), no head
dev = "cuda:0"
layers = [ops.Layer(torch.randn(n, k, device=dev) / k ** 0.5, torch.randn(n, device=dev) * 0.1, a)
          for k, n, a in zip(dims[:-1], dims[1:], acts)]
slab = sum(n * k + n for k, n in zip(dims[:-1], dims[1:]))
for m in (65536,):
    x = torch.randn(m, dims[0], device=dev)
    ys = ops.mlp_fwd(x, layers)
    gy = torch.randn(m, dims[-1], device=dev); gx = torch.empty(m, dims[0], device=dev)
    for _ in range(3): ops.mlp_bwd(ys, layers, gy, gx)
    torch.cuda.synchronize()
    grid = min(256, (m + 127) // 128)
    st = ops._scratch(x.device)[grid * slab: grid * slab + grid * 16].view(grid, 16).cpu()
    print(f"m={m} grid={grid}: median stamps (cycles) {st.median(0).values.tolist()}  max {st.max(0).values.tolist()}")

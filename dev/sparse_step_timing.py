"""dev: where a sparse-mode optimizer step spends its time (gather26 shape): device time of the row-wise Adam by
HIP events vs host wall time of Adam.step()"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd.model import EmbeddingStage
from deeplearningrecommendationsystem_amd.optim import Adam

dev = "cuda:0"
with torch.device(dev):
    m = EmbeddingStage(26, 1_000_000, 16)
m.sparse_grads(True)
opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
g = torch.Generator().manual_seed(0)
idx = torch.randint(0, 1_000_000, (65536, 26), generator=g).to(dev)
gout = torch.randn(65536, 26 * 16, device=dev)
def bwd():
    out = m(idx); out.backward(gout)
for it in range(6):
    bwd(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(); opt.step(); b.record(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"opt.step: host {1e6*(t1-t0):8.1f} us   device {a.elapsed_time(b)*1e3:8.1f} us")

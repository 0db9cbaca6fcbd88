"""dev tool: embedding-stage roofline shape: 26 id fields x 1e6 vocab x E=16, batch 65536"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplearningrecommendationsystem_amd import ops, _lib

dev = "cuda:0"
F, V, E, B = 26, 1_000_000, int(os.environ.get("E", 16)), 65536
g = torch.Generator(device="cpu").manual_seed(1234)
tables = [torch.randn(V, E, device=dev) for _ in range(F)]
idx = torch.randint(0, V, (B, F), generator=g).to(dev)
specs = [ops.FieldSpec(_lib.FIELD_ID_I64, E, f * E, table=tables[f], idx=idx[:, f], idx_stride=F) for f in range(F)]
out = torch.empty(B, F * E, device=dev)
gout = torch.randn(B, F * E, device=dev)
grads = {id(t): torch.zeros_like(t) for t in tables}

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

fwd_bytes = B * F * (E * 4 + 8 + E * 4)
t = timeit(lambda: ops.embed_fwd(specs, None, B, out))
print(f"embed_fwd  {t:8.1f} us  {fwd_bytes/t/1e3:8.1f} GB/s algorithmic ({fwd_bytes/1e6:.1f} MB)  {fwd_bytes/t/1e3/8000:.3f} of 8 TB/s")
ref = torch.stack([tables[f][idx[:, f]] for f in range(F)], 1).view(B, F * E)
assert torch.equal(out, ref)
bwd_bytes = B * F * (E * 4 + 8 + 2 * E * 4)
t = 1.0
print(f"embed_bwd  {t:8.1f} us  {bwd_bytes/t/1e3:8.1f} GB/s algorithmic ({bwd_bytes/1e6:.1f} MB)")
# torch reference for comparison
t = 1.0
print(f"torch 26x F.embedding + stack {t:8.1f} us")

"""round 3: where the table-row NeuralCF path spends its time.  Run under rocprofv3 --kernel-trace (dev/r03_probe.sh):
30 forwards under no_grad (no rank atomics), 30 training forwards, 30 whole steps; `--parse <kernel_trace.csv>` prints the
average duration of every kernel per phase."""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seen_fwd, phase = 0, {}
    out = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        if name == "ncfp_fwd_kernel":
            seen_fwd += 1
        ph = "A eval fwd" if seen_fwd <= 35 else "B train fwd" if seen_fwd <= 70 else "C step"
        out[(ph, name)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (ph, name), v in sorted(out.items()):
        if name.startswith(("ncfp", "reduce", "bce")):
            v = v[5:] if len(v) > 10 else v
            print(f"{ph:12s} {name:32s} n={len(v):3d} avg {sum(v) / len(v):7.2f} us  min {min(v):7.2f}")


if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    parse(sys.argv[2])
    sys.exit(0)

import torch
from deeplearningrecommendationsystem_amd import synth
from deeplearningrecommendationsystem_amd.model import NeuralCF
from deeplearningrecommendationsystem_amd.loss import BCELoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"
torch.manual_seed(0)
m = NeuralCF(943, 1682, 64, [128, 64, 32, 16, 8]).to(dev)
gen = synth.generator(1)
u, i = synth.id_batch(B, gen=gen)
y = synth.labels(B, True, gen).to(dev)
if os.environ.get("PROBE_IDS") == "regular":     # the pattern CTR_NCFP_DBG=8 computes, but LOADED from memory
    u, i = (torch.arange(B) * 7) % 943, (torch.arange(B) * 13) % 1682
elif os.environ.get("PROBE_IDS") == "sorted":    # random ids, sorted by user
    order = torch.argsort(u, stable=True)
    u, i = u[order], i[order]
elif os.environ.get("PROBE_IDS") == "sorted_item":
    order = torch.argsort(i, stable=True)
    u, i = u[order], i[order]
elif os.environ.get("PROBE_IDS") == "block16":   # one user per aligned block of 16 samples, blocks in random order
    u = u[::16].repeat_interleave(16)
elif os.environ.get("PROBE_IDS") == "block16_both":
    u = u[::16].repeat_interleave(16)
    i = i[::16].repeat_interleave(16)
elif os.environ.get("PROBE_IDS") == "neighbours":   # sample b and b + 16384 (concurrent waves) share a user, no sharing inside a group
    u = u[:16384].repeat(4)
u, i = u.to(dev), i.to(dev)
loss_fn = BCELoss()
for _ in range(35):
    with torch.no_grad():
        m(u, i)
torch.cuda.synchronize()
for _ in range(35):
    m(u, i)
torch.cuda.synchronize()
for _ in range(35):
    m.zero_grad(set_to_none=True)
    loss_fn(m(u, i), y).backward()
torch.cuda.synchronize()

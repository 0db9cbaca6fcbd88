"""CPU tests of host-side logic that needs no GPU: the evaluator's metric definitions."""
import math

import numpy as np
import pytest
import torch


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_evaluator_equals_sklearn_on_thresholded_predictions(seed):
    """reference evaluator/evaluator.py:14-20: every metric, roc_auc_score included, is computed by
    sklearn from ``(y_pred >= 0.5).astype(int)``"""
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score, roc_auc_score
    from deeplearningrecommendationsystem_amd.evaluator import Evaluator
    rng = np.random.default_rng(seed)
    y = (rng.random(257) < 0.4).astype(np.float32)
    p = np.clip(0.35 * y + rng.random(257) * 0.7, 0, 1).astype(np.float32)
    p[:5] = 0.5  # the threshold itself counts as positive
    hard = (p >= 0.5).astype(int)
    want = [accuracy_score(y, hard), precision_score(y, hard), recall_score(y, hard), f1_score(y, hard),
            roc_auc_score(y, hard)]
    got = Evaluator.eval(torch.from_numpy(y).view(-1, 1), torch.from_numpy(p).view(-1, 1))
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
    # the extra: ranking AUC of the raw scores
    assert abs(Evaluator.score_auc(torch.from_numpy(y), torch.from_numpy(p)) - roc_auc_score(y, p)) < 1e-6


def test_evaluator_single_class_auc_is_nan():
    from deeplearningrecommendationsystem_amd.evaluator import Evaluator
    out = Evaluator.eval(torch.ones(4), torch.tensor([0.9, 0.2, 0.7, 0.6]))
    assert math.isnan(out[4]) and abs(out[0] - 0.75) < 1e-6


# the reference scripts' own import lines for the packages this build replaces (scripts/<m>.py:5-13:
# ``sys.path.append('../')`` then ``from model.<m> import <Class>``, ``from trainer.trainer import Trainer``)
# and the constructor call of each script, executed with compat/ first on the path
_SCRIPT_LINES = """
import sys
sys.path.append('../')
from model.mf import MatrixFactorization
from model.neuralcf import NeuralCF
from model.ffm import FFM
from model.pnn import PNN
from model.deepcrossing import DeepCrossing
from model.deepfm import DeepFM
from model.din import DIN
from model.dien import DIEN
from model.deepcross import DeepCross
from model.widedeep import WideDeep
from model.nfm import NFM
from model.afm import AFM
from model.lr import LogisticRegression
from model.autorec import AutoRec
from trainer.trainer import Trainer
from evaluator.evaluator import Evaluator
from sampler.sampler import Sampler
assert Sampler.__module__ == 'deeplearningrecommendationsystem_amd.sampler.sampler'
import torch.nn
from torch import optim
nu, ni = 943, 1682
models = [MatrixFactorization(nu, ni, 64), NeuralCF(nu, ni, 256, [512, 256, 128, 64, 32]), FFM(43, 32),
          PNN(256, [256, 128, 64, 32]), DeepCrossing(nu, ni, 32, [256, 128, 64, 32]),
          DeepFM(nu, ni, [512, 256, 128, 1], 128), DIN(ni, 64), DIEN(ni, 16),
          DeepCross(nu, ni, 3, [512, 256, 128, 1], 128), WideDeep(nu, ni, [512, 256, 128, 1], 128),
          NFM(nu, ni, [512, 256, 128, 1], 128), AFM(nu, ni, 128, 64), LogisticRegression(nu, ni, 43),
          AutoRec(ni, 500)]
for m in models:
    assert type(m).__module__.startswith('deeplearningrecommendationsystem_amd.model.'), type(m)
    t = Trainer(m, torch.nn.BCELoss(), optim.Adam(m.parameters(), lr=0.001, weight_decay=1e-5))
    assert t.model is m
import model.pnn, deeplearningrecommendationsystem_amd.model.pnn as real
assert model.pnn is real          # one module object under both names
print('ok', len(models))
"""


def test_reference_import_lines_resolve_to_the_mirrors(tmp_path):
    """``PYTHONPATH=<repo>/compat`` and the reference's scripts import the MI355X
    mirrors under the names they already use (model.*, trainer.trainer, evaluator.evaluator)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = tmp_path / "scripts"   # cwd = a scripts/ directory, as the reference runs them
    scripts.mkdir()
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "compat"), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", _SCRIPT_LINES], cwd=scripts, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip() == "ok 14"


def test_default_samplers_get_distinct_streams():
    """``Sampler()`` x 3 (train / valid / test in the reference's scripts) must not share a stream; ``Sampler(seed=)``
    stays reproducible"""
    from deeplearningrecommendationsystem_amd.sampler import Sampler
    seeds = {Sampler()._seed for _ in range(8)}
    assert len(seeds) == 8
    assert Sampler(seed=5)._seed == Sampler(seed=5)._seed == 5
    assert all(0 <= s < 2 ** 64 for s in seeds)


def test_kernel_profiler_mixed_shapes_and_roofline_bounds():
    """a label covering two shapes must report total work over total time (round 2 printed a
    fraction of 1.13), and a byte-dominant launch whose PMC traffic is far under its algorithmic bytes is priced
    against L2, not called an HBM fraction"""
    import importlib
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    from deeplearningrecommendationsystem_amd.ops import KernelProfiler
    # small fast stack first, big slow stack second, same label
    k = KernelProfiler.fold([("mlp", 1e6, 2e9, 20.0), ("mlp", 1e8, 2e11, 2000.0), ("x", 2e6, 0, 3.0)])
    assert k["mlp"]["calls"] == 2 and k["mlp"]["shapes"] == 2
    e = bench.roofline_entry("mlp", k["mlp"])
    assert e["bound"] == "mfma" and abs(e["achieved"] - (2e9 + 2e11) / 2020e-6 / 1e12) < 1e-6 and e["frac"] < 1.0
    # byte-dominant, no PMC evidence, long launch: hbm
    rec = dict(avg_us=50.0, bytes=2e8, flops=0)
    assert bench.roofline_entry("g", rec)["bound"] == "hbm"
    # PMC says 20 % of the algorithmic bytes reached DRAM: l2, fraction against the L2 rate
    t = {"hbm_bytes_raw": 3e7, "hbm_bytes_fetch_x2": 4e7}
    e = bench.roofline_entry("g", rec, t)
    assert e["bound"] == "l2" and e["peak"] == bench.L2_PEAK_GBS and abs(e["dram_gbs"] - 4e7 / 50e-6 / 1e9) < 1e-6
    # a 3 us launch moving 2 MB: latency
    assert bench.roofline_entry("x", k["x"])["bound"] == "latency"


def test_ncf_counts_holder_hands_out_a_clean_buffer_or_a_private_one():
    """ops.NcfCounts (the sample counters of ctr_ncf_proj_fwd): clean -> handed out; busy -> a private zeroed buffer for
    the second forward; dirty (a training forward without a backward) -> zero-filled before it is handed out again"""
    import torch
    from deeplearningrecommendationsystem_amd import _lib, ops
    h = ops.NcfCounts()
    cpu = torch.device("cpu")
    a, owns_a = h.take(10, cpu)
    assert owns_a and a.numel() == 10 * _lib.CTR_NCF_PROJ_COUNT_STRIDE and int(a.abs().sum()) == 0 and h.state == "busy"
    b, owns_b = h.take(10, cpu)                      # a second forward before the first one's backward
    assert not owns_b and b.data_ptr() != a.data_ptr() and int(b.abs().sum()) == 0 and h.state == "busy"
    h.state = "clean"                                # the owner's backward ran (its last launch zeroes the counters)
    c, owns_c = h.take(10, cpu)
    assert owns_c and c.data_ptr() == a.data_ptr()
    c[3] = 7                                         # ... a forward that never got its backward leaves counts behind
    h.state = "dirty"
    d, owns_d = h.take(10, cpu)
    assert owns_d and d.data_ptr() == a.data_ptr() and int(d.abs().sum()) == 0
    h.state = "clean"
    e, _ = h.take(20, cpu)                           # more table rows: a new buffer
    assert e.numel() == 20 * _lib.CTR_NCF_PROJ_COUNT_STRIDE and int(e.abs().sum()) == 0

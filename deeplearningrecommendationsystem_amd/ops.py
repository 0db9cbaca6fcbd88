"""Host-side wrappers over the C ABI (include/ctrhip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function
passes raw pointers + leading dimensions to libctrhip and enqueues on torch's
current stream.  2-D operands may be column slices of a wider buffer (stride(1)
must be 1), which is how the reference's ``torch.cat`` calls disappear: producers
write straight into the consumer's operand.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, FIELD_BAG, FIELD_DENSE, FIELD_ID_F32, FIELD_ID_I64,
                   FIELD_PROD_I64, Field)

__all__ = ["FieldSpec", "embed_fwd", "embed_bwd", "linear_fwd", "linear_bwd", "mf_fwd", "mf_bwd",
           "ACT_NONE", "ACT_RELU", "ACT_SIGMOID"]


# ---------------------------------------------------------------------------
# optional per-launch timing (bench.py): a HIP event pair on the stream the
# kernel is enqueued on, plus the launch's algorithmic bytes / flops
# ---------------------------------------------------------------------------
class KernelProfiler:
    """HIP-event timing of every C-ABI call of the steps run while it is installed.

    ``spacer_us`` > 0 keeps the GPU busy for about that long (``torch.cuda._sleep``) before each
    bracketed call: start event, launch and end event are then all queued before the GPU reaches
    them, so the pair measures the kernel(s) of the call and not the host's launch latency (without
    it a 70 us kernel reads ~85 us; with it the numbers agree with rocprofv3 to a few per cent)."""

    def __init__(self, spacer_us: float = 0.0):
        self.records = []
        self.spacer_cycles = 0
        if spacer_us > 0:
            # calibrate _sleep: cycles per microsecond of whatever clock it spins on
            probe = 200_000
            torch.cuda._sleep(probe)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            torch.cuda._sleep(probe)
            b.record()
            torch.cuda.synchronize()
            us = max(a.elapsed_time(b) * 1e3, 1e-3)
            self.spacer_cycles = max(1, int(probe / us * spacer_us))

    def add(self, label, nbytes, flops, start, end):
        self.records.append((label, nbytes, flops, start, end))

    def summary(self):
        """{label: dict(calls, total_us, avg_us, bytes, flops)} -- call after a sync.  ``bytes`` / ``flops`` are the
        AVERAGE per call of what the calls under that label declared, ``avg_us`` the average duration: a label that
        covers launches of different shapes (two stacks through ``mlp_fused_fwd``) then still gives a physically
        possible rate, total work over total time (round 2 divided the first call's work by the average time of
        all calls and printed a fraction above 1).  ``shapes`` counts the distinct (bytes, flops) pairs seen."""
        torch.cuda.synchronize()
        return self.fold([(label, nbytes, flops, start.elapsed_time(end) * 1e3)
                          for label, nbytes, flops, start, end in self.records])

    @staticmethod
    def fold(timed):
        """``summary`` over (label, bytes, flops, microseconds) records"""
        out = {}
        for label, nbytes, flops, us in timed:
            d = out.setdefault(label, dict(calls=0, total_us=0.0, total_bytes=0, total_flops=0, _shapes=set()))
            d["calls"] += 1
            d["total_us"] += us
            d["total_bytes"] += nbytes
            d["total_flops"] += flops
            d["_shapes"].add((nbytes, flops))
        for d in out.values():
            d["avg_us"] = d["total_us"] / d["calls"]
            d["bytes"] = d["total_bytes"] / d["calls"]
            d["flops"] = d["total_flops"] / d["calls"]
            d["shapes"] = len(d.pop("_shapes"))
        return out


_profiler: Optional[KernelProfiler] = None


def set_profiler(p: Optional[KernelProfiler]) -> None:
    global _profiler
    _profiler = p


def _timed(label, meta, fn, *args):
    """run one C call; when a profiler is installed bracket it with events"""
    p = _profiler
    if p is None:
        return fn(*args)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if p.spacer_cycles:
        torch.cuda._sleep(p.spacer_cycles)
    start.record()
    rc = fn(*args)
    end.record()
    nbytes, flops = meta()
    p.add(label, nbytes, flops, start, end)
    return rc


def _embed_bytes(specs, batch, backward):
    """algorithmic HBM bytes of one embedding-stage launch"""
    rows = idx = xcols = out = small = 0
    seen = set()
    for s in specs:
        out += s.width * 4
        for t in (s.idx, s.idx2):
            if t is not None and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                idx += 8
        if s.kind in (FIELD_ID_I64, FIELD_ID_F32):
            rows += s.width * 4
            xcols += 4 if s.kind == FIELD_ID_F32 else 0
        elif s.kind == FIELD_PROD_I64:
            rows += 2 * s.width * 4
        elif s.kind == FIELD_BAG:
            xcols += 4 * s.bag_size
            small += s.bag_size * s.width * 4
        else:
            xcols += 4 * s.width
    if not backward:
        return batch * (rows + idx + xcols + out) + small
    # backward: read gout, idx, (PROD: both rows), read-modify-write each touched grad row
    prod_rows = sum(2 * s.width * 4 for s in specs if s.kind == FIELD_PROD_I64)
    return batch * (out + idx + xcols + prod_rows + 2 * rows) + 2 * small


SCRATCH_FLOATS = 16 * 1024 * 1024  # 64 MiB: per-workgroup partials of the weight-gradient reductions


def _scratch(device) -> torch.Tensor:
    """device scratch for one backward launch.  Taken from torch's caching allocator per
    call (a few microseconds, no hipMalloc in steady state), so it follows the usual
    stream-ordered lifetime rules instead of being shared global state."""
    return torch.empty(SCRATCH_FLOATS, dtype=torch.float32, device=device)


def _mat(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dim() != 2 or t.dtype != torch.float32 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{what}: expected a 2-D float32 tensor with unit inner stride, got "
                         f"{tuple(t.shape)} {t.dtype} strides {t.stride()}")
    _lib.require_device(t)
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


@dataclass
class FieldSpec:
    """one output slice of the embedding stage (mirror of ``ctr_field_t``)"""
    kind: int
    width: int
    out_col: int
    table: Optional[torch.Tensor] = None      # (vocab, width) parameter
    src_col: int = 0
    bag_size: int = 0
    idx: Optional[torch.Tensor] = None        # int64, (B,) or a column of (B,L)
    idx_stride: int = 1
    table2: Optional[torch.Tensor] = None
    idx2: Optional[torch.Tensor] = None


def _field_array(specs: Sequence[FieldSpec], grads=None):
    if len(specs) > _lib.CTR_MAX_FIELDS:
        raise ValueError(f"at most {_lib.CTR_MAX_FIELDS} fields per launch")
    arr = (Field * len(specs))()
    for k, s in enumerate(specs):
        f = arr[k]
        f.kind, f.width, f.out_col, f.src_col, f.bag_size = s.kind, s.width, s.out_col, s.src_col, s.bag_size
        f.idx_stride = s.idx_stride
        if s.table is not None:
            _lib.require_device(s.table)
            if s.table.dtype != torch.float32 or not s.table.is_contiguous() or s.table.shape[1] != s.width:
                raise ValueError("embedding table must be contiguous float32 (vocab, width)")
            f.vocab = s.table.shape[0]
            f.table = s.table.data_ptr()
        if s.idx is not None:
            _lib.require_device(s.idx)
            if s.idx.dtype != torch.int64:
                raise ValueError("indices must be int64")
            f.idx = s.idx.data_ptr()
        if s.table2 is not None:
            f.vocab2 = s.table2.shape[0]
            f.table2 = s.table2.data_ptr()
            f.idx2 = s.idx2.data_ptr()
        if grads is not None:
            g = grads.get(id(s.table)) if s.table is not None else None
            f.grad = None if g is None else g.data_ptr()
            g2 = grads.get(id(s.table2)) if s.table2 is not None else None
            f.grad2 = None if g2 is None else g2.data_ptr()
    return arr


def embed_fwd(specs: Sequence[FieldSpec], x: Optional[torch.Tensor], batch: int, out: torch.Tensor,
              err_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fused gather + bag pooling + concat into ``out`` (B, >= sum widths)"""
    out = _mat(out, "out")
    if x is not None:
        x = _mat(x, "x")
    arr = _field_array(specs)
    rc = _timed("embed_fwd", lambda: (_embed_bytes(specs, batch, False), 0),
                _lib.load().ctr_embed_fwd, arr, len(specs), _lib.ptr(x), _ld(x) if x is not None else 0, batch,
                out.data_ptr(), _ld(out), _lib.ptr(err_flag), _lib.stream_ptr())
    _lib.check(rc, "ctr_embed_fwd")
    return out


def embed_bwd(specs: Sequence[FieldSpec], x: Optional[torch.Tensor], batch: int, gout: torch.Tensor,
              grads: dict) -> None:
    """accumulate into ``grads[id(table)]`` (dense, same shape as the table)"""
    gout = _mat(gout, "gout")
    if x is not None:
        x = _mat(x, "x")
    arr = _field_array(specs, grads)
    ws = _scratch(gout.device)  # bag partials + the small-table sort buffers
    rc = _timed("embed_bwd", lambda: (_embed_bytes(specs, batch, True), 0),
                _lib.load().ctr_embed_bwd, arr, len(specs), _lib.ptr(x), _ld(x) if x is not None else 0, batch,
                gout.data_ptr(), _ld(gout), _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream_ptr())
    _lib.check(rc, "ctr_embed_bwd")


def linear_fwd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int = ACT_NONE,
               out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(x @ w.T + b (+ residual)); ``w`` is nn.Linear's (out, in) weight"""
    x, w = _mat(x, "x"), _mat(w, "w")
    m, k = x.shape
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError(f"linear: x has {k} columns, weight expects {w.shape[1]}")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    out = _mat(out, "out")
    if residual is not None:
        residual = _mat(residual, "residual")
    rc = _timed(f"linear_fwd[{m}x{n}x{k}]",
                lambda: (4 * (m * k + n * k + n + m * n * (2 if residual is not None else 1)), 2 * m * n * k),
                _lib.load().ctr_linear_fwd, x.data_ptr(), _ld(x), w.data_ptr(), _ld(w), _lib.ptr(b),
                _lib.ptr(residual), _ld(residual) if residual is not None else 0, out.data_ptr(), _ld(out),
                m, n, k, act, _lib.stream_ptr())
    _lib.check(rc, "ctr_linear_fwd")
    return out


def linear_bwd(x: torch.Tensor, w: torch.Tensor, y: Optional[torch.Tensor], gy: torch.Tensor, act: int,
               gx: Optional[torch.Tensor], gw: Optional[torch.Tensor], gb: Optional[torch.Tensor],
               accumulate_gx: bool = False) -> None:
    """gz = gy*act'(y); gx (=|+=) gz @ w; gw += gz.T @ x; gb += gz.sum(0)"""
    x, w, gy = _mat(x, "x"), _mat(w, "w"), _mat(gy, "gy")
    m, k = x.shape
    n = w.shape[0]
    if y is not None:
        y = _mat(y, "y")
    if gx is not None:
        gx = _mat(gx, "gx")
    fn = _lib.load().ctr_linear_bwd
    ygy = 4 * m * n * (2 if (y is not None and act != ACT_NONE) else 1)
    ws = _scratch(x.device) if gw is not None else None

    def call(gx_, gw_, gb_):
        return (x.data_ptr(), _ld(x), w.data_ptr(), _ld(w), _lib.ptr(y), _ld(y) if y is not None else 0,
                gy.data_ptr(), _ld(gy), _lib.ptr(gx_), _ld(gx_) if gx_ is not None else 0, int(accumulate_gx),
                _lib.ptr(gw_), _ld(gw_) if gw_ is not None else 0, _lib.ptr(gb_), m, n, k, act,
                _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream_ptr())

    if _profiler is None:
        _lib.check(fn(*call(gx, gw, gb)), "ctr_linear_bwd")
        return
    # profiling: one C call per kernel so each gets its own event pair
    if gx is not None:
        _lib.check(_timed(f"linear_bwd_dx[{m}x{n}x{k}]", lambda: (ygy + 4 * (n * k + m * k), 2 * m * n * k),
                          fn, *call(gx, None, None)), "ctr_linear_bwd")
    if gw is not None:
        _lib.check(_timed(f"linear_bwd_dw[{m}x{n}x{k}]", lambda: (ygy + 4 * (m * k + 2 * n * k), 2 * m * n * k),
                          fn, *call(None, gw, gb)), "ctr_linear_bwd")


def mf_fwd(user_table, item_table, user_idx, item_idx, err_flag=None) -> torch.Tensor:
    _lib.require_device(user_table, item_table, user_idx, item_idx)
    batch = user_idx.numel()
    prob = torch.empty(batch, dtype=torch.float32, device=user_table.device)
    dim = user_table.shape[1]
    rc = _timed("mf_fwd", lambda: (batch * (8 * dim + 16 + 4), 2 * batch * dim),
                _lib.load().ctr_mf_fwd, user_table.data_ptr(), user_table.shape[0], item_table.data_ptr(),
                item_table.shape[0], dim, user_idx.data_ptr(), item_idx.data_ptr(), batch, prob.data_ptr(),
                _lib.ptr(err_flag), _lib.stream_ptr())
    _lib.check(rc, "ctr_mf_fwd")
    return prob


def mf_bwd(user_table, item_table, user_idx, item_idx, prob, gprob, guser, gitem) -> None:
    dim, batch = user_table.shape[1], user_idx.numel()
    rc = _timed("mf_bwd", lambda: (batch * (8 * dim + 16 + 8 + 16 * dim), 4 * batch * dim),
                _lib.load().ctr_mf_bwd, user_table.data_ptr(), user_table.shape[0], item_table.data_ptr(),
                item_table.shape[0], dim, user_idx.data_ptr(), item_idx.data_ptr(), batch, prob.data_ptr(),
                gprob.data_ptr(), _lib.ptr(guser), _lib.ptr(gitem), _lib.stream_ptr())
    _lib.check(rc, "ctr_mf_bwd")


# ---------------------------------------------------------------------------
# a stack of nn.Linear(+activation) layers, forward and hand-written backward
# ---------------------------------------------------------------------------
@dataclass
class Layer:
    weight: torch.Tensor
    bias: Optional[torch.Tensor]
    act: int


def zero_grads(tensors: Sequence[torch.Tensor], lazy: bool = False) -> dict:
    """{id(t): zero tensor shaped like t} for every parameter of a backward pass, carved
    out of ONE flat buffer so a step issues a single memset instead of one per
    parameter (each piece starts 16-byte aligned).  ``lazy=True``: the flat buffer is NOT filled; it is returned
    under the key ``"flat"`` and whoever launches first has to clear it (``mlp_head_bwd(zero=...)``) or call
    ``.zero_()`` on it."""
    out = {}
    dense = []
    for t in tensors:
        st = getattr(t, "_ctr_sparse", None)
        if st is None:
            dense.append(t)
            continue
        # sparse mode (sparse.py): the table's persistent accumulation buffer, clean outside pending rows --
        # the scatter kernels add into it as they would into a fresh zero buffer, nothing is zero-filled
        if st.grad.device != t.device or st.grad.shape != t.shape:
            raise RuntimeError("sparse-mode state does not match its table (enable sparse_grads after .to(device))")
        out[id(t)] = st.grad
    if not dense:
        return out
    sizes = [(t.numel() + 3) // 4 * 4 for t in dense]
    flat = (torch.empty if lazy else torch.zeros)(sum(sizes), dtype=torch.float32, device=dense[0].device)
    if lazy:
        out["flat"] = flat
    off = 0
    for t, n in zip(dense, sizes):
        out[id(t)] = flat[off:off + t.numel()].view(t.shape)
        off += n
    return out


def fold_head_fwd(u_full: torch.Tensor, p: int, w: torch.Tensor, b: Optional[torch.Tensor],
                  b2: Optional[torch.Tensor], out=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(wfold (1, p+k), cfold (1,)) of the layer ``w, b`` (n x k) composed with the single-unit layer
    ``u_full`` (1, p+n) whose first p columns pass through (ctr_fold_head_fwd)"""
    n, k = w.shape
    if out is not None:
        wfold, cfold = out
    else:
        wfold = torch.empty((1, p + k), dtype=torch.float32, device=w.device)
        cfold = torch.empty(1, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().ctr_fold_head_fwd(u_full.data_ptr(), p, w.data_ptr(), _ld(w), _lib.ptr(b), _lib.ptr(b2), n, k,
                                             wfold.data_ptr(), cfold.data_ptr(), _lib.stream_ptr()), "ctr_fold_head_fwd")
    return wfold, cfold


def fold_head_bwd(u_full: torch.Tensor, p: int, w: torch.Tensor, b: Optional[torch.Tensor], gwfold: torch.Tensor,
                  gc: torch.Tensor, gu_full: Optional[torch.Tensor], gw: Optional[torch.Tensor],
                  gb: Optional[torch.Tensor], gb2: Optional[torch.Tensor]) -> None:
    """chain rule through ``fold_head_fwd``; every gradient is accumulated (ctr_fold_head_bwd)"""
    n, k = w.shape
    _lib.check(_lib.load().ctr_fold_head_bwd(u_full.data_ptr(), p, w.data_ptr(), _ld(w), _lib.ptr(b), n, k,
                                             gwfold.data_ptr(), gc.data_ptr(), _lib.ptr(gu_full), _lib.ptr(gw),
                                             _ld(gw) if gw is not None else 0, _lib.ptr(gb), _lib.ptr(gb2),
                                             _lib.stream_ptr()), "ctr_fold_head_bwd")


FUSED_MLP = True  # narrow stacks run as one launch (ctr_mlp_fwd/bwd) when their shape allows
_REFUSED = (-2, -4)  # CTR_ELIMIT / CTR_EALIGN: nothing was enqueued, take the per-layer path


def _mlp_layer_array(layers, ys, grads=None):
    arr = (_lib.MlpLayer * len(layers))()
    for k, (layer, y) in enumerate(zip(layers, ys)):
        e = arr[k]
        e.w, e.b = layer.weight.data_ptr(), _lib.ptr(layer.bias)
        e.y, e.ldy = y.data_ptr(), _ld(y)
        e.n, e.k, e.act = layer.weight.shape[0], layer.weight.shape[1], layer.act
        if grads is not None:
            e.gw, e.gb = grads[k][0].data_ptr(), _lib.ptr(grads[k][1])
    return arr


def _fusable(x, layers) -> bool:
    if not FUSED_MLP or len(layers) < 2 or len(layers) > 8:
        return False
    for layer in layers:
        n, k = layer.weight.shape
        if n > 128 or k > 128 or k % 8 or layer.bias is None or not layer.weight.is_contiguous():
            return False
    return x.shape[0] >= 1024


class Head:
    """single-unit layer on [x_extra | last activations] formed in the fused stack's epilogue
    (``ctr_mlp_head_t``): ``out = act(x_extra . w[:p] + y_last . w[p:] + c)``"""

    def __init__(self, x_extra: Optional[torch.Tensor], w: torch.Tensor, c: torch.Tensor, act: int):
        self.x_extra, self.w, self.c, self.act = x_extra, w, c, act
        self.out: Optional[torch.Tensor] = None  # (m, 1), set by mlp_fwd


def mlp_fwd(x: torch.Tensor, layers: Sequence[Layer], last_out: Optional[torch.Tensor] = None,
            head: Optional[Head] = None) -> List[torch.Tensor]:
    """returns [x, y_1, ..., y_n]; the last layer may write into ``last_out``.  With ``head`` the
    single-unit layer on top is computed too and left in ``head.out`` (in the same launch when the
    stack runs fused, else by ``linear_fwd`` on the concatenation-free operand, which then must be
    the columns right in front of ``last_out``)."""
    x = _mat(x, "x")
    m = x.shape[0]
    if _fusable(x, layers):
        ys = [torch.empty((m, layer.weight.shape[0]), dtype=torch.float32, device=x.device) for layer in layers[:-1]]
        ys.append(last_out if last_out is not None else
                  torch.empty((m, layers[-1].weight.shape[0]), dtype=torch.float32, device=x.device))
        arr = _mlp_layer_array(layers, ys)
        dims = [(layer.weight.shape[0], layer.weight.shape[1]) for layer in layers]
        lib = _lib.load()
        if head is not None:
            p = 0 if head.x_extra is None else head.x_extra.shape[1]
            out = torch.empty((m, 1), dtype=torch.float32, device=x.device)
            hd = _lib.MlpHead(_lib.ptr(head.x_extra), _ld(head.x_extra) if p else 0, head.w.data_ptr(),
                              head.c.data_ptr(), out.data_ptr(), 1, p, head.act)
            rc = _timed("mlp_fused_fwd", lambda: (4 * m * (dims[0][1] + p + 1 + sum(n for n, _ in dims)),
                                                  2 * m * (sum(n * k for n, k in dims) + p + dims[-1][0])),
                        lib.ctr_mlp_head_fwd, x.data_ptr(), _ld(x), m, arr, len(layers), C.byref(hd), _lib.stream_ptr())
            if rc not in _REFUSED:
                _lib.check(rc, "ctr_mlp_head_fwd")
                head.out = out
                return [x] + ys
            if _profiler is not None and _profiler.records and _profiler.records[-1][0] == "mlp_fused_fwd":
                _profiler.records.pop()  # refused: nothing ran
        rc = _timed("mlp_fused_fwd", lambda: (4 * m * (dims[0][1] + sum(n for n, _ in dims)),
                                              2 * m * sum(n * k for n, k in dims)),
                    lib.ctr_mlp_fwd, x.data_ptr(), _ld(x), m, arr, len(layers), _lib.stream_ptr())
        if rc not in _REFUSED:
            _lib.check(rc, "ctr_mlp_fwd")
            _head_unfused(head, ys[-1])
            return [x] + ys
        if _profiler is not None and _profiler.records and _profiler.records[-1][0] == "mlp_fused_fwd":
            _profiler.records.pop()  # refused: nothing ran
    acts = [x]
    for k, layer in enumerate(layers):
        out = last_out if (k == len(layers) - 1) else None
        acts.append(linear_fwd(acts[-1], layer.weight, layer.bias, layer.act, out=out))
    _head_unfused(head, acts[-1])
    return acts


def embed_mlp_head_fwd(specs: Sequence[FieldSpec], batch: int, buf: torch.Tensor, k0: int, layers: Sequence[Layer],
                       head: Head, last_out: torch.Tensor, err_flag: Optional[torch.Tensor] = None, write_x: bool = True,
                       fold: Optional[tuple] = None):
    """``embed_fwd(specs -> buf)`` and ``mlp_fwd(buf[:, :k0], layers, last_out, head)`` in ONE launch
    (ctr_embed_mlp_head_fwd) where the library has a kernel for the pattern (NeuralCF at BASELINE configs[1]); returns
    the activation list of ``mlp_fwd`` or None when it refused (nothing was enqueued: issue the two calls).
    ``write_x=False``: ``buf[:, :k0]`` is left untouched and the backward has to be ``embed_mlp_head_bwd``.
    ``fold=(u_full, w, b, b2)``: ``fold_head_fwd`` done by the same launch -- ``head.w`` / ``head.c`` are then OUTPUTS
    (uninitialised buffers of the folded shape) that the kernel fills for the backward."""
    buf = _mat(buf, "buf")
    x = buf[:, :k0]
    if not _fusable(x, layers) or head.x_extra is None:
        return None
    m = batch
    ys = [torch.empty((m, layer.weight.shape[0]), dtype=torch.float32, device=buf.device) for layer in layers[:-1]]
    ys.append(last_out)
    arr = _mlp_layer_array(layers, ys)
    farr = _field_array(specs)
    dims = [(layer.weight.shape[0], layer.weight.shape[1]) for layer in layers]
    p = head.x_extra.shape[1]
    out = torch.empty((m, 1), dtype=torch.float32, device=buf.device)
    hd = _lib.MlpHead(_lib.ptr(head.x_extra), _ld(head.x_extra), head.w.data_ptr(), head.c.data_ptr(), out.data_ptr(), 1, p,
                      head.act)
    fd = None
    if fold is not None:
        u_full, fw, fb, fb2 = fold
        fd = _lib.HeadFold(u_full.data_ptr(), fw.data_ptr(), _ld(fw), _lib.ptr(fb), _lib.ptr(fb2), head.w.data_ptr(),
                           head.c.data_ptr(), p, fw.shape[0], fw.shape[1], 0)
    rc = _timed("embed_mlp_fused_fwd",
                lambda: (_embed_bytes(specs, batch, False) + 4 * m * (1 + sum(n for n, _ in dims)),
                         2 * m * (sum(n * k for n, k in dims) + p + dims[-1][0])),
                _lib.load().ctr_embed_mlp_head_fwd, farr, len(specs), batch, buf.data_ptr(), _ld(buf), _lib.ptr(err_flag),
                1 if write_x else 0, arr, len(layers), C.byref(hd), C.byref(fd) if fd is not None else None,
                _lib.stream_ptr())
    if rc in _REFUSED:
        if _profiler is not None and _profiler.records and _profiler.records[-1][0] == "embed_mlp_fused_fwd":
            _profiler.records.pop()  # refused: nothing ran
        return None
    _lib.check(rc, "ctr_embed_mlp_head_fwd")
    head.out = out
    return [x] + ys


def _head_unfused(head: Optional[Head], y_last: torch.Tensor) -> None:
    if head is None:
        return
    if head.x_extra is None:
        operand = y_last
    else:
        # [x_extra | y_last] must already be adjacent columns of one buffer (no torch.cat on the hot path)
        p, n = head.x_extra.shape[1], y_last.shape[1]
        adjacent = (head.x_extra.stride(0) == y_last.stride(0) and
                    head.x_extra.data_ptr() + 4 * p == y_last.data_ptr())
        operand = (torch.as_strided(head.x_extra, (head.x_extra.shape[0], p + n), (head.x_extra.stride(0), 1))
                   if adjacent else torch.cat([head.x_extra, y_last], dim=1))
    head.out = linear_fwd(operand, head.w, head.c, head.act)


def mlp_head_bwd(acts: Sequence[torch.Tensor], layers: Sequence[Layer], head: Head, prob: torch.Tensor,
                 gprob: torch.Tensor, g_extra: torch.Tensor, gw_head: torch.Tensor, gc_head: torch.Tensor,
                 gx_first: torch.Tensor, zeros: dict, gather_specs: Optional[Sequence[FieldSpec]] = None,
                 fold_grad: Optional[tuple] = None, zero: Optional[torch.Tensor] = None):
    """backward of ``mlp_fwd(..., head=head)`` in one launch (ctr_mlp_head_bwd): the head's gz, the stack's
    backward, ``g_extra = gz * w[:p]`` and the head's weight / bias sums.  Returns the per-layer
    ``[(gw, gb)]`` or None when the library has no fused path for this stack (nothing was enqueued)."""
    if not _fusable(acts[0], layers) or head.x_extra is None:
        return None
    grads = [(zeros[id(layer.weight)], zeros[id(layer.bias)]) for layer in layers]
    x0 = _mat(acts[0], "x")
    m = x0.shape[0]
    arr = _mlp_layer_array(layers, acts[1:], grads)
    ws = _scratch(x0.device)
    dims = [(layer.weight.shape[0], layer.weight.shape[1]) for layer in layers]
    p = head.x_extra.shape[1]
    prob, gprob = prob.reshape(-1), gprob.reshape(-1)
    hg = _lib.MlpHeadGrad(prob.data_ptr(), prob.stride(0), gprob.data_ptr(), gprob.stride(0), head.x_extra.data_ptr(),
                          _ld(head.x_extra), head.w.data_ptr(), g_extra.data_ptr(), _ld(g_extra), gw_head.data_ptr(),
                          gc_head.data_ptr(), p, head.act)
    meta = lambda: (4 * m * (2 * dims[0][1] + sum(n for n, _ in dims) + 2 * p + 2),  # noqa: E731
                    4 * m * (sum(n * k for n, k in dims) + p + dims[-1][0]))
    # read: stack input, every saved activation, the extra columns, prob, gprob; written: gX, g_extra.
    # flops: dW + dX of every layer, the head's products and sums
    if gather_specs is not None:
        # the forward ran embed_mlp_head_fwd(write_x=False): the stack input is gathered again from the tables
        farr = _field_array(gather_specs)
        fg = None
        if fold_grad is not None:
            # fold_head_bwd in the reduction launch of the same call: (u_full, w, b, gu_full, gw, gb, gb2)
            fu, fw, fb, gu, gw_, gb_, gb2_ = fold_grad
            fg = _lib.HeadFoldGrad(fu.data_ptr(), fw.data_ptr(), _ld(fw), _lib.ptr(fb), _lib.ptr(gu), _lib.ptr(gw_),
                                   _ld(gw_) if gw_ is not None else 0, _lib.ptr(gb_), _lib.ptr(gb2_), p, fw.shape[0],
                                   fw.shape[1], 0)
        rc = _timed("mlp_fused_bwd", meta, _lib.load().ctr_embed_mlp_head_bwd, farr, len(gather_specs), m, arr, len(layers),
                    C.byref(hg), C.byref(fg) if fg is not None else None, gx_first.data_ptr(), _ld(gx_first), ws.data_ptr(),
                    ws.numel(), _lib.ptr(zero), zero.numel() if zero is not None else 0, _lib.stream_ptr())
        _lib.check(rc, "ctr_embed_mlp_head_bwd")  # a refusal is an error here: the input columns were never written
        return grads
    rc = _timed("mlp_fused_bwd", meta, _lib.load().ctr_mlp_head_bwd, x0.data_ptr(), _ld(x0), m, arr, len(layers),
                C.byref(hg), gx_first.data_ptr(), _ld(gx_first), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    if rc in _REFUSED:
        if _profiler is not None and _profiler.records and _profiler.records[-1][0] == "mlp_fused_bwd":
            _profiler.records.pop()
        return None
    _lib.check(rc, "ctr_mlp_head_bwd")
    return grads


def mlp_bwd(acts: Sequence[torch.Tensor], layers: Sequence[Layer], gy: torch.Tensor,
            gx_first: Optional[torch.Tensor], want_gx_first: bool = True, zeros: Optional[dict] = None):
    """backward through ``mlp_fwd``; returns ([(gw, gb) per layer], gx of the
    first layer input or None)"""
    grads = [None] * len(layers)
    if _fusable(acts[0], layers):
        # whole stack in one launch: dX chain in LDS, dW tiles in registers
        for k, layer in enumerate(layers):
            if zeros is not None:
                grads[k] = (zeros[id(layer.weight)], zeros[id(layer.bias)])
            else:
                grads[k] = (torch.zeros_like(layer.weight), torch.zeros_like(layer.bias))
        x0 = _mat(acts[0], "x")
        m = x0.shape[0]
        gx = gx_first if want_gx_first else None
        if want_gx_first and gx is None:
            gx = torch.empty((m, x0.shape[1]), dtype=torch.float32, device=x0.device)
        gy = _mat(gy, "gy")
        arr = _mlp_layer_array(layers, acts[1:], grads)
        ws = _scratch(x0.device)
        dims = [(layer.weight.shape[0], layer.weight.shape[1]) for layer in layers]
        # algorithmic bytes: x and every intermediate Y read once (as the next layer's input; the
        # activation mask is folded into the dX write-back), gY read, the last Y only when it has an
        # activation to mask, gX written
        last_mask = dims[-1][0] if layers[-1].act != ACT_NONE else 0
        rc = _timed("mlp_fused_bwd", lambda: (4 * m * (dims[0][1] + sum(n for n, _ in dims[:-1]) + dims[-1][0] +
                                                       last_mask + (dims[0][1] if gx is not None else 0)),
                                              4 * m * sum(n * k for n, k in dims)),
                    _lib.load().ctr_mlp_bwd, x0.data_ptr(), _ld(x0), m, arr, len(layers), gy.data_ptr(), _ld(gy),
                    _lib.ptr(gx), _ld(gx) if gx is not None else 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        if rc not in _REFUSED:
            _lib.check(rc, "ctr_mlp_bwd")
            return grads, gx
        if _profiler is not None and _profiler.records and _profiler.records[-1][0] == "mlp_fused_bwd":
            _profiler.records.pop()  # refused: nothing ran, the per-layer path below is what gets timed
    g = gy
    for k in range(len(layers) - 1, -1, -1):
        layer = layers[k]
        xin, yout = acts[k], acts[k + 1]
        if zeros is not None:
            gw = zeros[id(layer.weight)]
            gb = zeros[id(layer.bias)] if layer.bias is not None else None
        else:
            gw = torch.zeros_like(layer.weight)
            gb = torch.zeros_like(layer.bias) if layer.bias is not None else None
        if k > 0:
            gx = torch.empty((xin.shape[0], xin.shape[1]), dtype=torch.float32, device=xin.device)
        else:
            gx = gx_first if want_gx_first else None
            if want_gx_first and gx is None:
                gx = torch.empty((xin.shape[0], xin.shape[1]), dtype=torch.float32, device=xin.device)
        linear_bwd(xin, layer.weight, yout, g, layer.act, gx, gw, gb)
        grads[k] = (gw, gb)
        g = gx
    return grads, g


def rows_sum_act_fwd(table_a, idx_a, table_b, idx_b, act: int, out, err_flag=None) -> torch.Tensor:
    """out[b] = act(table_a[idx_a[b]] + table_b[idx_b[b]]) (ctr_rows_sum_act_fwd)"""
    out = _mat(out, "out")
    _lib.require_device(table_a, table_b, idx_a, idx_b)
    batch, width = idx_a.numel(), table_a.shape[1]
    rc = _timed("rows_sum_act_fwd", lambda: (batch * (16 + 12 * width), batch * width),
                _lib.load().ctr_rows_sum_act_fwd, table_a.data_ptr(), idx_a.data_ptr(), idx_a.stride(0) if batch > 1 else 1,
                table_a.shape[0], table_b.data_ptr(), idx_b.data_ptr(), idx_b.stride(0) if batch > 1 else 1, table_b.shape[0],
                batch, width, act, out.data_ptr(), _ld(out), _lib.ptr(err_flag), _lib.stream_ptr())
    _lib.check(rc, "ctr_rows_sum_act_fwd")
    return out


def act_mask_bwd(g, y, act: int) -> None:
    """g *= act'(y) in place (ctr_act_mask_bwd)"""
    g, y = _mat(g, "g"), _mat(y, "y")
    m, n = g.shape
    rc = _timed("act_mask_bwd", lambda: (12 * m * n, m * n), _lib.load().ctr_act_mask_bwd, g.data_ptr(), _ld(g),
                y.data_ptr(), _ld(y), m, n, act, _lib.stream_ptr())
    _lib.check(rc, "ctr_act_mask_bwd")


# ---------------------------------------------------------------------------
# NeuralCF with the first tower layer on the table rows (csrc/ncf_proj.hip)
# ---------------------------------------------------------------------------
class NcfCounts:
    """the per-row sample counters of ``ctr_ncf_proj_fwd`` (one holder per model).  The C entry points want them all
    zero at a training forward and leave them zero after the backward, so in the usual forward -> backward rhythm the
    buffer is filled once, here.  ``take`` hands it out when it is known to be clean; while a forward's counts still
    wait for their backward (two graphs alive at once), or after a training forward that never got one, a later
    forward gets a freshly zeroed buffer of its own instead."""

    def __init__(self):
        self.buf, self.state = None, "clean"

    def take(self, rows: int, device):
        n = rows * _lib.CTR_NCF_PROJ_COUNT_STRIDE      # a line per counter (ctrhip.h)
        if self.buf is None or self.buf.numel() < n or self.buf.device != device:
            self.buf, self.state = torch.zeros(n, dtype=torch.int32, device=device), "clean"
        if self.state == "dirty":            # a training forward whose graph was dropped without a backward
            self.buf.zero_()
            self.state = "clean"
        if self.state == "clean":
            self.state = "busy"
            return self.buf, True
        return torch.zeros(n, dtype=torch.int32, device=device), False


class NcfProj:
    """one forward of ``ctr_ncf_proj_fwd`` and what its backward needs.  ``tables`` = (gmf_u, gmf_i, mlp_u, mlp_i),
    ``hidden`` = the four tower layers, ``proj`` = (w, b) of ``linear``, ``head`` = (w, b) of ``linear2``."""

    def __init__(self, user_idx, item_idx, tables, hidden, proj, head, err_flag, training, counts: "NcfCounts" = None):
        self.user_idx, self.item_idx = user_idx, item_idx
        self.tables, self.hidden, self.proj, self.head = tables, hidden, proj, head
        gmf_u, gmf_i, mlp_u, mlp_i = tables
        dev = gmf_u.device
        self.batch, self.nu, self.ni = user_idx.numel(), gmf_u.shape[0], gmf_i.shape[0]
        m, rows = self.batch, self.nu + self.ni
        self.ptab = torch.empty((rows, hidden[0].weight.shape[0]), dtype=torch.float32, device=dev)
        self.wfold = torch.empty(76, dtype=torch.float32, device=dev)
        # every per-sample buffer has a spare row m: lanes without a sample store there (unconditional, countable stores)
        self.ys = [torch.empty((m + 1, layer.weight.shape[0]), dtype=torch.float32, device=dev) for layer in hidden[1:]]
        self.prob_buf = torch.empty((m + 1, 1), dtype=torch.float32, device=dev)
        self.prob = self.prob_buf[:m]
        self._holder, self._owns = None, False
        self.counts = self.ranks = None
        if training:
            self._holder = counts if counts is not None else NcfCounts()
            self.counts, self._owns = self._holder.take(rows, dev)
            self.ranks = torch.empty(2 * (m + 1), dtype=torch.int32, device=dev)
        self.err_flag, self.training = err_flag, training

    def __del__(self):
        # a training forward that never saw its backward leaves the counters dirty
        if getattr(self, "_owns", False) and self._holder.state == "busy":
            self._holder.state = "dirty"

    def _desc(self):
        gmf_u, gmf_i, mlp_u, mlp_i = self.tables
        d = _lib.NcfProj()
        d.user_idx, d.user_stride = self.user_idx.data_ptr(), self.user_idx.stride(0) if self.user_idx.numel() > 1 else 1
        d.item_idx, d.item_stride = self.item_idx.data_ptr(), self.item_idx.stride(0) if self.item_idx.numel() > 1 else 1
        d.batch, d.num_users, d.num_items = self.batch, self.nu, self.ni
        d.mlp_user, d.mlp_item, d.gmf_user, d.gmf_item = (t.data_ptr() for t in (mlp_u, mlp_i, gmf_u, gmf_i))
        d.mlp_dim, d.mf_dim = mlp_u.shape[1], gmf_u.shape[1]
        for k, layer in enumerate(self.hidden):
            e = d.layers[k]
            e.w, e.b, e.n, e.k, e.act = layer.weight.data_ptr(), _lib.ptr(layer.bias), layer.weight.shape[0], layer.weight.shape[1], layer.act
            if k:
                e.y, e.ldy = self.ys[k - 1].data_ptr(), _ld(self.ys[k - 1])
        pw, pb = self.proj
        hw, hb = self.head
        d.proj_w, d.ld_proj_w, d.proj_b, d.proj_n, d.proj_k = pw.data_ptr(), _ld(pw), _lib.ptr(pb), pw.shape[0], pw.shape[1]
        d.head_w, d.head_b, d.head_act = hw.data_ptr(), _lib.ptr(hb), ACT_SIGMOID
        d.prob, d.ldprob, d.err_flag = self.prob.data_ptr(), 1, _lib.ptr(self.err_flag)
        d.ptab, d.wfold, d.counts, d.ranks = self.ptab.data_ptr(), self.wfold.data_ptr(), _lib.ptr(self.counts), _lib.ptr(self.ranks)
        d.training = 1 if self.training else 0
        return d

    @staticmethod
    def supported(tables, hidden, proj, batch) -> bool:
        gmf_u, gmf_i, mlp_u, mlp_i = tables
        rows = gmf_u.shape[0] + gmf_i.shape[0]
        dims = [tuple(layer.weight.shape) for layer in hidden]
        return (dims == [(64, 128), (32, 64), (16, 32), (8, 16)] and gmf_u.shape[1] == 64 and mlp_u.shape[1] == 64 and
                tuple(proj[0].shape) == (64, 8) and all(layer.bias is not None for layer in hidden) and
                rows <= _lib.CTR_NCF_PROJ_MAX_ROWS and batch >= 4096 and batch >= 4 * rows and 2 * batch < 2 ** 31)

    # algorithmic bytes / flops of the launches: per sample unless said otherwise
    def _meta(self, which):
        m, rows = self.batch, self.nu + self.ni
        tower = 64 * 32 + 32 * 16 + 16 * 8
        return {
            # (rows, 64) tables read, (rows, 64) projected rows written; one 64 x 64 product per row
            "ncfp_prep": lambda: (rows * 512, 2 * rows * 64 * 64),
            # ids, four 256-byte rows (cache-resident tables), saved activations + prob written; tower + head
            # (+ in training, by the rank workgroups of the same launch: ids again, two returning atomics, an 8-byte record)
            "ncfp_fwd": lambda: (m * (16 + 4 * 256 + 4 * (32 + 16 + 8) + 4 + ((16 + 8 + 8) if self.training else 0)),
                                 2 * m * (tower + 72 + 64)),
            # ids, prob, gprob, ranks, two projected rows, saved activations read; gz0 row stored twice + two records
            "ncfp_bwd": lambda: (m * (16 + 16 + 2 * 256 + 4 * (32 + 16 + 8) + 2 * 256 + 32), 4 * m * tower + 2 * m * 8),
            # both buckets (row + record) and one partner row per slot read, (rows, 128) sums added
            "ncfp_segsum": lambda: (2 * m * (256 + 16 + 256) + rows * 512, 2 * 2 * m * 64 * 2),
            # sums + tables read, table gradients read-modify-written; three 64 x 64 products per row
            "ncfp_finish": lambda: (rows * (512 + 512 + 4 * 256), 3 * 2 * rows * 64 * 64),
        }[which]

    def forward(self) -> torch.Tensor:
        d = self._desc()
        fn = _lib.load().ctr_ncf_proj_fwd
        if _profiler is None:
            rc = fn(C.byref(d), _lib.stream_ptr())
        else:   # one call per launch, so that each gets its own event pair
            rc = 0
            for phase, label in ((1, "ncfp_prep"), (2, "ncfp_fwd")):
                d.phases = phase
                rc = rc or _timed(label, self._meta(label), fn, C.byref(d), _lib.stream_ptr())
        if rc in _REFUSED:
            return None
        _lib.check(rc, "ctr_ncf_proj_fwd")
        return self.prob

    def backward(self, gprob: torch.Tensor, grads: dict, zero: Optional[torch.Tensor]) -> None:
        """``grads[id(param)]``: where each parameter's gradient accumulates; ``zero``: the flat buffer behind them,
        cleared by the call's first launch"""
        if getattr(self, "_spent", False):
            # the call's last launch clears the sample counters for the next forward: a second backward over the same
            # forward would bucket with empty counters
            raise RuntimeError("NeuralCF backward consumes what its forward saved: run the forward again before a second "
                               "backward (retain_graph is not supported)")
        self._spent = True
        gmf_u, gmf_i, mlp_u, mlp_i = self.tables
        d = self._desc()
        g = _lib.NcfProjGrad()
        gprob = gprob.reshape(-1)
        g.gprob, g.ldgprob = gprob.data_ptr(), gprob.stride(0) if gprob.numel() > 1 else 1
        for k, layer in enumerate(self.hidden):
            g.layers[k].gw, g.layers[k].gb = grads[id(layer.weight)].data_ptr(), grads[id(layer.bias)].data_ptr()
        g.g_mlp_user, g.g_mlp_item, g.g_gmf_user, g.g_gmf_item = (grads[id(t)].data_ptr() for t in (mlp_u, mlp_i, gmf_u, gmf_i))
        pw, pb = self.proj
        hw, hb = self.head
        g.g_proj_w, g.ld_g_proj_w, g.g_proj_b = grads[id(pw)].data_ptr(), _ld(grads[id(pw)]), _lib.ptr(grads.get(id(pb)))
        g.g_head_w, g.g_head_b = grads[id(hw)].data_ptr(), _lib.ptr(grads.get(id(hb)))
        need = C.c_int64(0)
        _lib.check(_lib.load().ctr_ncf_proj_workspace_floats(self.batch, self.nu, self.ni, C.byref(need)), "workspace")
        ws = _scratch(gmf_u.device) if need.value <= SCRATCH_FLOATS else torch.empty(need.value, dtype=torch.float32,
                                                                                      device=gmf_u.device)
        g.workspace, g.workspace_floats = ws.data_ptr(), ws.numel()
        g.zero_buf, g.zero_floats = _lib.ptr(zero), zero.numel() if zero is not None else 0
        fn = _lib.load().ctr_ncf_proj_bwd
        if _profiler is None:
            rc = fn(C.byref(d), C.byref(g), _lib.stream_ptr())
        else:
            rc = 0
            for phase, label in ((1, "ncfp_bwd"), (2, "ncfp_segsum"), (4, "ncfp_finish")):
                g.phases = phase
                rc = rc or _timed(label, self._meta(label), fn, C.byref(d), C.byref(g), _lib.stream_ptr())
        _lib.check(rc, "ctr_ncf_proj_bwd")
        if self._owns:
            self._holder.state = "clean"     # (the call's last launch zeroed the counters)
        self._owns = False


# ---------------------------------------------------------------------------
# feature interactions
# ---------------------------------------------------------------------------
USER_COL, ITEM_COL, DENSE_COL0, NUM_DENSE = 0, 1, 2, 43  # the (B,45) layout of data/reader.py:98-112


def allpairs_fwd(emb: torch.Tensor, nvec: int, dim: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    emb = _mat(emb, "emb")
    batch = emb.shape[0]
    npairs = nvec * (nvec - 1) // 2
    if out is None:
        out = torch.empty((batch, npairs), dtype=torch.float32, device=emb.device)
    many = nvec > 8 and dim in (8, 16, 32, 64)   # lane-group kernel (fields.hip) instead of the six-field LDS tiles
    rc = _timed("allpairs_fwd", lambda: (4 * batch * (nvec * dim + npairs), 2 * batch * npairs * dim),
                _lib.load().ctr_fields_pairs_fwd if many else _lib.load().ctr_allpairs_fwd, emb.data_ptr(), _ld(emb),
                batch, nvec, dim, out.data_ptr(), _ld(out), _lib.stream_ptr())
    if many and rc in _REFUSED:
        rc = _lib.load().ctr_allpairs_fwd(emb.data_ptr(), _ld(emb), batch, nvec, dim, out.data_ptr(), _ld(out),
                                          _lib.stream_ptr())
    _lib.check(rc, "ctr_allpairs_fwd")
    return out


def allpairs_bwd(emb, nvec, dim, gp, gemb, accumulate: bool) -> None:
    emb, gp, gemb = _mat(emb, "emb"), _mat(gp, "gp"), _mat(gemb, "gemb")
    batch = emb.shape[0]
    npairs = nvec * (nvec - 1) // 2
    many = nvec > 8 and dim in (8, 16, 32, 64)
    rc = _timed("allpairs_bwd", lambda: (4 * batch * ((2 + int(accumulate)) * nvec * dim + npairs),
                                         2 * batch * npairs * dim * 2),
                _lib.load().ctr_fields_pairs_bwd if many else _lib.load().ctr_allpairs_bwd, emb.data_ptr(), _ld(emb),
                batch, nvec, dim, gp.data_ptr(), _ld(gp), gemb.data_ptr(), _ld(gemb), int(accumulate), _lib.stream_ptr())
    if many and rc in _REFUSED:
        rc = _lib.load().ctr_allpairs_bwd(emb.data_ptr(), _ld(emb), batch, nvec, dim, gp.data_ptr(), _ld(gp),
                                          gemb.data_ptr(), _ld(gemb), int(accumulate), _lib.stream_ptr())
    _lib.check(rc, "ctr_allpairs_bwd")


def _wide_args(x, user1, item1, w, b):
    x = _mat(x, "x")
    _lib.require_device(user1, item1, w, b)
    return (x.data_ptr(), _ld(x), USER_COL, ITEM_COL, DENSE_COL0, w.shape[1], user1.data_ptr(), user1.shape[0],
            item1.data_ptr(), item1.shape[0], w.data_ptr(), b.data_ptr())


def fm_wide_fwd(emb, nvec, dim, x, user1, item1, wide_w, wide_b, out, err_flag=None) -> None:
    emb, out = _mat(emb, "emb"), _mat(out, "out")
    batch = emb.shape[0]
    rc = _timed("fm_wide_fwd", lambda: (4 * batch * (nvec * dim + 45 + 3), 4 * batch * nvec * dim),
                _lib.load().ctr_fm_wide_fwd, emb.data_ptr(), _ld(emb), batch, nvec, dim,
                *_wide_args(x, user1, item1, wide_w, wide_b), out.data_ptr(), _ld(out), _lib.ptr(err_flag),
                _lib.stream_ptr())
    _lib.check(rc, "ctr_fm_wide_fwd")


ROWS1_IN_LDS = os.environ.get("CTR_ROWS1_LDS", "1") != "0"   # 0: the interaction kernels' own per-sample atomics (A/B)


def _rows1_small(x, guser1, gitem1) -> bool:
    """the (V, 1) first-order gradients of SMALL tables leave ``fm_wide_bwd``: per-sample 4-byte atomics on a few dozen cache
    lines queue up at the memory side (csrc/rows_sum.hip, ctr_rows1_scatter).  Measured (profiles/r03_rows1_ab.txt): logistic
    regression 66.5 -> 58.0 us per step, Wide & Deep / AFM unchanged; under ``ffm_fused_bwd`` the atomics only cost 5 us and
    the extra launch 10, so FFM keeps them in the kernel."""
    if not ROWS1_IN_LDS or x.shape[0] < 4096 or (guser1 is None and gitem1 is None):
        return False
    rows = (guser1.shape[0] if guser1 is not None else 1) + (gitem1.shape[0] if gitem1 is not None else 1)
    return rows <= _lib.CTR_ROWS1_MAX_ROWS


def rows1_scatter(x, g, prob, guser1, gitem1) -> None:
    """guser1[x[:, 0]] += v, gitem1[x[:, 1]] += v, v = g (* p (1 - p) when ``prob`` is given)"""
    x, g = _mat(x, "x"), _mat(g, "g")
    batch = x.shape[0]
    nu = guser1.shape[0] if guser1 is not None else 1
    ni = gitem1.shape[0] if gitem1 is not None else 1
    rc = _timed("rows1_scatter", lambda: (batch * (8 + 4 + (4 if prob is not None else 0)) + 8 * (nu + ni), 2 * batch),
                _lib.load().ctr_rows1_scatter, x.data_ptr(), _ld(x), USER_COL, ITEM_COL, g.data_ptr(), _ld(g),
                _lib.ptr(prob), _ld(prob) if prob is not None else 0, batch, _lib.ptr(guser1), nu, _lib.ptr(gitem1), ni,
                _lib.stream_ptr())
    _lib.check(rc, "ctr_rows1_scatter")


def fm_wide_bwd(emb, nvec, dim, x, user1, item1, wide_w, wide_b, gout, guser1, gitem1, gwide_w, gwide_b, gemb,
                accumulate: bool) -> None:
    emb, gout = _mat(emb, "emb"), _mat(gout, "gout")
    batch = emb.shape[0]
    ws = _scratch(emb.device)
    if _rows1_small(x, guser1, gitem1):
        rows1_scatter(x, gout, None, guser1, gitem1)
        guser1 = gitem1 = None
    rc = _timed("fm_wide_bwd", lambda: (4 * batch * ((2 + int(accumulate)) * nvec * dim + 45 + 5),
                                        4 * batch * nvec * dim),
                _lib.load().ctr_fm_wide_bwd, emb.data_ptr(), _ld(emb), batch, nvec, dim,
                *_wide_args(x, user1, item1, wide_w, wide_b), gout.data_ptr(), _ld(gout), _lib.ptr(guser1),
                _lib.ptr(gitem1), _lib.ptr(gwide_w), _lib.ptr(gwide_b), _lib.ptr(gemb),
                _ld(gemb) if gemb is not None else 0, int(accumulate), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "ctr_fm_wide_bwd")


def _ptr_array(tensors):
    """host array of device pointers (NULL for None); None -> NULL array"""
    if tensors is None:
        return None
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def fields_fm_fwd(idx, tables, first, bias, emb, fm, err_flag=None) -> None:
    """N id fields: gather into ``emb`` (B, F*E) + FM second order + first-order sum into ``fm`` (B,1) -- one launch"""
    _lib.require_device(idx, emb, fm, *tables)
    emb, fm = _mat(emb, "emb"), _mat(fm, "fm")
    if idx.dtype != torch.int64 or idx.dim() != 2 or idx.stride(1) != 1:
        raise ValueError("fields_fm_fwd: idx must be a (B,F) int64 matrix with unit inner stride")
    batch, nf = idx.shape
    dim = tables[0].shape[1]
    vocabs = (C.c_int64 * nf)(*[t.shape[0] for t in tables])
    n1 = 0 if first is None else sum(1 for t in first if t is not None)
    rc = _timed("fields_fm_fwd", lambda: (batch * (nf * (dim * 8 + 8) + 4 * n1 + 4), 4 * batch * nf * dim),
                _lib.load().ctr_fields_fm_fwd, idx.data_ptr(), idx.stride(0), batch, nf, dim, _ptr_array(tables), vocabs,
                _ptr_array(first), _lib.ptr(bias), emb.data_ptr(), _ld(emb), fm.data_ptr(), _ld(fm),
                _lib.ptr(err_flag), _lib.stream_ptr())
    _lib.check(rc, "ctr_fields_fm_fwd")


def fields_fm_bwd(idx, vocabs, dim, emb, gdeep, gfm, gtables, gfirst, gbias) -> None:
    """backward of ``fields_fm_fwd``: scatter-add ``gdeep + gfm * (S - v)`` into the dense table gradients"""
    emb = _mat(emb, "emb")
    batch, nf = idx.shape
    gdeep = _mat(gdeep, "gdeep") if gdeep is not None else None
    gfm = _mat(gfm, "gfm") if gfm is not None else None
    ws = _scratch(emb.device) if gbias is not None else None
    n1 = 0 if gfirst is None else sum(1 for t in gfirst if t is not None)
    rc = _timed("fields_fm_bwd", lambda: (batch * (nf * (dim * 4 * (2 + (gdeep is not None)) + 8) + 8 * n1 + 4),
                                          3 * batch * nf * dim),
                _lib.load().ctr_fields_fm_bwd, idx.data_ptr(), idx.stride(0), batch, nf, dim,
                (C.c_int64 * nf)(*vocabs), emb.data_ptr(), _ld(emb), _lib.ptr(gdeep),
                _ld(gdeep) if gdeep is not None else 0, _lib.ptr(gfm), _ld(gfm) if gfm is not None else 0,
                _ptr_array(gtables), _ptr_array(gfirst), _lib.ptr(gbias), _lib.ptr(ws),
                ws.numel() if ws is not None else 0, _lib.stream_ptr())
    _lib.check(rc, "ctr_fields_fm_bwd")


def _pair_array(pairs):
    flat = [v for p in pairs for v in p]
    return (C.c_int32 * len(flat))(*flat)


def ffm_fused_fwd(x, tables, user1, item1, lin_w, lin_b, emb, prob, err_flag=None) -> None:
    """FFM forward in one launch: the 12 field-aware vectors into ``emb`` (B, 12k), the probability into ``prob``"""
    x, emb, prob = _mat(x, "x"), _mat(emb, "emb"), _mat(prob, "prob")
    _lib.require_device(user1, item1, lin_w, lin_b, *tables)
    batch, dim = x.shape[0], tables[0].shape[1]
    rc = _timed("ffm_fused_fwd", lambda: (4 * batch * (45 + 4 * dim + 2 + 12 * dim + 1), 2 * batch * (15 + 86) * dim),
                _lib.load().ctr_ffm_fused_fwd, x.data_ptr(), _ld(x), batch, dim, _ptr_array(tables), user1.shape[0],
                item1.shape[0], user1.data_ptr(), item1.data_ptr(), lin_w.data_ptr(), lin_b.data_ptr(), emb.data_ptr(),
                _ld(emb), prob.data_ptr(), _ld(prob), _lib.ptr(err_flag), _lib.stream_ptr())
    _lib.check(rc, "ctr_ffm_fused_fwd")


def ffm_fused_bwd(x, emb, num_users, num_items, lin_w, prob, gprob, guser1, gitem1, glin_w, glin_b, gemb) -> None:
    """backward of ``ffm_fused_fwd``'s head + dots: small gradients directly, ``gemb`` for the embedding backward"""
    x, emb, prob, gprob, gemb = _mat(x, "x"), _mat(emb, "emb"), _mat(prob, "prob"), _mat(gprob, "gprob"), _mat(gemb, "gemb")
    batch, dim = x.shape[0], emb.shape[1] // 12
    ws = _scratch(emb.device)
    rc = _timed("ffm_fused_bwd", lambda: (4 * batch * (45 + 24 * dim + 4), 4 * batch * 15 * dim),
                _lib.load().ctr_ffm_fused_bwd, x.data_ptr(), _ld(x), batch, dim, emb.data_ptr(), _ld(emb), num_users,
                num_items, lin_w.data_ptr(), prob.data_ptr(), _ld(prob), gprob.data_ptr(), _ld(gprob), _lib.ptr(guser1),
                _lib.ptr(gitem1), _lib.ptr(glin_w), _lib.ptr(glin_b), gemb.data_ptr(), _ld(gemb), ws.data_ptr(),
                ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "ctr_ffm_fused_bwd")


def ffm_head_fwd(emb, nvec, dim, pairs, x, user1, item1, lin_w, lin_b, prob, err_flag=None) -> None:
    emb, prob = _mat(emb, "emb"), _mat(prob, "prob")
    batch = emb.shape[0]
    rc = _timed("ffm_head_fwd", lambda: (4 * batch * (nvec * dim + 45 + 3), 2 * batch * len(pairs) * dim),
                _lib.load().ctr_ffm_head_fwd, emb.data_ptr(), _ld(emb), batch, nvec, dim, _pair_array(pairs),
                len(pairs), *_wide_args(x, user1, item1, lin_w, lin_b), prob.data_ptr(), _ld(prob),
                _lib.ptr(err_flag), _lib.stream_ptr())
    _lib.check(rc, "ctr_ffm_head_fwd")


def ffm_head_bwd(emb, nvec, dim, pairs, x, user1, item1, lin_w, lin_b, prob, gprob, guser1, gitem1, glin_w, glin_b,
                 gemb) -> None:
    emb, prob, gprob, gemb = _mat(emb, "emb"), _mat(prob, "prob"), _mat(gprob, "gprob"), _mat(gemb, "gemb")
    batch = emb.shape[0]
    ws = _scratch(emb.device)
    rc = _timed("ffm_head_bwd", lambda: (4 * batch * (2 * nvec * dim + 45 + 6), 4 * batch * len(pairs) * dim),
                _lib.load().ctr_ffm_head_bwd, emb.data_ptr(), _ld(emb), batch, nvec, dim, _pair_array(pairs),
                len(pairs), *_wide_args(x, user1, item1, lin_w, lin_b), prob.data_ptr(), _ld(prob),
                gprob.data_ptr(), _ld(gprob), _lib.ptr(guser1), _lib.ptr(gitem1), _lib.ptr(glin_w),
                _lib.ptr(glin_b), gemb.data_ptr(), _ld(gemb), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "ctr_ffm_head_bwd")


def act_bwd(y, gy, act: int, out, accumulate: bool) -> None:
    y, gy, out = _mat(y, "y"), _mat(gy, "gy"), _mat(out, "out")
    m, n = y.shape
    rc = _timed(f"act_bwd[{m}x{n}]", lambda: (4 * m * n * (3 + int(accumulate)), m * n),
                _lib.load().ctr_act_bwd, y.data_ptr(), _ld(y), gy.data_ptr(), _ld(gy), out.data_ptr(), _ld(out),
                m, n, act, int(accumulate), _lib.stream_ptr())
    _lib.check(rc, "ctr_act_bwd")


def biinteract_fwd(emb, nvec, dim, out) -> torch.Tensor:
    """NFM bi-interaction: out[b, e] = sum_{i<j} v_i[e] * v_j[e]"""
    emb, out = _mat(emb, "emb"), _mat(out, "out")
    batch = emb.shape[0]
    rc = _timed("biinteract_fwd", lambda: (4 * batch * (nvec + 1) * dim, batch * dim * nvec * (nvec - 1)),
                _lib.load().ctr_biinteract_fwd, emb.data_ptr(), _ld(emb), batch, nvec, dim, out.data_ptr(), _ld(out),
                _lib.stream_ptr())
    _lib.check(rc, "ctr_biinteract_fwd")
    return out


def biinteract_bwd(emb, nvec, dim, gout, gemb, accumulate: bool) -> None:
    emb, gout, gemb = _mat(emb, "emb"), _mat(gout, "gout"), _mat(gemb, "gemb")
    batch = emb.shape[0]
    rc = _timed("biinteract_bwd", lambda: (4 * batch * ((2 + int(accumulate)) * nvec + 1) * dim, 3 * batch * dim * nvec),
                _lib.load().ctr_biinteract_bwd, emb.data_ptr(), _ld(emb), batch, nvec, dim, gout.data_ptr(), _ld(gout),
                gemb.data_ptr(), _ld(gemb), int(accumulate), _lib.stream_ptr())
    _lib.check(rc, "ctr_biinteract_bwd")


def pairprod_fwd(emb, nvec, dim, out) -> torch.Tensor:
    """AFM: out[(b*np + idx(i,j)), :] = v_i * v_j for every pair i < j"""
    emb, out = _mat(emb, "emb"), _mat(out, "out")
    batch, npairs = emb.shape[0], nvec * (nvec - 1) // 2
    rc = _timed("pairprod_fwd", lambda: (4 * batch * (nvec + npairs) * dim, batch * npairs * dim),
                _lib.load().ctr_pairprod_fwd, emb.data_ptr(), _ld(emb), batch, nvec, dim, out.data_ptr(), _ld(out),
                _lib.stream_ptr())
    _lib.check(rc, "ctr_pairprod_fwd")
    return out


def pairprod_bwd(emb, nvec, dim, gp, attn, gpool, gemb, accumulate: bool) -> None:
    emb, gp, gemb = _mat(emb, "emb"), _mat(gp, "gp"), _mat(gemb, "gemb")
    if gpool is not None:
        gpool = _mat(gpool, "gpool")
    batch, npairs = emb.shape[0], nvec * (nvec - 1) // 2
    rc = _timed("pairprod_bwd", lambda: (4 * batch * (2 * nvec + npairs + 1) * dim, 4 * batch * npairs * dim),
                _lib.load().ctr_pairprod_bwd, emb.data_ptr(), _ld(emb), batch, nvec, dim, gp.data_ptr(), _ld(gp),
                _lib.ptr(attn), _lib.ptr(gpool), _ld(gpool) if gpool is not None else 0, gemb.data_ptr(), _ld(gemb),
                int(accumulate), _lib.stream_ptr())
    _lib.check(rc, "ctr_pairprod_bwd")


def cross_fwd(x0, u, xl, bias, out) -> torch.Tensor:
    """Deep & Cross combine: out = x0 * u + bias + xl"""
    x0, u, xl, out = _mat(x0, "x0"), _mat(u, "u"), _mat(xl, "xl"), _mat(out, "out")
    m, d = x0.shape
    rc = _timed(f"cross_fwd[{m}x{d}]", lambda: (16 * m * d, 3 * m * d),
                _lib.load().ctr_cross_fwd, x0.data_ptr(), _ld(x0), u.data_ptr(), _ld(u), xl.data_ptr(), _ld(xl),
                bias.data_ptr(), out.data_ptr(), _ld(out), m, d, _lib.stream_ptr())
    _lib.check(rc, "ctr_cross_fwd")
    return out


def cross_bwd(x0, u, gy, gu, gx0, gbias) -> None:
    """gu = gy * x0; gx0 += gy * u; gbias += gy.sum(0)"""
    x0, u, gy, gu, gx0 = _mat(x0, "x0"), _mat(u, "u"), _mat(gy, "gy"), _mat(gu, "gu"), _mat(gx0, "gx0")
    m, d = x0.shape
    ws = _scratch(x0.device)
    rc = _timed(f"cross_bwd[{m}x{d}]", lambda: (24 * m * d, 3 * m * d),
                _lib.load().ctr_cross_bwd, x0.data_ptr(), _ld(x0), u.data_ptr(), _ld(u), gy.data_ptr(), _ld(gy),
                gu.data_ptr(), _ld(gu), gx0.data_ptr(), _ld(gx0), gbias.data_ptr(), m, d, ws.data_ptr(), ws.numel(),
                _lib.stream_ptr())
    _lib.check(rc, "ctr_cross_bwd")


# ---------------------------------------------------------------------------
# DIN / DIEN sequence attention and GRU
# ---------------------------------------------------------------------------
def linear_group_fwd(x, w, bias, res, group: int, act: int = ACT_NONE, out=None, sign_bits=None) -> torch.Tensor:
    """out = act(x @ w.T + bias + res[row // group]): one residual row per ``group`` consecutive rows (DIN: the
    per-sample term of the first attention layer on the E-wide operand).  ``sign_bits``: an int32 (m, n/32) tensor
    that receives bit (j & 31) of word j // 32 = (out[i, j] > 0) -- the ReLU derivative for ``linear_dx_masked``."""
    x, w, res = _mat(x, "x"), _mat(w, "w"), _mat(res, "res")
    m, k = x.shape
    n = w.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    out = _mat(out, "out")
    if sign_bits is not None:
        _lib.require_device(sign_bits)
        if sign_bits.dtype != torch.int32 or sign_bits.dim() != 2 or sign_bits.stride(1) != 1 or n % 32 or \
                sign_bits.shape[0] != m or sign_bits.shape[1] < n // 32:
            raise ValueError("linear_group_fwd: sign_bits must be an int32 (m, n/32) tensor with n % 32 == 0")
    rc = _timed(f"linear_group_fwd[{m}x{n}x{k}]",
                lambda: (4 * (m * k + n * k + m * n + (m // group) * n) + (m * n // 8 if sign_bits is not None else 0),
                         2 * m * n * k),
                _lib.load().ctr_linear_group_fwd, x.data_ptr(), _ld(x), w.data_ptr(), _ld(w), _lib.ptr(bias),
                res.data_ptr(), _ld(res), group, out.data_ptr(), _ld(out), _lib.ptr(sign_bits),
                sign_bits.stride(0) if sign_bits is not None else 0, m, n, k, act, _lib.stream_ptr())
    _lib.check(rc, "ctr_linear_group_fwd")
    return out


def linear_dx_masked(w, y, gy, act: int, xin, act_in: int, gx, gsum=None, group: int = 0, sign_bits=None) -> None:
    """gx = ((gy * act'(y)) @ w) * act_in'(xin);  gsum[row // group] += gx[row] (optional).  With ``sign_bits`` (what
    ``linear_group_fwd`` wrote for xin, act_in = ReLU) xin is not read and may be None."""
    w, gy, gx = _mat(w, "w"), _mat(gy, "gy"), _mat(gx, "gx")
    m, n = gy.shape
    k = w.shape[1]
    if sign_bits is not None:
        _lib.require_device(sign_bits)
        if sign_bits.dtype != torch.int32 or sign_bits.dim() != 2 or sign_bits.stride(1) != 1 or \
                sign_bits.shape[0] != m or sign_bits.shape[1] * 32 < k:
            raise ValueError("linear_dx_masked: sign_bits must be an int32 (m, k/32) tensor")
        xin = None
    mask_bytes = (m * k // 8) if sign_bits is not None else (4 * m * k if act_in != ACT_NONE else 0)
    rc = _timed(f"linear_dx_masked[{m}x{n}x{k}]",
                lambda: (4 * (m * n * (2 if act != ACT_NONE else 1) + n * k + m * k) + mask_bytes, 2 * m * n * k),
                _lib.load().ctr_linear_dx_masked, w.data_ptr(), _ld(w), _lib.ptr(y), _ld(y) if y is not None else 0,
                gy.data_ptr(), _ld(gy), act, _lib.ptr(xin), _ld(xin) if xin is not None else 0, act_in,
                _lib.ptr(sign_bits), sign_bits.stride(0) if sign_bits is not None else 0, gx.data_ptr(),
                _ld(gx), _lib.ptr(gsum), _ld(gsum) if gsum is not None else 0, group, m, n, k, _lib.stream_ptr())
    _lib.check(rc, "ctr_linear_dx_masked")


def linear_fwd_dot(x, w, b, act: int, u, c):
    """(y, out) with y = act(x @ w.T + b) and out = y @ u.T + c for a single-unit ``u`` (1, n): the second layer is
    formed in the first one's epilogue (n <= 128), so y is not re-read for it"""
    x, w, u = _mat(x, "x"), _mat(w, "w"), _mat(u, "u")
    m, k = x.shape
    n = w.shape[0]
    if u.shape != (1, n):
        raise ValueError(f"linear_fwd_dot: u must be (1, {n})")
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    out = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    rc = _timed(f"linear_fwd_dot[{m}x{n}x{k}]", lambda: (4 * (m * k + n * k + n + m * n + m), 2 * m * n * (k + 1)),
                _lib.load().ctr_linear_fwd_dot, x.data_ptr(), _ld(x), w.data_ptr(), _ld(w), _lib.ptr(b), y.data_ptr(),
                _ld(y), u.data_ptr(), _lib.ptr(c), out.data_ptr(), 1, m, n, k, act, _lib.stream_ptr())
    _lib.check(rc, "ctr_linear_fwd_dot")
    return y, out


def linear_dx_scatter(w, gy, idx, attn, gpool, group: int, gtable) -> None:
    """gtable[idx[i]] += gy[i] @ w + attn[i] * gpool[i // group]: the input gradient of a layer whose input rows were
    gathered from ``gtable``'s table, added where the rows came from without storing the (m, k) gradient (DIN)"""
    w, gy, gpool, gtable = _mat(w, "w"), _mat(gy, "gy"), _mat(gpool, "gpool"), _mat(gtable, "gtable")
    _lib.require_device(idx, attn)
    m, n = gy.shape
    k = w.shape[1]
    if idx.dtype != torch.int64 or not idx.is_contiguous() or idx.numel() != m or attn.numel() != m or \
            not attn.is_contiguous() or attn.dtype != torch.float32:
        raise ValueError("linear_dx_scatter: idx (int64) and attn (float32) must be contiguous with one entry per row")
    if gtable.shape[1] != k or not gtable.is_contiguous():
        raise ValueError("linear_dx_scatter: gtable must be a contiguous (vocab, k) tensor")
    rc = _timed(f"linear_dx_scatter[{m}x{n}x{k}]", lambda: (4 * (m * n + n * k + 3 * m + 3 * m * k), 2 * m * n * k),
                _lib.load().ctr_linear_dx_scatter, w.data_ptr(), _ld(w), gy.data_ptr(), _ld(gy), idx.data_ptr(),
                attn.data_ptr(), gpool.data_ptr(), _ld(gpool), group, gtable.data_ptr(), gtable.shape[0], m, n, k,
                _lib.stream_ptr())
    _lib.check(rc, "ctr_linear_dx_scatter")


def linear_n1_bwd_masked(x, w, gy, act_in: int, gx, gw=None, gb=None) -> None:
    """backward of the single-unit layer y = x @ w.T + b whose input x is the previous layer's activation output:
    gx = (gy @ w) * act_in'(x) (``gx`` may be ``x`` itself), gw += gy.T @ x, gb += gy.sum()"""
    x, w, gy, gx = _mat(x, "x"), _mat(w, "w"), _mat(gy, "gy"), _mat(gx, "gx")
    m, k = x.shape
    if w.shape[0] != 1 or gy.shape != (m, 1) or gx.shape != (m, k):
        raise ValueError("linear_n1_bwd_masked: w must be (1, k), gy (m, 1), gx (m, k)")
    ws = _scratch(x.device) if gw is not None else None
    rc = _timed(f"linear_n1_bwd_masked[{m}x1x{k}]", lambda: (4 * (2 * m * k + m + 2 * k), 4 * m * k),
                _lib.load().ctr_linear_n1_bwd_masked, x.data_ptr(), _ld(x), w.data_ptr(), gy.data_ptr(), _ld(gy), act_in,
                gx.data_ptr(), _ld(gx), _lib.ptr(gw), _lib.ptr(gb), m, k, _lib.ptr(ws),
                ws.numel() if ws is not None else 0, _lib.stream_ptr())
    _lib.check(rc, "ctr_linear_n1_bwd_masked")


def din_scatter_bwd(hist, vocab, dim, gh, attn, gpool, summed: bool, gtable) -> None:
    """history positions' rows of the table gradient: gh + attn * gpool, scatter-added by hist ids"""
    gh, gpool = _mat(gh, "gh"), _mat(gpool, "gpool")
    batch, length = hist.shape
    rc = _timed("din_scatter_bwd", lambda: (batch * length * (8 + 4 + 4 * dim * (3 if summed else 4)), 0),
                _lib.load().ctr_din_scatter_bwd, hist.data_ptr(), vocab, batch, length, dim, gh.data_ptr(), _ld(gh),
                attn.data_ptr(), gpool.data_ptr(), _ld(gpool), int(summed), gtable.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "ctr_din_scatter_bwd")


def din_concat_fwd(table, hist, target, c, tvec, err_flag=None, pair: bool = False, h_only: bool = False) -> None:
    """attention operand per position: [h, h-t, t] (the reference's cat), ``pair``: [h, t], ``h_only``: [h]"""
    _lib.require_device(table, hist, target)
    c = _mat(c, "c")
    batch, length = hist.shape
    dim = table.shape[1]
    width = 1 if h_only else (2 if pair else 3)
    rc = _timed("din_concat_fwd", lambda: (batch * length * (dim * 4 + 8 + 4 * width * dim) + batch * (8 + 8 * dim), 0),
                _lib.load().ctr_din_concat_fwd, table.data_ptr(), table.shape[0], dim, hist.data_ptr(),
                target.data_ptr(), batch, length, c.data_ptr(), _ld(c), _lib.ptr(tvec),
                _ld(tvec) if tvec is not None else 0,
                _lib.DIN_H if h_only else (_lib.DIN_PAIR if pair else _lib.DIN_TRIPLE),
                _lib.ptr(err_flag), _lib.stream_ptr())
    _lib.check(rc, "ctr_din_concat_fwd")


def din_pool_fwd(score, hsrc, batch, length, dim, attn, out, summed: bool) -> None:
    hsrc, out = _mat(hsrc, "hsrc"), _mat(out, "out")
    rc = _timed("din_pool_fwd", lambda: (4 * batch * length * (2 + dim * (1 if summed else 2)), 2 * batch * length * dim),
                _lib.load().ctr_din_pool_fwd, score.data_ptr(), hsrc.data_ptr(), _ld(hsrc), batch, length, dim,
                attn.data_ptr(), out.data_ptr(), _ld(out), int(summed), _lib.stream_ptr())
    _lib.check(rc, "ctr_din_pool_fwd")


def din_pool_bwd(attn, hsrc, batch, length, dim, gout, summed: bool, gscore) -> None:
    hsrc, gout = _mat(hsrc, "hsrc"), _mat(gout, "gout")
    rc = _timed("din_pool_bwd", lambda: (4 * batch * length * (2 + dim * (1 if summed else 2)), 2 * batch * length * dim),
                _lib.load().ctr_din_pool_bwd, attn.data_ptr(), hsrc.data_ptr(), _ld(hsrc), batch, length, dim,
                gout.data_ptr(), _ld(gout), int(summed), gscore.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "ctr_din_pool_bwd")


def din_concat_bwd(hist, target, vocab, dim, gc, attn, gout, summed: bool, gt_extra, gtable,
                   pair: bool = False) -> None:
    gc, gout = _mat(gc, "gc"), _mat(gout, "gout")
    batch, length = hist.shape
    width = 2 if pair else 3
    rc = _timed("din_concat_bwd", lambda: (batch * length * (8 + 4 + 4 * width * dim + 8 * dim) + batch * 16 * dim, 0),
                _lib.load().ctr_din_concat_bwd, hist.data_ptr(), target.data_ptr(), vocab, batch, length, dim,
                gc.data_ptr(), _ld(gc), attn.data_ptr(), gout.data_ptr(), _ld(gout), int(summed),
                _lib.ptr(gt_extra), _ld(gt_extra) if gt_extra is not None else 0,
                _lib.DIN_PAIR if pair else _lib.DIN_TRIPLE, gtable.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "ctr_din_concat_bwd")


def gru_fwd(gi, w_hh, b_hh, batch, length, dim, hbuf, last) -> None:
    gi = _mat(gi, "gi")
    rc = _timed("gru_fwd", lambda: (4 * batch * length * 4 * dim, 6 * batch * length * dim * dim),
                _lib.load().ctr_gru_fwd, gi.data_ptr(), _ld(gi), w_hh.data_ptr(), b_hh.data_ptr(), batch, length, dim,
                hbuf.data_ptr(), _lib.ptr(last), _ld(last) if last is not None else 0, _lib.stream_ptr())
    _lib.check(rc, "ctr_gru_fwd")


def gru_fused_fwd(x, w_ih, b_ih, w_hh, b_hh, batch, length, dim, hbuf, last) -> bool:
    """the GRU with its input projection inside (x rows instead of gi).  False: shape not taken, nothing ran --
    form gi with ``linear_fwd`` and call ``gru_fwd``."""
    x = _mat(x, "x")
    rc = _timed("gru_fused_fwd", lambda: (4 * batch * length * 2 * dim, 12 * batch * length * dim * dim),
                _lib.load().ctr_gru_fused_fwd, x.data_ptr(), _ld(x), w_ih.data_ptr(), b_ih.data_ptr(), w_hh.data_ptr(),
                b_hh.data_ptr(), batch, length, dim, hbuf.data_ptr(), _lib.ptr(last),
                _ld(last) if last is not None else 0, _lib.stream_ptr())
    if rc in _REFUSED:
        if _profiler is not None and _profiler.records and _profiler.records[-1][0] == "gru_fused_fwd":
            _profiler.records.pop()
        return False
    _lib.check(rc, "ctr_gru_fused_fwd")
    return True


def gru_fused_bwd(x, w_ih, b_ih, w_hh, b_hh, hbuf, batch, length, dim, glast, gx, gw_ih, gb_ih, gw_hh, gb_hh) -> None:
    """backward of ``gru_fused_fwd``: gx = dgi W_ih and the four parameter gradients (accumulated), nothing of size
    3*dim per step stored"""
    x, glast, gx = _mat(x, "x"), _mat(glast, "glast"), _mat(gx, "gx")
    ws = _scratch(x.device)
    rc = _timed("gru_fused_bwd", lambda: (4 * batch * length * 3 * dim, 36 * batch * length * dim * dim),
                _lib.load().ctr_gru_fused_bwd, x.data_ptr(), _ld(x), w_ih.data_ptr(), b_ih.data_ptr(), w_hh.data_ptr(),
                b_hh.data_ptr(), hbuf.data_ptr(), batch, length, dim, glast.data_ptr(), _ld(glast), gx.data_ptr(),
                _ld(gx), gw_ih.data_ptr(), gb_ih.data_ptr(), gw_hh.data_ptr(), gb_hh.data_ptr(), ws.data_ptr(),
                ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "ctr_gru_fused_bwd")


def gru_bwd(gi, w_hh, b_hh, hbuf, batch, length, dim, glast, dgi, dgh) -> None:
    gi, glast = _mat(gi, "gi"), _mat(glast, "glast")
    rc = _timed("gru_bwd", lambda: (4 * batch * length * 10 * dim, 12 * batch * length * dim * dim),
                _lib.load().ctr_gru_bwd, gi.data_ptr(), _ld(gi), w_hh.data_ptr(), b_hh.data_ptr(), hbuf.data_ptr(),
                batch, length, dim, glast.data_ptr(), _ld(glast), dgi.data_ptr(), dgh.data_ptr(), _lib.stream_ptr())
    _lib.check(rc, "ctr_gru_bwd")


TOPK_MAX_K = 4096


def topk_rows(scores: torch.Tensor, k: int, dim: int = -1) -> torch.Tensor:
    """indices of the k best scores along ``dim`` of a 2-D float32 tensor, best first, ties by ascending index
    (csrc/topk.hip) -- the ranking step of every ``recommendation()`` (reference: ``torch.topk(...)[1]``,
    model/mf.py:28-35, neuralcf.py:61-72, pnn.py:133-143, din.py:55-66).  ``dim=0`` ranks columns (AutoRec's
    item-based variant) and returns (k, cols) like torch.  k up to 4096."""
    _lib.require_device(scores)
    if scores.dim() != 2 or scores.dtype != torch.float32:
        raise ValueError("topk_rows expects a 2-D float32 tensor")
    along = dim % 2
    rows, n = scores.shape[1 - along], scores.shape[along]
    if not 1 <= k <= n:
        raise RuntimeError(f"selected index k out of range: k = {k}, {n} candidates")
    out = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
    rc = _lib.load().ctr_topk_rows(_lib.ptr(scores) if rows else None, scores.stride(1 - along), scores.stride(along), rows, n, k,
                                   _lib.ptr(out) if rows else None, None, _lib.stream_ptr())
    _lib.check(rc, "ctr_topk_rows")
    return out if along == 1 else out.t()

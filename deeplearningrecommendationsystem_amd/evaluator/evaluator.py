"""Counterpart of the reference's evaluator/evaluator.py:14-20: accuracy, precision,
recall, F1 and ROC-AUC of 0.5-thresholded predictions -- computed on the device with
torch ops (the reference round-trips through sklearn on the host every epoch)."""
from __future__ import annotations

import torch


class Evaluator:
    @staticmethod
    def eval(y_true: torch.Tensor, y_pred: torch.Tensor):
        """returns [accuracy, precision, recall, f1, auc] as python floats"""
        y = y_true.detach().reshape(-1).float()
        p = y_pred.detach().reshape(-1).float()
        hard = (p >= 0.5).float()
        tp = (hard * y).sum()
        fp = (hard * (1 - y)).sum()
        fn = ((1 - hard) * y).sum()
        acc = (hard == y).float().mean()
        prec = tp / (tp + fp).clamp(min=1)
        rec = tp / (tp + fn).clamp(min=1)
        f1 = 2 * prec * rec / (prec + rec).clamp(min=1e-12)
        # AUC = P(score_pos > score_neg) + 0.5 P(tie), via average ranks
        order = torch.argsort(p)
        ranks = torch.empty_like(p)
        ranks[order] = torch.arange(1, p.numel() + 1, device=p.device, dtype=p.dtype)
        sorted_p = p[order]
        # average the ranks of tied scores
        uniq, inv, cnt = torch.unique_consecutive(sorted_p, return_inverse=True, return_counts=True)
        ends = torch.cumsum(cnt, 0).float()
        avg = ends - (cnt.float() - 1) / 2
        ranks[order] = avg[inv]
        npos, nneg = y.sum(), (1 - y).sum()
        auc = (ranks[y > 0.5].sum() - npos * (npos + 1) / 2) / (npos * nneg).clamp(min=1)
        return [float(v) for v in (acc, prec, rec, f1, auc)]

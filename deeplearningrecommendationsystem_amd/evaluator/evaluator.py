"""Counterpart of the reference's evaluator/evaluator.py:14-20: accuracy, precision,
recall, F1 and ROC-AUC of the 0.5-thresholded predictions -- computed on the device with
torch ops (the reference round-trips through sklearn on the host every epoch).

The reference thresholds FIRST (``y_pred = (y_pred >= 0.5).astype(int)``, evaluator.py:17) and
hands the hard labels to every sklearn metric, ``roc_auc_score`` included: its "ROC AUC" is
the AUC of a two-valued score, i.e. ``(TPR + TNR) / 2``.  ``eval`` reproduces exactly that;
the ranking AUC of the raw probabilities is the separately named ``score_auc``."""
from __future__ import annotations

import torch


class Evaluator:
    @staticmethod
    def eval(y_true: torch.Tensor, y_pred: torch.Tensor):
        """returns [accuracy, precision, recall, f1, auc] as python floats, each as sklearn
        computes it from the thresholded predictions (reference evaluator/evaluator.py:14-20)"""
        y = y_true.detach().reshape(-1).float()
        p = y_pred.detach().reshape(-1).float()
        hard = (p >= 0.5).float()
        tp = (hard * y).sum()
        fp = (hard * (1 - y)).sum()
        fn = ((1 - hard) * y).sum()
        tn = ((1 - hard) * (1 - y)).sum()
        acc = (hard == y).float().mean()
        prec = tp / (tp + fp).clamp(min=1)
        rec = tp / (tp + fn).clamp(min=1)
        f1 = 2 * prec * rec / (prec + rec).clamp(min=1e-12)
        # roc_auc_score on a {0,1} score: one interior ROC point (FPR, TPR) -> area (TPR + TNR) / 2;
        # nan when a class is missing from y_true (sklearn >= 1.7 warns and returns nan)
        auc = 0.5 * (tp / (tp + fn) + tn / (tn + fp))
        return [float(v) for v in (acc, prec, rec, f1, auc)]

    @staticmethod
    def score_auc(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:
        """extra, not in the reference: ROC-AUC of the raw probabilities,
        P(score_pos > score_neg) + 0.5 P(tie), via tie-averaged ranks"""
        y = y_true.detach().reshape(-1).float()
        p = y_pred.detach().reshape(-1).float()
        order = torch.argsort(p)
        sorted_p = p[order]
        _, inv, cnt = torch.unique_consecutive(sorted_p, return_inverse=True, return_counts=True)
        ends = torch.cumsum(cnt, 0).float()
        avg = ends - (cnt.float() - 1) / 2
        ranks = torch.empty_like(p)
        ranks[order] = avg[inv]
        npos, nneg = y.sum(), (1 - y).sum()
        return float((ranks[y > 0.5].sum() - npos * (npos + 1) / 2) / (npos * nneg))

"""``auc_score`` with the reference's signature (/root/reference/MIND_2020/evaluation.py:26-27) -- the only
metric on the path (the reference's dcg / ndcg / mrr helpers are unused by ``evaluate``: SURVEY section 2, out of scope).
It is what ``evaluate`` uses (train_eval.py:219-227); the reference delegates to
sklearn.metrics.roc_auc_score -- here it is the same Mann-Whitney statistic in numpy float64
(host) and in the HIP kernel ``nrms_impression_auc`` (device, used by train_eval.evaluate)."""
import numpy as np


def auc_score(y_true, y_pred):
    y_true = np.asarray(y_true)
    s = np.asarray(y_pred, dtype=np.float64)
    pos, neg = s[y_true == 1], s[y_true != 1]
    if pos.size == 0 or neg.size == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    neg_sorted = np.sort(neg)
    greater = np.searchsorted(neg_sorted, pos, side="left")          # negatives strictly below each positive
    equal = np.searchsorted(neg_sorted, pos, side="right") - greater
    return float((greater.sum() + 0.5 * equal.sum()) / (pos.size * neg.size))

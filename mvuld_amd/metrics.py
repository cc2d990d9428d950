"""Host-side meters and metrics of the train/validate loop (timm ``AverageMeter`` / ``accuracy`` as used at
main_bigvul.py:25,301-304,424; the P/R/F1 loop of :460-483; sklearn ``average_precision_score`` of :496)."""
import numpy as np
import torch


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """top-k accuracy in percent (timm.utils.accuracy)."""
    maxk = min(max(topk), output.size(1))
    _, pred = output.float().topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / target.size(0) for k in topk]


def binary_prf(target, predict):
    """precision / recall / F1 of the positive (vulnerable = 1) class, plus TP and FN."""
    target = np.asarray(target) == 1
    predict = np.asarray(predict).astype(bool)
    TP = int((predict & target).sum()); FP = int((predict & ~target).sum()); FN = int((~predict & target).sum())
    P = TP / (TP + FP) if TP + FP else 0.0
    R = TP / (TP + FN) if TP + FN else 0.0
    F1 = 2 * P * R / (P + R) if P + R else 0.0
    return P, R, F1, TP, FN


def average_precision(y_true, y_score):
    """AP = sum_n (R_n - R_{n-1}) P_n over distinct score thresholds (sklearn.metrics.average_precision_score)."""
    y_true = (np.asarray(y_true) == 1).astype(np.float64)
    y_score = np.asarray(y_score, dtype=np.float64)
    if y_true.sum() == 0:
        return 0.0
    order = np.argsort(-y_score, kind="mergesort")
    y_true, y_score = y_true[order], y_score[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true)[idx]
    fps = 1 + idx - tps
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))

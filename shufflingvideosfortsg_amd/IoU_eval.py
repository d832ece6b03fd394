"""R@1 / mIoU scorer for prediction ("submits") files -- reference grounding/IoU_eval.py.

Every (video, sentence-index) pair is its own group holding exactly one proposal, so the reference's
pandas group-by loop reduces to one vectorised IoU per row: R@1@m = mean(iou > m) with a strict '>',
mIoU = round(mean(iou)*100, 2), IoU union = len1 + len2 - inter (+1e-4).  The accumulator is
zero-initialised (the reference's ``np.empty`` can leak garbage, IoU_eval.py:131)."""
from __future__ import annotations

import json

import numpy as np

TIOU = (0.1, 0.3, 0.5, 0.7, 0.9)
pred_fields = ['results', 'version', 'external_data']


def segment_iou(pred: np.ndarray, gt: np.ndarray) -> np.ndarray:
    pred = np.asarray(pred, dtype=np.float64); gt = np.asarray(gt, dtype=np.float64)
    inter = (np.minimum(pred[:, 1], gt[:, 1]) - np.maximum(pred[:, 0], gt[:, 0])).clip(0)
    union = (gt[:, 1] - gt[:, 0]) + (pred[:, 1] - pred[:, 0]) - inter
    return inter / (union + 1e-4)


def score(pred, gt, thresholds=TIOU):
    """-> (mIoU in percent rounded to 2 dp, [R@1 per threshold in percent rounded to 2 dp])"""
    iou = segment_iou(pred, gt)
    return round(float(iou.mean()) * 100, 2), [round(float((iou > t).sum()) / len(iou) * 100, 2) for t in thresholds]


def load_submits(data):
    if not all(f in data for f in pred_fields):
        raise IOError('Please input a valid proposal file.')
    pred, gt = [], []
    for _, rows in data['results'].items():
        for r in rows:
            pred.append(r['timestamp']); gt.append(r['gt_timestamp'])
    return np.asarray(pred, dtype=np.float64), np.asarray(gt, dtype=np.float64)


def retrieval_eval(filename, verbose=True):
    """Score a submits JSON (path or already-loaded dict); prints the reference's table."""
    if isinstance(filename, dict):
        data = filename
    else:
        with open(filename, 'r') as f:
            data = json.load(f)
    miou, recall = score(*load_submits(data))
    if verbose:
        print('\tmIoU\t', '\t'.join(str(t) for t in TIOU))
        print(1, '\t', miou, '\t', '\t'.join(str(r) for r in recall))
        print('mIoU\t{:.4f}'.format(miou))
    return miou, recall

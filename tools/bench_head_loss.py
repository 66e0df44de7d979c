#!/usr/bin/env python
"""Timing of the training-side step after the decoder: Detr3DCriterion.loss (all layers, one host round trip) at the
BASELINE size (dev tool).  Prints wall time per call and the split: device kernels / host assignment."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import Detr3DCriterion  # noqa: E402
from graph_detr4d_amd import criterion as C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--queries', type=int, default=900)
    ap.add_argument('--layers', type=int, default=6)
    ap.add_argument('--gts', type=int, default=45)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = 'cuda'
    cls = (torch.randn(a.layers, 1, a.queries, 10, device=dev) * 2 - 2).requires_grad_()
    box = torch.randn(a.layers, 1, a.queries, 10, device=dev)
    box[..., 0:2] *= 30.
    box.requires_grad_()
    gt = torch.randn(a.gts, 9, device=dev)
    gt[:, 0:2] *= 30.
    gt[:, 3:6] = gt[:, 3:6].abs() * 2 + 0.3
    lab = torch.randint(0, 10, (a.gts,), device=dev)
    crit = Detr3DCriterion().to(dev)
    preds = dict(all_cls_scores=cls, all_bbox_preds=box)

    def step():
        cls.grad = box.grad = None
        losses = crit.loss([gt], [lab], preds)
        sum(losses.values()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters
    # host share: the assignment alone on a cost matrix already on the host
    packed = C.pack_ground_truth([gt], [lab], dev)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        crit.assigner.assign_layers(cls, box, [gt], [lab], packed)
    torch.cuda.synchronize()
    assign = (time.perf_counter() - t0) / a.iters
    print(f'{a.layers} layers x {a.queries} queries x {a.gts} boxes: loss + backward {wall * 1e3:.2f} ms per step '
          f'(cost launch + copy + {a.layers} host assignments + copy: {assign * 1e3:.2f} ms; 1 device synchronisation)')


if __name__ == '__main__':
    main()

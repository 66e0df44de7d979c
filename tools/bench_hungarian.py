#!/usr/bin/env python
"""Timing of gd4d_hungarian_assign_fwd (dev tool): NL x B problems of Q predictions x G boxes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_detr4d_amd import ops  # noqa: E402
from bench_late import timed  # noqa: E402


def main():
    nl, b, q = 6, 1, 900
    for g in (int(x) for x in (sys.argv[1:] or ['40', '100'])):
        rng = np.random.default_rng(g)
        cost = torch.from_numpy(rng.normal(0, 3, (nl * q * g * b,)).astype(np.float32)).cuda()
        start = torch.tensor([g * i for i in range(b + 1)], dtype=torch.int32).cuda()
        a, st = ops.hungarian_assign_fwd(cost, start, nl, b, q, g * b, g)
        ws = torch.empty(int(ops._lib.load().gd4d_hungarian_assign_workspace_bytes(nl, b, q, g)), device='cuda', dtype=torch.uint8)
        t = timed(lambda: ops.hungarian_assign_fwd(cost, start, nl, b, q, g * b, g, assigned=a, status=st, workspace=ws), 20, 1)
        print(f'{nl * b} problems of {q} x {g}: {t:.1f} us per launch')


if __name__ == '__main__':
    main()

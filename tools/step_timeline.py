#!/usr/bin/env python
"""Timeline of ONE replayed decoder step from a rocprofv3 kernel trace (dev tool).
usage: python tools/step_timeline.py <kernel_trace.csv> [marker substring of the step's first kernel]"""
import csv
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'pyramid_'          # the per-sample copy: gd4d::pyramid_slice_planar_kernel (or _channels_last_)
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '')))
                 for r in rows), key=lambda t: t[0])
    starts = [i for i, k in enumerate(ks) if marker in k[2]]
    if len(starts) < 3:
        print('marker kernel not found often enough:', len(starts))
        return
    pairs = [(a, b) for a, b in zip(starts, starts[1:]) if b - a >= 10]      # marker-to-marker spans that hold a whole step
    if not pairs:
        print('no complete step between two marker kernels')
        return
    a, b = min(pairs, key=lambda ab: ks[ab[1]][0] - ks[ab[0]][0])      # the shortest complete step: no profiler flush inside it
    step = ks[a:b]
    t0 = step[0][0]
    print(f'{len(step)} kernels, step span {(ks[b][0] - t0) / 1e3:.1f} us')
    prev_end = t0
    for s, e, name, q in step:
        short = name.split('(')[0].replace('void gd4d::', '').replace('gd4d::', '')[:60]
        print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}  q{q:>3}  {short}')
        prev_end = max(prev_end, e)


if __name__ == '__main__':
    main()

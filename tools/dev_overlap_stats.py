"""dev: from a rocprofv3 kernel trace of `bench.py --inflight N`: how much of each kernel class's time runs beside a
gather (cross_attn_agg_items) of ANOTHER request, and how long each class takes alone / beside one."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r['Kernel_Name']
    cls = ('gather' if 'agg_items' in n else 'chain' if 'row_chain' in n else 'mha' if 'mha_core' in n else
           'plan' if 'plan_kernel' in n else 'copy' if 'slice_planar' in n else None)
    if cls:
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), cls, r.get('Queue_Id') or r.get('Stream_Id')))
ev.sort()
t0 = ev[len(ev) // 3][0]
t1 = ev[2 * len(ev) // 3][0]
ev = [e for e in ev if t0 <= e[0] < t1]
g = [(s, e) for s, e, c, q in ev if c == 'gather']
span = (ev[-1][1] - ev[0][0]) / 1e3
print(f'{len(ev)} kernels in a window of {span:.0f} us; queues: {sorted(set(e[3] for e in ev))}')
busy = defaultdict(float)
stat = defaultdict(lambda: [0, 0.0, 0.0])
for s, e, c, q in ev:
    ov = 0
    for gs, ge in g:
        if (gs, ge) == (s, e):
            continue
        ov += max(0, min(e, ge) - max(s, gs))
    st = stat[c]
    st[0] += 1
    st[1] += (e - s) / 1e3
    st[2] += min(ov, e - s) / 1e3
for c, (n, tot, ov) in stat.items():
    print(f'{c:7s} {n:5d} launches, mean {tot / n:7.1f} us, {100 * ov / tot:5.1f} % of its time beside another gather; total {tot:9.0f} us = {100 * tot / span:5.1f} % of the window')
# union of gather intervals
u = 0
cur_s, cur_e = None, None
for s, e in sorted(g):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            u += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
u += cur_e - cur_s
print(f'some gather is running {100 * u / 1e3 / span:.1f} % of the window')

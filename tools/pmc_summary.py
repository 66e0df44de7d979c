#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch, per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get('Kernel_Name', '')
            if 'gd4d' not in name:
                continue
            acc[name.split('(')[0][-60:]][row['Counter_Name']].append(float(row['Counter_Value']))
out = []
for k, ctrs in acc.items():
    out.append(k)
    for c, v in sorted(ctrs.items()):
        out.append(f'  {c:32s} mean/dispatch {sum(v) / len(v):16.1f}   (n={len(v)})')
txt = '\n'.join(out)
print(txt)
open(os.path.join(root, 'pmc_summary.txt'), 'w').write(txt + '\n')

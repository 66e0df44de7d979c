#!/usr/bin/env python
"""Turn a tools/prof_pmc.sh summary of tools/bench_kernel.py into profiles/rNN_pmc_cross_attn.json, keyed to the kernel source
it was measured on (bench.py reports `roofline.traffic` only while that hash matches).
usage: python tools/make_pmc_record.py gpurun_out/<tag>/pmc_summary.txt profiles/r02_pmc_cross_attn.json "<workload note>" """
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary, out_path, note = sys.argv[1], sys.argv[2], sys.argv[3]
# optional 4th / 5th arguments: kernel-name substring and the source file it lives in (default: the projected-value gather)
KERNEL = sys.argv[4] if len(sys.argv) > 4 else 'cross_attn_fwd_block'
SOURCE = sys.argv[5] if len(sys.argv) > 5 else 'gd4d_cross_attn.hip' 
ctr, cur = {}, None
for line in open(summary):
    m = re.match(r'\s+(\S+)\s+mean/dispatch\s+([0-9.]+)', line)
    if not m:                                     # a kernel name (some come with a leading blank from the demangler)
        cur = line.strip()
        continue
    if cur and KERNEL in cur:
        ctr[m.group(1)] = float(m.group(2))
src = open(os.path.join(ROOT, 'graph-detr4d_amd', 'csrc', SOURCE), 'rb').read()
rec = {
    'kernel': 'gd4d::' + KERNEL + ' (' + SOURCE + ')',
    'kernel_source_sha256': hashlib.sha256(src).hexdigest(),
    'workload': note,
    'FETCH_SIZE_KB': ctr.get('FETCH_SIZE'), 'WRITE_SIZE_KB': ctr.get('WRITE_SIZE'),
    'TCC_HIT': ctr.get('TCC_HIT_sum'), 'TCC_MISS': ctr.get('TCC_MISS_sum'),
    'correction': 'gfx950: FETCH_SIZE tallies 128-byte line requests at 64 B (MI355X_MICROARCH.md, HBM section): '
                  'HBM read bytes = 2 x FETCH_SIZE x 1024',
    'hbm_read_bytes_per_launch': 2 * ctr['FETCH_SIZE'] * 1024,
    'hbm_write_bytes_per_launch': ctr['WRITE_SIZE'] * 1024,
    'counters': ctr,
}
json.dump(rec, open(out_path, 'w'), indent=1)
print(json.dumps({k: rec[k] for k in ('hbm_read_bytes_per_launch', 'hbm_write_bytes_per_launch', 'TCC_HIT', 'TCC_MISS')}))

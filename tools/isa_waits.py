#!/usr/bin/env python
"""Per kernel of a csrc/*.hip file: global loads, s_waitcnt vmcnt instructions and how many of them are vmcnt(0), registers.
Many vmcnt(0) next to few loads = loads the compiler waits for one at a time (a load under a select / `continue`, dependent
`t += p[i]` chains, an operand first used after stores: docs/design_notes_r01_r03.md section 4.7).  Compiles for gfx950 with the library's flags;
no GPU needed.  usage: python tools/isa_waits.py gd4d_linear.hip [name-substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'graph-detr4d_amd', 'csrc')


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
               '-fhip-fp32-correctly-rounded-divide-sqrt', '-munsafe-fp-atomics', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
               '-S', '--cuda-device-only', '-o', out, os.path.join(CSRC, src)]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    parts = re.split(r'\n(_ZN4gd4d[^\n:]*):\s', txt)
    for i in range(1, len(parts), 2):
        name, rest = parts[i], parts[i + 1]
        body = rest.split('.Lfunc_end')[0]
        loads = len(re.findall(r'\b(global_load|buffer_load)_', body))
        stores = len(re.findall(r'\b(global_store|buffer_store)_', body))
        waits = len(re.findall(r's_waitcnt vmcnt', body))
        w0 = len(re.findall(r's_waitcnt vmcnt\(0\)', body))
        regs = re.search(r'; NumVgprs: (\d+)', rest)
        occ = re.search(r'; Occupancy: (\d+)', rest)
        scratch = re.search(r'; ScratchSize: (\d+)', rest)
        try:
            short = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            short = name
        short = re.sub(r'\(.*', '', short)
        if want in short and loads + stores > 0:
            print(f'{short[:84]:84s} loads {loads:3d} stores {stores:3d} vmcnt waits {waits:3d} (to zero: {w0:3d})  '
                  f'vgprs {regs.group(1) if regs else "?":>3s} occupancy {occ.group(1) if occ else "?"} scratch {scratch.group(1) if scratch else "?"}')


if __name__ == '__main__':
    main()

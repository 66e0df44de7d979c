import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, pytest
import test_train_chains_gpu as T
from golden_io import Golden


class MP:
    def setenv(self, k, v): os.environ[k] = v
    def setattr(self, o, n, v): setattr(o, n, v)


g = Golden('decoder_deform')
tr = T._transformer(g)
reg = T._reg_branches(g.meta['num_layers']) if len(sys.argv) > 1 else None
from graph_detr4d_amd import fused_train
real = fused_train.run
a = T._run(tr, g, reg, True, MP())
fused_train.run = real
b = T._run(tr, g, reg, False, MP())
rel = lambda x, y: ((x - y).abs().max() / y.abs().max().clamp_min(1e-12)).item()
print('states', rel(a['states'], b['states']), 'refs', rel(a['refs'], b['refs']))
c = 256
print('qe pos half', rel(a['qe'][:, :c], b['qe'][:, :c]), 'query half', rel(a['qe'][:, c:], b['qe'][:, c:]))
for i, (x, y) in enumerate(zip(a['feats'], b['feats'])):
    print('feat', i, rel(x, y))
for k, y in b['params'].items():
    x = a['params'][k]
    if y is None or x is None:
        print(k, 'None', x is None, y is None)
        continue
    print(f'{k:60s} {rel(x, y):.2e}')

"""Loader for tests/golden/*.npz (written by tools/gen_golden.py from the reference itself)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.meta = json.loads(bytes(z['meta']).decode())
        self.arrays = {k: z[k] for k in z.files if k != 'meta'}

    def t(self, key):
        return torch.from_numpy(np.ascontiguousarray(self.arrays[key]))

    def has(self, key):
        return key in self.arrays

    def feats(self, dtype=torch.float32):
        """List of L tensors (B, N, C, H, W) on the exact k/32 grid."""
        out, i = [], 0
        while f'feat{i}@q' in self.arrays:
            out.append((torch.from_numpy(self.arrays[f'feat{i}@q']).float()
                        / self.meta['feat_scale']).to(dtype))
            i += 1
        return out

    def state(self, prefix='sd.'):
        """State dict (fp32 tensors) with the reference's key names."""
        sd = {}
        for k, v in self.arrays.items():
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if name.endswith('@q'):
                sd[name[:-2]] = torch.from_numpy(np.asarray(v.astype(np.float32) / self.meta['w_scale'], dtype=np.float32))
            else:
                sd[name] = torch.from_numpy(np.asarray(v, dtype=np.float32))
        return sd

    def img_metas(self):
        l2i = self.arrays['lidar2img']
        n = l2i.shape[0]
        shp = tuple(self.meta['img_shape'])
        return [dict(lidar2img=[l2i[i] for i in range(n)], img_shape=[shp] * n)
                for _ in range(self.meta['batch'])]


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}

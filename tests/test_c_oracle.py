"""Pin the plain-C oracle (oracle/gd4d_oracle.c) against the reference's golden vectors. CPU only."""
import numpy as np
import pytest
import torch

from golden_io import Golden
from oracle import c_oracle
from oracle import torch_oracle as O


@pytest.mark.parametrize('name', ['deform_n6', 'deform_n12_depth', 'deform_n24_b2', 'deform_edge'])
def test_c_cross_attn_matches_reference(name):
    g = Golden(name)
    m = g.meta
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    sd = g.state()
    flat, shapes = O.flatten_pyramid(g.feats())
    val = torch.nn.functional.linear(flat, sd['value_proj.weight'], sd['value_proj.bias'])
    val = val.view(b * n, -1, 8, 32).numpy()
    l2i = np.broadcast_to(g.arrays['lidar2img'][None], (b, n, 4, 4))
    out, mask, uv = c_oracle.cross_attn_fwd(
        val, shapes, g.arrays['reference_points'], g.arrays['offsets'].reshape(b, q, 8, 4, 3),
        g.arrays['attn_logits'].reshape(b, q, 8, 4, 4), g.arrays['cam_logits'], l2i, m['pc_range'],
        m['img_shape'][0], m['img_shape'][1])
    gmask = g.arrays['mask'].reshape(b, n, q, 8, 4, 4)[..., 0, :]
    guv = g.arrays['uv'].reshape(b, n, q, 8, 4, 4, 2)[..., 0, :, :]
    assert np.array_equal(mask, gmask)
    assert np.array_equal(uv, guv)
    np.testing.assert_allclose(out, g.arrays['agg'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name', ['detr3d_n6', 'detr3d_n12_b2'])
def test_c_detr3d_matches_reference(name):
    g = Golden(name)
    m = g.meta
    b, n, q = m['batch'], m['num_cams'], m['num_query']
    feats = [f.numpy() for f in g.feats()]
    l2i = np.broadcast_to(g.arrays['lidar2img'][None], (b, n, 4, 4))
    logits = g.arrays['attn_logits'].reshape(b, q, n, 1, len(feats))
    out, mask, uv, sampled = c_oracle.detr3d_fwd(feats, g.arrays['reference_points'], logits, l2i,
                                                 m['pc_range'], m['img_shape'][0], m['img_shape'][1],
                                                 want_sampled=True)
    gmask = g.arrays['fs_mask'][:, 0, :, :, 0, 0].transpose(0, 2, 1)      # (B,1,Q,N,1,1) -> (B,N,Q)
    assert np.array_equal(mask, gmask)
    np.testing.assert_allclose(sampled, g.arrays['fs_sampled'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out, g.arrays['agg'].transpose(1, 0, 2) if g.arrays['agg'].shape[0] == q
                               else g.arrays['agg'], rtol=1e-5, atol=2e-5)

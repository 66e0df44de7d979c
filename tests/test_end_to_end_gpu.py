"""Forward parity of the whole head pipeline on the GPU - feature position embedding -> Detr3DTransformer (2 layers,
Deform3DCrossAttn, reference-point refinement) -> per-layer head outputs -> NMSFreeCoder.decode - against the same
pipeline assembled from the oracle's pieces (each of which is pinned by a reference-generated fixture).  This is the
proxy for the reference's end metric (mAP/NDS needs the dataset): identical boxes, scores and labels for identical
inputs and weights.  GPU only."""
import pytest
import torch
import torch.nn as nn

import graph_detr4d_amd as G
from graph_detr4d_amd import functional as Fn
from graph_detr4d_amd import synthetic
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
POST_RANGE = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]


@pytest.mark.parametrize('case', ['small', 'mid', 'small-channels-last', 'full-channels-last'])
def test_features_to_boxes_matches_oracle_pipeline(case):
    """small-channels-last: the position-embedding stage hands the decoder channels-last levels (FeaturePositionEmbedding(
    channels_last_out=True): SE gate + fuse as one kernel, the levels gathered in place - no slice-planar copy in the decoder).
    small: 60 queries x 6 cameras x 2 layers - everything must agree element for element.  mid: 300 queries x 12 cameras
    (two frames) x 3 layers on 256 x 448 images - large enough that a sample or two sits within rounding of a visibility
    boundary (the two sides compute the offsets with different GEMM arithmetic): the decoder states are compared row by row
    with a counted number of outliers, the decoded detections as sets."""
    torch.manual_seed(7)
    channels_last = case.endswith('channels-last')
    case = case.split('-')[0]
    top = 100
    if case == 'small':
        n, q, nl, frames = 6, 60, 2, 1
        img_hw, levels = (128, 224), [(16, 28), (8, 14), (4, 7), (2, 4)]
    elif case == 'full':
        # the bench's size: 900 queries x 24 cameras (6 x T = 4) on the 900 x 1600 rig, the R50 pyramid 116x200 .. 15x25, top 300 -
        # two decoder layers (chained layers on i.i.d. synthetic features amplify a rounding difference ~3 x per layer:
        # tests/test_full_size_gpu.py checks all six with teacher forcing)
        n, q, nl, frames = 24, 900, 2, 4
        img_hw, levels, top = (900, 1600), list(synthetic.R50_LEVELS), 300
        torch.set_num_threads(16)
    else:
        n, q, nl, frames = 12, 300, 3, 2
        img_hw, levels = (256, 448), [(32, 56), (16, 28), (8, 14), (4, 7)]
    rig = synthetic.camera_rig(frames, img_hw)
    metas = synthetic.make_img_metas(rig, img_shape=(img_hw[0], img_hw[1], 3), pad_shape=(img_hw[0], img_hw[1], 3))
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(1, n, 256, h, w, generator=g) for h, w in levels]
    query_embed = torch.randn(q, 512, generator=g)

    pe = G.FeaturePositionEmbedding(pc_range=synthetic.PC_RANGE, channels_last_out=channels_last)
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=n,
        decoder=dict(type='Detr3DTransformerDecoder', num_layers=nl, return_intermediate=True,
                     transformerlayers=dict(
                         type='DetrTransformerDecoderLayer',
                         attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1),
                                    dict(type='Deform3DCrossAttn', num_cams=n, pc_range=synthetic.PC_RANGE,
                                         num_points=4, embed_dims=256)],
                         feedforward_channels=512, ffn_dropout=0.1,
                         operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
    cls_b = nn.ModuleList(nn.Sequential(nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                                        nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                                        nn.Linear(256, 10)) for _ in range(nl))
    reg_b = nn.ModuleList(nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                        nn.Linear(256, 10)) for _ in range(nl))
    for mod in (pe, tr, cls_b, reg_b):
        synthetic.randomise_all_(mod, seed=11, std=0.04)
        mod.eval()
    with torch.no_grad():                                       # spread the class logits so the top-k is well separated
        for b in cls_b:
            b[-1].weight.mul_(8)
            b[-1].bias.copy_(torch.linspace(-3, 1, 10))

    # ---- oracle pipeline (CPU) ----
    with torch.no_grad():
        sd_pe = {k: v.detach().clone() for k, v in pe.state_dict().items()}
        ofeats, _ = O.feature_position_embedding(sd_pe, feats, rig[None], [metas[0]['img_shape']], metas[0]['pad_shape'][0],
                                                 64, 1, synthetic.PC_RANGE)
        sd = {k: v.detach().clone() for k, v in tr.state_dict().items()}
        layers = [{k[len(f'decoder.layers.{i}.'):]: v for k, v in sd.items() if k.startswith(f'decoder.layers.{i}.')}
                  for i in range(nl)]
        states, init_ref, refs = O.transformer(sd, layers, ofeats, query_embed, metas, synthetic.PC_RANGE,
                                               reg_branches=[reg_b[i] for i in range(nl)], cross='Deform3DCrossAttn',
                                               num_points=4)
        hs = states.permute(0, 2, 1, 3)
        exp_cls = torch.stack([cls_b[i](hs[i]) for i in range(nl)])
        exp_box = torch.stack([O.box_head(reg_b[i](hs[i]), init_ref if i == 0 else refs[i - 1], synthetic.PC_RANGE)
                               for i in range(nl)])
        exp = O.nms_free_decode({'all_cls_scores': exp_cls, 'all_bbox_preds': exp_box}, POST_RANGE, top, 10)[0]

    # ---- HIP pipeline ----
    for mod in (pe, tr, cls_b, reg_b):
        mod.cuda()
    coder = G.build_bbox_coder(dict(type='NMSFreeCoder', pc_range=synthetic.PC_RANGE, post_center_range=POST_RANGE,
                                    max_num=top, num_classes=10))
    with torch.no_grad():
        gfeats = pe([f.cuda() for f in feats], metas)
        from graph_detr4d_amd import ops
        assert all(ops.PyramidView.is_channels_last_level(f) for f in gfeats) == channels_last
        copies = []
        real_copy = ops.pyramid_slice_planar_fwd
        ops.pyramid_slice_planar_fwd = lambda *a, **k: (copies.append(1), real_copy(*a, **k))[1]
        try:
            gstates, ginit, grefs = tr(gfeats, query_embed.cuda(), reg_branches=reg_b, img_metas=metas)
        finally:
            ops.pyramid_slice_planar_fwd = real_copy
        assert bool(copies) != channels_last, 'the decoder copies NCHW levels once per sample and reads channels-last levels in place'
        outs = Fn.head_outputs(gstates, ginit, grefs, cls_b, reg_b, synthetic.PC_RANGE)
        got = coder.decode(outs)[0]

    for a, b in zip(gfeats, ofeats):
        torch.testing.assert_close(a.cpu(), b, rtol=2e-4, atol=2e-4)
    if case == 'small':
        torch.testing.assert_close(gstates.cpu(), states, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(outs['all_cls_scores'].cpu(), exp_cls, rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(outs['all_bbox_preds'].cpu(), exp_box, rtol=2e-3, atol=2e-3)
    else:
        row_err = (gstates.cpu() - states).abs().amax(dim=(2, 3))            # (layers, queries)
        assert (row_err[0] > 1e-3).sum().item() <= 2 and (row_err > 2e-3).float().mean().item() <= 0.03, \
            ((row_err[0] > 1e-3).sum().item(), (row_err > 2e-3).float().mean().item())
        assert row_err.median().item() < 2e-4
    # decoded detections: same count, scores agree rank by rank; as sets (near-ties may swap ranks, and the last few of the
    # top-k may differ) every detection has a partner with the same label, score and box
    assert got['scores'].shape == exp['scores'].shape and got['scores'].numel() > 10
    gs, es = got['scores'].cpu(), exp['scores']
    if case == 'small':
        torch.testing.assert_close(gs, es, rtol=0, atol=2e-3)
    gb, gl = got['bboxes'].cpu(), got['labels'].cpu()
    same = (gl[:, None] == exp['labels'][None]) & ((gs[:, None] - es[None]).abs() < 2e-3) & \
        ((gb[:, None] - exp['bboxes'][None]).abs().amax(-1) < 5e-3)
    need = 0.97 if case == 'small' else 0.93
    if case == 'full':
        print(f'full size: decoder rows off by > 1e-3 in layer 0: {(row_err[0] > 1e-3).sum().item()} of {q}; median row error '
              f'{row_err.median().item():.2e}; top-{top} detections with a partner: {same.any(1).float().mean().item():.3f} / '
              f'{same.any(0).float().mean().item():.3f}')
    assert same.any(1).float().mean().item() >= need
    assert same.any(0).float().mean().item() >= need

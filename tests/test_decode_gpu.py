"""gd4d_nms_free_decode_fwd / gd4d_box_head_fwd (through NMSFreeCoder and functional.head_outputs) against the
reference-generated fixtures and the oracle.  GPU only."""
import pytest
import torch
import torch.nn as nn

from golden_io import Golden
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu

POST_RANGE = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]
PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def _coder(**kw):
    from graph_detr4d_amd import build_bbox_coder
    cfg = dict(type='NMSFreeCoder', pc_range=PC_RANGE, post_center_range=POST_RANGE, max_num=300, num_classes=10)
    cfg.update(kw)
    return build_bbox_coder(cfg)


def _check(got, exp):
    assert got['bboxes'].shape == exp['bboxes'].shape
    assert got['labels'].dtype == torch.int64
    assert torch.equal(got['labels'].cpu(), exp['labels'])
    # expf / atan2f on the device differ from glibc's by an ulp or two
    torch.testing.assert_close(got['scores'].cpu(), exp['scores'], rtol=0, atol=2e-7)
    torch.testing.assert_close(got['bboxes'].cpu(), exp['bboxes'], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize('name', ['decode', 'decode_thr', 'decode_code8'])
def test_decode_matches_reference_fixture(name):
    g = Golden(name)
    m = g.meta
    coder = _coder(max_num=m['max_num'], score_threshold=m['score_threshold'], post_center_range=m['post_center_range'],
                   pc_range=m['pc_range'])
    preds = {'all_cls_scores': g.t('all_cls_scores').cuda(), 'all_bbox_preds': g.t('all_bbox_preds').cuda()}
    out = coder.decode(preds)
    assert len(out) == m['batch']
    for b, d in enumerate(out):
        _check(d, {k: g.t(f'{k}{b}') for k in ('bboxes', 'scores', 'labels')})
    single = coder.decode_single(preds['all_cls_scores'][-1, 0], preds['all_bbox_preds'][-1, 0])
    _check(single, {k: g.t(f'{k}0') for k in ('bboxes', 'scores', 'labels')})


@pytest.mark.parametrize('q,batch,k', [(900, 2, 300), (2700, 1, 300), (900, 1, 1024), (31, 3, 7)])
def test_decode_full_size_matches_oracle(q, batch, k):
    g = torch.Generator().manual_seed(q + k)
    cls = torch.randn(1, batch, q, 10, generator=g) * 2 - 2
    box = torch.randn(1, batch, q, 10, generator=g)
    box[..., 0:2] *= 40.
    box[..., 4] *= 6.
    out = _coder(max_num=k).decode({'all_cls_scores': cls.cuda(), 'all_bbox_preds': box.cuda()})
    exp = O.nms_free_decode({'all_cls_scores': cls, 'all_bbox_preds': box}, POST_RANGE, k, 10)
    for d, e in zip(out, exp):
        _check(d, e)
        s = d['scores']
        assert bool((s[:-1] >= s[1:]).all()), 'scores must come out sorted'


def test_decode_ties_take_lowest_indices():
    """All-equal and partially-equal scores: the K survivors are the lowest flat indices, in index order."""
    from graph_detr4d_amd import ops
    q, c, k = 64, 10, 100
    cls = torch.zeros(1, q, c)
    box = torch.zeros(1, q, 10)
    box[0, :, 8] = torch.arange(q).float()                       # vx carries the query index
    boxes, scores, labels, keep = ops.nms_free_decode_fwd(cls.cuda(), box.cuda(), POST_RANGE, k)
    idx = torch.arange(k)
    assert torch.equal(labels[0].cpu().long(), idx % c)
    assert torch.equal(boxes[0, :, 7].cpu().long(), idx // c)
    assert torch.equal(scores.cpu(), torch.full((1, k), 0.5))
    assert bool(keep.all())
    cls[0, 40:50, 3] = 1.0                                       # ten clear winners, then the tie
    boxes, scores, labels, keep = ops.nms_free_decode_fwd(cls.cuda(), box.cuda(), POST_RANGE, k)
    assert torch.equal(boxes[0, :10, 7].cpu().long(), torch.arange(40, 50))
    assert bool((labels[0, :10] == 3).all())
    rest = torch.tensor([i for i in range(q * c) if not (400 <= i < 500 and i % c == 3)][:k - 10])
    assert torch.equal(labels[0, 10:].cpu().long(), rest % c)
    assert torch.equal(boxes[0, 10:, 7].cpu().long(), rest // c)


def test_decode_errors():
    from graph_detr4d_amd import ops
    from graph_detr4d_amd._lib import Gd4dError
    cls, box = torch.zeros(1, 5, 10).cuda(), torch.zeros(1, 5, 10).cuda()
    with pytest.raises(RuntimeError, match='out of range'):       # torch.topk's message in the reference
        ops.nms_free_decode_fwd(cls, box, POST_RANGE, 300)
    big = torch.zeros(1, 900, 10).cuda()
    with pytest.raises(Gd4dError):
        ops.nms_free_decode_fwd(big, big, POST_RANGE, 2000)       # K > 1024 is not supported
    with pytest.raises(Gd4dError):
        ops.nms_free_decode_fwd(big, torch.zeros(1, 900, 9).cuda(), POST_RANGE, 300)
    with pytest.raises(Gd4dError):
        ops.nms_free_decode_fwd(cls.cpu(), box.cpu(), POST_RANGE, 3)
    with pytest.raises(NotImplementedError):
        _coder(post_center_range=None).decode_single(cls[0], box[0])


def _branches(num_layers, seed):
    torch.manual_seed(seed)

    def cls_branch():
        return nn.Sequential(nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                             nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True), nn.Linear(256, 10))

    def reg_branch():
        return nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 10))
    return (nn.ModuleList(cls_branch() for _ in range(num_layers)),
            nn.ModuleList(reg_branch() for _ in range(num_layers)))


@pytest.mark.parametrize('q,batch,depth_factor', [(900, 1, None), (64, 2, 1.25)])
def test_head_outputs_match_oracle(q, batch, depth_factor):
    """cls / reg branches (detr3d_head_pe.py:368-388 shapes) + box epilogue for every decoder layer, then decode."""
    from graph_detr4d_amd import functional as Fn
    nl = 3
    cls_b, reg_b = _branches(nl, 7)
    g = torch.Generator().manual_seed(11)
    hs = torch.randn(nl, q, batch, 256, generator=g)
    init_ref = torch.rand(batch, q, 3, generator=g)
    init_ref[0, 0] = torch.tensor([0., 1., 0.5])                    # the inverse_sigmoid clamps
    inter = torch.rand(nl, batch, q, 3, generator=g)
    with torch.no_grad():
        exp_cls, exp_box = [], []
        for lvl in range(nl):
            x = hs[lvl].permute(1, 0, 2)
            ref = init_ref if lvl == 0 else inter[lvl - 1]
            exp_cls.append(cls_b[lvl](x))
            exp_box.append(O.box_head(reg_b[lvl](x), ref, PC_RANGE, depth_factor))
        exp = {'all_cls_scores': torch.stack(exp_cls), 'all_bbox_preds': torch.stack(exp_box)}
        cls_b.cuda(), reg_b.cuda()
        got = Fn.head_outputs(hs.cuda(), init_ref.cuda(), inter.cuda(), cls_b, reg_b, PC_RANGE, depth_factor)
    assert got['enc_cls_scores'] is None and got['enc_bbox_preds'] is None
    torch.testing.assert_close(got['all_cls_scores'].cpu(), exp['all_cls_scores'], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got['all_bbox_preds'].cpu(), exp['all_bbox_preds'], rtol=1e-4, atol=1e-4)
    # decoding the device-side head outputs reproduces the oracle's decode of the same tensors
    preds = {k: got[k] for k in ('all_cls_scores', 'all_bbox_preds')}
    dec = _coder(max_num=100).decode(preds)
    ref_dec = O.nms_free_decode({k: v.cpu() for k, v in preds.items()}, POST_RANGE, 100, 10)
    for d, e in zip(dec, ref_dec):
        _check(d, e)

"""CPU: the oracle's NMSFreeCoder / head-epilogue restatement against reference-generated fixtures."""
import pytest
import torch

from golden_io import Golden, sub
from oracle import torch_oracle as O

DECODE_CASES = ['decode', 'decode_thr', 'decode_code8']


@pytest.mark.parametrize('name', DECODE_CASES)
def test_nms_free_decode_matches_reference(name):
    g = Golden(name)
    m = g.meta
    preds = {'all_cls_scores': g.t('all_cls_scores'), 'all_bbox_preds': g.t('all_bbox_preds')}
    out = O.nms_free_decode(preds, m['post_center_range'], m['max_num'], m['num_classes'], m['score_threshold'])
    assert len(out) == m['batch']
    for b, d in enumerate(out):
        assert d['bboxes'].shape == g.t(f'bboxes{b}').shape
        assert 0 < d['bboxes'].shape[0] < m['max_num'], 'fixture must exercise the range filter'
        assert torch.equal(d['labels'], g.t(f'labels{b}'))
        torch.testing.assert_close(d['scores'], g.t(f'scores{b}'), rtol=0, atol=0)
        torch.testing.assert_close(d['bboxes'], g.t(f'bboxes{b}'), rtol=0, atol=0)


def test_decode_k_larger_than_scores_raises():
    with pytest.raises(RuntimeError):
        O.nms_free_decode_single(torch.zeros(5, 10), torch.zeros(5, 10), [-1, -1, -1, 1, 1, 1], 300, 10)


@pytest.mark.parametrize('name', ['decoder_deform', 'decoder_detr3d'])
def test_box_head_consistent_with_reference_refinement(name):
    """The head's sigmoid(tmp + inverse_sigmoid(ref)) (detr3d_head_pe.py:585-588) is the same expression as the
    decoder's reference-point refinement (detr3d_transformer.py:201-214), which the decoder fixtures hold from the
    reference itself: un-scaling the oracle's box centres must reproduce inter_references."""
    g = Golden(name)
    m = g.meta
    reg_sd = g.state(prefix='reg.')
    F = torch.nn.functional
    hs = g.t('inter_states')                               # (nl, Q, B, C)
    refs_in = [g.t('init_reference')] + list(g.t('inter_references'))[:-1]
    lo = torch.tensor(m['pc_range'][:3])
    span = torch.tensor(m['pc_range'][3:]) - lo
    for lvl in range(m['num_layers']):
        p = sub(reg_sd, f'{lvl}.')
        x = hs[lvl].permute(1, 0, 2)
        tmp = F.linear(F.relu(F.linear(F.relu(F.linear(x, p['0.weight'], p['0.bias'])), p['2.weight'], p['2.bias'])),
                       p['4.weight'], p['4.bias'])
        box = O.box_head(tmp, refs_in[lvl], m['pc_range'])
        centre = torch.stack([box[..., 0], box[..., 1], box[..., 4]], -1)
        torch.testing.assert_close((centre - lo) / span, g.t('inter_references')[lvl], rtol=1e-4, atol=1e-5)
        untouched = [2, 3, 5, 6, 7, 8, 9]
        assert torch.equal(box[..., untouched], tmp[..., untouched])
        scaled = O.box_head(tmp, refs_in[lvl], m['pc_range'], depth_factor=1.5)
        torch.testing.assert_close(scaled[..., [0, 1, 4]], box[..., [0, 1, 4]] * 1.5)

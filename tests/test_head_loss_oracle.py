"""CPU: the oracle's restatement of the head loss (Hungarian assignment, focal / L1 terms, normalisers) against
fixtures captured from the reference's own Detr3DHeadPE.loss + HungarianAssigner3D (tools/gen_golden.py:case_head_loss)."""
import pytest
import torch

from golden_io import Golden
from oracle import torch_oracle as O

CASES = ['head_loss', 'head_loss_b2', 'head_loss_degenerate', 'head_loss_b2_both']


def _gt(g):
    b = g.meta['batch']
    return [g.t(f'gt_boxes{i}') for i in range(b)], [g.t(f'gt_labels{i}') for i in range(b)]


@pytest.mark.parametrize('name', CASES)
def test_cost_and_assignment_match_reference(name):
    g = Golden(name)
    cls, box = g.t('all_cls_scores'), g.t('all_bbox_preds')
    boxes, labels = _gt(g)
    for l in range(g.meta['num_layers']):
        for b in range(g.meta['batch']):
            if g.meta['gts'][b] > 0:
                cost = O.hungarian_cost(box[l, b], cls[l, b], boxes[b], labels[b])
                torch.testing.assert_close(cost, g.t(f'cost_l{l}_b{b}'), rtol=0, atol=0)
            inds = O.hungarian_assign(box[l, b], cls[l, b], boxes[b], labels[b])
            assert torch.equal(inds, g.t(f'assigned_l{l}_b{b}'))
            assert int((inds > 0).sum()) == g.meta['gts'][b]


@pytest.mark.parametrize('name', CASES)
def test_losses_and_gradients_match_reference(name):
    g = Golden(name)
    cls = g.t('all_cls_scores').requires_grad_()
    box = g.t('all_bbox_preds').requires_grad_()
    boxes, labels = _gt(g)
    losses, _ = O.head_loss(cls, box, boxes, labels, torch.tensor(g.meta['code_weights']))
    assert list(losses.keys()) == g.meta['loss_keys']
    for k, v in losses.items():
        torch.testing.assert_close(v, g.t('loss.' + k).reshape(()), rtol=1e-6, atol=1e-7)
    sum(losses.values()).backward()
    torch.testing.assert_close(cls.grad, g.t('grad_cls'), rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(box.grad, g.t('grad_box'), rtol=1e-5, atol=1e-8)


def test_degenerate_fixture_exercises_the_non_finite_paths():
    g = Golden('head_loss_degenerate')
    cost = g.t('cost_l0_b0')
    assert (cost[:, 1] == 100.0).all(), 'log(0) width: the whole column is +inf -> 100'

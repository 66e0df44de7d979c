"""gd4d_match_cost_fwd / gd4d_head_loss_fwd_bwd and their host modules (HungarianAssigner3D, Detr3DCriterion) against the
fixtures captured from the reference's own Detr3DHeadPE.loss + HungarianAssigner3D and against the oracle.  GPU only."""
import pytest
import torch

from golden_io import Golden
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
CASES = ['head_loss', 'head_loss_b2', 'head_loss_degenerate', 'head_loss_b2_both']


def _gt(g, dev='cuda'):
    b = g.meta['batch']
    return [g.t(f'gt_boxes{i}').to(dev) for i in range(b)], [g.t(f'gt_labels{i}').to(dev) for i in range(b)]


@pytest.mark.parametrize('name', CASES)
def test_cost_matrices_match_reference(name):
    from graph_detr4d_amd import ops
    from graph_detr4d_amd.criterion import pack_ground_truth
    g = Golden(name)
    cls, box = g.t('all_cls_scores').cuda(), g.t('all_bbox_preds').cuda()
    boxes, labels = _gt(g)
    gt, lab, start_dev, start, counts = pack_ground_truth(boxes, labels, 'cuda')
    cost = ops.match_cost_fwd(cls, box, gt, lab, start_dev, max(counts)).cpu()
    q, sum_gt = cls.shape[2], int(start[-1])
    for l in range(g.meta['num_layers']):
        for b in range(g.meta['batch']):
            n = counts[b]
            if n == 0:
                continue
            off = q * (l * sum_gt + int(start[b]))
            got = cost[off:off + q * n].view(q, n)
            want = g.t(f'cost_l{l}_b{b}')
            assert torch.equal(got == 100.0, want == 100.0)                  # the nan_to_num entries, exactly
            torch.testing.assert_close(got, want, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize('name', CASES)
def test_assigner_matches_reference(name):
    from graph_detr4d_amd import HungarianAssigner3D
    g = Golden(name)
    cls, box = g.t('all_cls_scores').cuda(), g.t('all_bbox_preds').cuda()
    boxes, labels = _gt(g)
    asg = HungarianAssigner3D(cls_cost=dict(type='FocalLossCost', weight=2.0), reg_cost=dict(type='BBox3DL1Cost', weight=0.25),
                              iou_cost=dict(type='IoUCost', weight=0.0), pc_range=g.meta['pc_range'])
    assigned = asg.assign_layers(cls, box, boxes, labels).cpu()
    start = 0
    for b in range(g.meta['batch']):
        for l in range(g.meta['num_layers']):
            want = g.t(f'assigned_l{l}_b{b}')                               # 0 = background, g + 1 = matched
            got = assigned[l, b].long()
            got = torch.where(got >= 0, got - start + 1, torch.zeros_like(got))
            if name == 'head_loss_degenerate':
                # every query costs 100 for the degenerate box: which one takes it is a tie; the others are pinned
                keep = (want != 2) & (got != 2)
                assert torch.equal(got[keep], want[keep]) and int((got == 2).sum()) == 1
            else:
                assert torch.equal(got, want)
            # the reference's per-call entry point
            r = asg.assign(box[l, b], cls[l, b], boxes[b], labels[b])
            assert r.num_gts == g.meta['gts'][b] and torch.equal(r.gt_inds.cpu(), got)
        start += g.meta['gts'][b]


@pytest.mark.parametrize('name', ['head_loss', 'head_loss_b2', 'head_loss_b2_both'])
def test_criterion_losses_and_gradients_match_reference(name):
    from graph_detr4d_amd import Detr3DCriterion
    g = Golden(name)
    cls = g.t('all_cls_scores').cuda().requires_grad_()
    box = g.t('all_bbox_preds').cuda().requires_grad_()
    boxes, labels = _gt(g)
    crit = Detr3DCriterion(code_weights=g.meta['code_weights'], pc_range=g.meta['pc_range']).cuda()
    losses = crit.loss(boxes, labels, dict(all_cls_scores=cls, all_bbox_preds=box, enc_cls_scores=None, enc_bbox_preds=None))
    assert list(losses.keys()) == g.meta['loss_keys']
    for k, v in losses.items():
        torch.testing.assert_close(v.cpu(), g.t('loss.' + k).reshape(()), rtol=1e-5, atol=1e-6)
    sum(losses.values()).backward()
    torch.testing.assert_close(cls.grad.cpu(), g.t('grad_cls'), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(box.grad.cpu(), g.t('grad_box'), rtol=1e-5, atol=1e-8)


def test_degenerate_box_is_dropped_from_the_l1_term():
    """log(0) width: the oracle on the SAME assignment (the tie among queries is free) gives the same losses."""
    from graph_detr4d_amd import Detr3DCriterion
    g = Golden('head_loss_degenerate')
    cls, box = g.t('all_cls_scores').cuda(), g.t('all_bbox_preds').cuda()
    boxes, labels = _gt(g)
    crit = Detr3DCriterion(code_weights=g.meta['code_weights']).cuda()
    losses = crit.loss(boxes, labels, dict(all_cls_scores=cls, all_bbox_preds=box))
    assigned = crit.last_assigned.cpu().long()
    for l, (kc, kb) in enumerate([('d0.loss_cls', 'd0.loss_bbox'), ('loss_cls', 'loss_bbox')]):
        a = assigned[l, 0]
        lab = torch.full((a.numel(),), 10, dtype=torch.long)
        lab[a >= 0] = labels[0].cpu()[a[a >= 0]].long()
        want_cls = 2.0 * O.sigmoid_focal_loss_sum(cls[l, 0].cpu(), lab, 10) / 6.0
        tgt = O.normalize_bbox(boxes[0].cpu()[a[a >= 0]])
        ok = torch.isfinite(tgt).all(-1)
        assert int((~ok).sum()) == 1
        w = torch.tensor(g.meta['code_weights'])
        want_box = 0.25 * ((box[l, 0].cpu()[a >= 0][ok] - tgt[ok]).abs() * w).sum() / 6.0
        torch.testing.assert_close(losses[kc].cpu(), want_cls, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(losses[kb].cpu(), want_box, rtol=1e-5, atol=1e-6)


def test_full_size_one_sync_step():
    """900 queries x 6 layers x 45 boxes: assignment is a permutation-consistent matching and agrees with the oracle's
    per-layer loop; losses agree with the oracle."""
    from graph_detr4d_amd import Detr3DCriterion
    torch.manual_seed(5)
    nl, q, n = 6, 900, 45
    cls = (torch.randn(nl, 1, q, 10) * 2 - 2).cuda().requires_grad_()
    box = torch.randn(nl, 1, q, 10)
    box[..., 0:2] *= 30.
    box = box.cuda().requires_grad_()
    gt = torch.randn(n, 9)
    gt[:, 0:2] *= 30.
    gt[:, 3:6] = gt[:, 3:6].abs() * 2 + 0.3
    lab = torch.randint(0, 10, (n,))
    crit = Detr3DCriterion().cuda()
    losses = crit.loss([gt.cuda()], [lab.cuda()], dict(all_cls_scores=cls, all_bbox_preds=box))
    want, assigned = O.head_loss(cls.detach().cpu(), box.detach().cpu(), [gt], [lab], torch.tensor([1.] * 8 + [.2, .2]))
    got = crit.last_assigned.cpu().long()
    for l in range(nl):
        assert torch.equal(got[l, 0] + 1, assigned[l][0])
    for k in want:
        torch.testing.assert_close(losses[k].cpu(), want[k], rtol=1e-5, atol=1e-6)
    sum(losses.values()).backward()
    assert torch.isfinite(cls.grad).all() and torch.isfinite(box.grad).all() and box.grad.abs().sum() > 0


def test_head_epilogue_and_loss_train_end_to_end():
    """hs -> cls / reg branches -> box epilogue (functional.head_outputs, autograd path) -> Detr3DCriterion -> backward,
    against the oracle's box_head + head_loss with torch autograd on the CPU."""
    import copy
    from graph_detr4d_amd import Detr3DCriterion
    from graph_detr4d_amd import functional as Fn
    torch.manual_seed(9)
    nl, q, c = 2, 50, 256
    pc_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    mk_cls = lambda: torch.nn.Sequential(torch.nn.Linear(c, c), torch.nn.LayerNorm(c), torch.nn.ReLU(inplace=True),
                                         torch.nn.Linear(c, 10))
    mk_reg = lambda: torch.nn.Sequential(torch.nn.Linear(c, c), torch.nn.ReLU(), torch.nn.Linear(c, 10))
    cls_b = torch.nn.ModuleList(mk_cls() for _ in range(nl))
    reg_b = torch.nn.ModuleList(mk_reg() for _ in range(nl))
    hs = torch.randn(nl, q, 1, c)
    init_ref, inter_ref = torch.rand(1, q, 3), torch.rand(nl, 1, q, 3)
    gt = torch.randn(6, 9)
    gt[:, 0:2] *= 30.
    gt[:, 3:6] = gt[:, 3:6].abs() * 2 + 0.3
    lab = torch.randint(0, 10, (6,))
    # oracle on the CPU
    hs_c = hs.clone().requires_grad_()
    cls_c, reg_c = copy.deepcopy(cls_b), copy.deepcopy(reg_b)
    x = hs_c.permute(0, 2, 1, 3)
    all_cls = torch.stack([cls_c[l](x[l]) for l in range(nl)])
    all_box = torch.stack([O.box_head(reg_c[l](x[l]), init_ref if l == 0 else inter_ref[l - 1], pc_range) for l in range(nl)])
    want, _ = O.head_loss(all_cls, all_box, [gt], [lab], torch.tensor([1.] * 8 + [.2, .2]))
    sum(want.values()).backward()
    # product path on the GPU
    hs_g = hs.cuda().requires_grad_()
    cls_g, reg_g = cls_b.cuda(), reg_b.cuda()
    outs = Fn.head_outputs(hs_g, init_ref.cuda(), inter_ref.cuda(), cls_g, reg_g, pc_range)
    got = Detr3DCriterion().cuda().loss([gt.cuda()], [lab.cuda()], outs)
    for k in want:
        torch.testing.assert_close(got[k].cpu(), want[k].detach(), rtol=1e-4, atol=1e-5)
    sum(got.values()).backward()
    torch.testing.assert_close(hs_g.grad.cpu(), hs_c.grad, rtol=1e-3, atol=1e-6)
    for pg, pc in zip(list(cls_g.parameters()) + list(reg_g.parameters()), list(cls_c.parameters()) + list(reg_c.parameters())):
        torch.testing.assert_close(pg.grad.cpu(), pc.grad, rtol=1e-3, atol=1e-5)


def test_no_ground_truth_at_all():
    """Every query is background: focal term only, normalisers clamp to 1, no host round trip."""
    from graph_detr4d_amd import Detr3DCriterion, HungarianAssigner3D
    torch.manual_seed(3)
    cls = (torch.randn(2, 1, 30, 10) - 2).cuda().requires_grad_()
    box = torch.randn(2, 1, 30, 10).cuda().requires_grad_()
    gt, lab = torch.zeros(0, 9), torch.zeros(0, dtype=torch.long)
    crit = Detr3DCriterion().cuda()
    got = crit.loss([gt.cuda()], [lab.cuda()], dict(all_cls_scores=cls, all_bbox_preds=box))
    want, _ = O.head_loss(cls.detach().cpu(), box.detach().cpu(), [gt], [lab], torch.tensor([1.] * 8 + [.2, .2]))
    for k in want:
        torch.testing.assert_close(got[k].cpu(), want[k], rtol=1e-5, atol=1e-6)
    assert float(got['loss_bbox'].detach()) == 0.0
    sum(got.values()).backward()
    assert float(box.grad.abs().sum()) == 0.0 and float(cls.grad.abs().sum()) > 0
    r = HungarianAssigner3D(cls_cost=dict(type='FocalLossCost', weight=2.0), reg_cost=dict(type='BBox3DL1Cost', weight=0.25)) \
        .assign(box[0, 0].detach(), cls[0, 0].detach(), gt.cuda(), lab.cuda())
    assert r.num_gts == 0 and int(r.gt_inds.abs().sum()) == 0

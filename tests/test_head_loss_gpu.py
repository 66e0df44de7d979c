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


def _solve_both(cost_blocks, q):
    """cost_blocks: list of (Q, G_b) float32 CPU tensors, one per sample (one layer).  Returns (device assignment, host
    assignment) as lists of (Q,) int64 with the per-sample column or -1, and the device status."""
    import numpy as np
    from graph_detr4d_amd import ops
    counts = [int(c.shape[1]) for c in cost_blocks]
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    sum_gt = int(start[-1])
    flat = torch.cat([c.contiguous().reshape(-1) for c in cost_blocks]) if sum_gt else torch.zeros(1)
    b = len(cost_blocks)
    assigned, status = ops.hungarian_assign_fwd(flat.cuda(), torch.from_numpy(start).cuda(), 1, b, q, sum_gt, max(counts + [0]))
    problems = [(q * int(start[i]), q, counts[i]) for i in range(b)]
    host = ops.linear_sum_assignment_batch(flat.numpy(), problems, num_threads=1)
    dev = assigned[0].cpu().long()
    dev_cols = [torch.where(dev[i] >= 0, dev[i] - int(start[i]), dev[i]) for i in range(b)]
    return dev_cols, [torch.from_numpy(np.asarray(h)).long() for h in host], status.cpu()


def test_device_assignment_equals_the_host_solver_on_a_thousand_problems():
    """gd4d_hungarian_assign_fwd against gd4d_linear_sum_assignment_batch (the scipy algorithm) - index work, so
    BIT-EXACT: 1000 random problems of 900 predictions x 0..100 boxes, in batches of 50 samples per launch; costs with the
    nan_to_num clamps (+-100) in them, with exact ties (costs rounded to a coarse grid, duplicated boxes, all-equal blocks)."""
    import numpy as np
    rng = np.random.default_rng(42)
    q = 900
    total = 0
    for batch in range(20):
        blocks = []
        for k in range(50):
            g = int(rng.integers(0, 101))
            kind = (batch * 50 + k) % 5
            c = rng.normal(0, 3, (q, g)).astype(np.float32)
            if kind == 1:                                   # coarse grid: many exact ties
                c = np.round(c * 2) / 2
            elif kind == 2 and g:                           # clamps: +100 / -100 entries (nan_to_num), whole clamped columns
                m = rng.random((q, g))
                c[m < 0.05] = 100.0
                c[m > 0.99] = -100.0
                c[:, rng.integers(0, g)] = 100.0
            elif kind == 3 and g > 1:                       # duplicated boxes: identical columns
                c[:, 1::2] = c[:, 0:-1:2][:, :c[:, 1::2].shape[1]]
            elif kind == 4:                                 # a constant block: everything ties
                c[:] = 1.5
            blocks.append(torch.from_numpy(np.ascontiguousarray(c)))
        dev, host, status = _solve_both(blocks, q)
        assert int(status.abs().sum()) == 0
        for d, h, blk in zip(dev, host, blocks):
            assert torch.equal(d, h), f'batch {batch}: device and host matchings differ ({blk.shape})'
            g = blk.shape[1]
            assert int((d >= 0).sum()) == min(g, q) and (g == 0 or sorted(d[d >= 0].tolist()) == list(range(g)))
            total += 1
    assert total == 1000


@pytest.mark.parametrize('q,g', [(7, 30), (64, 64), (1, 5), (5, 1), (130, 129), (2700, 60)])
def test_device_assignment_other_shapes(q, g):
    """More boxes than predictions (the untransposed orientation), square problems, H-DETR's 2700 queries; a NaN block is reported
    (status 1, nothing matched) - the marker gd4d_match_cost_fwd leaves for a label outside [0, classes)."""
    import numpy as np
    rng = np.random.default_rng(q * 1000 + g)
    blocks = [torch.from_numpy(np.round(rng.normal(0, 2, (q, g)).astype(np.float32) * 4) / 4) for _ in range(3)]
    dev, host, status = _solve_both(blocks, q)
    assert int(status.abs().sum()) == 0
    for d, h in zip(dev, host):
        assert torch.equal(d, h)
    bad = [b.clone() for b in blocks]
    bad[1][q // 2, g // 2] = float('nan')
    dev, _, status = _solve_both(bad, q)
    assert status.tolist() == [0, 1, 0] and int((dev[1] >= 0).sum()) == 0 and torch.equal(dev[0], host[0]) and torch.equal(dev[2], host[2])


@pytest.mark.parametrize('name', CASES)
def test_device_assigner_equals_the_host_route_on_the_reference_fixtures(name):
    """assign_layers (device route, the default) == assign_layers(host=True) on the reference's own head-loss fixtures."""
    from graph_detr4d_amd import HungarianAssigner3D
    g = Golden(name)
    cls, box = g.t('all_cls_scores').cuda(), g.t('all_bbox_preds').cuda()
    boxes, labels = _gt(g)
    asg = HungarianAssigner3D(cls_cost=dict(type='FocalLossCost', weight=2.0), reg_cost=dict(type='BBox3DL1Cost', weight=0.25),
                              iou_cost=dict(type='IoUCost', weight=0.0), pc_range=g.meta['pc_range'])
    a_dev = asg.assign_layers(cls, box, boxes, labels)
    asg.check_status()
    a_host = asg.assign_layers(cls, box, boxes, labels, host=True)
    assert torch.equal(a_dev.cpu(), a_host.cpu())


def test_a_label_out_of_range_raises_on_the_device_route_too():
    from graph_detr4d_amd import HungarianAssigner3D
    torch.manual_seed(0)
    cls, box = torch.randn(2, 1, 20, 10).cuda(), torch.randn(2, 1, 20, 10).cuda()
    gt = torch.randn(3, 9)
    gt[:, 3:6] = gt[:, 3:6].abs() + 0.3
    asg = HungarianAssigner3D(cls_cost=dict(type='FocalLossCost', weight=2.0), reg_cost=dict(type='BBox3DL1Cost', weight=0.25))
    a = asg.assign_layers(cls, box, [gt.cuda()], [torch.tensor([1, 12, 3]).cuda()])
    assert int((a >= 0).sum()) == 0
    with pytest.raises(IndexError):
        asg.check_status()
    with pytest.raises(IndexError):
        asg.assign(box[0, 0], cls[0, 0], gt.cuda(), torch.tensor([1, 12, 3]).cuda())


def test_flat_adamw_with_clipping_equals_torch():
    """dist.FlatGradAllReducer.adamw_step (gd4d_adamw_flat: clip_grad_norm_(35) + AdamW lr 2e-4 wd 0.01, the reference's recipe,
    ...ceph.py:205-213) against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over five steps, with gradients large enough
    for the clip to bite and small enough not to; the norm it reports = torch's."""
    import copy
    from graph_detr4d_amd import dist as D
    torch.manual_seed(0)
    # (no normalisation after the last Linear: sum(LayerNorm(.)^2) is a constant, its gradient rounding noise that Adam would amplify)
    net = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.LayerNorm(96), torch.nn.ReLU(), torch.nn.Linear(96, 33)).cuda()
    ref = copy.deepcopy(net)
    red = D.FlatGradAllReducer(list(net.parameters()), align=4)
    red.bind()
    opt = torch.optim.AdamW(ref.parameters(), lr=2e-4, weight_decay=0.01)
    for step in range(5):
        x = torch.randn(16, 64, device='cuda') * (300.0 if step % 2 == 0 else 0.01)
        red.zero_grad()
        opt.zero_grad()
        net(x).square().sum().backward()
        ref(x).square().sum().backward()
        want_norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 35.0)
        opt.step()
        red.adamw_step(lr=2e-4, weight_decay=0.01, max_norm=35.0)
        torch.testing.assert_close(red.last_grad_norm, want_norm, rtol=1e-5, atol=0)
        for p, r in zip(net.parameters(), ref.parameters()):
            torch.testing.assert_close(p, r, rtol=1e-5, atol=2e-7)


def test_flat_adamw_inside_a_replayed_graph_keeps_its_state():
    """adamw_step allocates its state lazily: inside a capture the zero-fills would be recorded and EVERY replay would reset m, v and
    the step counter (each step Adam's first).  A first call inside a capture raises; with adamw_state() before the capture, N
    replays equal N eager steps; after_replays() bumps the parameter versions the weight-image caches are keyed by."""
    import copy
    from graph_detr4d_amd import dist as D
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.ReLU(), torch.nn.Linear(96, 32)).cuda()
    ref = copy.deepcopy(net)
    x = torch.randn(16, 64, device='cuda')
    red, red_ref = D.FlatGradAllReducer(list(net.parameters()), align=4), D.FlatGradAllReducer(list(ref.parameters()), align=4)
    red.bind()
    red_ref.bind()

    def step(model, r):
        r.zero_grad()
        model(x).square().sum().backward()
        r.adamw_step(lr=1e-2, weight_decay=0.01, max_norm=35.0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net(x).square().sum().backward()                          # warm-up of autograd on the capture stream (no optimizer step)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g_bad = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match='adamw_state'):
        with torch.cuda.graph(g_bad, capture_error_mode='thread_local'):
            step(net, red)
    torch.cuda.synchronize()
    red.adamw_state()
    versions = [p._version for p in net.parameters()]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        step(net, red)
    for _ in range(4):
        graph.replay()
    for _ in range(4):
        step(ref, red_ref)
    torch.cuda.synchronize()
    for p, r in zip(net.parameters(), ref.parameters()):
        torch.testing.assert_close(p, r, rtol=1e-5, atol=1e-7)
    assert int(red._adam[2][0].item()) == 4                        # the device step counter advanced with every replay
    red.after_replays()
    assert all(p._version > v for p, v in zip(net.parameters(), versions))

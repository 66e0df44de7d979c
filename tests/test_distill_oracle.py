"""The instance distillation loss of the teacher - student step (BASELINE configs[4]) against vectors captured from the
reference's own `MixDistill.get_instance_distill_loss` (distillation/distillers/mix_distill.py:140-168, imported
unmodified by tools/gen_golden.py::case_distill): the oracle's stage-by-stage restatement and the package's one-pass
form (`criterion.instance_distill_loss`, plain torch ops - what `bench.py --mode distill` adds to the student's loss),
values and gradients.  Runs on CPU; the GPU run of the same function is in tests/test_configs_gpu.py."""
import pytest
import torch

from golden_io import Golden


def _run(fn_kind, g, device='cpu'):
    m = g.meta
    t_cls, t_box = g.t('t_cls').to(device), g.t('t_box').to(device)
    s_cls, s_box = g.t('s_cls').to(device).requires_grad_(), g.t('s_box').to(device).requires_grad_()
    kw = dict(loss_cls_weight=m['loss_cls_weight'], loss_reg_weight=m['loss_reg_weight'], reweight_score=m['reweight_score'])
    if fn_kind == 'oracle':
        from oracle import torch_oracle as O
        losses = O.instance_distill_loss(t_cls, t_box, s_cls, s_box, **kw)
    else:
        from graph_detr4d_amd.criterion import instance_distill_loss
        losses = instance_distill_loss(dict(all_cls_scores=t_cls, all_bbox_preds=t_box),
                                       dict(guided_cls_scores=s_cls, guided_bbox_preds=s_box), **kw)
    assert list(losses.keys()) == m['loss_keys']                       # the reference's names, in the reference's order
    sum(losses.values()).backward()
    return torch.stack([losses[k].detach() for k in m['loss_keys']]).cpu(), s_cls.grad.cpu(), s_box.grad.cpu()


@pytest.mark.parametrize('name', ['distill_loss', 'distill_loss_b2_mean'])
@pytest.mark.parametrize('fn_kind', ['oracle', 'package'])
def test_instance_distill_loss_matches_reference(name, fn_kind):
    g = Golden(name)
    losses, g_cls, g_box = _run(fn_kind, g)
    tol = dict(rtol=1e-6, atol=1e-7) if fn_kind == 'oracle' else dict(rtol=1e-5, atol=1e-7)   # one-pass sums: other order
    torch.testing.assert_close(losses, g.t('losses'), **tol)
    torch.testing.assert_close(g_cls, g.t('grad_s_cls'), **tol)
    torch.testing.assert_close(g_box, g.t('grad_s_box'), **tol)
    assert float(g.t('losses').abs().min()) > 0


def test_teacher_side_gets_no_gradient():
    from graph_detr4d_amd.criterion import instance_distill_loss
    g = Golden('distill_loss')
    t_cls, t_box = g.t('t_cls').requires_grad_(), g.t('t_box').requires_grad_()
    s_cls, s_box = g.t('s_cls').requires_grad_(), g.t('s_box').requires_grad_()
    losses = instance_distill_loss(dict(all_cls_scores=t_cls, all_bbox_preds=t_box), dict(guided_cls_scores=s_cls, guided_bbox_preds=s_box))
    sum(losses.values()).backward()
    assert t_cls.grad is None and t_box.grad is None and s_cls.grad is not None      # mix_distill.py:150 detaches the teacher

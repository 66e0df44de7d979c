"""Hungarian assignment and the head's per-layer losses: the training-side step right after the path (SURVEY.md §8f
rank 4).

Mirrors of
  * `HungarianAssigner3D` (projects/mmdet3d_plugin/core/bbox/assigners/hungarian_assigner_3d.py:25-144): same
    constructor keywords and `assign(bbox_pred, cls_pred, gt_bboxes, gt_labels)` contract, plus `assign_layers` for all
    decoder layers and samples at once;
  * `Detr3DHeadPE.loss` / `loss_single` (projects/mmdet3d_plugin/models/dense_heads/detr3d_head_pe.py:782-845,
    1014-1094): `Detr3DCriterion.loss(gt_bboxes_list, gt_labels_list, preds_dicts)` returns the same dictionary
    (`loss_cls`, `loss_bbox`, `d{i}.loss_cls`, `d{i}.loss_bbox`).

What changes is the schedule.  The reference walks the decoder layers one by one and in each: builds the cost matrix
with ~15 small kernels, copies it to the host (a device synchronisation), runs scipy, copies the matches back, builds
targets, all-reduces two scalars and calls .item() on one of them (another synchronisation).  Here one launch
(gd4d_match_cost_fwd) produces every layer's cost matrix, ONE copy brings them to the host,
gd4d_linear_sum_assignment_batch (the algorithm scipy ships, in the library) solves them, ONE copy
returns the matches, one launch (gd4d_head_loss_fwd_bwd) produces all the loss terms and their gradients.  The two
normalisers depend on the ground-truth counts only (every ground-truth box is matched exactly once when Q >= G), so
they are known before the forward pass and need one 2-element all-reduce per step instead of twelve scalar ones.
"""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .registry import BBOX_ASSIGNERS


class AssignResult:
    """What HungarianAssigner3D.assign returns (mmdet's AssignResult: num_gts, gt_inds, max_overlaps, labels)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


def _weight(cfg, default):
    return float((cfg or {}).get('weight', default))


def gt_tensor(gt_bboxes):
    """`loss` takes LiDARInstance3DBoxes (:1059-1061: gravity centre + the remaining columns); plain (G, 9) tensors in
    that form pass through."""
    if torch.is_tensor(gt_bboxes):
        return gt_bboxes
    return torch.cat((gt_bboxes.gravity_center, gt_bboxes.tensor[:, 3:]), dim=1)


def pack_ground_truth(gt_bboxes_list, gt_labels_list, device):
    """Concatenate the per-sample ground truth for the kernels: boxes (sumG, D) fp32, labels (sumG) int32, prefix
    offsets (B + 1) int32 on the device and as a host list.  None when there is no ground truth at all."""
    counts = [int(g.shape[0]) for g in gt_bboxes_list]
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    if start[-1] == 0:
        return None
    boxes = torch.cat([gt_tensor(g).to(device=device, dtype=torch.float32) for g in gt_bboxes_list]).contiguous()
    labels = torch.cat([lab.to(device=device, dtype=torch.int32) for lab in gt_labels_list]).contiguous()
    return boxes, labels, torch.from_numpy(start).to(device), start, counts


@BBOX_ASSIGNERS.register_module()
class HungarianAssigner3D:
    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), reg_cost=dict(type='BBoxL1Cost', weight=1.0),
                 iou_cost=dict(type='IoUCost', weight=0.0), pc_range=None):
        if (cls_cost or {}).get('type', 'FocalLossCost') != 'FocalLossCost' or \
                (reg_cost or {}).get('type', 'BBox3DL1Cost') != 'BBox3DL1Cost':
            raise NotImplementedError('gd4d_match_cost_fwd implements the shipped configs\' costs: FocalLossCost + '
                                      'BBox3DL1Cost (…ceph.py:131-136)')
        self.cls_weight = _weight(cls_cost, 1.)
        self.reg_weight = _weight(reg_cost, 1.)
        self.alpha = float((cls_cost or {}).get('alpha', 0.25))
        if float((cls_cost or {}).get('gamma', 2)) != 2.0:
            raise NotImplementedError('FocalLossCost gamma is fixed at 2 in the kernel')
        self.pc_range = pc_range

    def assign_layers(self, all_cls_scores, all_bbox_preds, gt_bboxes_list, gt_labels_list, packed=None, host=False):
        """All decoder layers and samples at once.  all_cls_scores (NL, B, Q, C), all_bbox_preds (NL, B, Q, code);
        per-sample ground truth.  Returns assigned (NL, B, Q) int32 on the device: index into the concatenated ground
        truth (pack_ground_truth) or -1 for background.
        Default: two launches (gd4d_match_cost_fwd, gd4d_hungarian_assign_fwd) and NO host synchronisation - the matching is
        the host solver's, bit for bit (ties included).  A label outside [0, num_classes) - the reference's indexing raises on it
        (core/bbox/match_costs/match_cost.py:17-30) - leaves its sample unmatched and is reported by check_status(), which the
        next call polls without blocking (and which raises IndexError).  host=True: round 4's route (cost matrix to the host,
        gd4d_linear_sum_assignment_batch, back) - kept as the comparison the tests make."""
        nl, b, q, _ = all_cls_scores.shape
        dev = all_cls_scores.device
        packed = packed or pack_ground_truth(gt_bboxes_list, gt_labels_list, dev)
        if packed is None or q == 0:
            return torch.full((nl, b, q), -1, dtype=torch.int32, device=dev)
        boxes, labels, start_dev, start, counts = packed
        cost = ops.match_cost_fwd(all_cls_scores.detach().contiguous().float(), all_bbox_preds.detach().contiguous().float(),
                                  boxes, labels, start_dev, max(counts), self.cls_weight, self.reg_weight, self.alpha)
        sum_gt = int(start[-1])
        if not host:
            self.poll_status()
            assigned, status = ops.hungarian_assign_fwd(cost, start_dev, nl, b, q, sum_gt, max(counts))
            self._status = (status, all_cls_scores.shape[-1])
            return assigned
        cost = cost.cpu().numpy()                            # THE synchronisation of the step
        if np.isnan(cost).any():                             # the kernel's marker for a label outside [0, num_classes)
            raise IndexError(f'gt_labels must lie in [0, {all_cls_scores.shape[-1]}): the reference indexes '
                             'cls_pred[:, gt_labels] with them (core/bbox/match_costs/match_cost.py:17-30)')
        problems = [(q * (l * sum_gt + int(start[i])), q, counts[i]) for l in range(nl) for i in range(b)]
        # host threads pay off only when a problem is worth more than starting one (measured: 6 x (900 x 45) takes
        # 1.0 ms on one thread, 2.5 ms on six)
        work = max(min(q, g) ** 2 * max(q, g) for g in counts)
        matched = ops.linear_sum_assignment_batch(cost, problems, num_threads=min(len(problems), 8) if work > 2e7 else 1)
        assigned = np.stack(matched).reshape(nl, b, q)
        for i in range(b):                                   # per-sample column index -> index into the packed arrays
            if start[i]:
                a = assigned[:, i]
                a[a >= 0] += int(start[i])
        return torch.from_numpy(assigned).to(dev, non_blocking=True)

    _status = None          # (status tensor of the last device assignment, num_classes)
    _pending = None         # (pinned copy, event, num_classes) requested by poll_status
    _pinned = None          # the pinned host buffer poll_status copies into, kept

    def check_status(self):
        """Blocking: raises if the last device assignment met a label outside [0, num_classes) (IndexError, as the reference's
        indexing) or an infeasible problem."""
        if self._status is None:
            return
        status, ncls = self._status
        self._status = self._pending = None
        self._raise_for(status.cpu(), ncls)

    @staticmethod
    def _raise_for(st, ncls):
        if bool((st == 1).any()):
            raise IndexError(f'gt_labels must lie in [0, {ncls}): the reference indexes cls_pred[:, gt_labels] with them '
                             '(core/bbox/match_costs/match_cost.py:17-30)')
        if bool((st == 2).any()):
            raise ops._lib.Gd4dError('gd4d_hungarian_assign_fwd: an assignment problem was infeasible')

    def poll_status(self):
        """Non-blocking: looks at the copy of the previous call's status if it has arrived, requests one of the current."""
        if torch.cuda.is_current_stream_capturing():
            return
        if self._pending is not None and self._pending[1].query():
            pinned, _, ncls = self._pending
            self._pending = None
            self._raise_for(pinned, ncls)
        if self._pending is None and self._status is not None:
            status, ncls = self._status
            pinned = self._pinned                                     # ONE pinned buffer per assigner (pin_memory() is a slow path)
            if pinned is None or pinned.shape != status.shape or pinned.dtype != status.dtype:
                pinned = self._pinned = torch.empty(status.shape, dtype=status.dtype).pin_memory()
            pinned.copy_(status, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (pinned, ev, ncls)

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_bboxes_ignore=None, eps=1e-7):
        """The reference's per-layer, per-sample entry point (:62-144): gt_inds 0 = background, g + 1 = matched."""
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_gts, q = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((q,), -1, dtype=torch.long)
        lab = bbox_pred.new_full((q,), -1, dtype=torch.long)
        if num_gts == 0 or q == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=lab)
        a = self.assign_layers(cls_pred[None, None], bbox_pred[None, None], [gt_bboxes], [gt_labels])[0, 0].long()
        self.check_status()                                              # (the reference's entry point raises where it stands)
        gt_inds = a + 1                                                  # -1 -> 0 background, g -> g + 1
        pos = a >= 0
        lab[pos] = gt_labels[a[pos]].long()
        return AssignResult(num_gts, gt_inds, None, labels=lab)


class _HeadLossFunction(torch.autograd.Function):
    """loss (NL, 2) from gd4d_head_loss_fwd_bwd; the gradients come out of the same launch."""

    @staticmethod
    def forward(ctx, cls, box, assigned, boxes, labels, code_weights, avg, alpha, wc, wb):
        loss, gcls, gbox = ops.head_loss_fwd_bwd(cls.contiguous(), box.contiguous(), assigned, boxes, labels,
                                                 code_weights, avg, alpha, wc, wb)
        ctx.save_for_backward(gcls, gbox)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        gcls, gbox = ctx.saved_tensors
        gl = grad_loss.view(-1, 2)
        return (gcls * gl[:, 0].view(-1, 1, 1, 1), gbox * gl[:, 1].view(-1, 1, 1, 1)) + (None,) * 8


class Detr3DCriterion(nn.Module):
    """The loss part of Detr3DHeadPE (:303-424 constructor keywords that matter here, `loss` :1014-1094)."""

    def __init__(self, num_classes=10, code_weights=None, sync_cls_avg_factor=True, bg_cls_weight=0.0,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
                 loss_bbox=dict(type='L1Loss', loss_weight=0.25), assigner=None, pc_range=None):
        super().__init__()
        if loss_cls.get('type') != 'FocalLoss' or not loss_cls.get('use_sigmoid', True) or \
                float(loss_cls.get('gamma', 2.0)) != 2.0 or loss_bbox.get('type') != 'L1Loss':
            raise NotImplementedError('gd4d_head_loss_fwd_bwd implements the shipped configs\' losses: sigmoid FocalLoss '
                                      '(gamma 2) + L1Loss (…ceph.py:117-123)')
        self.num_classes = num_classes
        self.sync_cls_avg_factor, self.bg_cls_weight = sync_cls_avg_factor, float(bg_cls_weight)
        self.alpha = float(loss_cls.get('alpha', 0.25))
        self.loss_cls_weight = float(loss_cls.get('loss_weight', 1.0))
        self.loss_bbox_weight = float(loss_bbox.get('loss_weight', 1.0))
        cw = code_weights if code_weights is not None else [1.0] * 8 + [0.2, 0.2]             # :337-341
        self.code_weights = nn.Parameter(torch.tensor(cw, dtype=torch.float32), requires_grad=False)
        if assigner is None:
            assigner = dict(type='HungarianAssigner3D', cls_cost=dict(type='FocalLossCost', weight=2.0),
                            reg_cost=dict(type='BBox3DL1Cost', weight=0.25), iou_cost=dict(type='IoUCost', weight=0.0),
                            pc_range=pc_range)
        if isinstance(assigner, dict):
            assigner = dict(assigner)
            assigner.pop('type', None)
            assigner = HungarianAssigner3D(**assigner)
        self.assigner = assigner

    def normalisers(self, counts, num_query, device):
        """(cls_avg_factor, num_total_pos) as a 2-element device tensor, averaged over the ranks (`reduce_mean`,
        :825-826 and :835) - known from the ground-truth counts alone: one collective per step, no .item()."""
        num_pos = float(sum(min(c, num_query) for c in counts))
        num_neg = float(len(counts) * num_query) - num_pos
        avg = torch.tensor([num_pos + num_neg * self.bg_cls_weight, num_pos], dtype=torch.float32, device=device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            local_cls = avg[0].clone()
            dist.all_reduce(avg)
            avg /= dist.get_world_size()
            if not self.sync_cls_avg_factor:                 # :824: only num_total_pos is always synchronised
                avg[0] = local_cls
        return avg

    def prepare_ground_truth(self, gt_bboxes_list, gt_labels_list, num_query, device):
        """What `loss` derives from the ground truth alone - the packed boxes / labels / offsets on the device and the two
        normalisers - as one object to hand back to it (`prepared=`): a step whose ground truth is resident (a captured hipGraph
        cannot contain the host -> device copies) computes it once, outside the capture."""
        gts = [gt_tensor(g) for g in gt_bboxes_list]
        # The labels are checked HERE, once, where a host look is acceptable (outside any capture): on the device a label outside
        # [0, num_classes) is only reported through the assigner's status word - a step or more later, after an optimizer step that
        # treated the sample as all background, and inside a replayed graph only if its owner calls check_status().
        for lab in gt_labels_list:
            if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) >= self.num_classes):
                raise IndexError(f'gt_labels must lie in [0, {self.num_classes}): the reference indexes cls_pred[:, gt_labels] with them '
                                 '(core/bbox/match_costs/match_cost.py:17-30)')
        packed = pack_ground_truth(gts, gt_labels_list, device)
        return gts, packed, self.normalisers([int(g.shape[0]) for g in gts], num_query, device)

    def loss(self, gt_bboxes_list, gt_labels_list, preds_dicts, gt_bboxes_ignore=None, prepared=None):
        assert gt_bboxes_ignore is None, f'{self.__class__.__name__} only supports for gt_bboxes_ignore setting to None.'
        if preds_dicts.get('enc_cls_scores') is not None:
            raise NotImplementedError('two-stage proposals are not used by the shipped configs')
        cls, box = preds_dicts['all_cls_scores'], preds_dicts['all_bbox_preds']
        nl, b, q, c = cls.shape
        dev = cls.device
        gts, packed, avg = prepared if prepared is not None else self.prepare_ground_truth(gt_bboxes_list, gt_labels_list, q, dev)
        if packed is None:                                   # no ground truth anywhere: every query is background
            boxes = torch.ones(1, 9, device=dev)
            labels = torch.zeros(1, dtype=torch.int32, device=dev)
            assigned = torch.full((nl, b, q), -1, dtype=torch.int32, device=dev)
        else:
            boxes, labels = packed[0], packed[1]
            assigned = self.assigner.assign_layers(cls, box, gts, gt_labels_list, packed)
        loss = _HeadLossFunction.apply(cls.float(), box.float(), assigned, boxes, labels, self.code_weights, avg,
                                       self.alpha, self.loss_cls_weight, self.loss_bbox_weight)
        out = {'loss_cls': loss[-1, 0], 'loss_bbox': loss[-1, 1]}
        for i in range(nl - 1):
            out[f'd{i}.loss_cls'], out[f'd{i}.loss_bbox'] = loss[i, 0], loss[i, 1]
        self.last_assigned = assigned
        return out

    forward = loss


def instance_distill_loss(teacher_outs, student_outs, loss_cls_weight=1.0, loss_reg_weight=1.0, reweight_score=True):
    """The instance distillation terms of the teacher - student step (BASELINE configs[4]):
    `MixDistill.get_instance_distill_loss` (projects/mmdet3d_plugin/distillation/distillers/mix_distill.py:140-168).

    teacher_outs: dict with `all_cls_scores` (NL, B, Q, classes) and `all_bbox_preds` (NL, B, Q, code) of the teacher's head
    (detached here, as :150 does); student_outs: dict with `guided_cls_scores` / `guided_bbox_preds` - the student's head on
    the TEACHER's queries (detr3d_head_pe.py:617-625).  Per decoder stage: BCE-with-logits of the student's logits against
    the teacher's sigmoid scores and L1 between the box codes, both weighted per query by the teacher's best class score
    and normalised by `sum(q_score) * classes + 1e-10` (reweight_score, :160-162; note the SAME normaliser for the box term
    although it has `code` columns), or plain means.  Returns the reference's dictionary
    (`distill_loss_cls.{i}`, `distill_loss_reg.{i}`).  All stages in one pass of torch ops (the reference loops)."""
    t_cls = teacher_outs['all_cls_scores'].detach().sigmoid()
    t_box = teacher_outs['all_bbox_preds'].detach()
    s_cls, s_box = student_outs['guided_cls_scores'], student_outs['guided_bbox_preds']
    nl, ncls = t_cls.shape[0], t_cls.shape[-1]
    bce = torch.nn.functional.binary_cross_entropy_with_logits(s_cls, t_cls, reduction='none')
    l1 = (s_box - t_box).abs()
    if reweight_score:
        qs = t_cls.max(dim=-1, keepdim=True)[0]
        den = qs.flatten(1).sum(1) * ncls + 1e-10
        cls_terms = (qs * bce).flatten(1).sum(1) / den
        reg_terms = (qs * l1).flatten(1).sum(1) / den
    else:
        cls_terms, reg_terms = bce.flatten(1).mean(1), l1.flatten(1).mean(1)
    out = {}
    for i in range(nl):
        out[f'distill_loss_cls.{i}'] = cls_terms[i] * loss_cls_weight
        out[f'distill_loss_reg.{i}'] = reg_terms[i] * loss_reg_weight
    return out

"""Deform3DCrossAttnMP: the multi-point variant of the 3-D deformable cross-attention.

Mirror of projects/mmdet3d_plugin/models/utils/deform3d_cross_attn_multi_point.py:35-453 (config
projects/configs/detr4d/detr4d_res50_deform_pe_mp_testaug_2subset_12e.py:76): same constructor keywords, `forward`
signature and state-dict keys (`attention_weights_neighbor`, `deform_sampling_offsets_neighbor`, `output_weight` on top
of Deform3DCrossAttn's).  Differences of the reference that are reproduced:
  * `query_pos` is NOT added to the query (:211-222);
  * `reference_points` carries 9 points per query: Q centres, then 8 blocks of Q neighbour points (:228, :373);
  * the neighbour pass samples every neighbour point once per level, without offsets (:384-417), with attention logits
    taken from a raw view of a (Q, 256) Linear output as (8Q, heads, levels) (:375-376), camera weights that are NOT
    passed through the sigmoid (:424-430), summed over cameras and neighbours (:431-433);
  * centre and neighbour results are blended with a 2-way softmax whose logits are summed over the queries, and the
    weights of sample 0 are used for the whole batch (:435-438).

Both passes run on the RAW pyramid, like Deform3DCrossAttn's (aggregate-then-project: functional.LateValues in inference,
autograd.CrossAttnRawFunction in training - value_proj (:339-341, :415-417) is applied to the per-head aggregates, not to
the pixels): the neighbour pass as 8Q pseudo-queries with zero offsets in the kernels' one-point-per-level form and
GD4D_CA_RAW_CAM_WEIGHTS; in inference its 8 blocks are summed BEFORE the projection (value_proj is linear: Q rows
instead of 8Q).  The training kernels take four points per head: the neighbour pass is padded with three points that are
never visible and weigh nothing (functional.pad_points).  GD4D_PROJECT=early / GD4D_TRAIN_VALUES=projected keep the
projected-value kernels (gd4d_value_proj_fwd + gd4d_cross_attn_fwd / _bwd).
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .deform3d_cross_attn import Deform3DCrossAttn
from .registry import ATTENTION


@ATTENTION.register_module()
class Deform3DCrossAttnMP(Deform3DCrossAttn):
    def __init__(self, *args, multi_points=True, **kwargs):
        super().__init__(*args, **kwargs)
        if self.num_points != 4:
            raise ValueError('Deform3DCrossAttnMP: num_points must be 4 (the shipped config; the kernels sample 4 points)')
        self.multi_points = multi_points
        if multi_points:
            e = self.embed_dims
            self.attention_weights_neighbor = nn.Linear(e, self.num_heads * self.num_levels * self.num_points * 8 // 4)
            self.deform_sampling_offsets_neighbor = nn.Linear(e, self.num_heads * 1 * self.num_points * 3)   # unused (:386-389)
            self.output_weight = nn.Linear(e * 2, 2)

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        """query (Q, B, C); value = list of L maps (B, N, C, H_l, W_l); reference_points (B, 9Q, 3) in [0,1] (Q when
        multi_points=False); kwargs['img_metas'] mandatory.  Returns (Q, B, C)."""
        if residual is not None:
            raise NameError('Deform3DCrossAttnMP: residual must be None (as in the reference, :218-219)')
        if value is None or torch.is_tensor(value):
            raise TypeError('value must be the list of multi-camera feature maps (B, N, C, H, W)')
        img_metas = kwargs['img_metas']
        Fn.require_gpu(query, 'query')
        if Fn.wants_grad(self, query, reference_points, *value):
            cached = (kwargs.get(Fn.VALUE_CACHE_KEY) or {}).get(id(self))
            hit = cached is not None and cached[2] is value and len(cached) > 3
            return self._forward_autograd(query, value, reference_points, img_metas, cached[3] if hit else None)
        q, b, c = query.shape
        hh, npt, nl, n = self.num_heads, self.num_points, self.num_levels, self.num_cams
        if len(value) != nl or value[0].shape[1] != n:
            raise ValueError(f'expected {nl} levels x {n} cameras')
        want = 9 * q if self.multi_points else q
        if reference_points.shape[1] != want:
            raise ValueError(f'reference_points must hold {want} points ({q} queries), got {reference_points.shape[1]}')
        xq = query if b == 1 else query.permute(1, 0, 2).contiguous()         # rows in (B, Q) order; no query_pos
        mods = [self.cam_attention_weights, self.deform_sampling_offsets, self.attention_weights]
        if self.multi_points:
            mods.append(self.attention_weights_neighbor)
        outs = ops.linear_group_fwd(xq.contiguous(), [m_.weight.contiguous() for m_ in mods], [m_.bias for m_ in mods])
        cam_logits = outs[0].view(b, q, n)
        offsets = outs[1].view(b, q, hh, npt, 3)
        attn_logits = outs[2].view(b, q, hh, nl, npt)
        lidar2img = Fn.lidar2img_device(img_metas, query)
        img_h, img_w = Fn.img_hw(img_metas)
        centre = reference_points[:, :q].contiguous()
        order = Fn.query_order(centre, self.pc_range)
        late = kwargs.get(Fn.LATE_VALUES_KEY)
        if late is not None and late.value is not value:
            late = None
        own = None
        if late is None and Fn.LateValues.applicable([self], value):
            late = own = Fn.LateValues(value, self.value_dtype)   # a stand-alone call: its own channels-last copy
        if late is not None:
            agg = late.sample_aggregate(self, centre, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=order)
        else:
            val, shapes = Fn.value_projection(value, self.value_proj.weight, self.value_proj.bias, hh, self.value_dtype)
            agg = Fn.sample_aggregate(val, shapes, centre, offsets, attn_logits, cam_logits, lidar2img, self.pc_range,
                                      img_h, img_w, order=order)
        if self.multi_points:
            nbr = reference_points[:, q:].contiguous()                          # (B, 8Q, 3), block j = neighbour j
            # raw view of the (B, Q, 256) logits as (B, 8Q, heads, levels): one point per level (the kernel's P = 1 form)
            logits_n = outs[3].view(b, 8 * q, hh, nl, 1)
            zero_off = torch.zeros(b, 8 * q, hh, 1, 3, device=query.device)
            cam_n = cam_logits.repeat(1, 8, 1).contiguous()                     # cam_attention_weights(query.repeat(1,8,1))
            if late is not None:
                # the 8 blocks are summed before value_proj (linear in the aggregates and in the weight sums that carry its bias)
                raw_n, wsum_n = late.aggregate(self, nbr, zero_off, logits_n, cam_n, lidar2img, img_h, img_w,
                                               order=Fn.query_order(nbr, self.pc_range), raw_cam_weights=True)
                bias = self.value_proj.bias
                agg_n = ops.value_proj_heads_fwd(raw_n.view(b, 8, q, hh, c).sum(1), wsum_n.view(b, 8, q, hh).sum(1),
                                                 self.value_proj.weight.contiguous(), None if bias is None else bias.contiguous())
            else:
                head_major = val.shape[2] == sum(h * w for h, w in shapes) and val.shape[1] != val.shape[2]
                agg_n = ops.cross_attn_fwd(val, shapes, nbr, zero_off, logits_n, cam_n, lidar2img,
                                           self.pc_range, img_h, img_w, head_major=head_major, raw_cam_weights=True)
                agg_n = agg_n.view(b, 8, q, c).sum(1)
            blend = Fn.linear(torch.cat([agg, agg_n], -1), self.output_weight.weight, self.output_weight.bias)
            wts = blend.sum(1).softmax(-1)                                       # (B, 2); sample 0's are used (:438)
            agg = agg * wts[0][0] + agg_n * wts[0][1]
        if own is not None:
            own.finish()
        pos_feat = Fn.position_encoder(self.position_encoder, centre)
        if b == 1 and not self.training:
            return Fn.linear(agg.contiguous(), self.output_proj.weight, self.output_proj.bias,
                             r1=query.view(1, q, c), r2=pos_feat).view(q, 1, c)
        out = Fn.linear(agg.contiguous(), self.output_proj.weight, self.output_proj.bias).permute(1, 0, 2)
        return self.dropout(out) + query + pos_feat.permute(1, 0, 2)

    def _forward_autograd(self, query, value, reference_points, img_metas, cl=None):
        """Training path: the same maths with autograd.  On the raw pyramid (cl = (RawPyramid, token) from the decoder, or this
        call's own): both passes are autograd.CrossAttnRawFunction nodes; else the gathers on gd4d_cross_attn_fwd / _bwd and
        value_proj on the HIP forward with its GEMM backward.  Dense layers and the blend are differentiable ops."""
        from .autograd import CrossAttnFunction, CrossAttnRawFunction, ValueProjFunction
        q, b, c = query.shape
        hh, npt, nl, n = self.num_heads, self.num_points, self.num_levels, self.num_cams
        if len(value) != nl or value[0].shape[1] != n:
            raise ValueError(f'expected {nl} levels x {n} cameras')
        want = 9 * q if self.multi_points else q
        if reference_points.shape[1] != want:
            raise ValueError(f'reference_points must hold {want} points ({q} queries), got {reference_points.shape[1]}')
        if self.value_dtype != torch.float32:
            raise NotImplementedError('training needs value_dtype="fp32"')
        x = query.permute(1, 0, 2).contiguous()                               # (B, Q, C); no query_pos (:211-222)
        cam_logits = Fn.sequential_autograd(self.cam_attention_weights, x)
        offsets = Fn.sequential_autograd(self.deform_sampling_offsets, x).view(b, q, hh, npt, 3)
        attn_logits = Fn.sequential_autograd(self.attention_weights, x).view(b, q, hh, nl, npt)
        shapes = [tuple(v.shape[-2:]) for v in value]
        lidar2img = Fn.lidar2img_device(img_metas, query)
        img_h, img_w = Fn.img_hw(img_metas)
        centre = reference_points[:, :q].contiguous()
        if cl is None:
            own = Fn.raw_pyramid_for_training([self], value)
            cl = None if own is None else own[id(self)][3]
        if isinstance(cl, tuple):
            raw, token = cl
            w_v, b_v = self.value_proj.weight, self.value_proj.bias
            agg = CrossAttnRawFunction.apply(token, centre, offsets, attn_logits, cam_logits, lidar2img, w_v, b_v, raw,
                                             self.pc_range, img_h, img_w)
            if self.multi_points:
                nbr = reference_points[:, q:].contiguous()
                logits_n = Fn.sequential_autograd(self.attention_weights_neighbor, x).reshape(b, 8 * q, hh, nl, 1)
                zero_off, logits_n = Fn.pad_points(torch.zeros(b, 8 * q, hh, 1, 3, device=query.device), logits_n, (4,),
                                                   'Deform3DCrossAttnMP (training)')
                agg_n = CrossAttnRawFunction.apply(token, nbr, zero_off, logits_n, cam_logits.repeat(1, 8, 1), lidar2img, w_v, b_v,
                                                   raw, self.pc_range, img_h, img_w, True)
                agg_n = agg_n.view(b, 8, q, c).sum(1)
                blend = Fn.sequential_autograd(self.output_weight, torch.cat([agg, agg_n], -1))
                wts = blend.sum(1).softmax(-1)                                   # (B, 2); sample 0's are used (:438)
                agg = agg * wts[0][0] + agg_n * wts[0][1]
            pos_feat = Fn.sequential_autograd(self.position_encoder, Fn.inverse_sigmoid(centre)).permute(1, 0, 2)
            out = Fn.sequential_autograd(self.output_proj, agg).permute(1, 0, 2)
            return self.dropout(out) + query + pos_feat
        val = ValueProjFunction.apply(self.value_proj.weight, self.value_proj.bias, *value)
        val = val.view(val.shape[0], -1, hh, c // hh)
        agg = CrossAttnFunction.apply(val, centre, offsets, attn_logits, cam_logits, lidar2img, shapes, self.pc_range,
                                      img_h, img_w)
        if self.multi_points:
            nbr = reference_points[:, q:].contiguous()
            logits_n = Fn.sequential_autograd(self.attention_weights_neighbor, x).reshape(b, 8 * q, hh, nl, 1)
            logits_n = torch.nn.functional.pad(logits_n, (0, npt - 1), value=-1e30)     # three points of zero weight
            zero_off = torch.zeros(b, 8 * q, hh, npt, 3, device=query.device)
            agg_n = CrossAttnFunction.apply(val, nbr, zero_off, logits_n, cam_logits.repeat(1, 8, 1), lidar2img, shapes,
                                            self.pc_range, img_h, img_w, None, None, None, True)
            agg_n = agg_n.view(b, 8, q, c).sum(1)
            blend = Fn.sequential_autograd(self.output_weight, torch.cat([agg, agg_n], -1))
            wts = blend.sum(1).softmax(-1)                                       # (B, 2); sample 0's are used (:438)
            agg = agg * wts[0][0] + agg_n * wts[0][1]
        pos_feat = Fn.sequential_autograd(self.position_encoder, Fn.inverse_sigmoid(centre)).permute(1, 0, 2)
        out = Fn.sequential_autograd(self.output_proj, agg).permute(1, 0, 2)
        return self.dropout(out) + query + pos_feat

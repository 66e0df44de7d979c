"""`Deform3DCrossAttn` - Graph-DETR4D's decoder cross-attention, MI355X-native.

Drop-in for the reference class of the same name (projects/mmdet3d_plugin/models/utils/
deform3d_cross_attn.py:33-339): same ATTENTION-registry type name, constructor keywords,
parameter / state-dict names (:100-121), `init_weight()` (:129-150) and `forward` signature
(:152-162).  What differs is how forward computes: the projection, mask, masked softmax, the mmcv
MSDA gather and the camera-weighted sum are ONE HIP kernel (gd4d_cross_attn_fwd) instead of ~15
elementwise launches + a third-party CUDA kernel + a (N, Q, 256) intermediate.
"""
import math
import warnings

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .registry import ATTENTION


@ATTENTION.register_module()
class Deform3DCrossAttn(nn.Module):
    """See the reference docstring for argument meaning; extra keyword `value_dtype`
    ('fp32' | 'bf16') selects the storage type of the projected value tensor (fp32 accumulate)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=5, num_cams=6,
                 im2col_step=64, pc_range=None, dropout=0.1, norm_cfg=None, init_cfg=None,
                 batch_first=False, fix_offset=False, depth_encode=False, value_dtype='fp32'):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        dim_per_head = embed_dims // num_heads
        if dim_per_head & (dim_per_head - 1):
            warnings.warn('embed_dims // num_heads should be a power of 2 for the gfx950 kernels')
        if value_dtype not in ('fp32', 'bf16'):
            raise ValueError(f"value_dtype must be 'fp32' or 'bf16', got {value_dtype!r}")
        self.norm_cfg = norm_cfg
        self.init_cfg = init_cfg
        self.pc_range = pc_range
        self.fix_offset = fix_offset
        self.depth_encode = depth_encode
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_cams = num_cams
        self.batch_first = batch_first
        self.value_dtype = torch.float32 if value_dtype == 'fp32' else torch.bfloat16

        self.dropout = nn.Dropout(dropout)
        self.cam_attention_weights = nn.Linear(embed_dims, num_cams)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.position_encoder = nn.Sequential(
            nn.Linear(4 if depth_encode else 3, embed_dims), nn.LayerNorm(embed_dims),
            nn.ReLU(inplace=True),
            nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True))
        # one 3-D offset per (head, point), shared by all levels (:116-117)
        self.deform_sampling_offsets = nn.Linear(embed_dims, num_heads * 1 * num_points * 3)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weight()
        if fix_offset:
            self.deform_sampling_offsets.weight.requires_grad = False
            self.deform_sampling_offsets.bias.requires_grad = False

    def init_weight(self):
        """Reference :129-150: zero logits, xavier projections, per-head direction x (i+1) metres."""
        for lin in (self.cam_attention_weights, self.attention_weights):
            nn.init.constant_(lin.weight, 0.)
            nn.init.constant_(lin.bias, 0.)
        for lin in (self.output_proj, self.value_proj):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0.)
        nn.init.constant_(self.deform_sampling_offsets.weight, 0.)
        theta = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        direction = torch.stack([theta.cos(), theta.sin(), theta.cos()], -1)
        direction = direction / direction.abs().max(-1, keepdim=True)[0]
        steps = torch.arange(1, self.num_points + 1, dtype=torch.float32)
        grid = direction[:, None, :] * steps[None, :, None]          # (heads, points, 3)
        with torch.no_grad():
            self.deform_sampling_offsets.bias.copy_(grid.reshape(-1))

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        """query (Q, B, C); value = list of L maps (B, N, C, H_l, W_l); reference_points (B, Q, 3)
        in [0,1]; kwargs['img_metas'] mandatory.  Returns (Q, B, C)."""
        if residual is not None:
            # the reference leaves `inp_residual` undefined in this case (:201-202 -> NameError)
            raise NameError('Deform3DCrossAttn: residual must be None (as in the reference)')
        if value is None or torch.is_tensor(value):
            raise TypeError('value must be the list of multi-camera feature maps (B, N, C, H, W)')
        img_metas = kwargs['img_metas']
        Fn.require_gpu(query, 'query')
        if Fn.wants_grad(self, query, query_pos, reference_points, *value):
            cached = (kwargs.get(Fn.VALUE_CACHE_KEY) or {}).get(id(self))
            hit = cached is not None and cached[2] is value
            return self._forward_autograd(query, value, query_pos, reference_points, img_metas, cached[0] if hit else None,
                                          cached[3] if hit and len(cached) > 3 else None, offers=kwargs)

        inp_residual = query
        q_len, b, c = query.shape

        # position_encoder depends on the reference points only: fork it onto the auxiliary stream, it runs next to the
        # query-side linears and the gather and is joined before output_proj adds it (:331-336)
        main = torch.cuda.current_stream(query.device)
        aux = Fn.aux_stream(query.device)
        ev_pos = None
        if aux is not None:
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(aux):
                aux.wait_event(fork)
                pos_feat = self._position_features(reference_points)
                ev_pos = torch.cuda.Event()
                ev_pos.record(aux)
        ref_event = kwargs.get(Fn.REF_EVENT_KEY)     # reference points refined on the auxiliary stream by the decoder
        if ref_event is not None:
            main.wait_event(ref_event)
        q = q_len
        hh, npt, nl = self.num_heads, self.num_points, self.num_levels
        n = self.num_cams
        if len(value) != nl:
            raise ValueError(f'expected {nl} feature levels, got {len(value)}')
        if value[0].shape[1] != n:
            raise ValueError(f'expected {n} cameras, got {value[0].shape[1]}')

        # query-side projections of (query + query_pos), the add fused into each GEMM's load (:204-228,281).
        # Rows are (q, b)-ordered in memory; for B = 1 that IS the reference's (B, Q, C) permute (:207).
        if b == 1:
            xq, xp = query, query_pos
        else:
            xq = query.permute(1, 0, 2).contiguous()
            xp = None if query_pos is None else query_pos.permute(1, 0, 2).contiguous()
        # one launch for the three Linears of (query + query_pos) (:211, :227, :281)
        mods = (self.cam_attention_weights, self.deform_sampling_offsets, self.attention_weights)
        cam_logits, offsets, attn_logits = ops.linear_group_fwd(
            xq.contiguous(), [m_.weight.contiguous() for m_ in mods], [m_.bias for m_ in mods],
            x2=None if xp is None else xp.contiguous())
        cam_logits = cam_logits.view(b, q, n)                                        # un-scrambled
        offsets = offsets.view(b, q, hh, npt, 3)
        attn_logits = attn_logits.view(b, q, hh, nl, npt)

        lidar2img = Fn.lidar2img_device(img_metas, query)
        img_h, img_w = Fn.img_hw(img_metas)
        order = kwargs.get(Fn.QUERY_ORDER_KEY)
        if order is None or order.numel() != b * q:
            order = Fn.query_order(reference_points, self.pc_range)
        pipeline = kwargs.get(Fn.VALUE_PIPELINE_KEY)
        taken = pipeline.take(self, value) if pipeline is not None else None
        cached = (kwargs.get(Fn.VALUE_CACHE_KEY) or {}).get(id(self))
        late = kwargs.get(Fn.LATE_VALUES_KEY)
        if late is not None and late.value is not value:
            late = None
        if late is None and taken is None and (cached is None or cached[2] is not value) \
                and Fn.LateValues.applicable([self], value):
            late = Fn.LateValues(value, self.value_dtype)   # a stand-alone call: its own channels-last copy
        if late is not None:
            # aggregate-then-project (csrc/gd4d_cross_attn_late.hip): raw features gathered per head, value_proj applied
            # to the Q x Hh aggregates - no projected value tensor
            agg = late.sample_aggregate(self, reference_points, offsets, attn_logits, cam_logits, lidar2img,
                                        img_h, img_w, order=order)                        # (B, Q, C)
        else:
            if taken is not None:
                val, shapes = taken                  # projected on the side stream underneath the previous layer
            elif cached is not None and cached[2] is value:
                val, shapes = cached[0], cached[1]   # projected by the decoder for all layers at once
            else:
                val, shapes = Fn.value_projection(value, self.value_proj.weight, self.value_proj.bias,
                                                  hh, self.value_dtype)
            agg = Fn.sample_aggregate(val, shapes, reference_points, offsets, attn_logits, cam_logits,
                                      lidar2img, self.pc_range, img_h, img_w, order=order)    # (B, Q, C)
            if taken is not None:
                del val, taken
                pipeline.gather_enqueued(self)

        if ev_pos is None:
            pos_feat = self._position_features(reference_points)         # (B, Q, C)
        else:
            main.wait_event(ev_pos)
        if b == 1 and not self.training:
            # output_proj with both residuals of :336 in its epilogue; (B=1,Q,C) and (Q,1,C) share memory
            fused = Fn.take_fused_norm(kwargs)
            if fused is not None and Fn.rowblock_ok(agg, self.output_proj.weight, fused['norm']):
                fused['done'] = True                 # ... and the layer's LayerNorm that follows
                return Fn.linear_norm(agg, self.output_proj.weight, self.output_proj.bias, fused['norm'],
                                      r1=inp_residual.view(1, q, c), r2=pos_feat).view(q, 1, c)
            return Fn.linear(agg, self.output_proj.weight, self.output_proj.bias,
                             r1=inp_residual.view(1, q, c), r2=pos_feat).view(q, 1, c)
        out = Fn.linear(agg, self.output_proj.weight, self.output_proj.bias).permute(1, 0, 2)
        return self.dropout(out) + inp_residual + pos_feat.permute(1, 0, 2)

    def _position_features(self, reference_points):
        ref3d = reference_points
        if self.depth_encode:                                             # :331-333
            depth = (ref3d[..., 0:1] ** 2 + ref3d[..., 1:2] ** 2) ** 0.5
            ref3d = torch.cat([ref3d, depth], dim=-1)
        return Fn.position_encoder(self.position_encoder, ref3d)

    def _forward_autograd(self, query, value, query_pos, reference_points, img_metas, projected=None, cl=None, offers=None):
        """Training path: the same maths with autograd.  The gather runs gd4d_cross_attn_fwd/_bwd, value_proj
        runs the HIP forward with a GEMM backward, the small dense layers are torch ops."""
        from .autograd import CrossAttnFunction, ValueProjFunction
        inp_residual = query
        hh, npt, nl = self.num_heads, self.num_points, self.num_levels
        three = [self.cam_attention_weights, self.deform_sampling_offsets, self.attention_weights]
        if Fn.can_group_linears(query, query_pos, three):
            # the three Linears of (query + query_pos) as one node: one launch forward (which also forms the sum), chained
            # input gradients backward
            xq, xp = query.permute(1, 0, 2), query_pos.permute(1, 0, 2)   # (B, Q, C)
            b, q, c = xq.shape
            cam_logits, offsets, attn_logits = Fn.linear_group_autograd(xq.contiguous(), xp.contiguous(), three)
            offsets, attn_logits = offsets.view(b, q, hh, npt, 3), attn_logits.view(b, q, hh, nl, npt)
        else:
            x = query if query_pos is None else query + query_pos
            x = x.permute(1, 0, 2)                                        # (B, Q, C)
            b, q, c = x.shape
            x = x.contiguous()
            cam_logits = Fn.sequential_autograd(self.cam_attention_weights, x)   # un-scrambled (B, Q, N)
            offsets = Fn.sequential_autograd(self.deform_sampling_offsets, x).view(b, q, hh, npt, 3)
            attn_logits = Fn.sequential_autograd(self.attention_weights, x).view(b, q, hh, nl, npt)
        shapes = [tuple(v.shape[-2:]) for v in value]
        lidar2img = Fn.lidar2img_device(img_metas, query)
        img_h, img_w = Fn.img_hw(img_metas)
        from .autograd import CrossAttnRawFunction
        own = None
        if projected is None and cl is None:     # a stand-alone call: its own copy of the pyramid, if the raw path applies
            own = Fn.raw_pyramid_for_training([self], value)
            cl = None if own is None else own[id(self)][3]
        # (the training kernels take 4 points per head, or 8 with 8 heads; other counts are padded with points that are never
        #  visible and weigh nothing - functional.pad_points; autograd slices the padding's gradients away)
        offsets, attn_logits = Fn.pad_points(offsets, attn_logits, (4, 8) if (hh == 8 and isinstance(cl, tuple)) else (4,),
                                             'Deform3DCrossAttn (training)')
        if isinstance(cl, tuple):                # (RawPyramid, token)
            # the inference step's kernels behind autograd: no projected value tensor (gd4d_cross_attn_sliced_bwd.hip)
            agg = CrossAttnRawFunction.apply(cl[1], reference_points, offsets, attn_logits, cam_logits, lidar2img,
                                             self.value_proj.weight, self.value_proj.bias, cl[0], self.pc_range, img_h, img_w)
            return self._finish_autograd(agg, reference_points, inp_residual, offers)
        if projected is None:
            val = ValueProjFunction.apply(self.value_proj.weight, self.value_proj.bias, *value)
            val = val.view(val.shape[0], -1, hh, c // hh)
        else:
            val = projected                      # the decoder projected for all its layers (one autograd node)
        if cl is not None:       # the decoder made a channels-last copy: value_proj's weight gradient comes from this node
            agg = CrossAttnFunction.apply(val, reference_points, offsets, attn_logits, cam_logits, lidar2img,
                                          shapes, self.pc_range, img_h, img_w, cl, self.value_proj.weight, self.value_proj.bias)
        else:
            agg = CrossAttnFunction.apply(val, reference_points, offsets, attn_logits, cam_logits, lidar2img,
                                          shapes, self.pc_range, img_h, img_w)
        return self._finish_autograd(agg, reference_points, inp_residual, offers)

    def _finish_autograd(self, agg, reference_points, inp_residual, offers=None):
        out = Fn.sequential_autograd(self.output_proj, agg).permute(1, 0, 2)
        ref3d = reference_points
        if self.depth_encode:
            depth = (ref3d[..., 0:1] ** 2 + ref3d[..., 1:2] ** 2) ** 0.5
            ref3d = torch.cat([ref3d, depth], dim=-1)
        # (the decoder hands over DETACHED reference points: their inverse_sigmoid is then one launch instead of six ATen ones)
        isig = ops.inverse_sigmoid_fwd(ref3d.contiguous()) if ref3d.is_cuda and ref3d.dtype == torch.float32 \
            and not ref3d.requires_grad else Fn.inverse_sigmoid(ref3d)
        pos_feat = Fn.sequential_autograd(self.position_encoder, isig).permute(1, 0, 2)
        fused = Fn.take_fused_norm(offers or {}, autograd=True)
        if fused is not None:        # the layer's LayerNorm comes next: its kernel adds inp_residual (:336's first sum)
            fused['done'] = True
            return Fn.layer_norm_autograd(inp_residual, fused['norm'], res=self.dropout(out) + pos_feat)
        return self.dropout(out) + inp_residual + pos_feat

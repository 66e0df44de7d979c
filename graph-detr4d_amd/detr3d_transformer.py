"""DETR3D-side modules of the hot path: `Detr3DCrossAtten`, `feature_sampling`,
`Detr3DTransformerDecoder`, `Detr3DTransformer`.

Drop-ins for the reference classes of the same names in
projects/mmdet3d_plugin/models/utils/detr3d_transformer.py (:46, :153, :229, :397): same registry
type names, constructor keywords, state-dict keys and call signatures.
"""
import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .deform3d_cross_attn import Deform3DCrossAttn
from .registry import (ATTENTION, TRANSFORMER, TRANSFORMER_LAYER_SEQUENCE,
                       build_transformer_layer_sequence)
from .transformer_layers import TransformerLayerSequence

inverse_sigmoid = Fn.inverse_sigmoid


def feature_sampling(mlvl_feats, reference_points, pc_range, img_metas):
    """Reference: detr3d_transformer.py:397-438.  Returns (reference_points_3d (B,Q,3),
    sampled_feats (B,C,Q,N,1,L), mask (B,1,Q,N,1,1) bool) computed by gd4d_detr3d_fwd."""
    Fn.require_gpu(reference_points, 'reference_points')
    Fn.require_inference(reference_points, *mlvl_feats)
    lidar2img = Fn.lidar2img_device(img_metas, reference_points)
    img_h, img_w = Fn.img_hw(img_metas)
    b, q = reference_points.shape[:2]
    n, nl = mlvl_feats[0].shape[1], len(mlvl_feats)
    dummy = reference_points.new_zeros(b, q, n, 1, nl)
    r = ops.detr3d_fwd([f.contiguous() for f in mlvl_feats], reference_points.contiguous(), dummy,
                       lidar2img, pc_range, img_h, img_w, want_out=False, want_mask=True,
                       want_sampled=True)
    mask = r['mask'].bool().permute(0, 2, 1).reshape(b, 1, q, n, 1, 1)
    return reference_points.clone(), r['sampled'], mask


@ATTENTION.register_module()
class Detr3DCrossAtten(nn.Module):
    """DETR3D baseline cross-attention (reference :229-390): one sampling point per query, no
    heads, sigmoid weights over cameras x levels."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=5, num_cams=6,
                 im2col_step=64, pc_range=None, dropout=0.1, norm_cfg=None, init_cfg=None,
                 batch_first=False):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, '
                             f'but got {embed_dims} and {num_heads}')
        self.norm_cfg = norm_cfg
        self.init_cfg = init_cfg
        self.pc_range = pc_range
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_cams = num_cams
        self.batch_first = batch_first
        self.dropout = nn.Dropout(dropout)
        self.attention_weights = nn.Linear(embed_dims, num_cams * num_levels * num_points)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.position_encoder = nn.Sequential(
            nn.Linear(3, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True),
            nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True))
        self.init_weight()

    def init_weight(self):
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if residual is not None:
            raise NameError('Detr3DCrossAtten: residual must be None (as in the reference)')
        img_metas = kwargs['img_metas']
        Fn.require_gpu(query, 'query')
        if self.num_points != 1:
            raise NotImplementedError('Detr3DCrossAtten: the gfx950 kernel is built for '
                                      'num_points=1 (every reference config)')
        if Fn.wants_grad(self, query, query_pos, reference_points, *value):
            return self._forward_autograd(query, value, query_pos, reference_points, img_metas)
        inp_residual = query
        q, b, c = query.shape
        if b == 1:
            xq, xp = query, query_pos
        else:
            xq = query.permute(1, 0, 2).contiguous()
            xp = None if query_pos is None else query_pos.permute(1, 0, 2).contiguous()
        logits = Fn.linear(xq, self.attention_weights.weight, self.attention_weights.bias,
                           **(dict(x2=xp) if xp is not None else {})).view(b, q, -1)
        lidar2img = Fn.lidar2img_device(img_metas, query)
        img_h, img_w = Fn.img_hw(img_metas)
        agg = ops.detr3d_fwd([f.contiguous() for f in value], reference_points.contiguous(),
                             logits.contiguous(), lidar2img, self.pc_range, img_h, img_w)['out']
        pos_feat = Fn.position_encoder(self.position_encoder, reference_points)
        if b == 1 and not self.training:
            return Fn.linear(agg, self.output_proj.weight, self.output_proj.bias,
                             r1=inp_residual.view(1, q, c), r2=pos_feat).view(q, 1, c)
        out = Fn.linear(agg, self.output_proj.weight, self.output_proj.bias).permute(1, 0, 2)
        return self.dropout(out) + inp_residual + pos_feat.permute(1, 0, 2)


    def _forward_autograd(self, query, value, query_pos, reference_points, img_metas):
        """Training path (the DETR3D configs train this module): the sampling core (:373-383, feature_sampling :397-438) on its
        HIP kernels both ways (autograd.Detr3DSampleFunction: gd4d_detr3d_fwd / gd4d_detr3d_bwd - gradients reach the feature
        maps, the attention logits and the reference points), the dense layers on the HIP kernels' autograd functions.
        GD4D_TORCH_OPS=1: the reference's operations as differentiable torch ops on the GPU (num_points != 1 needs it) -
        projection, F.grid_sample per level (bilinear, zero padding, align_corners=False), sigmoid weights x mask, sums."""
        inp_residual = query
        x = query if query_pos is None else query + query_pos
        x = x.permute(1, 0, 2).contiguous()                                   # (B, Q, C)
        b, q, c = x.shape
        n, nl = self.num_cams, self.num_levels
        logits = Fn.sequential_autograd(self.attention_weights, x).view(b, 1, q, n, self.num_points, nl)
        lidar2img = Fn.lidar2img_device(img_metas, query)                     # (B, N, 4, 4)
        img_h, img_w = Fn.img_hw(img_metas)
        if not Fn.torch_ops_route(f'Detr3DCrossAtten training with num_points = {self.num_points}, {len(value)} levels, {n} cameras',
                                  self.num_points == 1 and len(value) <= 8 and n <= 256
                                  and all(v.is_cuda and v.dtype == torch.float32 for v in value)
                                  and reference_points.dtype == torch.float32, module=self):
            # the sampling core on its HIP kernels both ways (gd4d_detr3d_fwd / gd4d_detr3d_bwd)
            from .autograd import Detr3DSampleFunction
            agg = Detr3DSampleFunction.apply(reference_points, logits.reshape(b, q, n, 1, nl), lidar2img, self.pc_range, img_h, img_w,
                                             *value)
            out = Fn.sequential_autograd(self.output_proj, agg).permute(1, 0, 2)
            pos_feat = Fn.sequential_autograd(self.position_encoder, Fn.inverse_sigmoid(reference_points)).permute(1, 0, 2)
            return self.dropout(out) + inp_residual + pos_feat
        rng = self.pc_range
        lo = reference_points.new_tensor(rng[:3])
        scale = reference_points.new_tensor([rng[3] - rng[0], rng[4] - rng[1], rng[5] - rng[2]])
        pts = reference_points * scale + lo                                   # metres (:402-405)
        pts = torch.cat([pts, torch.ones_like(pts[..., :1])], dim=-1)         # (B, Q, 4)
        cam = torch.matmul(lidar2img.view(b, n, 1, 4, 4), pts.view(b, 1, q, 4, 1)).squeeze(-1)   # (B, N, Q, 4)
        eps = 1e-5
        mask = cam[..., 2:3] > eps
        xy = cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.full_like(cam[..., 2:3], eps))
        xy = torch.stack([xy[..., 0] / img_w, xy[..., 1] / img_h], dim=-1)
        xy = (xy - 0.5) * 2
        mask = mask & (xy[..., 0:1] > -1.0) & (xy[..., 0:1] < 1.0) & (xy[..., 1:2] > -1.0) & (xy[..., 1:2] < 1.0)
        mask = mask.view(b, n, 1, q, 1, 1).permute(0, 2, 3, 1, 4, 5)          # (B, 1, Q, N, 1, 1)
        sampled = []
        for feat in value:
            bn, cc, h, w = feat.shape[0] * feat.shape[1], feat.shape[2], feat.shape[3], feat.shape[4]
            s_ = torch.nn.functional.grid_sample(feat.reshape(bn, cc, h, w), xy.reshape(bn, q, 1, 2), align_corners=False)
            sampled.append(s_.view(b, n, cc, q, 1).permute(0, 2, 3, 1, 4))   # (B, C, Q, N, 1)
        out = torch.stack(sampled, -1)                                        # (B, C, Q, N, 1, L)
        out = torch.nan_to_num(out)
        weights = logits.sigmoid() * mask.to(logits.dtype)
        out = (out * weights).sum(-1).sum(-1).sum(-1).permute(2, 0, 1)       # (Q, B, C)
        out = Fn.sequential_autograd(self.output_proj, out.contiguous())
        pos_feat = Fn.sequential_autograd(self.position_encoder, Fn.inverse_sigmoid(reference_points)).permute(1, 0, 2)
        return self.dropout(out) + inp_residual + pos_feat


@ATTENTION.register_module()
class Detr3DCrossAttenV2(nn.Module):
    """2-D-offset deformable variant (reference :441-710; registered by the reference, used by no shipped config): per
    (camera, head, level, point) pixel offsets around the projected reference point, softmax over level x point per
    (camera, head), samples of the head's channel slice of the raw NCHW maps.  Same constructor keywords and
    state-dict keys (`attention_weights`, `sampling_offsets`, `output_proj`, `position_encoder`).  The sampling core is
    gd4d_detr3d_v2_fwd (incl. the reference's level/point weight transposition); batch 1, num_points == num_levels as
    the reference's broadcasts require (:611, :698-700).  With autograd on: the same operations as differentiable torch
    ops on the GPU (_forward_autograd)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=5, num_cams=6, im2col_step=64,
                 pc_range=None, dropout=0.1, norm_cfg=None, init_cfg=None, batch_first=False):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}')
        self.norm_cfg, self.init_cfg, self.pc_range, self.im2col_step = norm_cfg, init_cfg, pc_range, im2col_step
        self.embed_dims, self.num_levels, self.num_heads = embed_dims, num_levels, num_heads
        self.num_points, self.num_cams, self.batch_first = num_points, num_cams, batch_first
        self.dropout = nn.Dropout(dropout)
        self.attention_weights = nn.Linear(embed_dims, num_cams * num_heads * num_levels * num_points)
        self.sampling_offsets = nn.Linear(embed_dims, num_cams * num_heads * num_levels * num_points * 2)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.position_encoder = nn.Sequential(
            nn.Linear(3, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True),
            nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True))
        self.init_weight()

    def init_weight(self):
        """Reference :527-543: zero offset weights, per-head direction x (i+1) pixels as bias, zero logits."""
        import math
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        theta = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([theta.cos(), theta.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(1, self.num_heads, 1, 1, 2) \
            .repeat(self.num_cams, 1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        if residual is not None:
            raise NameError('Detr3DCrossAttenV2: residual must be None (as in the reference)')
        img_metas = kwargs['img_metas']
        Fn.require_gpu(query, 'query')
        q, b, c = query.shape
        if b != 1:
            raise RuntimeError('Detr3DCrossAttenV2: batch size must be 1 (the reference broadcasts (B*N, Q) against '
                               '(B*heads, N, Q) coordinates)')
        if self.num_points != self.num_levels:
            raise RuntimeError('Detr3DCrossAttenV2: num_points must equal num_levels (the reference multiplies '
                               '(..., point, level) samples with (..., level, point) weights)')
        if Fn.wants_grad(self, query, query_pos, reference_points, *value):
            return self._forward_autograd(query, value, query_pos, reference_points, img_metas)
        n, hh, nl, npt = self.num_cams, self.num_heads, self.num_levels, self.num_points
        logits, offsets = ops.linear_group_fwd(
            query.contiguous(), [self.attention_weights.weight.contiguous(), self.sampling_offsets.weight.contiguous()],
            [self.attention_weights.bias, self.sampling_offsets.bias],
            x2=None if query_pos is None else query_pos.contiguous())
        lidar2img = Fn.lidar2img_device(img_metas, query)
        img_h, img_w = Fn.img_hw(img_metas)
        agg = ops.detr3d_v2_fwd([f.contiguous() for f in value], reference_points.contiguous(),
                                logits.view(b, q, n, hh, nl * npt), offsets.view(b, q, n, hh, nl, npt, 2), lidar2img,
                                self.pc_range, img_h, img_w, hh)
        pos_feat = Fn.position_encoder(self.position_encoder, reference_points)
        if not self.training:
            return Fn.linear(agg, self.output_proj.weight, self.output_proj.bias,
                             r1=query.view(1, q, c), r2=pos_feat).view(q, 1, c)
        out = Fn.linear(agg, self.output_proj.weight, self.output_proj.bias).permute(1, 0, 2)
        return self.dropout(out) + query + pos_feat.permute(1, 0, 2)

    def _forward_autograd(self, query, value, query_pos, reference_points, img_metas):
        """Training path.  The sampling (reference :597-710: projected reference point + per-head pixel offsets, softmax over
        level x point, the (point, level) x (level, point) pairing of :611 / :705-707, visibility, the sums) runs on
        gd4d_detr3d_v2_fwd / gd4d_detr3d_v2_bwd behind autograd.Detr3DV2SampleFunction (embed_dims <= 256; wider models raise;
        GD4D_TORCH_OPS=1: the same maths as differentiable torch ops, below); the Linears and LayerNorms are the package's
        autograd Functions."""
        if not Fn.torch_ops_route(f'Detr3DCrossAttenV2 training with embed_dims = {self.embed_dims}', self.embed_dims <= 256, module=self):
            from .autograd import Detr3DV2SampleFunction
            x = (query if query_pos is None else query + query_pos).permute(1, 0, 2).contiguous()      # (1, Q, C)
            b, q, c = x.shape
            n, hh, nl, npt = self.num_cams, self.num_heads, self.num_levels, self.num_points
            logits = Fn.sequential_autograd(self.attention_weights, x).view(b, q, n, hh, nl * npt)
            off = Fn.sequential_autograd(self.sampling_offsets, x).view(b, q, n, hh, nl, npt, 2)
            lidar2img = Fn.lidar2img_device(img_metas, query)
            img_h, img_w = Fn.img_hw(img_metas)
            agg = Detr3DV2SampleFunction.apply(reference_points, logits, off, lidar2img, self.pc_range, img_h, img_w, hh, *value)
            res = Fn.sequential_autograd(self.output_proj, agg.reshape(q, 1, c))
            pos_feat = Fn.sequential_autograd(self.position_encoder, Fn.inverse_sigmoid(reference_points)).permute(1, 0, 2)
            return self.dropout(res) + query + pos_feat
        x = query if query_pos is None else query + query_pos
        x = x.permute(1, 0, 2).contiguous()                                   # (1, Q, C)
        b, q, c = x.shape
        n, hh, nl, npt = self.num_cams, self.num_heads, self.num_levels, self.num_points
        d = c // hh
        w = Fn.sequential_autograd(self.attention_weights, x).view(b, q, n, hh, nl * npt).softmax(-1)
        w = w.view(b, q, n, hh, nl, npt)
        off = Fn.sequential_autograd(self.sampling_offsets, x).view(b, q, n, hh, nl, npt, 2)
        lidar2img = Fn.lidar2img_device(img_metas, query)
        img_h, img_w = Fn.img_hw(img_metas)
        rng = self.pc_range
        lo = reference_points.new_tensor(rng[:3])
        scale = reference_points.new_tensor([rng[3] - rng[0], rng[4] - rng[1], rng[5] - rng[2]])
        pts = reference_points * scale + lo
        pts = torch.cat([pts, torch.ones_like(pts[..., :1])], dim=-1)
        cam = torch.matmul(lidar2img.view(b, n, 1, 4, 4), pts.view(b, 1, q, 4, 1)).squeeze(-1)   # (1, N, Q, 4)
        eps = 1e-5
        z_ok = cam[..., 2] > eps
        xy = cam[..., 0:2] / torch.maximum(cam[..., 2:3], torch.full_like(cam[..., 2:3], eps))
        g = (torch.stack([xy[..., 0] / img_w, xy[..., 1] / img_h], dim=-1) - 0.5) * 2
        mask = (z_ok & (g[..., 0] > -1.0) & (g[..., 0] < 1.0) & (g[..., 1] > -1.0) & (g[..., 1] < 1.0)).to(x.dtype)
        out = x.new_zeros(q, hh, d)
        for lvl, feat in enumerate(value):
            h_l, w_l = feat.shape[-2:]
            f = feat.view(b, n, hh, d, h_l, w_l).transpose(1, 2).flatten(0, 2)             # (heads * N, d, H, W)
            o = off[:, :, :, :, lvl].permute(0, 3, 2, 1, 4, 5).flatten(0, 1)                # (heads, N, Q, P, 2)
            loc = g.view(b * n, q, 1, 2)[None] + o / o.new_tensor([w_l, h_l])
            s_ = torch.nn.functional.grid_sample(f, loc.flatten(0, 1), mode='bilinear', padding_mode='zeros',
                                                 align_corners=False).view(hh, n, d, q, npt)
            wq = w[0, :, :, :, :, lvl].permute(2, 1, 0, 3) * mask[0].view(1, n, q, 1)       # weight (level i, point lvl)
            out = out + torch.einsum('hndqp,hnqp->qhd', s_, wq)
        res = Fn.sequential_autograd(self.output_proj, out.reshape(q, 1, c))
        pos_feat = Fn.sequential_autograd(self.position_encoder, Fn.inverse_sigmoid(reference_points)).permute(1, 0, 2)
        return self.dropout(res) + query + pos_feat


def _watch_weight_updates(module):
    """The fused decoder loop caches MFMA-fragment images of its GEMM weights (ops.chain_weight_image), keyed on the
    tensors' version counters.  Writes through `.data` (checkpoint loaders, mmcv's EMA hook) do not bump those: forget
    the images whenever a state dict is loaded; train() / eval() do the same (below)."""
    module.register_load_state_dict_post_hook(_forget_chain_images)     # (a module-level function: torch.save(model) pickles hooks)


def _forget_chain_images(module, incompatible_keys):
    ops.invalidate_chain_images()


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class Detr3DTransformerDecoder(TransformerLayerSequence):
    """Reference :153-225: loop over layers, refine reference points with reg_branches."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        _watch_weight_updates(self)
        self.return_intermediate = return_intermediate

    def train(self, mode=True):
        ops.invalidate_chain_images()        # mode switches bracket the weight updates that do not bump version counters (EMA swaps)
        return super().train(mode)

    def _preproject_values(self, kwargs):
        """All layers get the same `value` pyramid and value_proj does not depend on the queries, so the decoder can
        project for every layer up front.  GD4D_PREPROJECT selects how:
          'auto' (default) - Fn.ValuePipeline over groups of two layers ('g2,2,2' for six): measured best on MI355X at
                       the headline size (2.68 ms per step with the fused decoder loop; 'g3,3' 2.66-2.73, one launch 2.98)
          '1'      - one gd4d_value_proj_multi_fwd launch on the main stream: a wave keeps its pixel tile in
                       registers for all the layers, the pyramid is read (and split) once - 1.98 ms for six layers against
                       6 x 0.48 ms one by one (NL value tensors alive: 4.5 GB at the headline size)
          'g3,3', 'g2,2,2', ... - Fn.ValuePipeline over layer GROUPS: one multi-layer launch per group on a second HIP
                       stream, group g+1 underneath the query-side kernels of group g
          'stream' - = 'g1,1,...': the round-1 per-layer pipeline (two value tensors alive)
          '0'      - off: every layer projects when it runs (the reference's order)
        Returns (kwargs, pipeline or None)."""
        mode = os.environ.get('GD4D_PREPROJECT', 'auto')
        value = kwargs.get('value')
        if kwargs.get(Fn.VALUE_CACHE_KEY) is not None:       # the caller projected already (Detr3DTransformer.forward_shared)
            return kwargs, None
        if mode == '0' or not isinstance(value, (list, tuple)) or len(value) == 0 or not value[0].is_cuda:
            return kwargs, None
        mods = [a for layer in self.layers for a in layer.attentions if isinstance(a, Deform3DCrossAttn)]
        if len(mods) < 2 or len(mods) > 8 or len({(m.num_heads, m.value_dtype, m.embed_dims) for m in mods}) != 1:
            return kwargs, None
        if Fn.wants_grad(mods[0], *value):
            # training: the raw-pyramid path (one copy + one gradient pass for all layers) ...
            raw = Fn.raw_pyramid_for_training(mods, value)
            if raw is not None:
                kwargs = dict(kwargs)
                kwargs[Fn.VALUE_CACHE_KEY] = raw
                return kwargs, None
            # ... or one autograd node for all the layers' projections (the pyramid's gradient is summed in place)
            if any(m.value_proj.bias is None or m.value_dtype != torch.float32 for m in mods) \
                    or value[0].shape[0] != 1 or any(v.dtype != torch.float32 for v in value):
                return kwargs, None
            kwargs = dict(kwargs)
            kwargs[Fn.VALUE_CACHE_KEY] = Fn.project_values_for_layers_autograd(mods, value)
            return kwargs, None
        Fn.require_inference(*value)
        kwargs = dict(kwargs)
        groups = Fn.pipeline_groups(mode, len(mods))
        if groups is None:                                   # '1' (and anything unparsable): one launch for all layers
            kwargs[Fn.VALUE_CACHE_KEY] = Fn.project_values_for_layers(mods, value)
            return kwargs, None
        pipeline = kwargs[Fn.VALUE_PIPELINE_KEY] = Fn.ValuePipeline(mods, value, groups)
        return kwargs, pipeline

    def _order_queries(self, kwargs, reference_points):
        """Locality order of the queries (Fn.query_order) for the first layer's fused kernel; later layers get theirs
        from the fused refinement launch (without reg_branches the points, and so the order, never change)."""
        mods = [a for layer in self.layers for a in layer.attentions if isinstance(a, Deform3DCrossAttn)]
        if not mods or reference_points is None or not reference_points.is_cuda or Fn.QUERY_ORDER_KEY in kwargs:
            return kwargs
        order = Fn.query_order(reference_points, mods[0].pc_range)
        if order is None:
            return kwargs
        self._order_pc_range = mods[0].pc_range
        kwargs = dict(kwargs)
        kwargs[Fn.QUERY_ORDER_KEY] = order
        return kwargs

    def _refine(self, branch, output, reference_points, kwargs, want_order):
        """reg branch + reference-point refinement (:201-214) on the HIP kernels; with a locality order in use the
        refinement launch also produces the next layer's order (kwargs is updated in place)."""
        tmp = Fn.run_branch(branch, output.permute(1, 0, 2).contiguous())
        if want_order and Fn.QUERY_ORDER_KEY in kwargs:
            new_ref, kwargs[Fn.QUERY_ORDER_KEY] = Fn.refine_reference_order(tmp, reference_points,
                                                                            self._order_pc_range)
            return new_ref
        return Fn.refine_reference(tmp, reference_points).detach()

    def forward(self, query, *args, reference_points=None, reg_branches=None, **kwargs):
        output = query
        intermediate, intermediate_reference_points = [], []
        from . import fused_decoder
        fused = not args and kwargs.get('key') is None and kwargs.get('query_pos') is not None and 'img_metas' in kwargs \
            and kwargs.get('key_padding_mask') is None and kwargs.get('query_key_padding_mask') is None \
            and fused_decoder.applicable(self, query, kwargs.get('value'), reference_points, reg_branches,
                                         kwargs.get('attn_masks'), query_pos=kwargs.get('query_pos'))
        late, own_late, pipeline = kwargs.get(Fn.LATE_VALUES_KEY), False, None
        if late is not None and late.value is not kwargs.get('value'):
            late = None
        cross = [a for layer in self.layers for a in layer.attentions if getattr(a, 'operation_name', '') == 'cross_attn']
        if late is None and kwargs.get(Fn.VALUE_CACHE_KEY) is None and cross \
                and all(isinstance(a, Deform3DCrossAttn) for a in cross) and isinstance(kwargs.get('value'), (list, tuple)) \
                and Fn.LateValues.applicable(cross, kwargs['value']) and not Fn.wants_grad(self, query, *kwargs['value']):
            # aggregate-then-project (GD4D_PROJECT=late, default): no per-layer value tensors, ONE channels-last copy of
            # the pyramid for all layers (made on the side stream next to layer 0's self-attention)
            single = fused and all((query.shape[-1] // a.num_heads) % 32 == 0 and not a.depth_encode for a in cross)
            late, own_late = Fn.LateValues(kwargs['value'], cross[0].value_dtype, coarse_for=cross if single else None), True
            kwargs = dict(kwargs)
            kwargs[Fn.LATE_VALUES_KEY] = late
        if late is None:
            kwargs, pipeline = self._preproject_values(kwargs)
        kwargs = self._order_queries(kwargs, reference_points)
        if not fused and query.is_cuda and torch.is_grad_enabled():
            # training on the raw pyramid: the whole loop as ONE autograd node whose forward and backward are row chains
            from . import fused_train
            if fused_train.applicable(self, query, kwargs.get('query_pos'), kwargs.get('value'), reference_points, reg_branches,
                                      kwargs.get('attn_masks'), kwargs.get(Fn.VALUE_CACHE_KEY), args, kwargs):
                return fused_train.run(self, query, kwargs['query_pos'], kwargs['value'], reference_points, reg_branches,
                                       kwargs['img_metas'], kwargs.get('attn_masks'), kwargs[Fn.VALUE_CACHE_KEY])
        if fused:
            # every layer is the post-norm (self-attention, Deform3DCrossAttn, FFN) layer: 4 launches per layer
            outs, refs = fused_decoder.run(
                self, query, kwargs['query_pos'], kwargs['value'], reference_points, reg_branches,
                kwargs['img_metas'], kwargs.get('attn_masks'), pipeline, kwargs.get(Fn.VALUE_CACHE_KEY),
                kwargs.get(Fn.QUERY_ORDER_KEY), getattr(self, '_order_pc_range', None), self.return_intermediate, late=late)
            if pipeline is not None:
                pipeline.finish()
            if own_late:
                late.finish()
            if self.return_intermediate:
                return outs, refs
            return outs[0], refs[0]
        # generic path: dense (Q, B, C) rows once (every kernel downstream wants them; the layers would re-copy otherwise)
        output = query = query.contiguous()
        if kwargs.get('query_pos') is not None and not kwargs['query_pos'].is_contiguous():
            kwargs = dict(kwargs)
            kwargs['query_pos'] = kwargs['query_pos'].contiguous()
        aux = None
        cross = [a for layer in self.layers for a in layer.attentions if getattr(a, 'operation_name', '') == 'cross_attn']
        deform_only = bool(cross) and all(isinstance(a, Deform3DCrossAttn) for a in cross)
        for lid, layer in enumerate(self.layers):
            output = layer(output, *args, reference_points=reference_points, **kwargs)
            if reg_branches is not None:
                assert reference_points.shape[-1] == 3
                grad = Fn.wants_grad(reg_branches[lid], output)
                fast = output.is_cuda and output.dtype == torch.float32 and not grad
                if fast and deform_only and lid + 1 < len(self.layers):
                    # the reg branch and the refinement feed the NEXT layer's cross-attention only: run them on the
                    # auxiliary stream next to that layer's self-attention; its gather waits for the event
                    # (Deform3DCrossAttn honours kwargs[REF_EVENT_KEY]; other cross-attentions stay on one stream)
                    aux = Fn.aux_stream(output.device)
                if fast and aux is not None and lid + 1 < len(self.layers):
                    main = torch.cuda.current_stream(output.device)
                    done = torch.cuda.Event()
                    done.record(main)
                    kwargs = dict(kwargs)
                    with torch.cuda.stream(aux):
                        aux.wait_event(done)
                        reference_points = self._refine(reg_branches[lid], output, reference_points, kwargs, True)
                        ev = torch.cuda.Event()
                        ev.record(aux)
                    kwargs[Fn.REF_EVENT_KEY] = ev
                elif fast:
                    reference_points = self._refine(reg_branches[lid], output, reference_points, kwargs,
                                                    lid + 1 < len(self.layers))
                elif output.is_cuda and output.dtype == torch.float32 and reference_points.dtype == torch.float32:
                    # training: the refined points are DETACHED (:213) - nothing of this branch call reaches a loss (the head
                    # applies reg_branches to the layer outputs itself, with gradients) - so it runs without autograd on the
                    # inference kernels, like the branch above
                    with torch.no_grad():
                        reference_points = self._refine(reg_branches[lid], output.detach(), reference_points.detach(),
                                                        dict(kwargs), False)
                else:
                    tmp = reg_branches[lid](output.permute(1, 0, 2))
                    new_ref = torch.zeros_like(reference_points)
                    new_ref[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points[..., :2])
                    new_ref[..., 2:3] = tmp[..., 4:5] + inverse_sigmoid(reference_points[..., 2:3])
                    reference_points = new_ref.sigmoid().detach()
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if pipeline is not None:
            pipeline.finish()
        if own_late:
            late.finish()
        if aux is not None:
            torch.cuda.current_stream(output.device).wait_stream(aux)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points


@TRANSFORMER.register_module()
class Detr3DTransformer(nn.Module):
    """Reference :46-150: split query_embed, predict initial reference points, run the decoder."""

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300,
                 decoder=None, init_cfg=None, **kwargs):
        super().__init__()
        self.decoder = build_transformer_layer_sequence(decoder)
        self.embed_dims = self.decoder.embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.two_stage_num_proposals = two_stage_num_proposals
        self.reference_points = nn.Linear(self.embed_dims, 3)
        _watch_weight_updates(self)

    def train(self, mode=True):
        ops.invalidate_chain_images()
        return super().train(mode)

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (Detr3DCrossAtten, Deform3DCrossAttn)):
                m.init_weight()
        nn.init.xavier_uniform_(self.reference_points.weight)
        nn.init.constant_(self.reference_points.bias, 0.)

    def forward(self, mlvl_feats, query_embed, reg_branches=None, **kwargs):
        assert query_embed is not None
        bs = mlvl_feats[0].size(0)
        query_pos, query = torch.split(query_embed, self.embed_dims, dim=1)
        from . import fused_decoder
        own_late = None
        fast = fused_decoder.fast_input(self, query_embed, mlvl_feats)
        if fast and kwargs.get(Fn.LATE_VALUES_KEY) is None and kwargs.get(Fn.VALUE_CACHE_KEY) is None:
            # the channels-last copy of the pyramid needs nothing but the pyramid: fork it first, before the query side
            cross = [a for layer in self.decoder.layers for a in layer.attentions if getattr(a, 'operation_name', '') == 'cross_attn']
            if cross and all(isinstance(a, Deform3DCrossAttn) for a in cross) and Fn.LateValues.applicable(cross, mlvl_feats):
                # (the loop that gathers the coarse levels from projected rows: the first layer's projection is forked with the copy)
                single = fused_decoder.takes_single_stream_loop(self.decoder, query.unsqueeze(1), mlvl_feats, query_pos[None, :, :3],
                                                                reg_branches, kwargs.get('attn_masks'), query_pos.unsqueeze(1))
                own_late = Fn.LateValues(mlvl_feats, cross[0].value_dtype, coarse_for=cross if single else None)
                kwargs = dict(kwargs)
                kwargs[Fn.LATE_VALUES_KEY] = own_late
        if fast:
            # batch 1, inference: no copies - the column slices of query_embed go to the decoder as strided (Q, 1, C) views
            # and the initial reference points come from one chain launch (no library GEMM / sigmoid / cat in the step)
            reference_points = fused_decoder.initial_reference(self.reference_points, query_pos)
            q_in, pos_in = query.unsqueeze(1), query_pos.unsqueeze(1)
        else:
            query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
            query = query.unsqueeze(0).expand(bs, -1, -1)
            # (training: the library's Linear forward / input gradient, its weight gradient queued with the decoder's - through
            #  torch this one 256 -> 3 Linear was a 25-us library GEMM in the backward pass)
            reference_points = Fn.linear_autograd(query_pos, self.reference_points.weight, self.reference_points.bias).sigmoid()
            q_in, pos_in = query.permute(1, 0, 2), query_pos.permute(1, 0, 2)
        init_reference_out = reference_points
        inter_states, inter_references = self.decoder(
            query=q_in, key=None, value=mlvl_feats, query_pos=pos_in, reference_points=reference_points,
            reg_branches=reg_branches, **kwargs)
        if own_late is not None:
            own_late.finish()
        return inter_states, init_reference_out, inter_references


    def forward_shared(self, mlvl_feats, query_embeds, reg_branches=None, **kwargs):
        """Several query sets through the decoder over ONE feature pyramid: [forward(mlvl_feats, qe, ...) for qe in
        query_embeds], with the layers' value tensors projected once and shared by all the passes.

        This is the distillation step's student side (distillation/distillers/mix_distill.py:92-106): the head runs the
        transformer on its own `query_embedding` and then again on the teacher's (`teacher_queries`,
        dense_heads/detr3d_head_pe.py:560-566 and :617-625) - same weights, same `mlvl_feats`, so value_proj (the
        dominant dense contraction, deform3d_cross_attn.py:264-280) of the second pass is redundant.  With autograd on,
        the shared tensors sit behind one autograd node and receive the gradients of every pass."""
        mods = [a for layer in self.decoder.layers for a in layer.attentions if isinstance(a, Deform3DCrossAttn)]
        share = len(mods) >= 1 and len(mods) <= 8 and mlvl_feats[0].is_cuda and \
            len({(m.num_heads, m.value_dtype, m.embed_dims) for m in mods}) == 1
        if share and Fn.wants_grad(mods[0], *mlvl_feats):
            cache = Fn.raw_pyramid_for_training(mods, mlvl_feats)
            if cache is not None:
                return [self.forward(mlvl_feats, qe, reg_branches=reg_branches, **{Fn.VALUE_CACHE_KEY: cache}, **kwargs)
                        for qe in query_embeds]
            share = all(m.value_proj.bias is not None and m.value_dtype == torch.float32 for m in mods) and \
                mlvl_feats[0].shape[0] == 1 and all(v.dtype == torch.float32 for v in mlvl_feats)
            cache = Fn.project_values_for_layers_autograd(mods, mlvl_feats) if share else None
        elif share and Fn.LateValues.applicable(mods, mlvl_feats):
            # inference, aggregate-then-project: what the passes share is the channels-last copy of the pyramid
            Fn.require_inference(*mlvl_feats)
            late = Fn.LateValues(mlvl_feats, mods[0].value_dtype)
            outs = [self.forward(mlvl_feats, qe, reg_branches=reg_branches, **{Fn.LATE_VALUES_KEY: late}, **kwargs)
                    for qe in query_embeds]
            late.finish()
            return outs
        elif share:
            Fn.require_inference(*mlvl_feats)
            cache = Fn.project_values_for_layers(mods, mlvl_feats)
        else:
            cache = None
        extra = {} if cache is None else {Fn.VALUE_CACHE_KEY: cache}
        return [self.forward(mlvl_feats, qe, reg_branches=reg_branches, **extra, **kwargs) for qe in query_embeds]


@TRANSFORMER.register_module()
class HDetr3DTransformer(Detr3DTransformer):
    """H-DETR variant (reference: utils/h_detr3d_transformer.py:48-160): identical to Detr3DTransformer
    except that `forward(mlvl_feats, query_embed, reg_branches=None, decoder_self_attn_mask=None)` hands a
    per-attention list of masks - [one-to-one / one-to-many block mask, None]
    (dense_heads/h_detr3d_head_pe.py:299-314) - to the decoder as `attn_masks`.  The boolean mask is
    applied inside gd4d_mha_core_fwd (True = masked)."""

    def forward(self, mlvl_feats, query_embed, reg_branches=None, decoder_self_attn_mask=None, **kwargs):
        return super().forward(mlvl_feats, query_embed, reg_branches=reg_branches,
                               attn_masks=decoder_self_attn_mask, **kwargs)

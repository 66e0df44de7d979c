"""autograd.Function wrappers used when gradients are requested (training).

Inference takes the fully fused HIP path (functional.py); with autograd enabled the modules switch to:
  * CrossAttnFunction  - gd4d_cross_attn_fwd / gd4d_cross_attn_bwd (hand-written HIP both ways);
  * ValueProjFunction  - gd4d_value_proj_fwd forward; backward = two library GEMMs (torch.matmul);
  * small dense layers - torch ops, so autograd sees them (their HIP kernels are forward-only).
"""
import torch

from . import ops


class CrossAttnFunction(torch.autograd.Function):
    """out (B,Q,C) = fused projection + mask + softmax + gather + camera-weighted sum
    (deform3d_cross_attn.py:220-324); backward replaces mmcv's ms_deformable_col2im + the elementwise chain."""

    @staticmethod
    def forward(ctx, value, ref, offsets, attn_logits, cam_logits, lidar2img, shapes, pc_range, img_h, img_w):
        value, ref, offsets = value.contiguous(), ref.contiguous(), offsets.contiguous()
        attn_logits, cam_logits = attn_logits.contiguous(), cam_logits.contiguous()
        from . import functional as Fn
        order = Fn.query_order(ref, pc_range)                   # locality order: forward reads and backward atomics
        out = ops.cross_attn_fwd(value, shapes, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range,
                                 img_h, img_w, query_order=order)
        ctx.order = order
        ctx.save_for_backward(value, ref, offsets, attn_logits, cam_logits, lidar2img)
        ctx.meta = (shapes, pc_range, img_h, img_w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        value, ref, offsets, attn_logits, cam_logits, lidar2img = ctx.saved_tensors
        shapes, pc_range, img_h, img_w = ctx.meta
        if ref.shape[0] != 1:
            raise NotImplementedError('gd4d_cross_attn_bwd supports batch 1 per GPU (samples_per_gpu=1)')
        if value.dtype != torch.float32:
            raise NotImplementedError('training needs the fp32 value tensor (value_dtype="fp32")')
        gv, gr, go, ga, gc = ops.cross_attn_bwd(value, shapes, ref, offsets, attn_logits, cam_logits, lidar2img,
                                                pc_range, img_h, img_w, grad_out.contiguous(),
                                                query_order=ctx.order)
        return gv, gr, go, ga.view_as(attn_logits), gc, None, None, None, None, None


class ValueProjFunction(torch.autograd.Function):
    """(B*N, S, C) = value_proj over the NCHW pyramid (deform3d_cross_attn.py:264-280)."""

    @staticmethod
    def forward(ctx, weight, bias, *feats):
        feats = [f.contiguous() for f in feats]
        out = ops.value_proj_fwd(feats, weight.contiguous(), None if bias is None else bias.contiguous())
        ctx.save_for_backward(weight, *feats)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, *feats = ctx.saved_tensors
        c = weight.shape[0]
        go = grad_out.reshape(-1, grad_out.shape[-2], c)                       # (R, S, C)
        r = go.shape[0]
        flat = torch.cat([f.reshape(r, c, -1) for f in feats], dim=2)           # (R, C, S)
        grad_w = torch.einsum('rso,rcs->oc', go, flat) if ctx.needs_input_grad[0] else None
        grad_b = go.sum(dim=(0, 1)) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        grads = [None] * len(feats)
        if any(ctx.needs_input_grad[2:]):
            gin = torch.matmul(go, weight).transpose(1, 2)                      # (R, C, S)
            parts = gin.split([f.shape[-2] * f.shape[-1] for f in feats], dim=2)
            grads = [p.reshape(f.shape) for p, f in zip(parts, feats)]
        return (grad_w, grad_b, *grads)

"""autograd.Function wrappers used when gradients are requested (training).

Inference takes the fully fused HIP path (functional.py); with autograd enabled the modules switch to:
  * PyramidSourceFunction + CrossAttnRawFunction - the inference step's kernels forward (slice-planar copy once, then per
                         layer plan + channel-sliced gather + value_proj of the aggregates) and a backward on the RAW
                         pyramid (gd4d_cross_attn_sliced_bwd.hip): no projected value tensor in either direction (default);
  * CrossAttnFunction  - gd4d_cross_attn_fwd / gd4d_cross_attn_bwd on projected values (GD4D_TRAIN_VALUES=projected, bf16);
  * ValueProjFunction  - gd4d_value_proj_fwd / gd4d_value_proj_bwd_input + _bwd_weight (HIP both ways);
  * LinearFunction     - gd4d_linear_fwd forward, gd4d_linear_fwd with the transposed-weight flag for the input
                         gradient, gd4d_linear_bwd_weight for the weight / bias gradient;
  * LayerNormFunction  - gd4d_layernorm_fwd / gd4d_layernorm_bwd (optionally with the ReLU that follows);
  * MhaCoreFunction    - gd4d_mha_core_fwd (saving the log-sum-exp) / gd4d_mha_core_bwd.
What is left to ATen in a training step: residual adds, ReLU between two Linears, dropout, autograd's own accumulations.
"""
import os

import torch

from . import ops


class CrossAttnFunction(torch.autograd.Function):
    """out (B,Q,C) = fused projection + mask + softmax + gather + camera-weighted sum
    (deform3d_cross_attn.py:220-324); backward replaces mmcv's ms_deformable_col2im + the elementwise chain.

    With `cl` (the channels-last copy of the raw pyramid, no gradient) and the layer's value_proj weight / bias as extra
    inputs, the node ALSO returns the gradient of value_proj's parameters: value_proj is linear, so
        d W_h = sum_q grad_out[q, h]^T (x) agg[q, h],   d b_h = sum_q grad_out[q, h] wsum[q, h]
    with agg / wsum the per-head aggregates of the raw features (gd4d_cross_attn_agg_fwd on the same query-side inputs:
    0.15 ms) - a 0.1-GFLOP contraction over 900 x 8 rows instead of gd4d_value_proj_bwd_weight's 97-GFLOP contraction
    over 739 800 pixel rows (0.39 ms + 757 MB read twice).  ValueProjMultiFunction then skips its weight gradients."""

    @staticmethod
    def forward(ctx, value, ref, offsets, attn_logits, cam_logits, lidar2img, shapes, pc_range, img_h, img_w, cl=None,
                vp_weight=None, vp_bias=None, raw_cam=False):
        """raw_cam: the camera weights are the raw logits, no sigmoid (Deform3DCrossAttnMP's neighbour pass)."""
        value, ref, offsets = value.contiguous(), ref.contiguous(), offsets.contiguous()
        attn_logits, cam_logits = attn_logits.contiguous(), cam_logits.contiguous()
        from . import functional as Fn
        order = Fn.query_order(ref, pc_range)                   # locality order: forward reads and backward atomics
        out = ops.cross_attn_fwd(value, shapes, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range,
                                 img_h, img_w, query_order=order, raw_cam_weights=raw_cam)
        ctx.order = order
        ctx.raw_cam = bool(raw_cam)
        ctx.save_for_backward(value, ref, offsets, attn_logits, cam_logits, lidar2img)
        ctx.meta = (shapes, pc_range, img_h, img_w)
        ctx.cl = cl                                              # (no gradient, not an autograd tensor of this node)
        ctx.has_bias = vp_bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        value, ref, offsets, attn_logits, cam_logits, lidar2img = ctx.saved_tensors
        shapes, pc_range, img_h, img_w = ctx.meta
        if value.dtype != torch.float32:
            raise NotImplementedError('training needs the fp32 value tensor (value_dtype="fp32")')
        grad_out = grad_out.contiguous()
        gv, gr, go, ga, gc = ops.cross_attn_bwd(value, shapes, ref, offsets, attn_logits, cam_logits, lidar2img,
                                                pc_range, img_h, img_w, grad_out, query_order=ctx.order,
                                                raw_cam_weights=ctx.raw_cam)
        gw = gb = None
        if ctx.cl is not None and (ctx.needs_input_grad[11] or ctx.needs_input_grad[12]):
            hh = value.shape[2]
            b, q, c = grad_out.shape
            if isinstance(ctx.cl, ops.PyramidView):          # plan + channel-sliced gather (the inference step's kernels)
                plan = ops.cross_attn_plan_fwd(ctx.cl, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, hh,
                                               query_order=ctx.order, raw_cam_weights=ctx.raw_cam)
                agg, wsum = ops.cross_attn_agg_sliced_fwd(plan), plan.wsum
            else:
                agg, wsum = ops.cross_attn_agg_fwd(ctx.cl, shapes, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range,
                                                   img_h, img_w, hh, query_order=ctx.order, raw_cam_weights=ctx.raw_cam)
            g = grad_out.view(b * q, hh, c // hh)
            gw = torch.einsum('qhd,qhc->hdc', g, agg.view(b * q, hh, c)).reshape(c, c)
            if ctx.has_bias:
                gb = (g * wsum.view(b * q, hh, 1)).sum(0).reshape(c)
        return gv, gr, go, ga.view_as(attn_logits), gc, None, None, None, None, None, None, gw, gb, None


class BoxHeadFunction(torch.autograd.Function):
    """bbox_preds of ALL decoder layers from the reg branches' raw outputs (dense_heads/detr3d_head_pe.py:585-605): columns 0, 1, 4 of
    tmp (.., code) get inverse_sigmoid(reference) added, a sigmoid, the point-cloud range and the depth factor; the others pass.
    forward: gd4d_box_head_fwd on the stacked layers (one launch); backward: a handful of elementwise ops on the same stack -
    the per-layer torch-op epilogue was ~50 launches per layer and step with autograd on.  The reference points carry a gradient
    only where they are not detached (layer 0's `init_reference`; every layer without refinement)."""

    @staticmethod
    def forward(ctx, tmp, ref, pc_range, scale):
        out = ops.box_head_fwd(tmp.contiguous(), ref.contiguous(), pc_range, scale)
        ctx.save_for_backward(out, ref)
        ctx.pc_range, ctx.scale = [float(v) for v in pc_range], float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        out, ref = ctx.saved_tensors
        cols = (0, 1, 4)
        gt = g.clone()
        ds = []
        for i, col in enumerate(cols):                                       # (python scalars only: no host -> device copy inside a capture)
            span = (ctx.pc_range[3 + i] - ctx.pc_range[i]) * ctx.scale
            s = (out[..., col] - ctx.pc_range[i] * ctx.scale) / span         # the sigmoid's value
            ds.append(g[..., col] * (span * (s * (1.0 - s))))
            gt[..., col] = ds[-1]
        d = torch.stack(ds, dim=-1)
        gref = None
        if ctx.needs_input_grad[1]:
            eps = 1e-5                                                       # inverse_sigmoid (deform3d_cross_attn.py:16-31): clamp to [0, 1],
            inside = (ref >= 0) & (ref <= 1)                                 # log(max(x, eps) / max(1 - x, eps))
            dinv = torch.where(ref > eps, 1.0 / ref.clamp(min=eps), torch.zeros_like(ref)) + \
                torch.where((1.0 - ref) > eps, 1.0 / (1.0 - ref).clamp(min=eps), torch.zeros_like(ref))
            gref = torch.where(inside, d * dinv, torch.zeros_like(ref))
        return gt, gref, None, None


class RawPyramid:
    """What the layers of one training step share on the raw-pyramid path: the slice-planar copy (PyramidView) and the sink
    that turns the layers' (pixel, weight, grad_agg row) records into the pyramid's gradient (ops.PyramidGrad), plus the
    per-slice dot-product workspace.  (The token that ties every layer's gather node to the pyramid's node is NOT kept here:
    the autograd nodes hold this object, and a reference back to the graph would keep a step's graph alive into the next.)

    Only the last pass over the records needs gradients: a layer's records are counted right after its plan in the FORWARD
    pass, scan / fill / sort are issued by the first backward node, the reduction as soon as the last layer's grad_agg rows
    exist.  Everything runs on the step's one stream: side streams for the counts, the bookkeeping and the copy were built and
    measured in rounds 3 - 4 and were slower inside a replayed hipGraph (docs/measurements_r04.md section 7); they are gone."""

    def __init__(self):
        self.pyramid = self.shapes = None
        self.layers = self.pending = 0
        self.layer_q = []                              # queries per sample of every registered layer
        self.needs_grad = self.channels_last = False
        self.copy_dtype = torch.float32
        self.sink = self.grads = self._dpart = None
        self._prepared = False
        self.fills_ride = False                        # set by the chain training path (fused_train)

    def register(self, q):
        """A layer (one CrossAttnRawFunction node) with `q` queries per sample: the passes that share this pyramid
        (Detr3DTransformer.forward_shared: student queries, then teacher_queries) need not agree on it."""
        self.layers += 1
        self.pending += 1
        self.layer_q.append(int(q))
        return self.layers - 1

    def _sink_for(self, layer, plan):
        if self.sink is None:
            self.sink = ops.PyramidGrad(self.pyramid, 0, plan.b, plan.q, plan.num_heads, points=plan.points)
        if (plan.b, plan.num_heads) != (self.sink.b, self.sink.hh) or self.layer_q[layer] != plan.q:
            raise ops._lib.Gd4dError(f'RawPyramid: layer {layer} registered with {self.layer_q[layer]} queries hands in a plan of '
                                     f'(B, Q, Hh) = ({plan.b}, {plan.q}, {plan.num_heads}); the sink has B = {self.sink.b}, Hh = {self.sink.hh}')
        return self.sink

    def count_with_gather(self, layer, plan):
        """Forward pass on ONE stream: the records' slots are handed out by the forward gather's launch
        (ops.cross_attn_agg_sliced_fwd(count=...)).  Returns the aggregates; None when the plan / pyramid is not of the kind
        that launch takes (count() + the gather then)."""
        if (not self.needs_grad or plan.items or plan.items_buf is None
                or plan.num_heads != 8 or len(self.pyramid.level_hw) != 4 or self.pyramid.dtype != torch.float32):
            return None
        return ops.cross_attn_agg_sliced_fwd(plan, count=(self._sink_for(layer, plan), layer))

    def count(self, layer, plan):
        """Forward pass: hand the records of `plan` their slots."""
        if not self.needs_grad:
            return
        self._sink_for(layer, plan).add_layer(layer, plan)

    def begin_backward(self):
        """First backward node of the step: the table for every registered layer; scan + fill + sort."""
        if not self.needs_grad or self._prepared or self.sink is None:
            return
        self.sink.alloc_table(self.layer_q)
        if self.fills_ride:
            # one stream, the chain training path: the fills ride in the attention backward's launches (fills_for_launch); what is
            # left of them, and the sort, run before the reduction (_reduce)
            self.sink.scan()
            # launches in front of the reduction that can carry fills: every layer's first backward chain (fills_ride = 'chain'), or
            # the attention backward launches of all layers but the last one walked
            self._launches_left = len(self.layer_q) if self.fills_ride == 'chain' else max(len(self.layer_q) - 1, 0)
            self._prepared = True
            return
        self.sink.prepare()
        self._prepared = True

    def fills_for_launch(self):
        """What the next attention backward carries (ops.mha_core_bwd(fills=...)), or None."""
        if not self.needs_grad or self.sink is None or self.sink._scanned is None:
            return None
        left = getattr(self, '_launches_left', 0)
        if left <= 0 or not self.sink.plans:
            return None
        self._launches_left = left - 1
        return self.sink.take_fills(-(-len(self.sink.plans) // left))

    def layer_done(self):
        """A layer's grad_agg rows are written; after the last one the reduction starts."""
        self.pending -= 1
        if self.pending == 0:
            self._reduce()

    def _reduce(self):
        if not self.needs_grad or self.sink is None or self.grads is not None:
            return
        self.begin_backward()
        if self.sink._scanned is not None:
            self.sink.finish_prepare()               # the fills nobody carried, and the sort
        py = self.pyramid
        grads = [torch.empty((py.rows, h, w, 256) if self.channels_last else (py.rows, 256, h, w), device=py.device, dtype=torch.float32)
                 for h, w in py.level_hw]
        grads = self.sink.reduce(grads, channels_last=self.channels_last)
        self.grads = grads

    def dpart(self, nbytes):
        if self._dpart is None or self._dpart.numel() < nbytes:
            self._dpart = torch.empty(nbytes, device=self.pyramid.device, dtype=torch.uint8)
        return self._dpart


class PyramidSourceFunction(torch.autograd.Function):
    """token = apply(raw, *levels): the slice-planar copy of the NCHW levels (the reference's flatten / transpose / cat,
    deform3d_cross_attn.py:264-276, once for all layers) lands in raw.pyramid; the returned 1-element token is what the
    layers' CrossAttnRawFunction nodes take as their input, so this node's backward runs after ALL of them: it hands out
    the gradient of the levels that the layers' records and grad_agg rows add up to (gd4d_pyramid_grad_*)."""

    @staticmethod
    def forward(ctx, raw, *feats):
        from . import functional as Fn
        if all(ops.PyramidView.is_channels_last_level(f) for f in feats):
            # levels stored (.., H, W, C): gathered in place, no copy; their gradient comes back in the same layout
            raw.pyramid = ops.PyramidView.channels_last_levels(list(feats))
            raw.channels_last = True
            sp = feats[0]
        else:
            sp, hw = ops.pyramid_slice_planar_fwd([f.contiguous() for f in feats], out_dtype=raw.copy_dtype)
            raw.pyramid = ops.PyramidView.slice_planar(sp, hw)
        raw.shapes = [tuple(f.shape) for f in feats]
        raw.needs_grad = any(ctx.needs_input_grad[1:])
        ctx.raw = raw
        ctx.set_materialize_grads(False)
        return torch.empty(1, device=feats[0].device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, _):
        raw = ctx.raw
        if raw.sink is None:
            return (None,) * (1 + len(raw.shapes))
        raw._reduce()                                # (already issued unless a layer's output was never used)
        grads, raw.grads, raw.sink, raw._dpart = raw.grads, None, None, None
        return (None, *[g.unflatten(0, shape[:-3]) if len(shape) > 4 else g for g, shape in zip(grads, raw.shapes)])


class CrossAttnRawFunction(torch.autograd.Function):
    """out (B, Q, C) = value_proj of the per-head aggregates of the RAW pyramid (deform3d_cross_attn.py:220-324 with
    value_proj commuted past the linear sampler): plan + channel-sliced gather + gd4d_value_proj_heads_fwd forward;
    backward: gd4d_value_proj_heads_bwd, gd4d_cross_attn_dot_sliced, gd4d_cross_attn_plan_bwd for the query side, the
    weight / bias gradient of value_proj from the saved aggregates, and the layer's grad_agg rows for the pyramid's gradient
    (assembled over the layers by RawPyramid / PyramidSourceFunction's backward)."""

    @staticmethod
    def forward(ctx, token, ref, offsets, attn_logits, cam_logits, lidar2img, vp_weight, vp_bias, raw, pc_range, img_h, img_w,
                raw_cam=False):
        from . import functional as Fn
        mw, mb = Fn.main_grad(vp_weight), Fn.main_grad(vp_bias)
        ctx.main = (mw, mb) if mw is not None and (vp_bias is None or mb is not None) else None
        ref, offsets = ref.contiguous(), offsets.contiguous()
        attn_logits, cam_logits = attn_logits.contiguous(), cam_logits.contiguous()
        hh = offsets.shape[2]
        order = Fn.query_order(ref, pc_range)
        plan = ops.cross_attn_plan_fwd(raw.pyramid, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, hh,
                                       query_order=order, raw_cam_weights=raw_cam)
        ctx.layer = raw.register(plan.q)
        if ctx.needs_input_grad[0]:
            raw.count(ctx.layer, plan)
        agg = ops.cross_attn_agg_sliced_fwd(plan)
        vp_weight = vp_weight.contiguous()
        out = ops.value_proj_heads_fwd(agg, plan.wsum, vp_weight, None if vp_bias is None else vp_bias.contiguous())
        ctx.save_for_backward(ref, offsets, attn_logits, cam_logits, lidar2img, vp_weight, vp_bias, agg, plan.wsum)
        ctx.plan, ctx.raw = plan, raw
        ctx.meta = (pc_range, img_h, img_w, bool(raw_cam))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ref, offsets, attn_logits, cam_logits, lidar2img, vp_weight, vp_bias, agg, wsum = ctx.saved_tensors
        pc_range, img_h, img_w, raw_cam = ctx.meta
        plan, raw = ctx.plan, ctx.raw
        if plan is None:
            raise RuntimeError('graph-detr4d_amd: the raw-pyramid training path keeps a layer\'s plan and records for ONE backward '
                               'pass; a second backward through the same graph (retain_graph) is not supported - run the forward '
                               'again, or set GD4D_TRAIN_VALUES=projected')
        b, q, c = grad_out.shape
        hh = plan.num_heads
        grad_out = grad_out.contiguous()
        want_pyramid = ctx.needs_input_grad[0] and raw.sink is not None
        if want_pyramid:
            raw.begin_backward()
        gagg, beta = ops.value_proj_heads_bwd(grad_out, vp_weight, None if vp_bias is None else vp_bias.contiguous(), hh,
                                              grad_agg=raw.sink.grad_agg_rows(ctx.layer) if want_pyramid else None)
        raw.layer_done()
        n = plan.pyramid.rows // b
        dpart = ops.cross_attn_dot_sliced(plan, gagg, dpart=raw.dpart(ops.cross_attn_dot_bytes(b, n, q, hh, plan.points)))
        gr, go, ga, gc = ops.cross_attn_plan_bwd(plan, dpart, beta, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range,
                                                 img_h, img_w, raw_cam_weights=raw_cam)
        gw = gb = None
        if ctx.needs_input_grad[6] or (vp_bias is not None and ctx.needs_input_grad[7]):
            if ctx.main is not None and _deferring() is not None and agg.shape[-1] == 256:
                _queue_deferred('vp', (grad_out, (agg, wsum), ctx.main[0], ctx.main[1] if vp_bias is not None else None), _VP_GROUP)
            elif ctx.main is not None:
                ops.value_proj_heads_bwd_weight(grad_out, agg, wsum, want_bias=vp_bias is not None, into=ctx.main)
            else:
                gw, gb = ops.value_proj_heads_bwd_weight(grad_out, agg, wsum, want_bias=vp_bias is not None)
        ctx.plan = None
        return None, gr, go, ga.view_as(attn_logits), gc, None, gw, gb, None, None, None, None, None


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
        go = grad_out.reshape(-1, grad_out.shape[-2], c).contiguous()          # (R, S, C)
        grad_w = grad_b = None
        if ctx.needs_input_grad[0] or (ctx.has_bias and ctx.needs_input_grad[1]):
            grad_w, grad_b = ops.value_proj_bwd_weight(go, feats, want_bias=ctx.has_bias)
        grads = [None] * len(feats)
        if any(ctx.needs_input_grad[2:]):
            gin = ops.value_proj_bwd_input(go, weight, [tuple(f.shape[-2:]) for f in feats])
            grads = [g.view(f.shape) for g, f in zip(gin, feats)]
        return (grad_w, grad_b, *grads)


class ValueProjMultiFunction(torch.autograd.Function):
    """value_proj of every decoder layer over the one pyramid they share (detr3d_transformer.py:192-198): one
    gd4d_value_proj_multi_fwd launch forward; backward runs once, when the gradients of all the layers' value tensors
    exist, and accumulates the pyramid's gradient in place (gd4d_value_proj_bwd_input with accumulate) instead of
    leaving autograd to add NL full-size tensors.

    apply(nl, w_0 .. w_{nl-1}, b_0 .. b_{nl-1}, feat_0 .. feat_{L-1}) -> nl tensors (R, S, C)."""

    @staticmethod
    def forward(ctx, nl, *args):
        ctx.weights_elsewhere = nl < 0            # nl < 0: the gather nodes return the weight / bias gradients (see
        nl = abs(nl)                              # CrossAttnFunction); this node then only propagates to the pyramid
        weights = [w.contiguous() for w in args[:nl]]
        biases = [b.contiguous() for b in args[nl:2 * nl]]
        feats = [f.contiguous() for f in args[2 * nl:]]
        outs = ops.value_proj_multi_fwd(feats, weights, biases)
        ctx.save_for_backward(*weights, *feats)
        ctx.nl = nl
        ctx.set_materialize_grads(False)          # a layer whose value tensor is unused arrives as None, not as zeros
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        nl = ctx.nl
        weights, feats = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]
        c = weights[0].shape[0]
        shapes = [tuple(f.shape[-2:]) for f in feats]
        need_feats = any(ctx.needs_input_grad[1 + 2 * nl:])
        gws, gbs = [None] * nl, [None] * nl
        gin = None
        for i, go in enumerate(grad_outs):
            if go is None:
                continue
            go = go.reshape(-1, go.shape[-2], c).contiguous()
            if (ctx.needs_input_grad[1 + i] or ctx.needs_input_grad[1 + nl + i]) and not ctx.weights_elsewhere:
                gws[i], gbs[i] = ops.value_proj_bwd_weight(go, feats)
            if need_feats:
                gin = ops.value_proj_bwd_input(go, weights[i], shapes, grads=gin, accumulate=gin is not None)
        grads = [None] * len(feats) if gin is None else [g.view(f.shape) for g, f in zip(gin, feats)]
        return (None, *gws, *gbs, *grads)


# Weight gradients that go straight into the flat gradient buffer (ctx.main) are read by nobody before the optimizer: they
# are queued during the backward pass and issued sixteen per launch (gd4d_linear_bwd_weight_group) - at the latest from a
# callback the autograd engine runs when the backward pass ends (inside a hipGraph capture that is still inside the capture).
_WGRAD_GROUP, _LN_GROUP, _VP_GROUP = 16, 32, 8
_MAX_TASKS = 2
# One pair of queues per backward pass (autograd graph-task id): a nested / re-entrant backward (reentrant checkpointing,
# autograd.grad inside a hook) has its own id and must neither flush nor drop the outer pass's entries.  A pass that raised
# never runs its callback and leaves its queues behind (pinning activations): they are never added to anything and are evicted,
# oldest id first, as soon as a later pass needs a queue and _MAX_TASKS exist - passes nest two deep at most in this package and a
# nested pass has the larger id, so what goes is never a live outer pass.
_DEFERRED = {}                # task id -> {'w': [...], 'ln': [...], 'vp': [...]}


def _issue(kind, entries):
    if kind == 'w':
        ops.linear_bwd_weight_group(entries, accumulate=True)
    elif kind == 'vp':                      # value_proj of the aggregates: (grad_out, (agg, wsum), main_w, main_b) per layer
        ops.value_proj_heads_bwd_weight_group([(g, aw[0], aw[1], mw, mb) for g, aw, mw, mb in entries], accumulate=True)
    else:
        ops.layernorm_bwd_reduce_group(entries, accumulate=True)


def _flush_deferred(task=None):
    """Issue what `task` (default: every pass) has queued.  Entries of one launch never share a target: see _queue_deferred."""
    for t in ([task] if task is not None else list(_DEFERRED)):
        queues = _DEFERRED.pop(t, None)
        if not queues:
            continue
        for kind, group in (('w', _WGRAD_GROUP), ('ln', _LN_GROUP), ('vp', _VP_GROUP)):
            q = queues[kind]
            for i in range(0, len(q), group):
                _issue(kind, q[i:i + group])


def take_queued_weight_grads():
    """The weight gradients the current backward pass has queued and not issued yet (<= 16, no two with one target), taken off the
    queue: for a launch that lets them ride (ops.cross_attn_dot_sliced(wgrads=...)).  [] when nothing is queued."""
    task = _deferring()
    queues = _DEFERRED.get(task) if task is not None else None
    if not queues or not queues['w']:
        return []
    q = queues['w'][:]
    del queues['w'][:]
    queues['busy']['w'].clear()
    return q


def _deferring():
    """The current backward pass (graph task id, 0 for the first pass of a process) if parameter gradients may be queued,
    else None - test with `is not None`."""
    if not hasattr(torch._C, '_current_graph_task_id'):
        return None
    task = torch._C._current_graph_task_id()
    return task if task >= 0 else None


def _targets(kind, entry):
    tensors = entry[2:4]                              # (.., main_w, main_b) / (.., main_gamma, main_beta)
    return [t.data_ptr() for t in tensors if t is not None]


def _queue_deferred(kind, entry, group):
    task = _deferring()
    queues = _DEFERRED.get(task)
    if queues is None:
        while len(_DEFERRED) >= _MAX_TASKS:
            del _DEFERRED[min(_DEFERRED)]             # left behind by a backward pass that raised
        queues = _DEFERRED[task] = {'w': [], 'ln': [], 'vp': [], 'busy': {'w': set(), 'ln': set(), 'vp': set()}}
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_deferred(task))
    q, busy = queues[kind], queues['busy'][kind]
    tg = _targets(kind, entry)
    # the grouped kernels add with a plain read-modify-write per problem: two problems of ONE launch that add into the same
    # gradient view (a module applied twice, the head's branches shared by all levels when with_box_refine=False,
    # detr3d_head_pe.py:410-413, tied weights) would race - such an entry starts a new launch
    if any(t in busy for t in tg) or len(q) >= group:
        _issue(kind, q[:])
        del q[:]
        busy.clear()
    q.append(entry)
    busy.update(tg)


def _queue_weight_grad(x, grad_y, main_w, main_b):
    if _deferring() is None:
        ops.linear_bwd_weight(x, grad_y, want_bias=main_b is not None, into=(main_w, main_b))
        return
    _queue_deferred('w', (x, grad_y, main_w, main_b), _WGRAD_GROUP)


class LinearFunction(torch.autograd.Function):
    """nn.Linear for the decoder's dense layers in training, all three products on the library's fp32 MFMA kernels:
    forward gd4d_linear_fwd, input gradient gd4d_linear_fwd with GD4D_LIN_WEIGHT_KN (y = grad W, no transposed copy of
    the weight), weight / bias gradient gd4d_linear_bwd_weight (a vendor GEMM runs that 256 x 256 x ~900 contraction on
    a single compute unit)."""

    @staticmethod
    def forward(ctx, x, weight, bias, main_w=None, main_b=None, relu=False):
        """main_w / main_b: where the weight / bias gradient is to be ADDED by the backward kernel itself (views of a flat
        gradient buffer, dist.FlatGradAllReducer.bind(fuse_weight_grads=True)); autograd then gets None for them.
        relu: y = max(x W^T + b, 0) in the kernel's epilogue (the Linear + ReLU pairs of the FFN and the branches: an
        nn.ReLU(inplace=True) applied to a view of this node's output makes autograd rebuild the base - a clone, two
        copies and a strided copy in the backward pass per pair)."""
        x, weight = x.contiguous(), weight.contiguous()
        y = ops.linear_fwd(x, weight, None if bias is None else bias.contiguous(), relu=relu)
        ctx.save_for_backward(x, weight, *((y,) if relu else ()))     # x: (M, K), see functional.linear_autograd
        ctx.has_bias = bias is not None
        ctx.relu = relu
        ctx.main = (main_w, main_b) if main_w is not None else None
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight = ctx.saved_tensors[:2]
        grad_y = grad_y.contiguous()
        if ctx.relu:
            grad_y = torch.ops.aten.threshold_backward(grad_y, ctx.saved_tensors[2], 0.0)
        gx = ops.linear_fwd(grad_y, weight, weight_kn=True) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if ctx.main is not None:
                _queue_weight_grad(x, grad_y, ctx.main[0], ctx.main[1] if ctx.has_bias else None)
            else:
                gw, gb = ops.linear_bwd_weight(x, grad_y, want_bias=ctx.has_bias)
        return gx, gw, gb, None, None, None


class LinearGroupFunction(torch.autograd.Function):
    """Up to four nn.Linear layers applied to the same x + x2 (Deform3DCrossAttn's camera logits, metre offsets and
    attention logits of query + query_pos, deform3d_cross_attn.py:211, :227, :281) as ONE node: forward = one
    gd4d_linear_group_fwd launch that also stores the sum it forms (kept for the weight gradients: no add launch);
    backward = the input gradients chained through the residual operand of gd4d_linear_fwd (y = grad_g W_g + previous:
    no autograd sums), one tensor for x and x2.  apply(x, x2, g, W_0.., b_0.., main_w_0.., main_b_0..) -> g tensors (M, N_g)."""

    @staticmethod
    def forward(ctx, x, x2, g, *args):
        weights = [w.contiguous() for w in args[:g]]
        biases = [None if b is None else b.contiguous() for b in args[g:2 * g]]
        ctx.mains = [(args[2 * g + i], args[3 * g + i]) if args[2 * g + i] is not None else None for i in range(g)]
        k = x.shape[-1]
        outs, xsum = ops.linear_group_fwd(x.contiguous().view(-1, k), weights, biases, x2=x2.contiguous().view(-1, k), want_xsum=True)
        ctx.save_for_backward(xsum, *weights)
        ctx.g, ctx.has_bias, ctx.xshape = g, [b is not None for b in biases], x.shape
        ctx.set_materialize_grads(False)                 # an unused output arrives as None, not as zeros
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        xsum, weights = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        g = ctx.g
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gx, gws, gbs = None, [None] * g, [None] * g
        for i, gy in enumerate(gys):
            if gy is None:
                continue
            gy = gy.contiguous()
            if need_x:
                gx = ops.linear_fwd(gy, weights[i], weight_kn=True, r1=gx)
            if ctx.needs_input_grad[3 + i] or (ctx.has_bias[i] and ctx.needs_input_grad[3 + g + i]):
                if ctx.mains[i] is not None:
                    _queue_weight_grad(xsum, gy, ctx.mains[i][0], ctx.mains[i][1] if ctx.has_bias[i] else None)
                else:
                    gws[i], gbs[i] = ops.linear_bwd_weight(xsum, gy, want_bias=ctx.has_bias[i])
        if gx is not None:
            gx = gx.view(ctx.xshape)
        return (gx if ctx.needs_input_grad[0] else None, gx if ctx.needs_input_grad[1] else None, None, *gws, *gbs,
                *([None] * (2 * g)))


class LayerNormFunction(torch.autograd.Function):
    """[ReLU] LayerNorm(x [+ res]) over the last dimension: gd4d_layernorm_fwd / gd4d_layernorm_bwd (the decoder layer's
    norms - with the residual sum that precedes them read by the kernel instead of a separate add -, position_encoder's
    LayerNorm + ReLU pairs, deform3d_cross_attn.py:104-111).  The gradient of res is the gradient of x (one tensor)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, main_g=None, main_b=None, res=None):
        x, gamma, beta = x.contiguous(), gamma.contiguous(), beta.contiguous()
        if res is not None:
            if res.shape != x.shape:
                raise ValueError('res must have the shape of x')
            res = res.contiguous()
        ctx.save_for_backward(x, gamma, beta, *(() if res is None else (res,)))
        ctx.eps, ctx.relu = eps, relu
        ctx.main = (main_g, main_b) if main_g is not None and main_b is not None else None
        return ops.layernorm_fwd(x, gamma, beta, eps, res=res, relu=relu)

    @staticmethod
    def backward(ctx, grad_y):
        x, gamma, beta = ctx.saved_tensors[:3]
        res = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        grad_y = grad_y.contiguous()
        dg = db = None
        if ctx.main is not None:                          # dgamma / dbeta added to the flat gradient buffer by the kernels
            if _deferring() is not None:                  # ... the column reduce queued with the weight gradients
                dx, ws, mc = ops.layernorm_bwd(x, gamma, beta, grad_y, ctx.eps, res=res, relu=ctx.relu, defer=True)
                _queue_deferred('ln', (ws, mc, ctx.main[0], ctx.main[1]), _LN_GROUP)
            else:
                dx, _, _ = ops.layernorm_bwd(x, gamma, beta, grad_y, ctx.eps, res=res, relu=ctx.relu, into=ctx.main)
        else:
            dx, dg, db = ops.layernorm_bwd(x, gamma, beta, grad_y, ctx.eps, res=res, relu=ctx.relu)
        return dx, dg, db, None, None, None, None, (dx if res is not None and ctx.needs_input_grad[7] else None)


class MhaCoreFunction(torch.autograd.Function):
    """softmax(q k^T / sqrt(d) + mask) v per head (the middle of nn.MultiheadAttention): gd4d_mha_core_fwd, which also
    saves the row log-sum-exp, and gd4d_mha_core_bwd.  q, k, v: (L, B, C) row-strided (e.g. thirds of a packed
    in-projection).  dropout_p: dropout of the probabilities (nn.MultiheadAttention in training); the mask is a function
    of a seed drawn here (ops.mha_dropout_seed) that the backward passes to the kernels again - nothing else is stored."""

    @staticmethod
    def forward(ctx, q, k, v, attn_mask, num_heads, dropout_p=0.):
        seed = ops.mha_dropout_seed(q.device) if dropout_p > 0. else None
        out, lse = ops.mha_core_fwd(q, k, v, num_heads, attn_mask, want_lse=True, dropout_p=dropout_p, seed=seed)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.mask, ctx.heads, ctx.drop = attn_mask, num_heads, (dropout_p, seed)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = ops.mha_core_bwd(q, k, v, out, grad_out.contiguous(), lse, ctx.heads, ctx.mask,
                                      dropout_p=ctx.drop[0], seed=ctx.drop[1])
        return dq, dk, dv, None, None, None


class MhaCorePackedFunction(torch.autograd.Function):
    """Self-attention core on a packed (L, B, 2C) projection qk = [q | k] (one GEMM for both, transformer_layers.py): the
    backward kernel writes dq and dk into the halves of ONE buffer, so autograd has no slices to put back together (two zero
    fills, two copies and an add per layer otherwise)."""

    @staticmethod
    def forward(ctx, qk, v, attn_mask, num_heads, dropout_p=0.):
        c = qk.shape[-1] // 2
        seed = ops.mha_dropout_seed(qk.device) if dropout_p > 0. else None
        out, lse = ops.mha_core_fwd(qk[..., :c], qk[..., c:], v, num_heads, attn_mask, want_lse=True, dropout_p=dropout_p,
                                    seed=seed)
        ctx.save_for_backward(qk, v, out, lse)
        ctx.mask, ctx.heads, ctx.drop = attn_mask, num_heads, (dropout_p, seed)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        qk, v, out, lse = ctx.saved_tensors
        c = qk.shape[-1] // 2
        dqk, dv = ops.mha_core_bwd(qk[..., :c], qk[..., c:], v, out, grad_out.contiguous(), lse, ctx.heads, ctx.mask,
                                   packed_qk=True, dropout_p=ctx.drop[0], seed=ctx.drop[1])
        return dqk, dv, None, None, None


class Detr3DSampleFunction(torch.autograd.Function):
    """out (B, Q, C) = sum over cameras and levels of sigmoid(logit) x mask x bilinear sample of the NCHW maps at the
    projected reference point (Detr3DCrossAtten, detr3d_transformer.py:373-383 + feature_sampling :397-438):
    gd4d_detr3d_fwd / gd4d_detr3d_bwd.  apply(ref, logits, lidar2img, pc_range, img_h, img_w, *feats)."""

    @staticmethod
    def forward(ctx, ref, logits, lidar2img, pc_range, img_h, img_w, *feats):
        ref, logits = ref.contiguous(), logits.contiguous()
        feats = [f.contiguous() for f in feats]
        ctx.save_for_backward(ref, logits, lidar2img, *feats)
        ctx.meta = (pc_range, img_h, img_w)
        return ops.detr3d_fwd(feats, ref, logits, lidar2img, pc_range, img_h, img_w)['out']

    @staticmethod
    def backward(ctx, grad_out):
        ref, logits, lidar2img, *feats = ctx.saved_tensors
        pc_range, img_h, img_w = ctx.meta
        gf, gl, gr = ops.detr3d_bwd(feats, ref, logits, lidar2img, pc_range, img_h, img_w, grad_out.contiguous(),
                                    want_feats=any(ctx.needs_input_grad[6:]), want_ref=ctx.needs_input_grad[0])
        return (gr, gl, None, None, None, None, *(gf if gf is not None else [None] * len(feats)))


class Detr3DV2SampleFunction(torch.autograd.Function):
    """out (B, Q, C) of Detr3DCrossAttenV2's sampling (detr3d_transformer.py:597-710: projected reference point + per-head
    2-D offsets, softmax over level x point, the (point, level) x (level, point) pairing, visibility, sums):
    gd4d_detr3d_v2_fwd / gd4d_detr3d_v2_bwd.  apply(ref, logits, offsets, lidar2img, pc_range, img_h, img_w, num_heads, *feats)."""

    @staticmethod
    def forward(ctx, ref, logits, offsets, lidar2img, pc_range, img_h, img_w, num_heads, *feats):
        ref, logits, offsets = ref.contiguous(), logits.contiguous(), offsets.contiguous()
        feats = [f.contiguous() for f in feats]
        ctx.save_for_backward(ref, logits, offsets, lidar2img, *feats)
        ctx.meta = (pc_range, img_h, img_w, int(num_heads))
        return ops.detr3d_v2_fwd(feats, ref, logits, offsets, lidar2img, pc_range, img_h, img_w, num_heads)

    @staticmethod
    def backward(ctx, grad_out):
        ref, logits, offsets, lidar2img, *feats = ctx.saved_tensors
        pc_range, img_h, img_w, hh = ctx.meta
        gf, gl, go, gr = ops.detr3d_v2_bwd(feats, ref, logits, offsets, lidar2img, pc_range, img_h, img_w, hh, grad_out.contiguous(),
                                           want_feats=any(ctx.needs_input_grad[8:]), want_ref=ctx.needs_input_grad[0])
        return (gr, gl.view_as(logits), go.view_as(offsets), None, None, None, None, None, *(gf if gf is not None else [None] * len(feats)))

"""The decoder's TRAINING step on row chains: forward with every intermediate kept, backward as chains.

The generic training path (transformer_layers.py, deform3d_cross_attn.py: one autograd node per Linear / LayerNorm / add) runs
a post-norm decoder layer as ~36 launches forward and ~45 backward, nearly all of them 900-row kernels that cost their
~5 us floor whatever they compute: 2.6 of the 6.2 ms of a step (docs/measurements_r04.md).  Between the points where a layer
needs all queries at once (attention core, plan / gather and their backward kernels) everything is row-local, forward AND
backward, so the inference chains (fused_decoder.py, csrc/gd4d_rowchain.hip) carry it here too:

    forward, layer l:   [mha core]  [chain A || position_encoder]  [order, plan, count, gather]  [chain B (+ in_proj of l + 1, reg branch, refine)]
    backward, layer l:  [chain B' bwd (+ position_encoder bwd)]  [heads bwd, gather-dot, plan bwd, value_proj wgrad]  [chain A bwd]
                        [mha bwd x 2]  [in_proj bwd]

The forward programs are the inference programs with `out=` on every operation whose result a gradient needs (the input of each
Linear, the input of each LayerNorm, the FFN's hidden activation); the backward programs are GEMMs over the TRANSPOSED weights'
images (GD4D_CHAIN_MASK_P2 at the FFN's ReLU), LN_BWD and ADD.  Weight / bias gradients are contractions over all rows of what
the chains wrote: queued to gd4d_linear_bwd_weight_group (sixteen per launch) exactly like the generic path; LayerNorm parameter
gradients leave LN_BWD as per-block partial sums in gd4d_layernorm_bwd's workspace layout and are added by
gd4d_layernorm_bwd_reduce_group.  The weights change every step: ops.ImageSet rebuilds all images (W and W^T) in ONE launch
at the start of the forward pass - inside a captured hipGraph that launch replays with the step.

One autograd node for the whole decoder (DecoderTrainFunction): sums of gradients that autograd would form with ATen adds
(x feeds the next layer three times, query_pos every layer twice) are operands of chain operations instead.
Reference: utils/detr3d_transformer.py:153-225 (loop, refinement, detach), utils/deform3d_cross_attn.py:196-339, mmcv's
BaseTransformerLayer / MultiheadAttention / FFN (config ...ceph.py:71-89); what is computed is autograd of exactly those.
GD4D_TRAIN_CHAINS=0 keeps the generic path.
"""
import os
import weakref

import torch

from . import functional as Fn
from . import ops
from .deform3d_cross_attn import Deform3DCrossAttn
from .fused_decoder import ORDER, _plain_reg_branch
from .transformer_layers import FFN, MultiheadAttention


def draw_seeds(n, device):
    """n 64-bit dropout seeds on the device (ops.mha_dropout_seed's recipe: torch's generator, so torch.manual_seed repeats the
    masks and a replayed hipGraph draws new ones) - one launch for all the dropout sites of a step."""
    return torch.randint(-2 ** 62, 2 ** 62, (n,), device=device, dtype=torch.int64)


def _dropouts(layer):
    """The layer's dropout probabilities where the chains apply them - (attention probabilities, after out_proj, after
    output_proj, after the FFN's ReLU, after its second Linear) - or None when the modules drop anywhere else.
    mmcv: MultiheadAttention = identity + dropout_layer(proj_drop(out)); Deform3DCrossAttn: dropout(output_proj(..)) + residuals
    (deform3d_cross_attn.py:336); FFN: Linear, ReLU, Dropout, Linear, Dropout, then identity + dropout_layer(out)."""
    if not layer.training:
        return (0., 0., 0., 0., 0.)
    sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
    prob = lambda m: float(getattr(m, 'p', 0.))          # noqa: E731
    if prob(sa.proj_drop) > 0. and prob(sa.dropout_layer) > 0.:
        return None                                       # two dropouts in a row: not one Bernoulli mask
    if prob(ffn.layers[2]) > 0. and prob(ffn.dropout_layer) > 0.:
        return None
    return (float(sa.attn_drop), max(prob(sa.proj_drop), prob(sa.dropout_layer)), prob(ca.dropout), prob(ffn.layers[0][2]),
            max(prob(ffn.layers[2]), prob(ffn.dropout_layer)))


def applicable(decoder, query, query_pos, value, reference_points, reg_branches, attn_masks, raw_entry, args, kwargs):
    if os.environ.get('GD4D_TRAIN_CHAINS', '1') == '0' or not torch.is_grad_enabled() or args:
        return False
    if raw_entry is None or not isinstance(raw_entry, dict) or query_pos is None or 'img_metas' not in kwargs:
        return False
    if kwargs.get('key') is not None or kwargs.get('key_padding_mask') is not None or kwargs.get('query_key_padding_mask') is not None:
        return False
    if not query.is_cuda or query.dtype != torch.float32 or query.dim() != 3 or query.shape[1] != 1 or query_pos.shape != query.shape:
        return False
    if reference_points is None or reference_points.shape[-1] != 3 or reference_points.shape[0] != 1:
        return False
    c = query.shape[-1]
    if c != 256 or not isinstance(value, (list, tuple)) or value[0].shape[0] != 1:
        return False
    if isinstance(attn_masks, (list, tuple)):           # [self-attention mask, cross-attention mask]: Deform3DCrossAttn takes none
        if len(attn_masks) != 2:
            return False
        attn_masks = attn_masks[0]
    if attn_masks is not None and not (torch.is_tensor(attn_masks) and attn_masks.dim() == 2):
        return False
    for layer in decoder.layers:
        if tuple(layer.operation_order) != ORDER or len(layer.attentions) != 2 or len(layer.ffns) != 1 or _dropouts(layer) is None:
            return False
        sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
        if not isinstance(sa, MultiheadAttention) or sa.batch_first or sa.embed_dims != c or c // sa.num_heads != 32:
            return False
        if type(ca) is not Deform3DCrossAttn or ca.embed_dims != c or len(value) != ca.num_levels or ca.depth_encode:
            return False
        if not (ca.num_points == 4 or (ca.num_points == 8 and ca.num_heads == 8)):      # (other counts: the per-module path pads them)
            return False
        entry = raw_entry.get(id(ca))
        if entry is None or len(entry) < 4 or not isinstance(entry[3], tuple):
            return False
        if not isinstance(ffn, FFN) or len(ffn.layers) != 3 or not ffn.add_identity or ffn.feedforward_channels % 64 \
                or ffn.feedforward_channels > 512 or len(ffn.layers[0]) != 3:
            return False
        lins = [sa.attn.out_proj, ca.cam_attention_weights, ca.deform_sampling_offsets, ca.attention_weights, ca.value_proj,
                ca.output_proj, ca.position_encoder[0], ca.position_encoder[3], ffn.layers[0][0], ffn.layers[1]]
        if any(not isinstance(m, torch.nn.Linear) or m.bias is None for m in lins) or sa.attn.in_proj_bias is None:
            return False
        if ca.position_encoder[0].in_features != 3:
            return False
    if reg_branches is not None and any(_plain_reg_branch(reg_branches[i], c) is None for i in range(len(decoder.layers))):
        return False
    return True


def _layer_params(layer):
    """The 32 parameters of a layer, in the order DecoderTrainFunction takes and returns gradients for them."""
    sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
    pe = ca.position_encoder
    mods = [sa.attn.out_proj, layer.norms[0], ca.cam_attention_weights, ca.deform_sampling_offsets, ca.attention_weights,
            pe[0], pe[1], pe[3], pe[4], ca.value_proj, ca.output_proj, layer.norms[1], ffn.layers[0][0], ffn.layers[1],
            layer.norms[2]]
    out = [sa.attn.in_proj_weight, sa.attn.in_proj_bias]
    for m in mods:
        out += [m.weight, m.bias]
    return out


NAMES = ['in_w', 'in_b', 'out_w', 'out_b', 'n0_w', 'n0_b', 'cam_w', 'cam_b', 'off_w', 'off_b', 'att_w', 'att_b', 'pe0_w', 'pe0_b',
         'pe1_w', 'pe1_b', 'pe3_w', 'pe3_b', 'pe4_w', 'pe4_b', 'vp_w', 'vp_b', 'op_w', 'op_b', 'n1_w', 'n1_b', 'f0_w', 'f0_b',
         'f1_w', 'f1_b', 'n2_w', 'n2_b']
PER_LAYER = len(NAMES)


class _Images:
    """The ImageSet of a decoder (+ its reg branches): made once per set of parameter tensors, refreshed every forward."""

    def __init__(self, decoder, reg_branches, device):
        self.set = ops.ImageSet(device)
        self.layers = []
        c = decoder.embed_dims
        for lid, layer in enumerate(decoder.layers):
            p = dict(zip(NAMES, _layer_params(layer)))
            add = self.set.add
            w_in = p['in_w'].detach()
            im = dict(inproj=add([w_in]), inproj_qk_t=add([w_in[:2 * c]], transposed=True), inproj_v_t=add([w_in[2 * c:]], transposed=True))
            three = [p['cam_w'].detach(), p['off_w'].detach(), p['att_w'].detach()]
            im['three'], im['three_t'] = add(three), add(three, transposed=True)
            im['three_b'] = self.set.add_concat([p['cam_b'].detach(), p['off_b'].detach(), p['att_b'].detach()])
            for key, name in (('outproj', 'out_w'), ('pos3', 'pe3_w'), ('outputproj', 'op_w'), ('ffn0', 'f0_w'), ('ffn1', 'f1_w')):
                im[key], im[key + '_t'] = add([p[name].detach()]), add([p[name].detach()], transposed=True)
            im['vp'] = add([p['vp_w'].detach()])
            if reg_branches is not None:
                im['reg'] = [add([lin.weight.detach()], exact=True) for lin in _plain_reg_branch(reg_branches[lid], c)]
            self.layers.append(im)
        self.signature = self.set.signature()


def _sources(decoder, reg_branches):
    """The data pointers an _Images job table was written from, in ImageSet.sources order."""
    c = decoder.embed_dims
    out = []
    for lid, layer in enumerate(decoder.layers):
        p = dict(zip(NAMES, _layer_params(layer)))
        w_in = p['in_w']
        out += [w_in.data_ptr(), w_in.data_ptr(), w_in[2 * c:].data_ptr()]
        out += [p[k].data_ptr() for k in ('cam_w', 'off_w', 'att_w')] * 2 + [p[k].data_ptr() for k in ('cam_b', 'off_b', 'att_b')]
        for name in ('out_w', 'pe3_w', 'op_w', 'f0_w', 'f1_w'):
            out += [p[name].data_ptr()] * 2
        out.append(p['vp_w'].data_ptr())
        if reg_branches is not None:
            out += [lin.weight.data_ptr() for lin in _plain_reg_branch(reg_branches[lid], c)]
    return tuple(out)


_IMAGE_SETS = weakref.WeakKeyDictionary()      # decoder -> (key, _Images): NOT an attribute of the module (a deepcopy / pickle of the
#                                                 model - EMA hooks, torch.save(model) - must not meet raw device pointers)


def _images(decoder, reg_branches, device):
    st = _IMAGE_SETS.get(decoder)
    key = (None if reg_branches is None else id(reg_branches), str(device))
    if st is not None and st[0] == key and _sources(decoder, reg_branches) == st[1].signature:
        return st[1]                                   # the parameters still live where the job table points
    imgs = _Images(decoder, reg_branches, device)
    _IMAGE_SETS[decoder] = (key, imgs)
    return imgs


class _Meta:
    """What DecoderTrainFunction needs besides tensors (one object: autograd passes it through untouched)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class DecoderTrainFunction(torch.autograd.Function):
    """(out_all (NL, Q, 1, C), ref_all (NL, 1, Q, 3)) = the decoder on `token`'s raw pyramid; see the module docstring.
    apply(meta, token, query, query_pos, reference_points, *parameters (32 per layer, _layer_params order))."""

    @staticmethod
    def forward(ctx, meta, token, query, query_pos, ref0, *params):
        dec, raw = meta.decoder, meta.raw
        layers = list(dec.layers)
        nl = len(layers)
        q, _, c = query.shape
        dev = query.device
        f32 = torch.float32
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=f32)    # noqa: E731
        imgs = _images(dec, meta.reg_branches, dev)
        imgs.set.refresh()
        x = query.contiguous().view(q, c)
        pos = query_pos.contiguous().view(q, c)
        ref = ref0.detach().contiguous()
        mask = meta.attn_mask
        want_pyramid = ctx.needs_input_grad[1]
        out_all, ref_all = new(nl, q, 1, c), new(nl, 1, q, 3)
        saved = []
        drops = [_dropouts(layer) for layer in layers]
        seeds = draw_seeds(5 * nl, dev) if any(any(pr > 0. for pr in d) for d in drops) else None
        beside = meta.reg_branches is not None and ops.handoff_enabled(dev, 'GD4D_TRAIN_REG_BESIDE')
        flags = ops.handoff_flags(dev, nl, (q + 15) // 16 // 8 * 8 + 16, ('train', Fn.slot_key(dev))) if beside else None   # hand-off flags per row block
        err = ops.handoff_error_word(dev) if beside else None    # a WAIT that gives up counts here (and poisons its rows): ops.check_handoff
        qkv, xp = new(q, 1, 3 * c), new(q, c)
        im0 = imgs.layers[0]
        p0 = dict(zip(NAMES, params[:PER_LAYER]))
        # K / V also as the attention core's split-bf16 operands (one pair of planes for all layers: a layer's core has read them
        # before the next in-projection writes them); the fp32 rows stay for the attention backward
        kv = None
        if (c == 256 and layers[0].attentions[0].num_heads == 8 and (mask is None or mask.dim() == 2)
                and os.environ.get('GD4D_MHA_FP32') != '1'):
            kv = ops.KVPlanes(q, c, dev, heads=8)
        ops.row_chain_fwd([ops.chain_load(0, x, pos, out=xp), ops.chain_load(1, x),
                           ops.chain_gemm_two_sources(0, 1, 2 * c, im0['inproj'], p0['in_b'], qkv.view(q, -1), kv=kv, keep_fp32=True)], q)
        # the locality order of the INITIAL reference points for every layer, as the inference loop (fused_decoder.run_single): the
        # refinements move a point little, a stale order costs a gather ~2 us, a fresh one a launch
        order = Fn.query_order(ref, layers[0].attentions[1].pc_range)
        for lid, layer in enumerate(layers):
            p = dict(zip(NAMES, params[lid * PER_LAYER:(lid + 1) * PER_LAYER]))
            im = imgs.layers[lid]
            sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
            hh, npt, nlv, ncam = ca.num_heads, ca.num_points, ca.num_levels, ca.num_cams
            fc = ffn.feedforward_channels
            last = lid + 1 == nl
            s = _Meta(x=x, xp=xp, qkv=qkv, ref=ref)
            # the dropout sites of the layer: (seed, p) each, or None
            s.drop = [(seeds[5 * lid + i:5 * lid + i + 1], pr) if pr > 0. else None for i, pr in enumerate(drops[lid])]
            qh, kh, vh = qkv.split(c, dim=-1)
            if kv is not None:
                s.o, s.lse = ops.mha_core_presplit_fwd(qh, kv, sa.num_heads, mask, want_lse=True,
                                                       dropout_p=drops[lid][0], seed=s.drop[0][0] if s.drop[0] else None)
            else:
                s.o, s.lse = ops.mha_core_fwd(qh, kh, vh, sa.num_heads, mask, want_lse=True,
                                              dropout_p=drops[lid][0], seed=s.drop[0][0] if s.drop[0] else None)
            s.y1, s.x1, s.x1p = new(q, c), new(q, c), new(q, c)
            s.cam, s.off, s.att = new(1, q, ncam), new(1, q, hh * npt * 3), new(1, q, hh * nlv * npt)
            prog_a = [ops.chain_load(0, s.o.view(q, c)),
                      ops.chain_gemm(0, im['outproj'], p['out_b'], dst=1, add=x, out=s.y1, dropout=s.drop[1]),
                      ops.chain_layernorm(1, layer.norms[0], dst=2, out=s.x1),
                      ops.chain_add(0, 2, c, add=pos, out=s.x1p),
                      ops.chain_gemm_three_outputs(0, [ca.cam_attention_weights, ca.deform_sampling_offsets, ca.attention_weights],
                                                   [s.cam.view(q, -1), s.off.view(q, -1), s.att.view(q, -1)],
                                                   stacked=(im['three'], im['three_b']))]
            pe = ca.position_encoder
            s.mid0, s.a1, s.mid1, s.pos_feat = new(q, c), new(q, c), new(q, c), new(q, c)
            s.isig = new(q, 3)                        # inverse_sigmoid(ref): the input of position_encoder's first Linear (its weight gradient)
            prog_p = [ops.chain_load(0, ref.view(q, 3), inv_sigmoid=True, out=s.isig),
                      ops.chain_small_linear(0, p['pe0_w'], p['pe0_b'], 1, out=s.mid0),
                      ops.chain_layernorm(1, pe[1], dst=2, relu=True, out=s.a1),
                      ops.chain_gemm(2, im['pos3'], p['pe3_b'], dst=1, out=s.mid1),
                      ops.chain_layernorm(1, pe[4], relu=True, out=s.pos_feat)]
            ops.row_chain2_fwd(prog_a, prog_p, q)
            # plan + gather on the raw pyramid (autograd.CrossAttnRawFunction's forward without its value_proj launch)
            s.plan = ops.cross_attn_plan_fwd(raw.pyramid, ref, s.off.view(1, q, hh, npt, 3), s.att.view(1, q, hh, nlv, npt), s.cam,
                                             meta.lidar2img, ca.pc_range, meta.img_h, meta.img_w, hh, query_order=order, both=True)
            s.layer = raw.register(s.plan.q)
            s.agg = raw.count_with_gather(s.layer, s.plan) if want_pyramid else None    # slots + gather in one launch
            if s.agg is None:
                if want_pyramid:
                    raw.count(s.layer, s.plan)
                s.agg = ops.cross_attn_agg_sliced_fwd(s.plan)
            s.v, s.y2, s.x2, s.h, s.y3 = new(q, c), new(q, c), new(q, c), new(q, fc), new(q, c)
            x3 = out_all[lid].view(q, c)
            prog = [ops.chain_headgemm(s.agg, s.plan.wsum, im['vp'], p['vp_b'], dst=0, out=s.v),
                    ops.chain_load(3, s.x1, s.pos_feat),
                    ops.chain_gemm(0, im['outputproj'], p['op_b'], dst=1, res=3, out=s.y2, dropout=s.drop[2]),
                    ops.chain_layernorm(1, layer.norms[1], dst=2, out=s.x2),
                    ops.chain_gemm(2, im['ffn0'], p['f0_b'], dst=0, relu=True, out=s.h, dropout=s.drop[3]),   # h: after the dropout
                    ops.chain_gemm(0, im['ffn1'], p['f1_b'], dst=1, res=2, out=s.y3, dropout=s.drop[4]),
                    ops.chain_layernorm(1, layer.norms[2], dst=3, out=x3)]
            if not last:
                qkv, xp = new(q, 1, 3 * c), new(q, c)
                pn = dict(zip(NAMES, params[(lid + 1) * PER_LAYER:(lid + 2) * PER_LAYER]))
                prog += [ops.chain_add(0, 3, c, add=pos, out=xp),
                         ops.chain_gemm_two_sources(0, 3, 2 * c, imgs.layers[lid + 1]['inproj'], pn['in_b'], qkv.view(q, -1), kv=kv, keep_fp32=True)]
            if meta.reg_branches is not None:
                # reg branch + refinement (:199-214); the refined points are DETACHED (:213): no gradient leaves this tail.  It needs
                # the layer's output only, so it runs as the launch's second program beside the next layer's in-projection: the first
                # program SIGNALs once its rows of x3 are stored (gd4d.h: the signalling program goes first).
                lins = _plain_reg_branch(meta.reg_branches[lid], c)
                new_ref = ref_all[lid]
                tail = [ops.chain_wait(flags[lid], err), ops.chain_load(3, x3)] if (beside and not last) else []
                src, tmp = 3, (1, 2)
                for i, (lin, wimg) in enumerate(zip(lins, im['reg'])):
                    tail.append(ops.chain_gemm(src, wimg, lin.bias, dst=tmp[i % 2], relu=i + 1 < len(lins), exact=True))
                    src = tmp[i % 2]
                tail.append(ops.chain_refine(src, ref, new_ref))
                if last or not beside:                       # nothing to run beside (or no hand-off on this device): the tail closes the chain
                    ops.row_chain_fwd(prog + tail, q)
                else:
                    at = 7                                  # after the LayerNorm that stores x3
                    ops.row_chain2_fwd(prog[:at] + [ops.chain_signal(flags[lid])] + prog[at:], tail, q)
            else:
                new_ref = ref
                ref_all[lid].copy_(ref)
                ops.row_chain_fwd(prog, q)
            saved.append(s)
            x, ref = x3, new_ref
        if beside:
            ops.poll_handoff(dev)                            # non-blocking; ops.check_handoff() is the blocking form
        ctx.meta, ctx.saved, ctx.params, ctx.imgs = meta, saved, params, imgs
        ctx.versions = [p_._version for p_ in params]        # the backward chains read the images of THESE weights (and their transposes)
        ctx.pos = pos
        ctx.set_materialize_grads(False)
        if not dec.return_intermediate:
            out_all, ref_all = out_all[-1:], ref_all[-1:]
        ctx.mark_non_differentiable(ref_all)
        return out_all, ref_all

    @staticmethod
    def backward(ctx, g_out_all, _g_ref):
        from .autograd import _LN_GROUP, _VP_GROUP, _WGRAD_GROUP, _deferring, _queue_deferred, take_queued_weight_grads
        meta, saved, params, imgs = ctx.meta, ctx.saved, ctx.params, ctx.imgs
        if saved is None:
            raise RuntimeError('graph-detr4d_amd: the chain training path keeps its activations for ONE backward pass '
                               '(retain_graph is not supported; GD4D_TRAIN_CHAINS=0 selects the generic path)')
        ctx.saved = None
        if [p_._version for p_ in params] != ctx.versions:
            raise RuntimeError('graph-detr4d_amd: a decoder parameter was modified in place between the forward and the backward pass '
                               '(the backward chains use the weight images the forward pass built) - as autograd itself refuses '
                               'a saved tensor that was modified in place')
        dec, raw = meta.decoder, meta.raw
        layers = list(dec.layers)
        nl = len(layers)
        pos = ctx.pos
        q, c = pos.shape
        dev = pos.device
        f32 = torch.float32
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=f32)    # noqa: E731
        blocks = (q + 15) // 16
        want_pyramid = ctx.needs_input_grad[1] and raw.sink is not None
        need = ctx.needs_input_grad[5:]
        grads = [None] * len(params)
        local_w, local_ln = [], []
        deferring = _deferring() is not None
        ride_wgrads = deferring                              # queued weight gradients ride in the gather-dots' launches

        def targets(iw, ib, rows):
            """Where the gradients of parameters iw / ib (a weight and its bias, or gamma and beta) go: their views of the flat
            gradient buffer (queued: added by the grouped launches at the end of the pass) or tensors returned to autograd."""
            mw = Fn.main_grad(params[iw], rows) if deferring else None
            mb = Fn.main_grad(params[ib], rows) if deferring else None
            if mw is not None and mb is not None:
                return mw, mb, True
            for i in (iw, ib):
                if grads[i] is None:
                    grads[i] = torch.empty_like(params[i])
            if rows is None:
                return grads[iw], grads[ib], False
            return grads[iw][rows[0]:rows[1]], grads[ib][rows[0]:rows[1]], False

        def wgrad(base, wname, x_in, gy, rows=None):
            iw, ib = base + NAMES.index(wname), base + NAMES.index(wname[:-1] + 'b')
            if not (need[iw] or need[ib]):
                return
            tw, tb, queued = targets(iw, ib, rows)
            if queued:
                _queue_deferred('w', (x_in, gy, tw, tb), _WGRAD_GROUP)
            else:
                local_w.append((x_in, gy, tw, tb))

        def lngrad(base, wname, ws):
            iw, ib = base + NAMES.index(wname), base + NAMES.index(wname[:-1] + 'b')
            if not (need[iw] or need[ib]):
                return
            tw, tb, queued = targets(iw, ib, None)
            if queued:
                _queue_deferred('ln', (ws, (q, c), tw, tb), _LN_GROUP)
            else:
                local_ln.append((ws, (q, c), tw, tb))

        part = lambda: new(blocks * 2 * c)                                   # noqa: E731
        g_next = None                        # gradient of a layer's OUTPUT coming from the layer after it
        gpos = None                          # running gradient of query_pos
        g_ref0 = padz = carry = carry_keep = None
        fills_in = os.environ.get('GD4D_FILLS_RIDE', 'chain')               # 'chain' | 'mha': which launches carry the record fills
        if want_pyramid:
            # the pyramid gradient's record fills ride as guests of every layer's first backward chain (57 of the 256 compute units
            # busy for 50-70 us); until round 6: of the attention backward's dk / dv launches, which they made 20-38 us longer
            raw.fills_ride = 'chain' if fills_in == 'chain' else True
            raw.begin_backward()
        for lid in range(nl - 1, -1, -1):
            layer, s, im = layers[lid], saved[lid], imgs.layers[lid]
            base = lid * PER_LAYER
            sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
            hh, ncam = ca.num_heads, ca.num_cams
            fc = ffn.feedforward_channels
            pe = ca.position_encoder
            g_here = None
            if g_out_all is not None:
                g_here = g_out_all[lid if dec.return_intermediate else 0].contiguous().view(q, c) \
                    if (dec.return_intermediate or lid == nl - 1) else None
            if g_here is None and g_next is None and carry is None:
                g_here = torch.zeros(q, c, device=dev, dtype=f32)
            if carry is None:
                first, second = (g_here, g_next) if g_here is not None else (g_next, None)
                head, g_src = [ops.chain_load(0, first, second)], 0
            elif g_here is not None:                   # the layer above's in-projection backward left its input gradient in buffer 3
                head, g_src = carry + [ops.chain_add(0, 3, c, add=g_here)], 0
            else:
                head, g_src = carry, 3
            gy3, ghp, gy2, gv, gmid1, gmid0 = new(q, c), new(q, fc), new(q, c), new(q, c), new(q, c), new(q, c)
            ws_n2, ws_n1, ws_p4, ws_p1, ws_n0 = part(), part(), part(), part(), part()
            # With dropout after a Linear the gradient at that Linear's output is the masked one (DROPMASK regenerates the
            # forward's mask); the unmasked one is still what the residual branch passes on.
            d_out, d_op, d_h, d_f1 = s.drop[1], s.drop[2], s.drop[3], s.drop[4]
            gy3m = new(q, c) if d_f1 else gy3
            gy2m = new(q, c) if d_op else gy2
            prog = head + [ops.chain_layernorm_bwd(g_src, s.y3, layer.norms[2], dst=0, out=None if d_f1 else gy3, part=ws_n2)]
            src = 0
            if d_f1:
                prog.append(ops.chain_dropmask(0, 1, c, d_f1[0], d_f1[1], out=gy3m))
                src = 1
            prog += [ops.chain_gemm(src, im['ffn1_t'], None, dst=2, mask=s.h, mask_scale=1.0 / (1.0 - d_h[1]) if d_h else 0., out=ghp),
                     ops.chain_gemm(2, im['ffn0_t'], None, dst=1, res=0),
                     ops.chain_layernorm_bwd(1, s.y2, layer.norms[1], dst=1, out=gy2, part=ws_n1)]
            src = 1
            if d_op:
                prog.append(ops.chain_dropmask(1, 0, c, d_op[0], d_op[1], out=gy2m))
                src = 0
            prog += [ops.chain_gemm(src, im['outputproj_t'], None, out=gv),
                     ops.chain_layernorm_bwd(1, s.mid1, pe[4], dst=2, relu=True, out=gmid1, part=ws_p4),
                     ops.chain_gemm(2, im['pos3_t'], None, dst=0),
                     ops.chain_layernorm_bwd(0, s.mid0, pe[1], dst=0, relu=True, out=gmid0, part=ws_p1)]
            ops.row_chain_fwd(prog, q, fills=raw.fills_for_launch() if want_pyramid and fills_in == 'chain' else None)
            wgrad(base, 'f1_w', s.h, gy3m)
            wgrad(base, 'f0_w', s.x2, ghp)
            wgrad(base, 'op_w', s.v, gy2m)
            wgrad(base, 'pe3_w', s.a1, gmid1)
            wgrad(base, 'pe0_w', s.isig, gmid0)
            lngrad(base, 'n2_w', ws_n2); lngrad(base, 'n1_w', ws_n1); lngrad(base, 'pe4_w', ws_p4); lngrad(base, 'pe1_w', ws_p1)
            # the gather's backward (autograd.CrossAttnRawFunction.backward)
            p = dict(zip(NAMES, params[base:base + PER_LAYER]))
            vp_w, vp_b = p['vp_w'].detach().contiguous(), p['vp_b'].detach().contiguous()
            n_cam_rows = s.plan.pyramid.rows
            # the weight gradients queued so far (this layer's chain B, the layer above's chain A / in-projection) ride in the
            # gather-dot's launch instead of waiting for the pass's end
            riders = take_queued_weight_grads() if ride_wgrads and ops.wgrads_ride_with(s.plan) else None
            gagg, beta = ops.value_proj_heads_bwd(gv.view(1, q, c), vp_w, vp_b, hh,
                                                  grad_agg=raw.sink.grad_agg_rows(s.layer) if want_pyramid else None)
            raw.layer_done()
            dpart = ops.cross_attn_dot_sliced(s.plan, gagg, dpart=raw.dpart(ops.cross_attn_dot_bytes(1, n_cam_rows, q, hh, s.plan.points)), wgrads=riders)
            off5, att5 = s.off.view(1, q, hh, ca.num_points, 3), s.att.view(1, q, hh, ca.num_levels, ca.num_points)
            gr, go, ga, gc = ops.cross_attn_plan_bwd(s.plan, dpart, beta, s.ref, off5, att5, s.cam, meta.lidar2img, ca.pc_range,
                                                     meta.img_h, meta.img_w)
            ivw, ivb = base + NAMES.index('vp_w'), base + NAMES.index('vp_b')
            if need[ivw] or need[ivb]:
                mw, mb = (Fn.main_grad(params[ivw]), Fn.main_grad(params[ivb])) if deferring else (None, None)
                if mw is not None and mb is not None:
                    _queue_deferred('vp', (gv.view(1, q, c), (s.agg, s.plan.wsum), mw, mb), _VP_GROUP)
                else:
                    grads[ivw], grads[ivb] = ops.value_proj_heads_bwd_weight(gv.view(1, q, c), s.agg, s.plan.wsum, want_bias=True)
            s.plan = None
            # chain A backward
            widths = [ncam, go.numel() // q, ga.numel() // q]
            kp = im['three_t'].k
            if kp > sum(widths) and (padz is None or padz.shape[1] != kp - sum(widths)):
                padz = torch.zeros(q, kp - sum(widths), device=dev, dtype=f32)      # the zero columns behind the stacked gradients
            gx1p, gy1, g_o = new(q, c), new(q, c), new(q, 1, c)
            gcat = [gc.contiguous().view(q, -1), go.contiguous().view(q, -1), ga.contiguous().view(q, -1)]
            prog, col = [], 0
            for t in gcat:
                prog.append(ops.chain_load(0, t, dst_col=col))
                col += t.shape[1]
            if kp > sum(widths):
                prog.append(ops.chain_load(0, padz, dst_col=col))
            prog += [ops.chain_gemm(0, im['three_t'], None, dst=2, out=gx1p),
                     ops.chain_add(1, 2, c, add=gy2),
                     ops.chain_layernorm_bwd(1, s.y1, layer.norms[0], dst=1, out=gy1, part=ws_n0)]
            gy1m = new(q, c) if d_out else gy1
            if d_out:
                prog.append(ops.chain_dropmask(1, 0, c, d_out[0], d_out[1], out=gy1m))
            prog.append(ops.chain_gemm(0 if d_out else 1, im['outproj_t'], None, out=g_o.view(q, c)))
            ops.row_chain_fwd(prog, q)
            wgrad(base, 'cam_w', s.x1p, gcat[0]); wgrad(base, 'off_w', s.x1p, gcat[1]); wgrad(base, 'att_w', s.x1p, gcat[2])
            wgrad(base, 'out_w', s.o.view(q, c), gy1m)
            lngrad(base, 'n0_w', ws_n0)
            qh, kh, vh = s.qkv.split(c, dim=-1)
            dqk, dv = ops.mha_core_bwd(qh, kh, vh, s.o, g_o, s.lse, sa.num_heads, meta.attn_mask, packed_qk=True,
                                       dropout_p=s.drop[0][1] if s.drop[0] else 0., seed=s.drop[0][0] if s.drop[0] else None,
                                       fills=raw.fills_for_launch() if want_pyramid and fills_in != 'chain' else None)
            # in-projection backward: launched with the NEXT layer's chain B' backward (its result stays in LDS), alone for layer 0
            gx = new(q, c) if lid == 0 else None
            gpos_new = new(q, c)
            prog = [ops.chain_load(0, dqk.view(q, 2 * c)), ops.chain_load(1, dv.view(q, c)),
                    ops.chain_gemm(0, im['inproj_qk_t'], None, dst=2),
                    ops.chain_gemm(1, im['inproj_v_t'], None, dst=3, res=2, add=gy1, out=gx),
                    ops.chain_load(0, gx1p, gpos),
                    ops.chain_add(0, 0, c, res=2, out=gpos_new)]
            if lid == 0:
                ops.row_chain_fwd(prog, q)
                carry = None
            else:
                carry, carry_keep = prog, (dqk, dv, gy1, gx1p, gpos, gpos_new)
            wgrad(base, 'in_w', s.xp, dqk.view(q, 2 * c), rows=(0, 2 * c))
            wgrad(base, 'in_w', s.x, dv.view(q, c), rows=(2 * c, 3 * c))
            g_next, gpos = gx, gpos_new
            if ctx.needs_input_grad[4] and (lid == 0 or meta.reg_branches is None):
                # The reference points a layer reads carry a gradient when they are the decoder's INPUT: layer 0's always, every
                # layer's without reg branches (no refinement, so no detach: detr3d_transformer.py:199-214).  Two parts: the
                # plan's, and position_encoder's through its first Linear and inverse_sigmoid (torch ops on 900 x 3 values).
                g_isig = ops.linear_fwd(gmid0, p['pe0_w'].detach().contiguous(), weight_kn=True)           # (Q, 3) = gmid0 W
                part_ref = ops.inverse_sigmoid_bwd(s.ref.view(q, 3), g_isig, add=gr.contiguous().view(q, 3)).view(1, q, 3)
                g_ref0 = part_ref if g_ref0 is None else g_ref0 + part_ref
        for i in range(0, len(local_w), 16):
            ops.linear_bwd_weight_group(local_w[i:i + 16], accumulate=False)
        for i in range(0, len(local_ln), 32):
            ops.layernorm_bwd_reduce_group(local_ln[i:i + 32], accumulate=False)
        grads = [g if n else None for g, n in zip(grads, need)]
        return (None, None, g_next.view(q, 1, c) if ctx.needs_input_grad[2] else None,
                gpos.view(q, 1, c) if ctx.needs_input_grad[3] else None, g_ref0, *grads)


CALLS = [0]          # how often run() was taken (bench.py reports which path trained)


def run(decoder, query, query_pos, value, reference_points, reg_branches, img_metas, attn_masks, raw_entry):
    """The decoder's forward behind DecoderTrainFunction.  Returns what Detr3DTransformerDecoder.forward returns."""
    CALLS[0] += 1
    layers = list(decoder.layers)
    ca0 = layers[0].attentions[1]
    raw, token = raw_entry[id(ca0)][3]
    if isinstance(attn_masks, (list, tuple)):
        attn_masks = attn_masks[0]
    img_h, img_w = Fn.img_hw(img_metas)
    meta = _Meta(decoder=decoder, raw=raw, reg_branches=reg_branches, lidar2img=Fn.lidar2img_device(img_metas, query),
                 img_h=img_h, img_w=img_w, attn_mask=attn_masks)
    params = [t for layer in layers for t in _layer_params(layer)]
    outs, refs = DecoderTrainFunction.apply(meta, token, query, query_pos, reference_points, *params)
    if reg_branches is None:
        # no refinement, so no detach (detr3d_transformer.py:199-214): every layer's reference points ARE the caller's tensor and
        # the box loss of every level reaches it through inverse_sigmoid(inter_references) - the Function's own copies are marked
        # non-differentiable (right for refined, detached points only)
        refs = reference_points.unsqueeze(0).expand(refs.shape[0], *reference_points.shape)
    if decoder.return_intermediate:
        return outs, refs
    return outs[0], refs[0]

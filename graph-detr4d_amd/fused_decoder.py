"""The decoder loop with the row-local work of every layer folded into two gd4d_row_chain_fwd launches.

Reference: Detr3DTransformerDecoder.forward (projects/mmdet3d_plugin/models/utils/detr3d_transformer.py:166-225) over
post-norm DetrTransformerDecoderLayers (config ...ceph.py:71-89: self_attn, norm, cross_attn, norm, ffn, norm) with
Deform3DCrossAttn (utils/deform3d_cross_attn.py:196-339).  Per layer the generic module path launches 14-18 kernels;
here a layer is

    attention core  ->  chain A  ->  fused sample-aggregate  ->  chain B          (+ position_encoder chain, query order:
                                                                                     off the critical path, aux stream)
    chain A: out_proj + residual, norms[0], the three Linears of Deform3DCrossAttn on query + query_pos
    chain B: output_proj + both residuals, norms[1], FFN, norms[2], the NEXT layer's in_proj, the reg branch and the
             reference-point refinement (:199-214)

Same arithmetic as the module path (fp32 MFMA products, two-pass LayerNorm), other summation order inside the LayerNorm
reductions only.  Used by Detr3DTransformerDecoder.forward when every layer has this shape, batch 1, fp32, no autograd
(GD4D_FUSED_DECODER=0 disables it); anything else takes the generic path.
"""
import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .deform3d_cross_attn import Deform3DCrossAttn
from .transformer_layers import FFN, MultiheadAttention

ORDER = ('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')
# The stacked GEMM that produces the sampling offsets (with the camera and attention logits) on six bf16 products (~2^-24) like the
# GEMMs that produce reference points: an offset in metres goes through a camera matrix before the visibility mask is decided
# (measured: docs/measurements_r05.md section 3)
OFFSETS_EXACT = True


def _plain_reg_branch(branch, c):
    """[Linear, ReLU]* Linear with c-wide hidden layers, as the heads build them (dense_heads/detr3d_head.py:58-75)."""
    if not isinstance(branch, nn.Sequential) or len(branch) % 2 == 0:
        return None
    lins = []
    for i, m in enumerate(branch):
        if i % 2 == 0:
            if not isinstance(m, nn.Linear) or m.in_features != c or (i + 1 < len(branch) and m.out_features != c):
                return None
            lins.append(m)
        elif not isinstance(m, nn.ReLU):
            return None
    return lins if lins[-1].out_features >= 5 and len(lins) <= 4 else None


def applicable(decoder, query, value, reference_points, reg_branches, attn_masks, query_pos=None):
    """The forward-only fused loop must not be taken when autograd has to see the call: a parameter, the queries, the
    reference points OR THE FEATURE MAPS (a frozen decoder on a backbone that is being fine-tuned, input-gradient
    analysis) requiring grad sends the call down the generic path, whose modules build the autograd graph."""
    if os.environ.get('GD4D_FUSED_DECODER', '1') == '0' or Fn.wants_grad(
            decoder, query, query_pos, reference_points, *(value if isinstance(value, (list, tuple)) else ())):
        return False
    if not query.is_cuda or query.dtype != torch.float32 or query.dim() != 3 or query.shape[1] != 1:
        return False
    if reference_points is None or reference_points.shape[-1] != 3 or not isinstance(value, (list, tuple)):
        return False
    c = query.shape[-1]
    if c % 64 or c > 512:
        return False
    for layer in decoder.layers:
        if tuple(layer.operation_order) != ORDER or layer.training or len(layer.attentions) != 2 or len(layer.ffns) != 1:
            return False
        sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
        if not isinstance(sa, MultiheadAttention) or sa.batch_first or sa.embed_dims != c:
            return False
        if type(ca) is not Deform3DCrossAttn or ca.embed_dims != c or len(value) != ca.num_levels:
            return False
        if ca.num_points not in ((1, 2, 4, 8) if ca.num_heads == 8 else (4,)):     # (other counts: the module path pads them)
            return False
        if not isinstance(ffn, FFN) or len(ffn.layers) != 3 or not ffn.add_identity or ffn.feedforward_channels % 64 \
                or ffn.feedforward_channels > 512:
            return False
    if attn_masks is not None and not (isinstance(attn_masks, (list, tuple)) or torch.is_tensor(attn_masks)):
        return False
    if reg_branches is not None and any(_plain_reg_branch(reg_branches[i], c) is None for i in range(len(decoder.layers))):
        return False
    return True


def takes_single_stream_loop(decoder, query, value, reference_points, reg_branches, attn_masks, query_pos):
    """Whether a Detr3DTransformerDecoder.forward with these arguments ends in run_single - the loop that gathers the coarse levels
    from projected rows (LateValues(coarse_for=...) is only worth preparing for it: another consumer copies the pyramid again)."""
    if not applicable(decoder, query, value, reference_points, reg_branches, attn_masks, query_pos=query_pos):
        return False
    c = query.shape[-1]
    return all((c // l.attentions[1].num_heads) % 32 == 0 and not l.attentions[1].depth_encode for l in decoder.layers)


def _in_proj_ops(sa, x_pos_buf, x_buf, qkv, kv=None):
    """The packed in-projection: q, k from (x + pos), v from x (mmcv MultiheadAttention semantics) - one GEMM operation whose
    last 256 columns read the other buffer (two operations when the width does not allow it; bit-identical).  kv (ops.KVPlanes):
    K and V also leave as the attention core's split-bf16 operands (_kv_planes decides)."""
    c = sa.embed_dims
    w, b = sa.attn.in_proj_weight, sa.attn.in_proj_bias
    if c % 128 == 0:
        return [ops.chain_gemm_two_sources(x_pos_buf, x_buf, 2 * c, w, b, qkv, kv=kv)]
    return [ops.chain_gemm(x_pos_buf, w[:2 * c], b[:2 * c], out=qkv[..., :2 * c]),
            ops.chain_gemm(x_buf, w[2 * c:], b[2 * c:], out=qkv[..., 2 * c:])]


def _kv_planes(layers, q, c, attn_mask, dev):
    """The K / V planes of run_single's attention launches, or None: one pair for all layers (a layer's attention core has read
    them before the in-projection of the next writes them - one stream).  Needs the one-operation in-projection, 256 channels
    (H-DETR's 2-D mask rides along)."""
    if c != 256 or (attn_mask is not None and attn_mask.dim() != 2) or os.environ.get('GD4D_MHA_FP32') == '1':
        return None
    if layers[0].attentions[0].num_heads != 8:
        return None
    return ops.KVPlanes(q, c, dev, heads=8)


def initial_reference(linear, query_pos):
    """sigmoid(Linear(query_pos)) (detr3d_transformer.py:133-134) as one chain launch, fp32-class products (GD4D_CHAIN_EXACT, as
    the reg branches below): a reference point is multiplied by the 102-m range and a focal length before it selects pixels,
    so the 2^-16 of a split-bf16 x3 product would move samples by hundredths of a pixel.  query_pos (Q, C), rows may be
    strided (a column slice of query_embed).  Returns (1, Q, 3)."""
    q = query_pos.shape[0]
    out = torch.empty(1, q, linear.out_features, device=query_pos.device, dtype=torch.float32)
    if query_pos.device.index != torch.cuda.current_device():
        with torch.cuda.device(query_pos.device):
            return initial_reference(linear, query_pos)
    ops.row_chain_fwd([ops.chain_load(0, query_pos),
                       ops.chain_gemm(0, linear.weight, linear.bias, out=out.view(q, -1), sigmoid=True, exact=True)], q)
    return out


def fast_input(module, query_embed, mlvl_feats):
    """The conditions under which Detr3DTransformer hands strided views to the decoder and uses initial_reference."""
    return (os.environ.get('GD4D_FUSED_DECODER', '1') != '0' and query_embed.is_cuda and query_embed.dtype == torch.float32
            and mlvl_feats[0].size(0) == 1 and query_embed.dim() == 2 and query_embed.stride(1) == 1
            and query_embed.shape[1] % 8 == 0 and (query_embed.shape[1] // 2) % 128 == 0
            and not Fn.wants_grad(module, query_embed, *mlvl_feats))


def _position_features(ca, reference_points, q):
    """position_encoder(inverse_sigmoid(ref [, depth])) (deform3d_cross_attn.py:104-111, 331-334) as one chain."""
    ref3d = reference_points
    if ca.depth_encode:
        depth = (ref3d[..., 0:1] ** 2 + ref3d[..., 1:2] ** 2) ** 0.5
        ref3d = torch.cat([ref3d, depth], dim=-1).contiguous()
    seq = ca.position_encoder
    out = torch.empty(1, q, ca.embed_dims, device=ref3d.device, dtype=torch.float32)
    prog = [ops.chain_load(0, ref3d.view(q, -1), inv_sigmoid=True),
            ops.chain_small_linear(0, seq[0].weight, seq[0].bias, 1),
            ops.chain_layernorm(1, seq[1], dst=0, relu=True),
            ops.chain_gemm(0, seq[3].weight, seq[3].bias, dst=1),
            ops.chain_layernorm(1, seq[4], relu=True, out=out.view(q, -1))]
    ops.row_chain_fwd(prog, q)
    return out, ref3d


def _position_ops(ca, out, ref=None, ref_buf=None):
    """position_encoder(inverse_sigmoid(ref)) (deform3d_cross_attn.py:104-111, 331-334) as chain operations writing `out`
    (Q, C): from the global reference points `ref` (Q, 3), or from the points a REFINE operation parked in LDS buffer
    `ref_buf` (columns 0..2).  depth_encode = False only."""
    seq = ca.position_encoder
    if ref_buf is None:
        head, src, inv = [ops.chain_load(0, ref, inv_sigmoid=True)], 0, False
    else:
        head, src, inv = [], ref_buf, True
    mid = 1 if src != 1 else 2
    nxt = [b for b in (0, 2, 3) if b not in (src, mid)][0]
    return head + [ops.chain_small_linear(src, seq[0].weight, seq[0].bias, mid, inv_sigmoid=inv),
                   ops.chain_layernorm(mid, seq[1], dst=nxt, relu=True),
                   ops.chain_gemm(nxt, seq[3].weight, seq[3].bias, dst=mid),
                   ops.chain_layernorm(mid, seq[4], relu=True, out=out)]


def run_single(decoder, query, query_pos, value, reference_points, reg_branches, img_metas, attn_masks, order,
               return_intermediate, late):
    """The decoder loop on ONE stream (besides the channels-last copy's).  A cross-stream dependency costs ~10 us in a
    replayed hipGraph against ~2 us for a kernel boundary on one stream (tools/trace_step.py), and the auxiliary-stream
    schedule of run() pays four of them per layer.  Here what used to run on the auxiliary stream - the previous layer's
    reg branch + reference-point refinement and this layer's position_encoder, all row-local and off the
    self-attention's path - is the SECOND program of chain A's launch (gd4d_row_chain2_fwd: twice the workgroups, each
    half on its own compute units):

        attention core -> [chain A | reg(l-1), refine, position_encoder(l)] -> aggregate -> chain B'

    The locality order of the queries is the one of the initial reference points for every layer (the refinements move a
    point little; a stale order costs ~2 us per aggregate launch, a fresh one a launch + boundary on the critical path).

    Round 4: position_encoder(l) leaves that second program.  It outlasted chain A by ~12 us per layer although only the
    refined points are needed before the gather; its rows are first read by chain B's SECOND operation.  It is now the second
    program of chain B's launch, handed over row block by row block with a SIGNAL / WAIT pair through the XCD's L2 (it is done
    long before HEADGEMM, chain B's first operation, is):

        attention core -> [chain A | reg(l-1), refine] -> plan -> gather -> [chain B' | position_encoder(l)]

    A WAIT that times out poisons its rows (NaN) and counts in the device's error word: ops.poll_handoff() below (eager calls),
    ops.check_handoff() after a replayed graph (the owner of the graph calls it where it synchronises anyway).  The hand-off needs
    workgroups j and j + 8 k on one XCD: ops.handoff_placement_ok() tests that once per device; where it does not hold, and with
    GD4D_POS_ENCODER=dual, round 3's schedule runs instead - [chain A | reg(l-1), refine, position_encoder(l)], chain B' alone
    (bit-identical results, tested)."""
    q, _, c = query.shape
    dev = query.device
    layers = list(decoder.layers)
    nl = len(layers)
    if attn_masks is None:
        attn_masks = [None, None]
    elif torch.is_tensor(attn_masks):
        attn_masks = [attn_masks, attn_masks]
    lidar2img = Fn.lidar2img_device(img_metas, query)
    img_h, img_w = Fn.img_hw(img_metas)
    x = query[:, 0, :]
    pos = query_pos[:, 0, :]
    keep = []
    qkv = torch.empty(q, 1, 3 * c, device=dev, dtype=torch.float32)
    kv = _kv_planes(layers, q, c, attn_masks[0], dev)
    # The two coarse pyramid levels are gathered from PROJECTED rows (LateValues.coarse_setup): layer l's value_proj over them is
    # the guest job of a chain launch that runs before gather l on compute units the chain leaves idle - layer 0's rides here,
    # layer l + 1's in chain B' of layer l (one buffer: gather l has read it by then).
    coarse = late.mode == 'sliced' and late.coarse_setup([l.attentions[1] for l in layers])
    first_guest = None
    if coarse and not late.take_first(layers[0].attentions[1]):      # (not enqueued beside the copy when `late` was made: rides here)
        first_guest = late.coarse_guest(layers[0].attentions[1])
    ops.row_chain_fwd([ops.chain_load(0, x, pos), ops.chain_load(1, x)] + _in_proj_ops(layers[0].attentions[0], 0, 1, qkv.view(q, -1), kv), q,
                      guest=first_guest)
    n_out = nl if return_intermediate else 1
    out_all = torch.empty(n_out, q, 1, c, device=dev, dtype=torch.float32)
    ref_all = torch.empty(n_out, 1, q, 3, device=dev, dtype=torch.float32)
    ref = reference_points.contiguous()
    pending = None                                       # (reg linears, x of the previous layer, its ref, where the new ref goes)
    pos_late = ops.handoff_enabled(dev, 'GD4D_POS_ENCODER')       # [chain A | reg branch], [chain B' | position_encoder]: the default
    if pos_late:
        blocks = (q + 15) // 16
        flags = ops.handoff_flags(dev, nl, (blocks + 7) // 8 * 8 + 8, Fn.slot_key(dev))       # row-block flags per layer (persistent: no fill in the graph)
        err = ops.handoff_error_word(dev)
        keep.append(flags)
    for lid, layer in enumerate(layers):
        sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
        hh, npt, nlv, ncam = ca.num_heads, ca.num_points, ca.num_levels, ca.num_cams
        last = lid + 1 == nl
        slot = lid if return_intermediate else 0
        qh, kh, vh = qkv.split(c, dim=-1)
        if kv is not None:
            o = ops.mha_core_presplit_fwd(qh, kv, sa.num_heads, attn_masks[0])
        else:
            o = ops.mha_core_fwd(qh, kh, vh, sa.num_heads, attn_masks[0])

        x1 = torch.empty(q, c, device=dev, dtype=torch.float32)
        cam = torch.empty(1, q, ncam, device=dev, dtype=torch.float32)
        off = torch.empty(1, q, hh * npt * 3, device=dev, dtype=torch.float32)
        att = torch.empty(1, q, hh * nlv * npt, device=dev, dtype=torch.float32)
        # (the residual x rides in out_proj's epilogue and x1 + pos leaves the LayerNorm as a second output: two operations
        #  - two barriers and two exposed round trips - fewer than LOAD, GEMM, LAYERNORM, ADD; same values bit for bit)
        prog_a = [ops.chain_load(0, o.view(q, c)),
                  ops.chain_gemm(0, sa.attn.out_proj.weight, sa.attn.out_proj.bias, dst=1, add=x),
                  ops.chain_layernorm(1, layer.norms[0], dst=2, out=x1, dst2=0, add=pos)]
        # the three Linears of query + query_pos as one GEMM over the stacked weights (248 columns: one pass)
        prog_a.append(ops.chain_gemm_three_outputs(
            0, [ca.cam_attention_weights, ca.deform_sampling_offsets, ca.attention_weights],
            [cam.view(q, -1), off.view(q, -1), att.view(q, -1)], exact=OFFSETS_EXACT))
        pos_feat = torch.empty(1, q, c, device=dev, dtype=torch.float32)
        if pending is not None:
            lins, x_prev, ref_prev, new_ref = pending
            prog_b, src, tmp = [ops.chain_load(3, x_prev)], 3, (1, 2)
            for i, lin in enumerate(lins):
                prog_b.append(ops.chain_gemm(src, lin.weight, lin.bias, dst=tmp[i % 2], relu=i + 1 < len(lins), exact=True))
                src = tmp[i % 2]
            park = 0 if src != 0 else 3
            prog_b.append(ops.chain_refine(src, ref_prev, new_ref, dst=-1 if pos_late else park))
            if not pos_late:
                prog_b += _position_ops(ca, pos_feat.view(q, c), ref_buf=park)
            keep += [x_prev, ref_prev]
            ref = new_ref
        else:
            prog_b = None if pos_late else _position_ops(ca, pos_feat.view(q, c), ref=ref.view(q, 3))
        if prog_b is None:                                      # layer 0: the initial reference points need no refinement
            ops.row_chain_fwd(prog_a, q)
        else:
            ops.row_chain2_fwd(prog_a, prog_b, q)

        if late.mode != 'sliced':
            # value_proj of the aggregates in the gather's epilogue: chain B' starts from one 1-KB row per query
            agg = late.sample_aggregate(ca, ref, off.view(1, q, hh, npt, 3), att.view(1, q, hh, nlv, npt), cam, lidar2img,
                                        img_h, img_w, order=order)
            first = ops.chain_load(0, agg.view(q, c))
            agg_raw, wsum = agg, agg
        elif coarse:
            agg_raw, wsum, pagg = late.aggregate(ca, ref, off.view(1, q, hh, npt, 3), att.view(1, q, hh, nlv, npt), cam, lidar2img,
                                                 img_h, img_w, order=order, coarse=True)
            first = ops.chain_headgemm(agg_raw, wsum, ca.value_proj.weight, ca.value_proj.bias, dst=0, addend=pagg.view(q, c))
            keep.append(pagg)
        else:
            agg_raw, wsum = late.aggregate(ca, ref, off.view(1, q, hh, npt, 3), att.view(1, q, hh, nlv, npt), cam, lidar2img,
                                           img_h, img_w, order=order)
            first = ops.chain_headgemm(agg_raw, wsum, ca.value_proj.weight, ca.value_proj.bias, dst=0)
        guest = late.coarse_guest(layers[lid + 1].attentions[1]) if coarse and not last else None
        x3 = out_all[slot]
        prog = [first] + ([ops.chain_wait(flags[lid], err)] if pos_late else []) + [
                ops.chain_load(3, x1, pos_feat.view(q, c)),                       # the two residuals of :336 (as GEMM addends: slower)
                ops.chain_gemm(0, ca.output_proj.weight, ca.output_proj.bias, dst=1, res=3),
                ops.chain_layernorm(1, layer.norms[1], dst=2),
                ops.chain_gemm(2, ffn.layers[0][0].weight, ffn.layers[0][0].bias, dst=0, relu=True),
                ops.chain_gemm(0, ffn.layers[1].weight, ffn.layers[1].bias, dst=1, res=2)]
        if not last:
            qkv = torch.empty(q, 1, 3 * c, device=dev, dtype=torch.float32)
            prog.append(ops.chain_layernorm(1, layer.norms[2], dst=3, out=x3.view(q, c), dst2=0, add=pos))
            prog += _in_proj_ops(layers[lid + 1].attentions[0], 0, 3, qkv.view(q, -1), kv)
        else:
            prog.append(ops.chain_layernorm(1, layer.norms[2], dst=3, out=x3.view(q, c)))
        pending = None
        if reg_branches is not None:
            lins = _plain_reg_branch(reg_branches[lid], c)
            new_ref = ref_all[slot]
            if last:                                     # nothing follows: the last refinement closes chain B'
                src, tmp = 3, (1, 2)
                for i, lin in enumerate(lins):
                    prog.append(ops.chain_gemm(src, lin.weight, lin.bias, dst=tmp[i % 2], relu=i + 1 < len(lins), exact=True))
                    src = tmp[i % 2]
                prog.append(ops.chain_refine(src, ref, new_ref))
            else:
                pending = (lins, x3.view(q, c), ref, new_ref)
        elif return_intermediate or last:
            ref_all[slot].copy_(ref)
        if pos_late:
            # position_encoder(l) on the refined points (in global memory since the dual launch), beside chain B'.  The
            # SIGNALling program goes FIRST: its workgroups are dispatched before the waiting ones (gd4d.h).
            ops.row_chain2_fwd(_position_ops(ca, pos_feat.view(q, c), ref=ref.view(q, 3)) + [ops.chain_signal(flags[lid])], prog, q,
                               guest=guest)
        else:
            ops.row_chain_fwd(prog, q, guest=guest)
        keep += [o, x1, cam, off, att, agg_raw, wsum, pos_feat, x]
        x = x3.view(q, c)
    if pos_late:
        ops.poll_handoff(dev)                                # non-blocking (eager calls); a graph's owner: ops.check_handoff()
    del keep
    return out_all, ref_all


def run(decoder, query, query_pos, value, reference_points, reg_branches, img_metas, attn_masks, pipeline, value_cache,
        order, order_pc_range, return_intermediate, late=None):
    """query / query_pos (Q, 1, C), rows may be strided; reference_points (1, Q, 3).  Returns the stacked per-layer
    outputs (NL, Q, 1, C) and reference points (NL, 1, Q, 3) (the last layer's only without return_intermediate)."""
    if query.device.index != torch.cuda.current_device():
        # the chain launches take no tensor argument their wrapper could read the device from: make the model's GPU current
        with torch.cuda.device(query.device):
            return run(decoder, query, query_pos, value, reference_points, reg_branches, img_metas, attn_masks, pipeline,
                       value_cache, order, order_pc_range, return_intermediate, late=late)
    q, _, c = query.shape
    dev = query.device
    layers = list(decoder.layers)
    nl = len(layers)
    if late is not None and all((c // l.attentions[1].num_heads) % 32 == 0 and not l.attentions[1].depth_encode for l in layers):
        if order is None or order.numel() != q:
            order = Fn.query_order(reference_points.contiguous(), layers[0].attentions[1].pc_range)
        return run_single(decoder, query, query_pos, value, reference_points, reg_branches, img_metas, attn_masks, order,
                          return_intermediate, late)
    main = torch.cuda.current_stream(dev)
    aux = Fn.aux_stream(dev)
    if attn_masks is None:
        attn_masks = [None, None]
    elif torch.is_tensor(attn_masks):
        attn_masks = [attn_masks, attn_masks]
    lidar2img = Fn.lidar2img_device(img_metas, query)
    img_h, img_w = Fn.img_hw(img_metas)
    x = query[:, 0, :]                                   # (Q, C) views; chain loads take a row stride
    pos = query_pos[:, 0, :]
    keep = []                                            # tensors the enqueued programs point to

    # layer 0's in-projection
    qkv = torch.empty(q, 1, 3 * c, device=dev, dtype=torch.float32)
    ops.row_chain_fwd([ops.chain_load(0, x, pos), ops.chain_load(1, x)] + _in_proj_ops(layers[0].attentions[0], 0, 1, qkv.view(q, -1)), q)

    n_out = nl if return_intermediate else 1
    out_all = torch.empty(n_out, q, 1, c, device=dev, dtype=torch.float32)       # what torch.stack would build
    ref_all = torch.empty(n_out, 1, q, 3, device=dev, dtype=torch.float32)
    ref = reference_points.contiguous()
    ref_event = None                                     # `ref` / `order` were produced on the aux stream
    for lid, layer in enumerate(layers):
        sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
        hh, npt, nlv, ncam = ca.num_heads, ca.num_points, ca.num_levels, ca.num_cams
        # position_encoder depends on the reference points only: aux stream, joined before chain B
        ev_pos = None
        if aux is not None:
            fork = torch.cuda.Event()
            fork.record(main)
            with torch.cuda.stream(aux):
                aux.wait_event(fork)
                pos_feat, ref_keep = _position_features(ca, ref, q)
                ev_pos = torch.cuda.Event()
                ev_pos.record(aux)
        else:
            if ref_event is not None:
                main.wait_event(ref_event)
            pos_feat, ref_keep = _position_features(ca, ref, q)
        keep += [pos_feat, ref_keep]

        # attention core
        qh, kh, vh = qkv.split(c, dim=-1)
        o = ops.mha_core_fwd(qh, kh, vh, sa.num_heads, attn_masks[0])

        # chain A
        x1 = torch.empty(q, c, device=dev, dtype=torch.float32)
        cam = torch.empty(1, q, ncam, device=dev, dtype=torch.float32)
        off = torch.empty(1, q, hh * npt * 3, device=dev, dtype=torch.float32)
        att = torch.empty(1, q, hh * nlv * npt, device=dev, dtype=torch.float32)
        prog = [ops.chain_load(0, o.view(q, c)),
                ops.chain_load(3, x),                                              # the residual of the self-attention
                ops.chain_gemm(0, sa.attn.out_proj.weight, sa.attn.out_proj.bias, dst=1, res=3),
                ops.chain_layernorm(1, layer.norms[0], dst=2, out=x1),
                ops.chain_add(0, 2, c, add=pos),
                ops.chain_gemm(0, ca.cam_attention_weights.weight, ca.cam_attention_weights.bias, out=cam.view(q, -1)),
                ops.chain_gemm(0, ca.deform_sampling_offsets.weight, ca.deform_sampling_offsets.bias, out=off.view(q, -1)),
                ops.chain_gemm(0, ca.attention_weights.weight, ca.attention_weights.bias, out=att.view(q, -1))]
        ops.row_chain_fwd(prog, q)

        if ref_event is not None:
            main.wait_event(ref_event)
        if order is None or order.numel() != q:
            order = Fn.query_order(ref, ca.pc_range)
        agg_raw = None
        if late is not None and late.mode == 'sliced' and (c // hh) % 32 == 0:
            # aggregate-then-project: gather the raw features per head; value_proj of the aggregates is chain B's first op
            agg_raw, wsum = late.aggregate(ca, ref, off.view(1, q, hh, npt, 3), att.view(1, q, hh, nlv, npt), cam, lidar2img,
                                           img_h, img_w, order=order)
            agg = agg_raw
        elif late is not None:
            agg = late.sample_aggregate(ca, ref, off.view(1, q, hh, npt, 3), att.view(1, q, hh, nlv, npt), cam, lidar2img,
                                        img_h, img_w, order=order)
        else:
            # projected values of this layer
            taken = pipeline.take(ca, value) if pipeline is not None else None
            cached = (value_cache or {}).get(id(ca))
            if taken is not None:
                val, shapes = taken
            elif cached is not None and cached[2] is value:
                val, shapes = cached[0], cached[1]
            else:
                val, shapes = Fn.value_projection(value, ca.value_proj.weight, ca.value_proj.bias, hh, ca.value_dtype)
            agg = Fn.sample_aggregate(val, shapes, ref, off.view(1, q, hh, npt, 3), att.view(1, q, hh, nlv, npt), cam,
                                      lidar2img, ca.pc_range, img_h, img_w, order=order)
            if taken is not None:
                del val, taken
                pipeline.gather_enqueued(ca)

        # chain B
        if ev_pos is not None:
            main.wait_event(ev_pos)
        last = lid + 1 == nl
        slot = lid if return_intermediate else 0
        x3 = out_all[slot]                                                         # (Q, 1, C)
        if agg_raw is not None:
            bias = ca.value_proj.bias
            first = ops.chain_headgemm(agg_raw, wsum, ca.value_proj.weight, bias, dst=0)
            keep.append(wsum)
        else:
            first = ops.chain_load(0, agg.view(q, c))
        prog = [first,
                ops.chain_load(3, x1, pos_feat.view(q, c)),                       # the two residuals of :336
                ops.chain_gemm(0, ca.output_proj.weight, ca.output_proj.bias, dst=1, res=3),
                ops.chain_layernorm(1, layer.norms[1], dst=2),                    # x2
                ops.chain_gemm(2, ffn.layers[0][0].weight, ffn.layers[0][0].bias, dst=0, relu=True),
                ops.chain_gemm(0, ffn.layers[1].weight, ffn.layers[1].bias, dst=1, res=2),
                ops.chain_layernorm(1, layer.norms[2], dst=3, out=x3.view(q, c))]  # x3 = the layer's output
        if not last:
            qkv = torch.empty(q, 1, 3 * c, device=dev, dtype=torch.float32)
            prog += [ops.chain_add(0, 3, c, add=pos)] + _in_proj_ops(layers[lid + 1].attentions[0], 0, 3, qkv.view(q, -1))
        new_ref = None
        reg_prog = []
        if reg_branches is not None:
            # reg branch + refinement (:199-214) feed the NEXT layer's gather and position_encoder only - not its
            # self-attention: with the auxiliary stream they are their own short chain next to that self-attention
            lins = _plain_reg_branch(reg_branches[lid], c)
            src, tmp = 3, (1, 2)
            for i, lin in enumerate(lins):
                reg_prog.append(ops.chain_gemm(src, lin.weight, lin.bias, dst=tmp[i % 2], relu=i + 1 < len(lins), exact=True))
                src = tmp[i % 2]
            new_ref = ref_all[slot]                                                 # (1, Q, 3)
            reg_prog.append(ops.chain_refine(src, ref, new_ref))
        split_reg = bool(reg_prog) and aux is not None
        if not split_reg:
            prog += reg_prog
        ops.row_chain_fwd(prog, q)
        keep += [o, x1, cam, off, att, agg, x]

        x = x3.view(q, c)
        if new_ref is not None:
            old_ref, ref = ref, new_ref
            ref_event = None
            if aux is not None and (split_reg or (not last and order is not None)):
                done = torch.cuda.Event()
                done.record(main)
                with torch.cuda.stream(aux):
                    aux.wait_event(done)
                    if split_reg:
                        ops.row_chain_fwd([ops.chain_load(3, x)] + reg_prog, q)
                        keep.append(old_ref)
                    if not last and order is not None:
                        # the next layer's locality order: one tiny launch next to that layer's self-attention
                        order = Fn.query_order(ref, order_pc_range)
                    ref_event = torch.cuda.Event()
                    ref_event.record(aux)
            elif not last and order is not None:
                order = Fn.query_order(ref, order_pc_range)
        if new_ref is None and (return_intermediate or last):
            ref_all[slot].copy_(ref)
    if aux is not None:
        main.wait_stream(aux)
    del keep
    return out_all, ref_all

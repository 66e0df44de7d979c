"""ORACLE - test infrastructure, not product code.

CPU (torch fp32) restatement of the reference's decoder hot path, written from the maths in
SURVEY.md Appendix A, each function citing the reference file:line it follows (paths relative
to the reference root, `projects/mmdet3d_plugin/models/utils/`).  Only tests/, the smoke check
in __graft_entry__.py and bench.py's `cpu_baseline` leg may import this file; the product path
(graph-detr4d_amd/) never does and fails loudly when its HIP library is missing.

Parity pin: the reference ships no tests and no golden vectors, so this oracle
is pinned against outputs of the reference itself run in the build container
(tools/gen_golden.py imports the reference's files unmodified under an mmcv stub and writes
tests/golden/*.npz; tests/test_oracle_golden.py checks every function here against them).
Third-party arithmetic that is NOT in the reference tree (mmcv 1.x MultiScaleDeformableAttn,
mmcv MultiheadAttention / FFN / BaseTransformerLayer, mmdet DetrTransformerDecoderLayer; no
version pinned anywhere in the reference) is restated from its published semantics; at that
boundary parity is pinned only by ATen's F.grid_sample / nn.MultiheadAttention in this
container.

All parameters are passed as a flat dict of tensors keyed by the reference's state-dict
names (e.g. 'value_proj.weight'), so a reference checkpoint slice can be fed directly.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def inverse_sigmoid(x, eps=1e-5):
    """deform3d_cross_attn.py:16-31 (detr3d_transformer.py:28-43 is the same on [0,1])."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps, max=1) / (1 - x).clamp(min=eps, max=1))


def _linear(x, p, name):
    return F.linear(x, p[name + '.weight'], p.get(name + '.bias'))


def lidar2img_tensor(img_metas, like):
    """deform3d_cross_attn.py:215-219: stack img_meta['lidar2img'] -> (B, N, 4, 4) fp32."""
    mats = np.asarray([m['lidar2img'] for m in img_metas])
    return like.new_tensor(mats)


def denormalise(ref, pc_range):
    """deform3d_cross_attn.py:222-224: ref*(hi-lo)+lo, (hi-lo) evaluated in Python floats."""
    out = ref.clone()
    for k in range(3):
        out[..., k:k + 1] = ref[..., k:k + 1] * (pc_range[k + 3] - pc_range[k]) + pc_range[k]
    return out


def project(points, lidar2img, img_h, img_w, eps=1e-5):
    """Projection block, deform3d_cross_attn.py:232-252 (same at detr3d_transformer.py:409-420).

    points (B, M, 3) metres; lidar2img (B, N, 4, 4).  Returns uv (B, N, M, 2) normalised by
    the image W/H and the bool mask (B, N, M) = z>eps & 0<u<1 & 0<v<1.  Arithmetic order is
    the reference's: batched matmul, IEEE division by max(z, eps), then by W and H.
    """
    b, m, _ = points.shape
    n = lidar2img.shape[1]
    hom = torch.cat((points, torch.ones_like(points[..., :1])), -1)
    hom = hom.view(b, 1, m, 4).repeat(1, n, 1, 1).unsqueeze(-1)
    mats = lidar2img.view(b, n, 1, 4, 4).repeat(1, 1, m, 1, 1)
    cam = torch.matmul(mats, hom).squeeze(-1)
    z = cam[..., 2:3]
    mask = z > eps
    uv = cam[..., 0:2] / torch.max(z, torch.ones_like(z) * eps)
    uv[..., 0] /= img_w
    uv[..., 1] /= img_h
    mask = (mask & (uv[..., 0:1] > 0.) & (uv[..., 0:1] < 1.0)
            & (uv[..., 1:2] > 0.) & (uv[..., 1:2] < 1.0))
    return uv, mask.squeeze(-1)


def flatten_pyramid(value):
    """deform3d_cross_attn.py:264-272: list of (B,N,C,H,W) -> (B*N, sum(HW), C) channels-last."""
    flat, shapes = [], []
    for v in value:
        b, n, c, h, w = v.shape
        shapes.append((h, w))
        flat.append(v.reshape(b * n, c, h * w).transpose(1, 2))
    return torch.cat(flat, 1), shapes


def bilinear_zero_pad(value_l, x, y):
    """Third-party MSDA sampling rule (mmcv ms_deform_attn kernel; == grid_sample bilinear,
    zeros padding, align_corners=False).  value_l (H, W, D); x, y pixel coords (...,) with the
    -0.5 shift already applied.  Returns (..., D).  Explicit-gather form, independent of ATen.
    """
    h, w, _ = value_l.shape
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    dx = x - x0
    dy = y - y0
    out = 0
    for (yy, xx, wt) in ((y0, x0, (1 - dy) * (1 - dx)), (y0, x0 + 1, (1 - dy) * dx),
                         (y0 + 1, x0, dy * (1 - dx)), (y0 + 1, x0 + 1, dy * dx)):
        ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
        xi = xx.clamp(0, w - 1).long()
        yi = yy.clamp(0, h - 1).long()
        out = out + value_l[yi, xi] * (wt * ok).unsqueeze(-1)
    return out


def msda(value, shapes, loc, weights):
    """Third-party MSDA (mmcv multi_scale_deformable_attn), call site deform3d_cross_attn.py:302-309.

    value (S, sum(HW), Hh, Dh); loc (S, Q, Hh, L, P, 2) in [0,1]; weights (S, Q, Hh, L*P).
    Returns (S, Q, Hh*Dh).  Uses ATen grid_sample (the arbiter for this boundary).
    """
    s, _, hh, dh = value.shape
    _, q, _, nl, npnt, _ = loc.shape
    parts = value.split([h * w for h, w in shapes], dim=1)
    acc = []
    for lvl, (h, w) in enumerate(shapes):
        v = parts[lvl].permute(0, 2, 3, 1).reshape(s * hh, dh, h, w)
        g = (2 * loc[:, :, :, lvl] - 1).permute(0, 2, 1, 3, 4).reshape(s * hh, q, npnt, 2)
        acc.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    samp = torch.stack(acc, dim=-2).flatten(-2)                       # (S*Hh, Dh, Q, L*P)
    wt = weights.permute(0, 2, 1, 3).reshape(s * hh, 1, q, nl * npnt)
    return (samp * wt).sum(-1).view(s, hh * dh, q).transpose(1, 2).contiguous()


def scrambled_cam_weights(cam_logits, num_cams):
    """deform3d_cross_attn.py:211-212: the (B,Q,N) Linear output is VIEWED as (B,N,Q,1)."""
    b, q, n = cam_logits.shape
    return cam_logits.reshape(b, num_cams, q, 1)


def sample_aggregate(value, shapes, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range,
                     img_h, img_w, raw_cam=False):
    """The fused kernel's contract (a3+a5+a6+a7-reduction; deform3d_cross_attn.py:220-324).

    value (B*N, sum(HW), Hh, Dh) already value_proj-ed; ref (B,Q,3) in [0,1];
    offsets (B,Q,Hh,P,3) metres; attn_logits (B,Q,Hh,L*P); cam_logits (B,Q,N) un-scrambled
    Linear output; lidar2img (B,N,4,4).
    Returns out (B,Q,Hh*Dh) = sum_n sigmoid(cam[b,n,q]) * MSDA_n, uv (B,N,Q,Hh,P,2) and
    mask (B,N,Q,Hh,P) bool.
    """
    b, q, hh, p, _ = offsets.shape
    n = lidar2img.shape[1]
    nl = len(shapes)
    pts = denormalise(ref, pc_range).view(b, q, 1, 1, 3) + offsets          # :229
    uv, mask = project(pts.reshape(b, q * hh * p, 3), lidar2img, img_h, img_w)
    uv = uv.view(b, n, q, hh, p, 2)
    mask = mask.view(b, n, q, hh, p)
    loc = uv.view(b * n, q, hh, 1, p, 2).expand(-1, -1, -1, nl, -1, -1)     # :228 repeat over L
    # :277,281-284  query.repeat(N,1,1) stacks rows as [b0,b1,..,b0,b1,..] while values/masks are
    # ordered b*N+n, so value row i=b*N+n is paired with the logits of batch (i % B).  Identity
    # for B=1 (every shipped config); reproduced for B>1 because it is what the reference computes.
    rows = torch.arange(b * n) % b
    w = attn_logits.softmax(-1)[rows].view(b, n, q, hh, nl, p)
    w = (w * mask.view(b, n, q, hh, 1, p)).reshape(b * n, q, hh, nl * p)
    per_cam = msda(value, shapes, loc, w).view(b, n, q, -1)                 # :302-304
    cam = scrambled_cam_weights(cam_logits, n)
    cam = cam if raw_cam else cam.sigmoid()                                 # :320 (MP neighbour pass: raw)
    return (per_cam * cam).sum(1), uv, mask                                  # :322-324


def _bf16r(t):
    return t.bfloat16().float()


def deform3d_cross_attn(p, query, value, query_pos, reference_points, img_metas, pc_range,
                        num_heads=8, num_points=4, depth_encode=False, return_parts=False, value_dtype='fp32'):
    """Deform3DCrossAttn.forward in eval mode, deform3d_cross_attn.py:196-339 (Appendix A.1).

    value_dtype='bf16' restates the build's opt-in reduced-precision mode (not a reference mode): value_proj on
    bf16-rounded features and weights (exact products, fp32 accumulate, fp32 bias) and the projected value tensor
    rounded to bf16; everything else fp32.  value_dtype='bf16_features': only the features are rounded to bf16."""
    x = query if query_pos is None else query + query_pos                   # :203-204
    x = x.permute(1, 0, 2)                                                  # :207
    b, q, c = x.shape
    nl = len(value)
    n = value[0].shape[1]
    l2i = lidar2img_tensor(img_metas, reference_points)
    img_h, img_w = img_metas[0]['img_shape'][0][0], img_metas[0]['img_shape'][0][1]
    cam_logits = _linear(x, p, 'cam_attention_weights')                     # :211
    offsets = _linear(x, p, 'deform_sampling_offsets').view(b, q, num_heads, num_points, 3)
    attn_logits = _linear(x, p, 'attention_weights').view(b, q, num_heads, nl * num_points)
    flat, shapes = flatten_pyramid(value)
    if value_dtype == 'bf16':
        val = _bf16r(F.linear(_bf16r(flat), _bf16r(p['value_proj.weight']), p['value_proj.bias']))
    elif value_dtype == 'bf16_features':
        # the aggregate-then-project form of the same opt-in mode: the FEATURES are what is stored in bf16 (the
        # channels-last copy); value_proj (linear, so it commutes with the weighted sum of the gather) stays fp32
        val = F.linear(_bf16r(flat), p['value_proj.weight'], p['value_proj.bias'])
    else:
        val = _linear(flat, p, 'value_proj')
    val = val.view(b * n, flat.shape[1], num_heads, c // num_heads)
    agg, uv, mask = sample_aggregate(val, shapes, reference_points, offsets, attn_logits,
                                     cam_logits, l2i, pc_range, img_h, img_w)
    out = _linear(agg, p, 'output_proj').permute(1, 0, 2)                   # :326-327
    ref3d = reference_points
    if depth_encode:                                                        # :331-333
        depth = (ref3d[..., 0:1] ** 2 + ref3d[..., 1:2] ** 2) ** 0.5
        ref3d = torch.cat([ref3d, depth], -1)
    pos = position_encoder(p, inverse_sigmoid(ref3d)).permute(1, 0, 2)      # :334
    res = out + query + pos                                                 # :336 (dropout = id)
    if return_parts:
        return res, dict(agg=agg, uv=uv, mask=mask, offsets=offsets, attn_logits=attn_logits,
                         cam_logits=cam_logits, value=val, pos=pos)
    return res


def deform3d_cross_attn_mp(p, query, value, reference_points, img_metas, pc_range, num_heads=8, num_points=4,
                           return_parts=False):
    """Deform3DCrossAttnMP.forward in eval mode with multi_points=True
    (utils/deform3d_cross_attn_multi_point.py:138-453).  reference_points (B, 9Q, 3): Q centres, then 8 blocks of Q
    neighbour points.  Differences to Deform3DCrossAttn: no query_pos (:211-222); a second sampling pass over the
    neighbours without offsets, one point per level (:373-417); neighbour camera weights are NOT sigmoided
    (:424-430); a 2-way softmax blend whose logits are summed over the queries of sample 0 (:434-439)."""
    x = query.permute(1, 0, 2)                                              # :222 (no query_pos)
    b, q, c = x.shape
    nl = len(value)
    n = value[0].shape[1]
    l2i = lidar2img_tensor(img_metas, reference_points)
    img_h, img_w = img_metas[0]['img_shape'][0][0], img_metas[0]['img_shape'][0][1]
    centre, nbr = reference_points[:, :q], reference_points[:, q:]          # :228, :373
    flat, shapes = flatten_pyramid(value)
    val = _linear(flat, p, 'value_proj').view(b * n, flat.shape[1], num_heads, c // num_heads)
    # centre pass: exactly Deform3DCrossAttn's
    agg, uv, mask = sample_aggregate(val, shapes, centre, _linear(x, p, 'deform_sampling_offsets')
                                     .view(b, q, num_heads, num_points, 3),
                                     _linear(x, p, 'attention_weights').view(b, q, num_heads, nl * num_points),
                                     _linear(x, p, 'cam_attention_weights'), l2i, pc_range, img_h, img_w)
    # neighbour pass.  The (B*N, Q, 256) Linear output is VIEWED as (B*N, 8Q, Hh, L*P/4) (:375-376): neighbour row r
    # takes the 32 logits [r % 8] of query r // 8, while point r of reference_points[:, Q:] belongs to query r % Q.
    logits_n = _linear(x, p, 'attention_weights_neighbor').reshape(b, 8 * q, num_heads, nl * num_points // 4)
    x_n = x.repeat(1, 8, 1)                                                 # :383
    cam_n = _linear(x_n, p, 'cam_attention_weights')                        # (B, 8Q, N), raw view below (:424-425)
    zero_off = torch.zeros(b, 8 * q, num_heads, num_points // 4, 3)
    agg_n, uv_n, mask_n = sample_aggregate(val, shapes, nbr, zero_off, logits_n, cam_n, l2i, pc_range, img_h, img_w,
                                           raw_cam=True)
    agg_n = agg_n.view(b, 8, q, c).sum(1)                                   # :431-433 (the sum over cameras is inside)
    blend = _linear(torch.cat([agg, agg_n], -1), p, 'output_weight')        # :435-436
    wts = blend.sum(1).softmax(-1)                                          # :437
    mixed = agg * wts[0][0] + agg_n * wts[0][1]                             # :438 (sample 0's weights for everyone)
    out = _linear(mixed, p, 'output_proj').permute(1, 0, 2)                 # :440-441
    pos = position_encoder(p, inverse_sigmoid(centre)).permute(1, 0, 2)     # :447
    res = out + query + pos                                                 # :449
    if return_parts:
        return res, dict(agg=agg, agg_n=agg_n, mask_n=mask_n, blend=blend, mixed=mixed, pos=pos, value=val)
    return res


def position_encoder(p, x, prefix='position_encoder'):
    """deform3d_cross_attn.py:104-111: Linear, LN, ReLU, Linear, LN, ReLU."""
    c = p[prefix + '.0.weight'].shape[0]
    x = F.linear(x, p[prefix + '.0.weight'], p[prefix + '.0.bias'])
    x = F.relu(F.layer_norm(x, (c,), p[prefix + '.1.weight'], p[prefix + '.1.bias']))
    x = F.linear(x, p[prefix + '.3.weight'], p[prefix + '.3.bias'])
    return F.relu(F.layer_norm(x, (c,), p[prefix + '.4.weight'], p[prefix + '.4.bias']))


def feature_sampling(mlvl_feats, reference_points, pc_range, img_metas):
    """feature_sampling, detr3d_transformer.py:397-438 (Appendix A.2).

    Returns (ref_3d (B,Q,3), sampled (B,C,Q,N,1,L), mask (B,1,Q,N,1,1) bool).
    """
    l2i = lidar2img_tensor(img_metas, reference_points)
    b, q, _ = reference_points.shape
    n = l2i.shape[1]
    img_h, img_w = img_metas[0]['img_shape'][0][0], img_metas[0]['img_shape'][0][1]
    pts = denormalise(reference_points, pc_range)
    hom = torch.cat((pts, torch.ones_like(pts[..., :1])), -1)
    hom = hom.view(b, 1, q, 4).repeat(1, n, 1, 1).unsqueeze(-1)
    cam = torch.matmul(l2i.view(b, n, 1, 4, 4).repeat(1, 1, q, 1, 1), hom).squeeze(-1)
    eps = 1e-5
    z = cam[..., 2:3]
    mask = z > eps
    uv = cam[..., 0:2] / torch.maximum(z, torch.ones_like(z) * eps)
    uv[..., 0] /= img_w
    uv[..., 1] /= img_h
    uv = (uv - 0.5) * 2                                                     # :421
    mask = (mask & (uv[..., 0:1] > -1.0) & (uv[..., 0:1] < 1.0)
            & (uv[..., 1:2] > -1.0) & (uv[..., 1:2] < 1.0))
    mask = mask.view(b, n, 1, q, 1, 1).permute(0, 2, 3, 1, 4, 5)
    sampled = []
    for feat in mlvl_feats:
        _, _, c, h, w = feat.shape
        s = F.grid_sample(feat.reshape(b * n, c, h, w), uv.view(b * n, q, 1, 2),
                          mode='bilinear', padding_mode='zeros', align_corners=False)
        sampled.append(s.view(b, n, c, q, 1).permute(0, 2, 3, 1, 4))
    sampled = torch.stack(sampled, -1).view(b, -1, q, n, 1, len(mlvl_feats))
    return reference_points.clone(), sampled, mask


def detr3d_cross_atten(p, query, value, query_pos, reference_points, img_metas, pc_range,
                       num_points=1):
    """Detr3DCrossAtten.forward in eval mode, detr3d_transformer.py:358-390 (Appendix A.2)."""
    x = query if query_pos is None else query + query_pos
    x = x.permute(1, 0, 2)
    b, q, _ = x.shape
    n = value[0].shape[1]
    nl = len(value)
    logits = _linear(x, p, 'attention_weights').view(b, 1, q, n, num_points, nl)   # :373-374
    ref3d, sampled, mask = feature_sampling(value, reference_points, pc_range, img_metas)
    sampled = torch.nan_to_num(sampled)
    w = logits.sigmoid() * mask                                             # :381
    out = (sampled * w).sum(-1).sum(-1).sum(-1).permute(2, 0, 1)            # :382-384
    out = _linear(out, p, 'output_proj')
    pos = position_encoder(p, inverse_sigmoid(ref3d)).permute(1, 0, 2)
    return out + query + pos                                                # :390


def multihead_self_attn(p, query, query_pos, num_heads=8, attn_mask=None, prefix='attn.'):
    """Decoder self-attention: third-party mmcv MultiheadAttention -> nn.MultiheadAttention
    (config call site: projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:74-78).

    q = k = query + query_pos, v = query; packed in_proj; softmax(q k^T / sqrt(d)) v; out_proj;
    returns identity + out (dropout = identity in eval).  query (Q, B, C).
    """
    nq, b, c = query.shape
    d = c // num_heads
    qk = query if query_pos is None else query + query_pos
    w, bias = p[prefix + 'in_proj_weight'], p[prefix + 'in_proj_bias']
    qh = F.linear(qk, w[:c], bias[:c])
    kh = F.linear(qk, w[c:2 * c], bias[c:2 * c])
    vh = F.linear(query, w[2 * c:], bias[2 * c:])

    def heads(t):
        return t.reshape(nq, b * num_heads, d).transpose(0, 1)              # (B*h, Q, d)
    qh, kh, vh = heads(qh), heads(kh), heads(vh)
    scores = torch.bmm(qh * (1.0 / math.sqrt(d)), kh.transpose(1, 2))
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            scores = scores.masked_fill(attn_mask, float('-inf'))
        else:
            scores = scores + attn_mask
    o = torch.bmm(scores.softmax(-1), vh).transpose(0, 1).reshape(nq, b, c)
    o = F.linear(o, p[prefix + 'out_proj.weight'], p[prefix + 'out_proj.bias'])
    return query + o


def ffn(p, x, prefix='ffns.0.'):
    """Third-party mmcv FFN (config :86-87): x + W2 relu(W1 x)."""
    h = F.relu(F.linear(x, p[prefix + 'layers.0.0.weight'], p[prefix + 'layers.0.0.bias']))
    return x + F.linear(h, p[prefix + 'layers.1.weight'], p[prefix + 'layers.1.bias'])


def _sub(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def decoder_layer(p, query, value, query_pos, reference_points, img_metas, pc_range,
                  cross='Deform3DCrossAttn', num_heads=8, num_points=4, attn_mask=None,
                  depth_encode=False, return_parts=False, value_dtype='fp32'):
    """Post-norm DetrTransformerDecoderLayer (third-party mmdet/mmcv), order
    self_attn, norm, cross_attn, norm, ffn, norm (config :88-89; Appendix A.4).
    return_parts (Deform3DCrossAttn only): also return the cross-attention's intermediates (mask, uv, ...)."""
    c = query.shape[-1]

    def ln(x, i):
        return F.layer_norm(x, (c,), p[f'norms.{i}.weight'], p[f'norms.{i}.bias'])
    x = multihead_self_attn(p, query, query_pos, num_heads, attn_mask, prefix='attentions.0.attn.')
    x = ln(x, 0)
    cp = _sub(p, 'attentions.1.')
    parts = None
    if cross == 'Deform3DCrossAttn':
        x = deform3d_cross_attn(cp, x, value, query_pos, reference_points, img_metas, pc_range,
                                num_heads, num_points, depth_encode, return_parts=return_parts,
                                value_dtype=value_dtype)
        if return_parts:
            x, parts = x
    else:
        x = detr3d_cross_atten(cp, x, value, query_pos, reference_points, img_metas, pc_range,
                               num_points)
    x = ln(x, 1)
    x = ffn(p, x)
    return (ln(x, 2), parts) if return_parts else ln(x, 2)


def decoder(layer_params, query, value, query_pos, reference_points, img_metas, pc_range,
            reg_branches=None, **kw):
    """Detr3DTransformerDecoder.forward with return_intermediate=True,
    detr3d_transformer.py:166-225.  reg_branches: list of callables (Q-first output -> (B,Q,>=5))."""
    out = query
    inter, inter_ref = [], []
    for lid, p in enumerate(layer_params):
        out = decoder_layer(p, out, value, query_pos, reference_points, img_metas, pc_range, **kw)
        if reg_branches is not None:
            tmp = reg_branches[lid](out.permute(1, 0, 2))                   # :199-202
            new = torch.zeros_like(reference_points)
            new[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points[..., :2])
            new[..., 2:3] = tmp[..., 4:5] + inverse_sigmoid(reference_points[..., 2:3])
            reference_points = new.sigmoid().detach()                       # :212-214
        inter.append(out)
        inter_ref.append(reference_points)
    return torch.stack(inter), torch.stack(inter_ref)


def transformer(p, layer_params, mlvl_feats, query_embed, img_metas, pc_range,
                reg_branches=None, **kw):
    """Detr3DTransformer.forward, detr3d_transformer.py:86-150."""
    b = mlvl_feats[0].shape[0]
    c = query_embed.shape[1] // 2
    query_pos, query = torch.split(query_embed, c, dim=1)                   # :130
    query_pos = query_pos.unsqueeze(0).expand(b, -1, -1)
    query = query.unsqueeze(0).expand(b, -1, -1)
    ref = _linear(query_pos, p, 'reference_points').sigmoid()               # :133-134
    states, refs = decoder(layer_params, query.permute(1, 0, 2), mlvl_feats,
                           query_pos.permute(1, 0, 2), ref, img_metas, pc_range,
                           reg_branches=reg_branches, **kw)
    return states, ref, refs


# ---------------------------------------------------------------------------------------------
# the step right after the decoder
# ---------------------------------------------------------------------------------------------
def box_head(tmp, reference, pc_range, depth_factor=None):
    """Per-layer box epilogue of Detr3DHeadPE.forward (models/dense_heads/detr3d_head_pe.py:571-600).

    Pinned by tests/golden/head_pe.npz: all_bbox_preds returned by the reference's own Detr3DHeadPE.forward run on a
    shell object (tools/gen_golden.py case_head_pe), tests/test_head_pe_oracle.py."""
    ref = inverse_sigmoid(reference)
    out = tmp.clone()
    out[..., 0:2] = (tmp[..., 0:2] + ref[..., 0:2]).sigmoid()
    out[..., 4:5] = (tmp[..., 4:5] + ref[..., 2:3]).sigmoid()
    for c, lo, hi in ((0, 0, 3), (1, 1, 4), (4, 2, 5)):
        v = out[..., c:c + 1] * (pc_range[hi] - pc_range[lo]) + pc_range[lo]
        out[..., c:c + 1] = v if depth_factor is None else v * depth_factor
    return out


def denormalize_bbox(b):
    """core/bbox/util.py:58-87: (cx, cy, log w, log l, cz, log h, sin, cos[, vx, vy]) -> (cx, cy, cz, w, l, h, rot[, vx, vy])."""
    rot = torch.atan2(b[..., 6:7], b[..., 7:8])
    cols = [b[..., 0:1], b[..., 1:2], b[..., 4:5], b[..., 2:3].exp(), b[..., 3:4].exp(), b[..., 5:6].exp(), rot]
    if b.shape[-1] > 8:
        cols += [b[..., 8:9], b[..., 9:10]]
    return torch.cat(cols, dim=-1)


def nms_free_decode_single(cls_scores, bbox_preds, post_center_range, max_num, num_classes, score_threshold=None):
    """core/bbox/coders/nms_free_coder.py:47-96.  Ties in the top-k are broken by the lower flat index."""
    s = cls_scores.sigmoid().view(-1)
    order = np.lexsort((np.arange(s.numel()), -s.numpy().astype(np.float64)))[:max_num]
    if max_num > s.numel():
        raise RuntimeError('selected index k out of range')
    idx = torch.from_numpy(order.copy())
    scores = s[idx]
    labels = idx % num_classes
    boxes = denormalize_bbox(bbox_preds[idx // num_classes])
    rng = torch.tensor(post_center_range, dtype=torch.float32)
    mask = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
    if score_threshold:
        mask &= scores > score_threshold
    return {'bboxes': boxes[mask], 'scores': scores[mask], 'labels': labels[mask]}


def nms_free_decode(preds, post_center_range, max_num, num_classes, score_threshold=None):
    """nms_free_coder.py:98-117: decode the last decoder layer, one dict per batch element."""
    cls, box = preds['all_cls_scores'][-1], preds['all_bbox_preds'][-1]
    return [nms_free_decode_single(cls[b], box[b], post_center_range, max_num, num_classes, score_threshold)
            for b in range(cls.shape[0])]


# ---------------------------------------------------------------------------------------------
# the step feeding the path: Detr3DHeadPE's feature position embedding
# paths below: projects/mmdet3d_plugin/models/dense_heads/detr3d_head_pe.py unless noted
# ---------------------------------------------------------------------------------------------
def frustum_depths(depth_num, depth_start, pc_range):
    """:450-453 (the LID branch is the one left active): d_i = start + bin * i * (i + 1)."""
    idx = torch.arange(depth_num, dtype=torch.float32)
    bin_size = (pc_range[3] - depth_start) / (depth_num * (1 + depth_num))
    return depth_start + bin_size * idx * (idx + 1)


def frustum_points(lidar2img, feat_hw, pad_hw, depth_num, depth_start, pc_range, eps=1e-5):
    """:438-476: pixel centres x depth bins -> lidar frame -> normalised by pc_range.

    lidar2img (B, N, 4, 4) array-like.  Returns coords3d (B, N, W, H, D, 3) in pc_range units and the
    'more than half of the depth bins fall outside the range' flag (B, N, W, H)."""
    h, w = feat_hw
    coords_h = torch.arange(h).float() * pad_hw[0] / h
    coords_w = torch.arange(w).float() * pad_hw[1] / w
    coords_d = frustum_depths(depth_num, depth_start, pc_range)
    d = coords_d.shape[0]
    coords = torch.stack(torch.meshgrid([coords_w, coords_h, coords_d], indexing='ij')).permute(1, 2, 3, 0)   # W, H, D, 3
    coords = torch.cat((coords, torch.ones_like(coords[..., :1])), -1)
    coords[..., :2] = coords[..., :2] * torch.maximum(coords[..., 2:3], torch.ones_like(coords[..., 2:3]) * eps)
    img2lidar = torch.from_numpy(np.linalg.inv(np.asarray(lidar2img, dtype=np.float64))).float()   # :459-465 (numpy inverse)
    img2lidar = img2lidar.to(coords.device)         # `coords.new_tensor(...)` in the reference
    b, n = img2lidar.shape[:2]
    coords = coords.view(1, 1, w, h, d, 4, 1).repeat(b, n, 1, 1, 1, 1, 1)
    mats = img2lidar.view(b, n, 1, 1, 1, 4, 4).repeat(1, 1, w, h, d, 1, 1)
    c3 = torch.matmul(mats, coords).squeeze(-1)[..., :3]
    for k in range(3):
        c3[..., k:k + 1] = (c3[..., k:k + 1] - pc_range[k]) / (pc_range[k + 3] - pc_range[k])
    outside = ((c3 > 1.0) | (c3 < 0.0)).flatten(-2).sum(-1) > (d * 0.5)
    return c3, outside


def conv1x1(x, w, b):
    return F.conv2d(x, w, b)


def frustum_position_embedding(p, lidar2img, masks, feat_shapes, pad_hw, depth_num, depth_start, pc_range):
    """`position_embeding` :427-491.  p: state dict with 'position_encoder.{0,2}.{weight,bias}'; masks: list of
    (B, N, H_l, W_l) bool; feat_shapes: [(H_l, W_l)].  Returns per level (B, N, C, H, W) embeddings and masks."""
    out, out_masks = [], []
    for lvl, (h, w) in enumerate(feat_shapes):
        c3, outside = frustum_points(lidar2img, (h, w), pad_hw, depth_num, depth_start, pc_range)
        b, n = c3.shape[:2]
        mask = masks[lvl] | outside.permute(0, 1, 3, 2)
        x = c3.permute(0, 1, 4, 5, 3, 2).contiguous().view(b * n, -1, h, w)          # channel = d * 3 + axis
        x = inverse_sigmoid(x)
        x = conv1x1(torch.relu(conv1x1(x, p['position_encoder.0.weight'], p['position_encoder.0.bias'])),
                    p['position_encoder.2.weight'], p['position_encoder.2.bias'])
        out.append(x.view(b, n, -1, h, w))
        out_masks.append(mask)
    return out, out_masks


def sine_positional_encoding_3d(mask, num_feats=128, temperature=10000, normalize=True, scale=2 * math.pi, eps=1e-6,
                                offset=-0.5):
    """models/utils/positional_encoding.py:58-100.  mask (B, N, H, W) bool -> (B, N, 3 * num_feats, H, W)."""
    not_mask = 1 - mask.to(torch.int)
    n_embed = not_mask.cumsum(1, dtype=torch.float32)
    y_embed = not_mask.cumsum(2, dtype=torch.float32)
    x_embed = not_mask.cumsum(3, dtype=torch.float32)
    if normalize:
        n_embed = (n_embed + offset) / (n_embed[:, -1:, :, :] + eps) * scale
        y_embed = (y_embed + offset) / (y_embed[:, :, -1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, :, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    b, n, h, w = mask.shape
    parts = []
    for e in (n_embed, y_embed, x_embed):
        pos = e[:, :, :, :, None] / dim_t
        parts.append(torch.stack((pos[..., 0::2].sin(), pos[..., 1::2].cos()), dim=4).view(b, n, h, w, -1))
    return torch.cat(parts, dim=4).permute(0, 1, 4, 2, 3)


def se_gate(p, x, x_se, prefix='fpe.'):
    """SELayer :231-243: x * sigmoid(conv_expand(relu(conv_reduce(x_se))))."""
    g = conv1x1(torch.relu(conv1x1(x_se, p[prefix + 'conv_reduce.weight'], p[prefix + 'conv_reduce.bias'])),
                p[prefix + 'conv_expand.weight'], p[prefix + 'conv_expand.bias'])
    return x * torch.sigmoid(g)


def feature_position_embedding(p, feats, lidar2img, img_shapes, pad_shape, depth_num, depth_start, pc_range):
    """Detr3DHeadPE.forward :525-557: padding masks, frustum position embedding, SE fusion with the features, sine
    3-D encoding through adapt_pos3d, added to the feature maps.  feats: list of (B, N, C, H_l, W_l);
    img_shapes: per camera (h, w, 3) of sample 0..B-1 (list of lists).  Returns (new feats, intermediates)."""
    b, n = feats[0].shape[:2]
    pad_h, pad_w = pad_shape[0], pad_shape[1]
    full = torch.ones(b, n, pad_h, pad_w, device=feats[0].device)
    for i in range(b):
        for c in range(n):
            ih, iw = img_shapes[i][c][0], img_shapes[i][c][1]
            full[i, c, :ih, :iw] = 0
    masks = [F.interpolate(full, size=f.shape[-2:]).to(torch.bool) for f in feats]
    shapes = [tuple(f.shape[-2:]) for f in feats]
    coords_pe, coords_masks = frustum_position_embedding(p, lidar2img, masks, shapes, (pad_h, pad_w), depth_num,
                                                         depth_start, pc_range)
    outs, sines = [], []
    for lvl, f in enumerate(feats):
        pe = se_gate(p, coords_pe[lvl].flatten(0, 1), f.flatten(0, 1)).view(f.shape)
        sine = sine_positional_encoding_3d(masks[lvl])
        s = conv1x1(torch.relu(conv1x1(sine.flatten(0, 1), p['adapt_pos3d.0.weight'], p['adapt_pos3d.0.bias'])),
                    p['adapt_pos3d.2.weight'], p['adapt_pos3d.2.bias']).view(f.shape)
        outs.append(f + (pe + s))
        sines.append(sine)
    return outs, dict(masks=masks, coords_pe=coords_pe, coords_masks=coords_masks, sine=sines)


# ---------------------------------------------------------------------------------------------
# DGCNNAttn (row a16; utils/dgcnn_attn.py), eval mode
# ---------------------------------------------------------------------------------------------
def dgcnn_edge_stage(p, x, k, prefix):
    """edge_feats + conv + BatchNorm(eval) + ReLU + max over the K neighbours (dgcnn_attn.py:72-74, 82-96).
    x (B, N, C).  Note the neighbours are the K FARTHEST points (topk of the distances, :84-86).
    Returns ((B, C, N) features, (B, N, K) neighbour indices)."""
    idx = torch.topk(torch.cdist(x, x), k=k, dim=2)[1]
    b, n, c = x.shape
    nbr = torch.gather(x.unsqueeze(1).expand(b, n, n, c), 2, idx.unsqueeze(-1).expand(b, n, k, c))   # (B, N, K, C)
    edge = torch.cat((nbr, x.unsqueeze(2).expand(b, n, k, c)), -1).permute(0, 3, 1, 2)               # (B, 2C, N, K)
    y = F.conv2d(edge, p[prefix + '0.weight'])
    y = F.batch_norm(y, p[prefix + '1.running_mean'], p[prefix + '1.running_var'], p[prefix + '1.weight'],
                     p[prefix + '1.bias'], training=False, eps=1e-5)
    return torch.relu(y).max(dim=-1)[0], idx


def dgcnn_attn(p, query, query_pos, k, return_parts=False):
    """DGCNNAttn.forward (:41-80): residual + f1 + f2; the second stage always uses K = 16 (default argument, :75)."""
    x = query if query_pos is None else query + query_pos
    x = x.permute(1, 0, 2)
    f1, idx1 = dgcnn_edge_stage(p, x, k, 'conv1.')
    f2, _ = dgcnn_edge_stage(p, f1.permute(0, 2, 1), 16, 'conv2.')
    out = query + (f1 + f2).permute(2, 0, 1)
    return (out, dict(f1=f1, idx1=idx1)) if return_parts else out


# ---------------------------------------------------------------------------------------------
# Detr3DCrossAttenV2 (row a11; utils/detr3d_transformer.py:441-710), eval mode, batch 1
# ---------------------------------------------------------------------------------------------
def detr3d_cross_atten_v2(p, query, value, query_pos, reference_points, img_metas, pc_range, num_heads=8, num_points=4,
                          return_parts=False):
    """Per (camera, head, level, point) 2-D offsets (in pixels of the level) around the projected reference point,
    softmax over level x point per (camera, head), bilinear samples of the head's channel slice of the raw NCHW maps
    (grid_sample, align_corners=False, zero padding), sum over cameras / levels / points.

    Quirk reproduced (:611-617 against :705-707): the samples are stacked as (..., point, level) but multiplied with
    weights laid out (..., level, point); with num_levels == num_points the product pairs the sample at
    (point i, level j) with the weight of (level i, point j)."""
    x = query if query_pos is None else query + query_pos
    x = x.permute(1, 0, 2)
    b, q, c = x.shape
    assert b == 1, 'the reference broadcasts (B*N, Q) against (B*heads, N, Q): batch 1 only (:698-700)'
    n, nl = value[0].shape[1], len(value)
    d = c // num_heads
    logits = _linear(x, p, 'attention_weights').view(b, q, n, num_heads, nl * num_points)
    w = logits.softmax(-1).view(b, q, n, num_heads, nl, num_points)
    off = _linear(x, p, 'sampling_offsets').view(b, q, n, num_heads, nl, num_points, 2)
    l2i = lidar2img_tensor(img_metas, reference_points)
    img_h, img_w = img_metas[0]['img_shape'][0][0], img_metas[0]['img_shape'][0][1]
    uv, _ = project(denormalise(reference_points, pc_range), l2i, img_h, img_w)      # (B, N, Q, 2) in [0, 1] units
    z_ok = _z_positive(denormalise(reference_points, pc_range), l2i)
    g = (uv - 0.5) * 2                                                               # :676
    mask = z_ok & (g[..., 0] > -1.0) & (g[..., 0] < 1.0) & (g[..., 1] > -1.0) & (g[..., 1] < 1.0)   # (B, N, Q)
    out = torch.zeros(q, num_heads, d)
    for lvl, feat in enumerate(value):
        h_l, w_l = feat.shape[-2:]
        f = feat.view(b, n, num_heads, d, h_l, w_l).transpose(1, 2).flatten(0, 2)    # (heads*N, d, H, W)
        o = off[:, :, :, :, lvl].permute(0, 3, 2, 1, 4, 5).flatten(0, 1)            # (heads, N, Q, P, 2)
        loc = g.view(b * n, q, 1, 2)[None] + o / torch.tensor([w_l, h_l], dtype=torch.float32)
        s = F.grid_sample(f, loc.flatten(0, 1), mode='bilinear', padding_mode='zeros', align_corners=False)
        s = s.view(num_heads, n, d, q, num_points)                                   # (heads, N, d, Q, P)
        # weight of the sample (point i, THIS level j = lvl): w[..., level i, point j]   (the transposition quirk)
        wq = w[0, :, :, :, :, lvl].permute(2, 1, 0, 3)                               # (heads, N, Q, i) = w[q,n,h,i,lvl]
        wq = wq * mask[0].view(1, n, q, 1)
        out += torch.einsum('hndqp,hnqp->qhd', s, wq)
    agg = out.reshape(q, 1, c)
    res = _linear(agg, p, 'output_proj')
    pos = position_encoder(p, inverse_sigmoid(reference_points)).permute(1, 0, 2)
    res = res + query + pos
    if return_parts:
        return res, dict(agg=agg, mask=mask, logits=logits, offsets=off)
    return res


def _z_positive(points, lidar2img, eps=1e-5):
    """Depth test of the projection block on its own (the [0,1] test of `project` does not apply to V2's [-1,1] one)."""
    b, m, _ = points.shape
    n = lidar2img.shape[1]
    hom = torch.cat((points, torch.ones_like(points[..., :1])), -1).view(b, 1, m, 4).repeat(1, n, 1, 1).unsqueeze(-1)
    cam = torch.matmul(lidar2img.view(b, n, 1, 4, 4).repeat(1, 1, m, 1, 1), hom).squeeze(-1)
    return cam[..., 2] > eps


# ----------------------------------------------------------------------------------------------------------------------
# Head loss: Hungarian assignment + per-layer focal / L1 losses.  Test infrastructure only.
# ----------------------------------------------------------------------------------------------------------------------
def normalize_bbox(bboxes):
    """core/bbox/util.py:38-58: (cx, cy, cz, w, l, h, rot[, vx, vy]) -> (cx, cy, log w, log l, cz, log h, sin, cos[, vx, vy])."""
    parts = [bboxes[..., 0:1], bboxes[..., 1:2], bboxes[..., 3:4].log(), bboxes[..., 4:5].log(), bboxes[..., 2:3],
             bboxes[..., 5:6].log(), bboxes[..., 6:7].sin(), bboxes[..., 6:7].cos()]
    if bboxes.size(-1) > 7:
        parts += [bboxes[..., 7:8], bboxes[..., 8:9]]
    return torch.cat(parts, dim=-1)


def focal_loss_cost(cls_pred, gt_labels, weight=2.0, alpha=0.25, gamma=2.0, eps=1e-12):
    """mmdet 2.x FocalLossCost (third-party, absent from /root/reference; published definition): (Q, C) logits and (G,)
    labels -> (Q, G).  Config: ...ceph.py:133 (`cls_cost=dict(type='FocalLossCost', weight=2.0)`)."""
    p = cls_pred.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    return (pos[:, gt_labels] - neg[:, gt_labels]) * weight


def hungarian_cost(bbox_pred, cls_pred, gt_bboxes, gt_labels, cls_weight=2.0, reg_weight=0.25):
    """hungarian_assigner_3d.py:117-130 with BBox3DL1Cost (match_costs/match_cost.py:17-30): the (Q, G) matrix handed to
    scipy, non-finite entries replaced (nan / +inf -> 100, -inf -> -100)."""
    cost = focal_loss_cost(cls_pred, gt_labels, cls_weight) + \
        torch.cdist(bbox_pred[:, :8], normalize_bbox(gt_bboxes)[:, :8], p=1) * reg_weight
    return torch.nan_to_num(cost.detach().cpu(), nan=100.0, posinf=100.0, neginf=-100.0)


def hungarian_assign(bbox_pred, cls_pred, gt_bboxes, gt_labels, cls_weight=2.0, reg_weight=0.25):
    """HungarianAssigner3D.assign (hungarian_assigner_3d.py:62-144): assigned_gt_inds (Q,) long, 0 = background,
    g + 1 = matched to ground truth g."""
    from scipy.optimize import linear_sum_assignment
    q, g = bbox_pred.size(0), gt_bboxes.size(0)
    inds = bbox_pred.new_zeros(q, dtype=torch.long)
    if g == 0 or q == 0:
        return inds
    rows, cols = linear_sum_assignment(hungarian_cost(bbox_pred, cls_pred, gt_bboxes, gt_labels, cls_weight, reg_weight))
    inds[torch.from_numpy(rows)] = torch.from_numpy(cols) + 1
    return inds


def sigmoid_focal_loss_sum(pred, labels, num_classes, alpha=0.25, gamma=2.0):
    """mmdet FocalLoss(use_sigmoid=True) before weighting / normalisation (published python path): labels ==
    num_classes are background."""
    t = F.one_hot(labels, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    w = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    return (F.binary_cross_entropy_with_logits(pred, t, reduction='none') * w).sum()


def head_loss_single(cls_scores, bbox_preds, gt_bboxes_list, gt_labels_list, code_weights, num_classes=10,
                     cls_cost_weight=2.0, reg_cost_weight=0.25, loss_cls_weight=2.0, loss_bbox_weight=0.25,
                     bg_cls_weight=0.0, world_mean=lambda t: t):
    """Detr3DHeadPE.loss_single (detr3d_head_pe.py:782-845) with get_targets / _get_target_single (:688-780):
    cls_scores (B, Q, C), bbox_preds (B, Q, 10), per-sample ground truth (G_b, 9) / (G_b,).  `world_mean` stands for
    mmdet's reduce_mean (identity in one process).  Returns (loss_cls, loss_bbox, assigned list)."""
    b, q, c = cls_scores.shape
    labels, targets, weights, assigned = [], [], [], []
    num_pos = num_neg = 0
    for i in range(b):
        inds = hungarian_assign(bbox_preds[i], cls_scores[i], gt_bboxes_list[i], gt_labels_list[i], cls_cost_weight,
                                reg_cost_weight)
        assigned.append(inds)
        pos = inds > 0
        lab = gt_labels_list[i].new_full((q,), num_classes, dtype=torch.long)
        lab[pos] = gt_labels_list[i][inds[pos] - 1].long()
        tgt = torch.zeros_like(bbox_preds[i])[..., :gt_bboxes_list[i].size(1)]
        tgt[pos] = gt_bboxes_list[i][inds[pos] - 1]
        w = torch.zeros_like(bbox_preds[i])
        w[pos] = 1.0
        labels.append(lab); targets.append(tgt); weights.append(w)
        num_pos += int(pos.sum()); num_neg += int((~pos).sum())
    labels, targets, weights = torch.cat(labels), torch.cat(targets), torch.cat(weights)
    cls_avg = world_mean(cls_scores.new_tensor([num_pos * 1.0 + num_neg * bg_cls_weight]))
    cls_avg = max(cls_avg, 1)
    loss_cls = loss_cls_weight * sigmoid_focal_loss_sum(cls_scores.reshape(-1, c), labels, num_classes) / cls_avg
    pos_avg = torch.clamp(world_mean(loss_cls.new_tensor([num_pos])), min=1).item()
    pred = bbox_preds.reshape(-1, bbox_preds.size(-1))
    norm = normalize_bbox(targets)
    ok = torch.isfinite(norm).all(dim=-1)
    weights = weights * code_weights
    if ok.any():
        loss_bbox = loss_bbox_weight * ((pred[ok, :10] - norm[ok, :10]).abs() * weights[ok, :10]).sum() / pos_avg
    else:
        loss_bbox = pred.sum() * 0
    return torch.nan_to_num(loss_cls).reshape(()), torch.nan_to_num(loss_bbox).reshape(()), assigned


def head_loss(all_cls_scores, all_bbox_preds, gt_bboxes_list, gt_labels_list, code_weights, **kw):
    """Detr3DHeadPE.loss (detr3d_head_pe.py:1014-1094) without the two-stage branch: loss_single per decoder layer; the
    last layer's terms are `loss_cls` / `loss_bbox`, the others `d{i}.loss_cls` / `d{i}.loss_bbox`."""
    out, per_layer = {}, []
    for l in range(all_cls_scores.shape[0]):
        per_layer.append(head_loss_single(all_cls_scores[l], all_bbox_preds[l], gt_bboxes_list, gt_labels_list,
                                          code_weights, **kw))
    out['loss_cls'], out['loss_bbox'] = per_layer[-1][0], per_layer[-1][1]
    for i, (lc, lb, _) in enumerate(per_layer[:-1]):
        out[f'd{i}.loss_cls'], out[f'd{i}.loss_bbox'] = lc, lb
    return out, [p[2] for p in per_layer]


def instance_distill_loss(t_cls_scores, t_bbox_preds, s_cls_scores, s_bbox_preds, loss_cls_weight=1.0, loss_reg_weight=1.0,
                          reweight_score=True):
    """ORACLE.  MixDistill.get_instance_distill_loss, projects/mmdet3d_plugin/distillation/distillers/mix_distill.py:140-168,
    stage by stage as the reference loops: teacher logits / boxes detached (:150), q_score = max class sigmoid (:153-154),
    BCE-with-logits against the teacher's sigmoid scores (:157), L1 on the box codes (:158), both normalised by
    sum(q_score) * num_class + 1e-10 when reweight_score (:161-162) else plain means (:164-165); weights :167-168.
    Pinned by tests/golden/distill_loss*.npz (the reference method itself, tools/gen_golden.py::case_distill)."""
    out = {}
    for i in range(len(t_cls_scores)):
        t_cls = t_cls_scores[i].detach().sigmoid()
        t_box = t_bbox_preds[i].detach()
        q_score = torch.max(t_cls, dim=-1, keepdim=True)[0]
        num_class = t_cls.shape[-1]
        cls_loss = F.binary_cross_entropy_with_logits(s_cls_scores[i], t_cls, reduction='none')
        reg_loss = F.l1_loss(s_bbox_preds[i], t_box, reduction='none')
        if reweight_score:
            cls_loss = torch.sum(q_score * cls_loss) / (torch.sum(q_score) * num_class + 1e-10)
            reg_loss = torch.sum(q_score * reg_loss) / (torch.sum(q_score) * num_class + 1e-10)
        else:
            cls_loss, reg_loss = torch.mean(cls_loss), torch.mean(reg_loss)
        out['distill_loss_cls.%d' % i] = cls_loss * loss_cls_weight
        out['distill_loss_reg.%d' % i] = reg_loss * loss_reg_weight
    return out

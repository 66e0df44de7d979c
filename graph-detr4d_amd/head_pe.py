"""Feature position embedding of Detr3DHeadPE - the step feeding the decoder path.

Mirror of the part of `Detr3DHeadPE` that rewrites the multi-camera feature maps before the transformer
(projects/mmdet3d_plugin/models/dense_heads/detr3d_head_pe.py: `_init_layers` :380-390, `position_embeding` :427-491,
`forward` :525-557): same submodule names (`position_encoder`, `adapt_pos3d`, `fpe.conv_reduce / conv_expand`), so the
corresponding slice of a head checkpoint loads with strict=True.

What runs where: the frustum geometry (+ inverse_sigmoid, written directly as the conv input, no (B,N,W,H,D,4,4)
temporaries), the sine / cosine expansion and the SE-gate + adds are HIP kernels (ops.frustum_pe_input_fwd,
ops.sine_pe3d_fwd, ops.se_fuse_fwd); the 1x1 convolutions are plain library GEMMs.  The sine branch depends on the
padding masks only, so its result is cached while the masks do not change (every sample of a dataset shares them).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from . import ops


class SELayer(nn.Module):
    """detr3d_head_pe.py:231-243 (parameter holder; the gate is applied by ops.se_fuse_fwd)."""

    def __init__(self, channels):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)

    def gate_logits(self, x_se):
        return self.conv_expand(F.relu(self.conv_reduce(x_se)))


class _HeadPEFunction(torch.autograd.Function):
    """The differentiable part of the stage on the library's kernels, forward and backward: the two position MLPs
    (Conv1x1 -> ReLU -> Conv1x1, detr3d_head_pe.py:380-390), the SE gate (:231-243) and the sum (:556), all in
    channels-last rows over every level side by side.

    forward:  hid = relu(x W0^T + b0), pe = hid W2^T + b2 (gd4d_gemm_bf16x3_fwd); the same for the sine input;
              g1 = conv_reduce(feats) (gd4d_value_proj_fwd, NCHW in), gate = relu(g1) We^T + be; out = feats + pe
              sigmoid(gate) + sine (gd4d_se_fuse_chlast_fwd).
    backward: gd4d_se_fuse_chlast_bwd (transposes grad_out, gate / pe derivatives in place), gd4d_gemm_tn_bf16x3 for every
              weight / bias gradient, gd4d_gemm_bf16x3_fwd with GD4D_GEMM_MASK_C for the gradients at the ReLUs (written
              over the saved activations: the buffers are used once, so a second backward through the same graph is
              refused), gd4d_value_proj_bwd_weight / _bwd_input for conv_reduce (NCHW side).
    Inputs x / xs (frustum coordinates, sine expansion) carry no gradient (the reference builds them under no_grad /
    from masks)."""

    @staticmethod
    def forward(ctx, x, xs, starts, nl, *args):
        feats = [f.contiguous() for f in args[:nl]]
        w0, b0, w2, b2, a0, ab0, a2, ab2, wr, br, we, be = args[nl:]
        r = feats[0].shape[0]
        s_tot = x.shape[1]
        c = w2.shape[0]
        flat = lambda w: w.detach().reshape(w.shape[0], -1).contiguous()
        split = lambda w: ops.split_bf16_fwd(flat(w))
        hid = ops.gemm_bf16x3_fwd(x.view(r * s_tot, -1), *split(w0), b0, relu=True)
        pe = ops.gemm_bf16x3_fwd(hid, *split(w2), b2).view(r, s_tot, c)
        hs = ops.gemm_bf16x3_fwd(xs.view(r * s_tot, -1), *split(a0), ab0, relu=True)
        sine = ops.gemm_bf16x3_fwd(hs, *split(a2), ab2).view(r, s_tot, c)
        g1 = ops.value_proj_fwd(feats, flat(wr), br.detach().contiguous())
        gate = ops.gemm_bf16x3_fwd(g1.view(r * s_tot, -1), *split(we), be, relu_in=True).view(r, s_tot, c)
        out = [ops.se_fuse_chlast_fwd(f, gate, pe, sine, st) for f, st in zip(feats, starts)]
        del sine
        ctx.bufs = dict(x=x, xs=xs, hid=hid, hs=hs, g1=g1, gate=gate, pe=pe, feats=feats)
        ctx.weights = (flat(w2), flat(a2), flat(wr), flat(we))
        ctx.shapes = [tuple(w.shape) for w in (w0, w2, a0, a2, wr, we)]
        ctx.starts, ctx.nl = starts, nl
        # what the backward reads is kept OUTSIDE save_for_backward (buffers are overwritten in place there; flat(w) are views of the
        # live parameters): an in-place update between the two passes (an optimizer step inside a captured step, EMA) would go
        # unnoticed - so the versions are recorded and compared, as fused_train does and as autograd does for its saved tensors
        ctx.watched = list(args)
        ctx.versions = [t._version for t in ctx.watched]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        if ctx.bufs is None:
            raise RuntimeError('FeaturePositionEmbedding: the saved activations were consumed by the first backward '
                               '(retain_graph / double backward are not supported on the HIP path; GD4D_TORCH_OPS=1 runs the torch-op route)')
        if [t._version for t in ctx.watched] != ctx.versions:
            raise RuntimeError('FeaturePositionEmbedding: a feature map or a parameter of the position embedding was modified in place '
                               'between the forward and the backward pass (the backward reads them) - as autograd itself refuses a '
                               'saved tensor that was modified in place')
        b, ctx.bufs = ctx.bufs, None
        ctx.watched = None
        nl, starts = ctx.nl, ctx.starts
        w2, a2, wr, we = ctx.weights
        feats = b['feats']
        r, s_tot, c = b['pe'].shape
        need = ctx.needs_input_grad[4:]
        need_f, need_p = need[:nl], need[nl:]
        dsine = torch.empty_like(b['pe'])
        dfeat = []
        for l in range(nl):
            g = grads[l]
            g = torch.zeros_like(feats[l]) if g is None else g.contiguous()
            ops.se_fuse_chlast_bwd(g, b['gate'], b['pe'], dsine, starts[l])       # gate / pe now hold their gradients
            dfeat.append(g)
        rows = r * s_tot
        dgate, dpe = b['gate'].view(rows, c), b['pe'].view(rows, c)
        out = [None] * 12
        tsplit = lambda w: ops.split_bf16_fwd(w.t().contiguous())

        def mlp(dy, hidden, inp, w_second, idx):
            """Conv -> ReLU -> Conv from the gradient of its output: fills out[idx .. idx + 3] (w0, b0, w2, b2)."""
            if not any(need_p[idx:idx + 4]):
                return
            gw, gb = ops.gemm_tn_bf16x3(dy, hidden)                               # second conv: (C, hidden)
            out[idx + 2], out[idx + 3] = gw, gb
            if need_p[idx] or need_p[idx + 1]:
                ops.gemm_bf16x3_fwd(dy, *tsplit(w_second), out=hidden, mask_out=True)   # gradient at the ReLU's input
                out[idx], out[idx + 1] = ops.gemm_tn_bf16x3(hidden, inp.view(rows, -1))

        mlp(dpe, b['hid'], b['x'], w2, 0)
        mlp(dsine.view(rows, c), b['hs'], b['xs'], a2, 4)
        del dsine
        g1 = b['g1'].view(rows, c)
        out[10], out[11] = ops.gemm_tn_bf16x3(dgate, g1, relu_b=True)
        if any(need_f) or need_p[8] or need_p[9]:
            ops.gemm_bf16x3_fwd(dgate, *tsplit(we), out=g1, mask_out=True)        # gradient of conv_reduce's output
            out[8], out[9] = ops.value_proj_bwd_weight(b['g1'], feats)
            if any(need_f):
                shapes = [tuple(f.shape[-2:]) for f in feats]
                dfeat = [g.clone() if g is go else g for g, go in zip(dfeat, grads)]     # grad_out is autograd's: not in place
                ops.value_proj_bwd_input(b['g1'], wr, shapes, grads=dfeat, accumulate=True)
        for i, sh in enumerate(ctx.shapes):                                       # conv weights are (out, in, 1, 1)
            if out[2 * i] is not None:
                out[2 * i] = out[2 * i].view(sh)
        shaped = [o if n else None for o, n in zip(out, need_p)]
        dfeat = [g if n else None for g, n in zip(dfeat, need_f)]
        return (None, None, None, None, *dfeat, *shaped)


class FeaturePositionEmbedding(nn.Module):
    def __init__(self, embed_dims=256, depth_num=64, depth_start=1, pc_range=None, num_feats=128, temperature=10000,
                 normalize=True, scale=2 * math.pi, eps=1e-6, offset=-0.5, with_detach=True, cams_per_frame=6,
                 channels_last_out=False):
        """with_detach (the head's keyword and default, detr3d_head_pe.py:326, :358): level 0's past-frame cameras - every
        camera after the first `cams_per_frame` = 6, hard-coded at :514-515 - reach this stage and the decoder DETACHED
        (:512-516): same values, no gradient back to the backbone through them.  A no-op for single-frame inputs.
        channels_last_out (not a keyword of the reference; inference): the returned (B, N, C, H, W) tensors hold the same values
        but their MEMORY is (B, N, H, W, C) - the decoder's gathers read such levels in place, so its per-sample slice-planar
        copy (0.27 ms of a 1.6-ms sample at 24 cameras) does not run.  Leave it off when other code takes `.view()`s of the maps."""
        super().__init__()
        if pc_range is None:
            raise ValueError('pc_range is required (the head takes it from its bbox coder)')
        self.with_detach, self.cams_per_frame = bool(with_detach), int(cams_per_frame)
        self.channels_last_out = bool(channels_last_out)
        self.cache_position_embedding = True      # inference: per camera, keyed by its matrix (757 MB at 24 cameras; False: recompute)
        self.embed_dims, self.depth_num, self.depth_start = embed_dims, depth_num, depth_start
        self.pc_range = list(pc_range)
        self.position_dim = 3 * depth_num
        self.position_encoder = nn.Sequential(nn.Conv2d(self.position_dim, embed_dims * 4, 1), nn.ReLU(),
                                              nn.Conv2d(embed_dims * 4, embed_dims, 1))
        self.adapt_pos3d = nn.Sequential(nn.Conv2d(embed_dims * 3 // 2, embed_dims * 4, 1), nn.ReLU(),
                                         nn.Conv2d(embed_dims * 4, embed_dims, 1))
        self.fpe = SELayer(embed_dims)
        # SinePositionalEncoding3D(num_feats, temperature, normalize, scale, eps, offset) has no parameters
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset
        self._sine_cache = None
        self._mask_cache = None
        self._split_cache = None
        self._pe_cache = None
        self._pe_stale = set()      # cameras whose rows of the kept embedding were not written by the last call (_forward_one_kernel)
        self._pe_moving = set()     # cameras whose matrix changed between the last two calls
        self._pe_read = {}          # stream -> event: the last read of the kept embedding on that stream
        self._pe_written = None     # event: the last (re)computation / in-place update of the kept embedding
        self._i2l = {}              # (device, request slot, R) -> [host matrices, (R, 4, 4) device tensor]: _matrices_device

    # ---- pieces -------------------------------------------------------------------------------------------------
    def padding_masks(self, img_metas, feats):
        """:525-544: 1 outside each camera's img_shape inside the padded canvas, nearest-resized to every level.
        A pure function of the shapes in img_metas: built on the device once per distinct key, then reused."""
        b, n = feats[0].shape[:2]
        pad_h, pad_w, _ = img_metas[0]['pad_shape'][0]
        key = (str(feats[0].device), pad_h, pad_w, tuple(tuple(f.shape[-2:]) for f in feats),
               tuple(tuple(tuple(s[:2]) for s in img_metas[i]['img_shape'][:n]) for i in range(b)))
        if self._mask_cache is not None and self._mask_cache[0] == key:
            return self._mask_cache[1], (pad_h, pad_w)
        full = torch.ones(b, n, pad_h, pad_w, device=feats[0].device)
        for i in range(b):
            for c in range(n):
                ih, iw, _ = img_metas[i]['img_shape'][c]
                full[i, c, :ih, :iw] = 0
        masks = [F.interpolate(full, size=f.shape[-2:]).to(torch.bool) for f in feats]
        self._mask_cache = (key, masks)
        return masks, (pad_h, pad_w)

    def _sine_embeds(self, mask):
        """The three cumulative-sum embeddings of SinePositionalEncoding3D.forward (positional_encoding.py:70-84) -
        tiny tensors, torch - and the dim_t divisors."""
        not_mask = 1 - mask.to(torch.int)
        embeds = [not_mask.cumsum(d, dtype=torch.float32) for d in (1, 2, 3)]
        if self.normalize:
            last = [embeds[0][:, -1:], embeds[1][:, :, -1:], embeds[2][:, :, :, -1:]]
            embeds = [(e + self.offset) / (l + self.eps) * self.scale for e, l in zip(embeds, last)]
        dim_t = torch.arange(self.num_feats, dtype=torch.float32)
        dim_t = (self.temperature ** (2 * (dim_t // 2) / self.num_feats)).to(mask.device)
        b, n, h, w = mask.shape
        return [e.reshape(b * n, h, w).contiguous() for e in embeds], dim_t

    def sine_embedding(self, mask):
        """SinePositionalEncoding3D.forward (positional_encoding.py:58-100): the 3 x num_feats sin / cos channels with
        ops.sine_pe3d_fwd, (B, N, 3 * num_feats, H, W)."""
        embeds, dim_t = self._sine_embeds(mask)
        b, n, h, w = mask.shape
        return ops.sine_pe3d_fwd(*embeds, dim_t).view(b, n, 3 * self.num_feats, h, w)

    def frustum_embedding(self, img_metas, masks, feats, pad_hw):
        """`position_embeding` (:427-491): returns per level ((B, N, C, H, W) embedding, (B, N, H, W) mask)."""
        b, n = feats[0].shape[:2]
        l2i = np.asarray([[np.asarray(m) for m in meta['lidar2img']] for meta in img_metas], dtype=np.float64)
        img2lidar = torch.from_numpy(np.linalg.inv(l2i)).float().view(b * n, 4, 4).to(feats[0].device)   # :459-465
        out, out_masks = [], []
        for lvl, f in enumerate(feats):
            h, w = f.shape[-2:]
            x, outside = ops.frustum_pe_input_fwd(img2lidar, (h, w), pad_hw, self.depth_num, self.depth_start,
                                                  self.pc_range)
            out.append(self.position_encoder(x).view(b, n, self.embed_dims, h, w))
            out_masks.append(masks[lvl] | outside.view(b, n, h, w))
        return out, out_masks

    def _sine_branch(self, masks, chlast=False):
        """adapt_pos3d(sine(mask)); cached while the masks (and the weights) do not change.  chlast=False: per level
        (B, N, C, H, W) through the library convolutions; chlast=True: one (B*N, S, C) channels-last tensor for all
        levels through gd4d_gemm_bf16x3_fwd (no library call at all)."""
        key = (chlast,) + tuple(p._version for p in self.adapt_pos3d.parameters())
        c = self._sine_cache
        if c is not None and c[0] == key and len(c[1]) == len(masks) and all(a is m for a, m in zip(c[1], masks)):
            return c[2]                              # same mask objects as last time (padding_masks caches by shapes)
        if chlast:
            b, n = masks[0].shape[:2]
            sizes = [m.shape[2] * m.shape[3] for m in masks]
            x = torch.empty(b * n, sum(sizes), 3 * self.num_feats, device=masks[0].device, dtype=torch.float32)
            start = 0
            for m, sz in zip(masks, sizes):
                embeds, dim_t = self._sine_embeds(m)
                ops.sine_pe3d_fwd(*embeds, dim_t, out=x, row_start=start)
                start += sz
            c0, c2 = self.adapt_pos3d[0], self.adapt_pos3d[2]
            w0 = ops.split_bf16_fwd(c0.weight.detach().view(c0.out_channels, -1).contiguous())
            w2 = ops.split_bf16_fwd(c2.weight.detach().view(c2.out_channels, -1).contiguous())
            hid = ops.gemm_bf16x3_fwd(x.view(-1, x.shape[-1]), *w0, c0.bias, relu=True)
            res = ops.gemm_bf16x3_fwd(hid, *w2, c2.bias).view(b * n, sum(sizes), -1)
        else:
            res = []
            for m in masks:
                s = self.sine_embedding(m)
                res.append(self.adapt_pos3d(s.flatten(0, 1)).view(m.shape[0], m.shape[1], self.embed_dims, *m.shape[2:]))
        if not any(p.requires_grad and torch.is_grad_enabled() for p in self.adapt_pos3d.parameters()):
            self._sine_cache = (key, list(masks), res)
        return res

    # ---- split-bf16 GEMM path ----------------------------------------------------------------------------------------
    def _split_weights(self):
        """bf16 (hi, lo) splits of the five 1x1-conv weights for gd4d_gemm_bf16x3_fwd, remade when a weight changes."""
        convs = dict(pe0=self.position_encoder[0], pe2=self.position_encoder[2], se1=self.fpe.conv_expand)
        cr = self.fpe.conv_reduce
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in convs.values()) + \
            (self.position_encoder[0].bias.data_ptr(), self.position_encoder[0].bias._version) + \
            (cr.weight.data_ptr(), cr.weight._version, cr.bias.data_ptr(), cr.bias._version)
        if self._split_cache is None or self._split_cache[0] != key:
            flat = lambda c: c.weight.detach().view(c.out_channels, c.in_channels).contiguous()      # noqa: E731
            cache = {k: ops.split_bf16_fwd(flat(c)) for k, c in convs.items()}
            pe0, pe2 = convs['pe0'], convs['pe2']
            # the position MLP as one kernel (gd4d_mlp2_bf16x3_fwd) where its shape allows: the image holds W1, b1 and W2
            cache['pe_mlp'] = ops.mlp2_image(flat(pe0), pe0.bias.detach(), flat(pe2)) \
                if ops.mlp2_supported(pe0.in_channels, pe0.out_channels, pe2.out_channels) else None
            # ... and with the frustum coordinates generated in its prologue (gd4d_mlp2_frustum_fwd; GD4D_PE_FRUSTUM=0: the two kernels)
            cache['pe_mlp_fr'] = ops.mlp2_frustum_image(flat(pe0), pe0.bias.detach(), flat(pe2)) \
                if cache['pe_mlp'] is not None and pe0.in_channels == 192 and self.depth_num == 64 and len(self.pc_range) == 6 \
                and os.environ.get('GD4D_PE_FRUSTUM', '1') != '0' else None
            # the SE gate's two convolutions (and the fuse behind them) as one kernel too: gd4d_mlp2_se_fuse_fwd
            ce = convs['se1']
            cache['se_mlp'] = ops.mlp2_image(flat(cr), cr.bias.detach(), flat(ce)) \
                if cr.in_channels == 256 and ops.mlp2_supported(cr.in_channels, cr.out_channels, ce.out_channels) else None
            cache['key'] = key
            self._split_cache = (key, cache)
        return self._split_cache[1]

    def _position_mlp(self, img2lidar, shapes, starts, s_tot, pad_hw, sw, out=None):
        """position_encoder(frustum) for the cameras of `img2lidar` (R', 4, 4) -> (R', S, C) channels-last rows (`out`: written there)."""
        r = img2lidar.shape[0]
        if sw.get('pe_mlp_fr') is not None and len(shapes) <= 4:
            # ONE kernel: the frustum inputs are generated in the MLP's prologue (no (R, S, 192) tensor)
            return ops.mlp2_frustum_fwd(img2lidar, shapes, pad_hw, self.depth_num, self.depth_start, self.pc_range, sw['pe_mlp_fr'],
                                        self.position_encoder[2].bias, out=out)
        if out is not None:
            out = out.view(r * s_tot, -1)
        x = torch.empty(r, s_tot, self.position_dim, device=img2lidar.device, dtype=torch.float32)
        for (h, w), st in zip(shapes, starts):
            ops.frustum_pe_input_fwd(img2lidar, (h, w), pad_hw, self.depth_num, self.depth_start, self.pc_range,
                                     out=x, row_start=st)
        pe0, pe2 = self.position_encoder[0], self.position_encoder[2]
        if sw.get('pe_mlp') is not None:
            # Conv1x1, ReLU, Conv1x1 in ONE kernel: the (R*S, 1024) hidden activation (3 GB at 24 cameras) stays in registers
            return ops.mlp2_bf16x3_fwd(x.view(r * s_tot, -1), sw['pe_mlp'], pe2.bias, out=out).view(r, s_tot, -1)
        hid = ops.gemm_bf16x3_fwd(x.view(r * s_tot, -1), *sw['pe0'], pe0.bias, relu=True)
        del x
        return ops.gemm_bf16x3_fwd(hid, *sw['pe2'], pe2.bias, out=out).view(r, s_tot, -1)

    def _forward_gemm(self, feats, img_metas, masks, pad_hw, sine):
        """The dense part on the bf16 matrix cores (three split products per output, fp32-class):
        frustum kernel (channels-last, all levels side by side) -> GEMM 192 -> 1024 (+ReLU) -> GEMM 1024 -> 256;
        SE gate: conv_reduce over the NCHW maps with the value_proj kernel (NCHW in, channels-last out), conv_expand
        as a GEMM with the ReLU applied on its input; one transposing pass fuses gate, embedding, sine branch, feats."""
        b, n = feats[0].shape[:2]
        r = b * n
        dev = feats[0].device
        shapes = [tuple(f.shape[-2:]) for f in feats]
        s_tot = sum(h * w for h, w in shapes)
        starts = [sum(h * w for h, w in shapes[:i]) for i in range(len(shapes))]
        mats = self._img2lidar(img_metas)                                                         # :459-465
        sw = self._split_weights()
        # The embedding of a camera is a function of its matrix, the level shapes and the MLP's weights - not of the features.  It is
        # kept per camera and recomputed for the cameras whose matrix changed: the current frame's cameras keep their calibration
        # from sample to sample (the past frames' matrices carry the ego motion and change every time).
        b2 = self.position_encoder[2].bias
        pkey = (str(dev), tuple(shapes), tuple(pad_hw), r, sw['key'], b2.data_ptr(), b2._version)
        if self.channels_last_out and sw.get('se_mlp') is not None and sw.get('pe_mlp_fr') is not None and len(feats) <= 4 \
                and os.environ.get('GD4D_PE_FUSED', '1') != '0':
            return self._forward_one_kernel(feats, mats, sw, sine, shapes, starts, s_tot, pad_hw, pkey)
        c = self._pe_cache if self.cache_position_embedding else None
        if c is not None and c[0] == pkey:
            changed = [i for i in range(r) if not np.array_equal(mats[i], c[1][i])]
        else:
            c, changed = None, list(range(r))
        cur = torch.cuda.current_stream(dev)
        capturing = torch.cuda.is_current_stream_capturing()
        if c is not None and self._pe_written is not None and not capturing:
            # whatever stream wrote the kept tensor last (a recomputation, an in-place update of some cameras): this call's kernels
            # come after it - also a call that finds every matrix unchanged and only READS the tensor (the host-side key says
            # "unchanged" from the moment the update was ENQUEUED)
            cur.wait_event(self._pe_written)
        i2l_all = self._matrices_device(mats, dev, capturing)
        if changed:
            run = changed[-1] - changed[0] + 1 == len(changed)
            if run:
                i2l = i2l_all[changed[0]:changed[-1] + 1]                # a view of the persistent buffer: a replayed graph reads what
            elif capturing:                                               # refresh_matrices() put there
                raise RuntimeError('FeaturePositionEmbedding under hipGraph capture: the cameras whose matrix changed must form one run '
                                   '(the past frames behind the current one) - or cache_position_embedding = False')
            else:
                i2l = i2l_all[torch.as_tensor(changed, device=dev)]
            if c is not None and not capturing:
                # the kept tensor is updated in place: after EVERY outstanding read of it (earlier calls, on whichever streams)
                for ev in self._pe_read.values():
                    cur.wait_event(ev)
            if c is None:
                pe = self._position_mlp(i2l, shapes, starts, s_tot, pad_hw, sw)
            elif run:
                # a run of cameras (the past frames follow the current one): the MLP writes their rows of the kept tensor
                pe = c[2]
                self._position_mlp(i2l, shapes, starts, s_tot, pad_hw, sw, out=pe[changed[0]:changed[-1] + 1])
            else:
                pe = c[2]
                pe.index_copy_(0, torch.as_tensor(changed, device=dev), self._position_mlp(i2l, shapes, starts, s_tot, pad_hw, sw))
            if self.cache_position_embedding:
                if c is None:
                    self._pe_read = {}                           # (a new tensor: the old one's readers hold it through record_stream)
                self._pe_cache = (pkey, mats, pe)
                if not capturing:
                    self._pe_written = torch.cuda.Event()
                    self._pe_written.record(cur)
            else:
                self._pe_cache = None
        else:
            pe = c[2]
        cr, ce = self.fpe.conv_reduce, self.fpe.conv_expand
        if self.channels_last_out and sw.get('se_mlp') is not None and len(feats) <= 4:
            # channels-last output: gate, sigmoid, product and the adds in the MLP kernel's epilogue (no gate tensor, no transposing pass)
            outs = ops.mlp2_se_fuse_fwd([f.flatten(0, 1).contiguous() for f in feats], sw['se_mlp'], ce.bias, pe.view(r, s_tot, -1), sine)
            self._mark_pe_read(dev)
            return [o.unflatten(0, (b, n)) for o in outs]
        g1 = ops.value_proj_fwd([f.contiguous() for f in feats], cr.weight.view(self.embed_dims, -1).contiguous(),
                                cr.bias.contiguous())                                             # (R, S, C) channels-last
        gate = ops.gemm_bf16x3_fwd(g1.view(r * s_tot, -1), *sw['se1'], ce.bias, relu_in=True)
        del g1
        pe, gate = pe.view(r, s_tot, -1), gate.view(r, s_tot, -1)
        out = []
        for lvl, (f, st) in enumerate(zip(feats, starts)):
            o = ops.se_fuse_chlast_fwd(f.flatten(0, 1).contiguous(), gate, pe, sine, st, out_channels_last=self.channels_last_out)
            out.append(o.unflatten(0, (b, n)) if self.channels_last_out else o.view(f.shape))
        self._mark_pe_read(dev)
        return out

    @staticmethod
    def _runs(idx):
        """Sorted camera indices -> the contiguous runs [a, e) they form."""
        runs = []
        for i in idx:
            if runs and runs[-1][1] == i:
                runs[-1][1] = i + 1
            else:
                runs.append([i, i + 1])
        return [tuple(x) for x in runs]

    def _forward_one_kernel(self, feats, mats, sw, sine, shapes, starts, s_tot, pad_hw, pkey):
        """The channels-last route with both MLPs in ONE kernel for the cameras that keep moving (gd4d_mlp2_pe_se_fwd: the embedding of
        such a camera is used once - it stays in the kernel's registers and is never stored).  Per camera:
          matrix unchanged, rows valid           -> kept: SE gate + fuse on the kept rows (gd4d_mlp2_se_fuse_fwd)
          changed now AND between the last two calls (the past frames of the temporal pattern: their matrices carry the ego motion)
                                                 -> one kernel, nothing stored; its kept rows are stale from then on
          changed for the first time, or stale and no longer moving
                                                 -> recomputed INTO the kept tensor (gd4d_mlp2_frustum_fwd), then as a kept camera
        cache_position_embedding = False: every camera through the one kernel, no (R, S, 256) tensor at all.  The same values bit for
        bit whichever way a camera goes (tested)."""
        b, n = feats[0].shape[:2]
        r = b * n
        dev = feats[0].device
        cur = torch.cuda.current_stream(dev)
        capturing = torch.cuda.is_current_stream_capturing()
        keep = self.cache_position_embedding
        c = self._pe_cache if keep else None
        if c is not None and c[0] != pkey:
            c = None
        if c is None:
            self._pe_stale, self._pe_moving = set(), set()
            really, need = set(), set(range(r))
        else:
            really = {i for i in range(r) if not np.array_equal(mats[i], c[1][i])}
            need = really | self._pe_stale
        fused = sorted(need) if not keep else sorted(i for i in need if i in really and i in self._pe_moving)
        store = sorted(need - set(fused))
        if c is not None and self._pe_written is not None and not capturing:
            cur.wait_event(self._pe_written)                     # (see _forward_gemm)
        i2l_all = self._matrices_device(mats, dev, capturing)
        pe = None if c is None else c[2]
        if store:
            if pe is None:
                pe = torch.empty(r, s_tot, self.embed_dims, device=dev, dtype=torch.float32)
                self._pe_read = {}
            elif not capturing:
                for ev in self._pe_read.values():                # in place: after every outstanding read of the kept tensor
                    cur.wait_event(ev)
            for a, e in self._runs(store):
                self._position_mlp(i2l_all[a:e], shapes, starts, s_tot, pad_hw, sw, out=pe[a:e])
            if not capturing:
                self._pe_written = torch.cuda.Event()
                self._pe_written.record(cur)
        if keep:
            self._pe_cache = (pkey, mats, pe)
            self._pe_moving, self._pe_stale = really, set(fused)
        else:
            self._pe_cache = None
        ce, b2 = self.fpe.conv_expand, self.position_encoder[2].bias
        flat = [f.flatten(0, 1).contiguous() for f in feats]
        outs = [torch.empty(r, h, w, self.embed_dims, device=dev, dtype=torch.float32) for h, w in shapes]
        sine = sine.view(r, s_tot, -1)
        for a, e in self._runs(fused):
            ops.mlp2_pe_se_fwd(i2l_all[a:e], [f[a:e] for f in flat], pad_hw, self.depth_num, self.depth_start, self.pc_range,
                               sw['pe_mlp_fr'], b2, sw['se_mlp'], ce.bias, sine[a:e], outs=[o[a:e] for o in outs])
        kept = sorted(set(range(r)) - set(fused))
        for a, e in self._runs(kept):
            ops.mlp2_se_fuse_fwd([f[a:e] for f in flat], sw['se_mlp'], ce.bias, pe[a:e], sine[a:e], outs=[o[a:e] for o in outs])
        if kept:
            self._mark_pe_read(dev)
        return [o.permute(0, 3, 1, 2).unflatten(0, (b, n)) for o in outs]

    def _matrices_device(self, mats, dev, capturing=False):
        """The (R, 4, 4) img2lidar matrices on the device: ONE persistent buffer per (device, request slot, R), refreshed in place when
        the host values differ - so that a hipGraph captured over this module keeps a valid address and a replay for a new sample
        only needs refresh_matrices() outside the graph (the pattern of functional.lidar2img_device).  Under capture the buffer
        must already hold these matrices (call the module, or refresh_matrices, eagerly first)."""
        key = (str(dev), Fn.slot_key(dev), mats.shape[0])
        ent = self._i2l.get(key)
        if ent is None or not np.array_equal(ent[0], mats):
            if capturing:
                if ent is None:
                    raise RuntimeError('FeaturePositionEmbedding under hipGraph capture: call the module (or refresh_matrices) once eagerly '
                                       'with these img_metas first - their matrices are not on the device yet')
                raise RuntimeError('FeaturePositionEmbedding under hipGraph capture: img_metas changed since the last eager call; '
                                   'refresh_matrices(img_metas) first')
            src = torch.from_numpy(mats).view(-1, 4, 4)
            if ent is None:
                ent = self._i2l[key] = [mats.copy(), src.to(dev)]
            else:
                ent[1].copy_(src)
                ent[0] = mats.copy()
        return ent[1]

    @staticmethod
    def _img2lidar(img_metas):
        l2i = np.asarray([[np.asarray(m) for m in meta['lidar2img']] for meta in img_metas], dtype=np.float64)
        return np.ascontiguousarray(np.linalg.inv(l2i).astype(np.float32).reshape(-1, 16))       # :459-465

    def refresh_matrices(self, img_metas, device):
        """For the owner of a hipGraph captured over this module: put the new sample's img2lidar matrices into the persistent device
        buffer the graph reads (outside the graph, before the replay).  The graph recomputes the cameras it recomputed when it was
        captured (e.g. the past frames' 18 of 24) from them; the cameras it kept must not have moved."""
        return self._matrices_device(self._img2lidar(img_metas), torch.device(device))

    def _mark_pe_read(self, dev):
        """This call's kernels have read the kept embedding on the current stream: an in-place update waits for them (per stream:
        several callers may read it concurrently), and the allocator does not hand its memory out again before they are done
        (a key change drops the tensor while another stream may still be reading it)."""
        if self.cache_position_embedding and self._pe_cache is not None and not torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event()
            ev.record(cur)
            self._pe_read[cur.cuda_stream] = ev
            self._pe_cache[2].record_stream(cur)

    # ---- training ---------------------------------------------------------------------------------------------------
    def _forward_autograd(self, feats, img_metas):
        """The stage with autograd on (the head trains through it: gradients reach the backbone's feature maps, the two
        position MLPs and the SE gate, detr3d_head_pe.py:546-557).  The geometry needs no gradient and stays on the HIP
        kernels (frustum coordinates -> conv input, sine / cosine expansion, padding masks); the differentiable part runs on
        _HeadPEFunction (the library's GEMMs forward and backward); shapes it does not take raise; with GD4D_TORCH_OPS=1 the 1x1
        convolutions, the gate and the adds are torch ops and autograd differentiates them."""
        with torch.no_grad():
            masks, pad_hw = self.padding_masks(img_metas, feats)
            b, n = feats[0].shape[:2]
            l2i = np.asarray([[np.asarray(m) for m in meta['lidar2img']] for meta in img_metas], dtype=np.float64)
            img2lidar = torch.from_numpy(np.linalg.inv(l2i)).float().view(b * n, 4, 4).to(feats[0].device)   # :459-465
        if not Fn.torch_ops_route(self._route_name(feats), self._gemm_ok(feats), module=self):
            return self._forward_hip_train(feats, masks, pad_hw, img2lidar)
        with torch.no_grad():
            xs = [ops.frustum_pe_input_fwd(img2lidar, tuple(f.shape[-2:]), pad_hw, self.depth_num, self.depth_start,
                                           self.pc_range)[0] for f in feats]
            sines = [self.sine_embedding(m) for m in masks]
        out = []
        for f, x, s in zip(feats, xs, sines):
            pe = self.position_encoder(x).view(f.shape)
            gate = self.fpe.gate_logits(f.flatten(0, 1)).view(f.shape)
            sine = self.adapt_pos3d(s.flatten(0, 1)).view(f.shape)
            out.append(f + (pe * torch.sigmoid(gate) + sine))
        return out

    def _route_name(self, feats):
        return (f'Detr3DHeadPE position embedding with embed_dims = {self.embed_dims}, position_dim = {self.position_dim}, '
                f'{feats[0].shape[2]} channels, num_feats = {self.num_feats} (kernels: 256 / a multiple of 32 / 256 / 3 num_feats a multiple of 32)')

    def _gemm_ok(self, feats):
        return self.embed_dims == 256 and self.position_dim % 32 == 0 and feats[0].shape[2] == 256 and \
            (3 * self.num_feats) % 32 == 0

    def _forward_hip_train(self, feats, masks, pad_hw, img2lidar):
        """Training on the library's own kernels (_HeadPEFunction): geometry as in _forward_gemm, channels-last, all
        levels side by side."""
        b, n = feats[0].shape[:2]
        r = b * n
        dev = feats[0].device
        shapes = [tuple(f.shape[-2:]) for f in feats]
        sizes = [h * w for h, w in shapes]
        starts = [sum(sizes[:i]) for i in range(len(sizes))]
        with torch.no_grad():
            x = torch.empty(r, sum(sizes), self.position_dim, device=dev, dtype=torch.float32)
            xs = torch.empty(r, sum(sizes), 3 * self.num_feats, device=dev, dtype=torch.float32)
            for (h, w), st, m in zip(shapes, starts, masks):
                ops.frustum_pe_input_fwd(img2lidar, (h, w), pad_hw, self.depth_num, self.depth_start, self.pc_range,
                                         out=x, row_start=st)
                embeds, dim_t = self._sine_embeds(m)
                ops.sine_pe3d_fwd(*embeds, dim_t, out=xs, row_start=st)
        pe0, pe2, a0, a2 = self.position_encoder[0], self.position_encoder[2], self.adapt_pos3d[0], self.adapt_pos3d[2]
        cr, ce = self.fpe.conv_reduce, self.fpe.conv_expand
        out = _HeadPEFunction.apply(x, xs, starts, len(feats), *[f.flatten(0, 1) for f in feats],
                                    pe0.weight, pe0.bias, pe2.weight, pe2.bias, a0.weight, a0.bias, a2.weight, a2.bias,
                                    cr.weight, cr.bias, ce.weight, ce.bias)
        return [o.view(f.shape) for o, f in zip(out, feats)]

    # ---- the stage ------------------------------------------------------------------------------------------------
    def train(self, mode=True):
        """A change of mode drops what inference keeps (the per-camera embedding, the weight splits / MLP images): training updates
        the weights - inside a replayed hipGraph without bumping the version counters the caches are keyed by."""
        if mode != self.training:
            self._pe_cache, self._pe_read, self._pe_written, self._split_cache = None, {}, None, None
            self._pe_stale, self._pe_moving = set(), set()
        return super().train(mode)

    def forward(self, mlvl_feats, img_metas):
        """mlvl_feats: list of (B, N, C, H_l, W_l) fp32 GPU tensors; returns the list with the position embedding
        added (:546-557).  With autograd on: _forward_autograd.  GD4D_TORCH_OPS=1 runs the 1x1 convolutions as torch ops (MIOpen)
        instead of gd4d_gemm_bf16x3_fwd; shapes the GEMM route does not take raise without it."""
        feats = list(mlvl_feats)
        Fn.require_gpu(feats[0], 'mlvl_feats')
        if self.with_detach and feats[0].shape[1] > self.cams_per_frame and feats[0].requires_grad and torch.is_grad_enabled():
            k = self.cams_per_frame                           # :512-516 (level 0 only, as the reference)
            feats[0] = torch.cat([feats[0][:, :k], feats[0][:, k:].detach()], 1)
        if Fn.wants_grad(self, *feats):
            return self._forward_autograd(feats, img_metas)
        with torch.no_grad():
            masks, pad_hw = self.padding_masks(img_metas, feats)
            if not Fn.torch_ops_route(self._route_name(feats), self._gemm_ok(feats), module=self):
                return self._forward_gemm(feats, img_metas, masks, pad_hw, self._sine_branch(masks, chlast=True))
            sine = self._sine_branch(masks)
            coords_pe, _ = self.frustum_embedding(img_metas, masks, feats, pad_hw)
            out = []
            for lvl, f in enumerate(feats):
                gate = self.fpe.gate_logits(f.flatten(0, 1)).view(f.shape)
                out.append(ops.se_fuse_fwd(f.contiguous(), gate.contiguous(), coords_pe[lvl].contiguous(),
                                           sine[lvl].contiguous()))
        return out

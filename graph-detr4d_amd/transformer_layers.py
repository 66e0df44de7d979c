"""Decoder-layer plumbing around the attention modules.

In the reference these classes are third-party (mmcv 1.x `MultiheadAttention`, `FFN`,
`BaseTransformerLayer`, `TransformerLayerSequence`; mmdet 2.x `DetrTransformerDecoderLayer`),
selected by the config at projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:71-89.
mmcv/mmdet are not installable on the ROCm box, so the package carries stand-ins with the same
type names, constructor keywords, calling convention and state-dict keys
(`attentions.{i}...`, `ffns.0.layers.0.0 / layers.1`, `norms.{i}`), registered in
the registry shim (and into real mmcv when it is importable).
"""
import copy
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from . import ops
from .registry import (ATTENTION, FEEDFORWARD_NETWORK, TRANSFORMER_LAYER, build_attention,
                       build_transformer_layer)


class _PackedAttnParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's state-dict keys
    (in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias)."""

    def __init__(self, embed_dims):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dims, embed_dims))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dims))
        self.out_proj = nn.Linear(embed_dims, embed_dims)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)


@ATTENTION.register_module()
class MultiheadAttention(nn.Module):
    """Decoder self-attention with mmcv 1.x `MultiheadAttention` semantics: q and k receive the
    positional encodings, v does not; returns identity + dropout(attn(q, k, v))."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=None,
                 init_cfg=None, batch_first=False, dropout=None, **kwargs):
        super().__init__()
        if dropout is not None:                    # deprecated alias used by the reference configs
            attn_drop = dropout
            dropout_layer = dict(type='Dropout', drop_prob=dropout)
        if embed_dims % num_heads:
            raise ValueError('embed_dims must be divisible by num_heads')
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn_drop = attn_drop
        self.attn = _PackedAttnParams(embed_dims)
        self.proj_drop = nn.Dropout(proj_drop)
        p = (dropout_layer or {}).get('drop_prob', 0.)
        self.dropout_layer = nn.Dropout(p) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        Fn.require_gpu(query, 'query')
        if key_padding_mask is not None:
            raise NotImplementedError('key_padding_mask is never set on the decoder self-attention path')
        grad = Fn.wants_grad(self, query, key, value, query_pos)
        eval_mode = not self.training and not grad
        if (eval_mode and not self.batch_first and key is query and value is query
                and query_pos is not None and key_pos is query_pos and identity.shape == query.shape):
            return self._packed_self_attention(query, query_pos, attn_mask, identity, Fn.take_fused_norm(kwargs))
        q_in = query if query_pos is None else query + query_pos
        k_in = q_in if (key is query and key_pos is query_pos) else (key if key_pos is None else key + key_pos)
        if self.batch_first:
            q_in, k_in, value = (t.transpose(0, 1) for t in (q_in, k_in, value))
        out = self._attention_autograd(q_in, k_in, value, attn_mask) if grad else \
            self._attention(q_in, k_in, value, attn_mask)
        if self.batch_first:
            out = out.transpose(0, 1)
        out = self.dropout_layer(self.proj_drop(out))
        return Fn.residual_norm_autograd(kwargs, identity, out) if grad else identity + out

    def _packed_self_attention(self, query, query_pos, attn_mask, identity, fused=None):
        """The decoder's case (q = k = query + query_pos, v = query): one in-projection GEMM with the
        positional add fused into its load, the attention core, and the out-projection with the
        residual fused into its epilogue - 3 launches."""
        c = self.embed_dims
        qkv = Fn.linear(query, self.attn.in_proj_weight, self.attn.in_proj_bias, x2=query_pos, n_split=2 * c)
        q, k, v = qkv.split(c, dim=-1)
        o = ops.mha_core_fwd(q, k, v, self.num_heads, attn_mask)
        if fused is not None and Fn.rowblock_ok(o, self.attn.out_proj.weight, fused['norm']):
            fused['done'] = True
            return Fn.linear_norm(o, self.attn.out_proj.weight, self.attn.out_proj.bias, fused['norm'], r1=identity)
        return Fn.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias, r1=identity)

    def _attention(self, q_in, k_in, v_in, attn_mask):
        """General (L, B, C) x3 -> (L, B, C): three in-projections, attention core, out_proj."""
        c = self.embed_dims
        w, bias = self.attn.in_proj_weight, self.attn.in_proj_bias
        qh = Fn.linear(q_in, w[:c], bias[:c])
        kh = Fn.linear(k_in, w[c:2 * c], bias[c:2 * c])
        vh = Fn.linear(v_in, w[2 * c:], bias[2 * c:])
        o = ops.mha_core_fwd(qh, kh, vh, self.num_heads, attn_mask)
        return Fn.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias)


    # training path (torch ops so autograd flows); attention-weight dropout as nn.MultiheadAttention applies it
    def _attention_autograd(self, q_in, k_in, v_in, attn_mask):
        c, h = self.embed_dims, self.num_heads
        d = c // h
        w, bias = self.attn.in_proj_weight, self.attn.in_proj_bias
        lq, b, _ = q_in.shape
        lk = k_in.shape[0]
        qk = qh_lin = kh_lin = None
        # (with dist.FlatGradAllReducer.bind(fuse_weight_grads=True) the gradient of a row slice of the packed in-projection is
        # added straight to the same rows of the parameter's view of the flat buffer)
        rows = lambda lo, hi: (Fn.main_grad(w, (lo, hi)), Fn.main_grad(bias, (lo, hi)))
        if k_in is q_in:                                  # decoder self-attention: one GEMM for q and k
            qk = Fn.linear_autograd(q_in.contiguous(), w[:2 * c], bias[:2 * c], main=rows(0, 2 * c))
            qh, kh = qk[..., :c], qk[..., c:]
        else:
            qh = qh_lin = Fn.linear_autograd(q_in.contiguous(), w[:c], bias[:c], main=rows(0, c))
            kh = kh_lin = Fn.linear_autograd(k_in.contiguous(), w[c:2 * c], bias[c:2 * c], main=rows(c, 2 * c))
        vh_lin = Fn.linear_autograd(v_in.contiguous(), w[2 * c:], bias[2 * c:], main=rows(2 * c, 3 * c))
        drop = float(self.attn_drop) if self.training else 0.
        if not Fn.torch_ops_route(f'MultiheadAttention training with head dim {d}, mask dim {None if attn_mask is None else attn_mask.dim()}',
                                  d == 32 and q_in.dtype == torch.float32 and (attn_mask is None or attn_mask.dim() == 2)
                                  and b * h * lq * lk < 2 ** 32, module=self):
            # the attention core with autograd on the HIP kernels; in train mode they drop probabilities as F.dropout does
            # inside nn.MultiheadAttention (same distribution, a different generator: gd4d_mha_dropout.h)
            from .autograd import MhaCoreFunction, MhaCorePackedFunction
            if k_in is q_in:
                o = MhaCorePackedFunction.apply(qk, vh_lin, attn_mask, h, drop)
            else:
                o = MhaCoreFunction.apply(qh_lin, kh_lin, vh_lin, attn_mask, h, drop)
            return Fn.linear_autograd(o, self.attn.out_proj.weight, self.attn.out_proj.bias)
        qh = qh.reshape(lq, b * h, d).transpose(0, 1)
        kh = kh.reshape(lk, b * h, d).transpose(0, 1)
        vh = vh_lin.reshape(lk, b * h, d).transpose(0, 1)
        scores = torch.bmm(qh * (1.0 / math.sqrt(d)), kh.transpose(1, 2))
        if attn_mask is not None:
            scores = scores.masked_fill(attn_mask, float('-inf')) if attn_mask.dtype == torch.bool \
                else scores + attn_mask
        attn = F.dropout(scores.softmax(-1), p=self.attn_drop, training=self.training)
        o = torch.bmm(attn, vh).transpose(0, 1).reshape(lq, b, c)
        return Fn.linear_autograd(o, self.attn.out_proj.weight, self.attn.out_proj.bias)


@FEEDFORWARD_NETWORK.register_module()
class FFN(nn.Module):
    """mmcv 1.x FFN: x + W2 relu(W1 x) (dropouts are identity in eval)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs >= 2
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(cin, feedforward_channels), nn.ReLU(inplace=True),
                                        nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        p = (dropout_layer or {}).get('drop_prob', 0.)
        self.dropout_layer = nn.Dropout(p) if dropout_layer else nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None, **kwargs):
        if not self.training and len(self.layers) == 3 and x.is_cuda and not Fn.wants_grad(self, x, identity):
            hdn = Fn.linear(x, self.layers[0][0].weight, self.layers[0][0].bias, relu=True)
            res = (x if identity is None else identity) if self.add_identity else None
            fused = Fn.take_fused_norm(kwargs)
            if fused is not None and Fn.rowblock_ok(hdn, self.layers[1].weight, fused['norm']):
                fused['done'] = True
                return Fn.linear_norm(hdn, self.layers[1].weight, self.layers[1].bias, fused['norm'], r1=res)
            return Fn.linear(hdn, self.layers[1].weight, self.layers[1].bias, r1=res)
        out = Fn.sequential_autograd(self.layers, x) if x.is_cuda else self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return Fn.residual_norm_autograd(kwargs, identity, self.dropout_layer(out)) if x.is_cuda else identity + self.dropout_layer(out)


@TRANSFORMER_LAYER.register_module()
class BaseTransformerLayer(nn.Module):
    """mmcv 1.x BaseTransformerLayer calling convention: attention modules are
    called as attn(query, key, value, identity_or_None, query_pos=, key_pos=, attn_mask=,
    key_padding_mask=, **kwargs); `batch_first` is injected into every attention config."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None, norm_cfg=dict(type='LN'),
                 init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        ffn_cfgs = dict(ffn_cfgs) if ffn_cfgs else dict(type='FFN', embed_dims=256,
                                                         feedforward_channels=1024, num_fcs=2,
                                                         ffn_drop=0.)
        for old, new in (('feedforward_channels', 'feedforward_channels'),
                         ('ffn_dropout', 'ffn_drop'), ('ffn_num_fcs', 'num_fcs')):
            if old in kwargs:                        # deprecated top-level keywords (mmdet configs)
                ffn_cfgs[new] = kwargs[old]
        assert set(operation_order) <= {'self_attn', 'norm', 'ffn', 'cross_attn'}
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        assert num_attn == len(attn_cfgs)
        self.batch_first = batch_first
        self.operation_order = operation_order
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = nn.ModuleList()
        idx = 0
        for op in operation_order:
            if op in ('self_attn', 'cross_attn'):
                cfg = copy.deepcopy(attn_cfgs[idx])
                cfg['batch_first'] = batch_first
                attn = build_attention(cfg)
                attn.operation_name = op
                self.attentions.append(attn)
                idx += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        for _ in range(operation_order.count('ffn')):
            cfg = dict(ffn_cfgs)
            cfg.pop('type', None)
            cfg.setdefault('embed_dims', self.embed_dims)
            self.ffns.append(FFN(**cfg))
        self.norms = nn.ModuleList(nn.LayerNorm(self.embed_dims)
                                   for _ in range(operation_order.count('norm')))

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        norm_i = attn_i = ffn_i = 0
        identity = query
        skip_norm = False
        if attn_masks is None:
            attn_masks = [None] * len(self.attentions)
        elif torch.is_tensor(attn_masks):
            attn_masks = [attn_masks for _ in range(len(self.attentions))]
        else:
            assert len(attn_masks) == len(self.attentions)
        order = self.operation_order
        for oi, op in enumerate(order):
            # a LayerNorm right after an attention / FFN can run in that module's last GEMM (Fn.linear_norm)
            holder = None
            if op != 'norm' and oi + 1 < len(order) and order[oi + 1] == 'norm' and query.is_cuda:
                if not self.training and not Fn.wants_grad(self.norms[norm_i], query):
                    holder = {'norm': self.norms[norm_i], 'done': False}
                elif not self.pre_norm:
                    # training path: the module may hand its residual sum to the LayerNorm kernel (Fn.residual_norm_autograd)
                    holder = {'norm': self.norms[norm_i], 'done': False, 'autograd': True}
            fuse = {Fn.NORM_KEY: holder} if holder is not None else {}
            if op == 'self_attn':
                query = self.attentions[attn_i](
                    query, query, query, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=attn_masks[attn_i],
                    key_padding_mask=query_key_padding_mask, **fuse, **kwargs)
                attn_i += 1
                identity = query
            elif op == 'norm':
                if skip_norm:
                    skip_norm = False
                else:
                    norm = self.norms[norm_i]
                    if not query.is_cuda:
                        query = norm(query)
                    elif Fn.wants_grad(norm, query):
                        query = Fn.layer_norm_autograd(query, norm)
                    else:
                        query = Fn.layer_norm(query, norm)
                norm_i += 1
            elif op == 'cross_attn':
                query = self.attentions[attn_i](
                    query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=attn_masks[attn_i],
                    key_padding_mask=key_padding_mask, **fuse, **kwargs)
                attn_i += 1
                identity = query
            elif op == 'ffn':
                query = self.ffns[ffn_i](query, identity if self.pre_norm else None, **fuse)
                ffn_i += 1
            if holder is not None and holder['done']:
                skip_norm = True                     # the module already applied the LayerNorm that follows
        return query


@TRANSFORMER_LAYER.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """mmdet 2.x DetrTransformerDecoderLayer (post-norm: self_attn, norm, cross_attn, norm, ffn, norm)."""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'), ffn_num_fcs=2,
                 **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, norm_cfg=norm_cfg,
                         ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6
        assert set(operation_order) == {'self_attn', 'norm', 'cross_attn', 'ffn'}


class TransformerLayerSequence(nn.Module):
    """mmcv 1.x TransformerLayerSequence: `num_layers` copies of `transformerlayers`."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__()
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        assert isinstance(transformerlayers, list) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList(build_transformer_layer(cfg) for cfg in transformerlayers)
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

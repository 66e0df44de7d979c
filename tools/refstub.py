"""Import the reference's hot-path files UNMODIFIED under a stub of mmcv/mmdet/mmdet3d.

Build-container only (needs /root/reference). Nothing here ships to the GPU box as
product code; it exists so `tools/gen_golden.py` can capture golden vectors from the real
reference implementation.

What is stubbed (the reference has no vendored copy of any of these):
  * mmcv.cnn.{xavier_init, constant_init}
  * mmcv.cnn.bricks.registry.{ATTENTION, TRANSFORMER_LAYER_SEQUENCE, ...}  (decorator registries)
  * mmcv.runner.base_module.BaseModule            (= nn.Module + init_cfg)
  * mmcv.ops.multi_scale_deform_attn.*            (third-party MSDA; restated from its
                                                   published semantics with F.grid_sample)
  * mmcv.cnn.bricks.transformer.{MultiheadAttention, FFN, BaseTransformerLayer,
                                 TransformerLayerSequence, build_transformer_layer_sequence}
  * mmdet.models.utils.builder.TRANSFORMER, mmdet DetrTransformerDecoderLayer
  * mmdet3d.core.bbox.structures.utils.rotation_3d_in_axis (imported, never called)

The reference's CPU branch of Deform3DCrossAttn.forward references an undefined local
(`sampling_locations`, deform3d_cross_attn.py:308-309).  We give the module a global of
that name (None) and let the MSDA stub pull the caller's `reference_points_cam` local out
of the calling frame; the reference file itself runs byte-for-byte unmodified.
"""
import copy
import importlib.util
import inspect
import math
import os
import sys
import types
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get('GD4D_REFERENCE_ROOT', '/root/reference')
_UTILS = 'projects/mmdet3d_plugin/models/utils'

# last frame-locals captured by the MSDA stub (intermediates of the reference forward)
CAPTURED = {}


class Registry:
    """Minimal mmcv-style registry: register_module() decorator + build(cfg)."""

    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _do(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _do(module)
        return _do

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        typ = cfg.pop('type')
        cls = self.module_dict[typ] if isinstance(typ, str) else typ
        return cls(**cfg)


ATTENTION = Registry('attention')
FEEDFORWARD_NETWORK = Registry('ffn')
TRANSFORMER_LAYER = Registry('transformerlayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
TRANSFORMER = Registry('Transformer')
POSITIONAL_ENCODING = Registry('position encoding')


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


# --------------------------------------------------------------------------------------
# third-party MSDA (mmcv.ops.multi_scale_deform_attn), restated from its public semantics
# --------------------------------------------------------------------------------------
def msda_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """out[b,q,h*D:(h+1)*D] = sum_{l,p} w[b,q,h,l,p] * bilinear(value_l[b,:,h,:], loc[b,q,h,l,p])

    value (bs, sum(HW), heads, D); sampling_locations (bs, Q, heads, L, P, 2) in [0,1];
    attention_weights (bs, Q, heads, L*P) or (bs, Q, heads, L, P).  Sampling is
    F.grid_sample(2*loc-1, bilinear, zeros padding, align_corners=False).
    """
    bs, _, heads, dim = value.shape
    _, nq, _, nl, npnt, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in value_spatial_shapes]
    per_level = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = per_level[lvl].flatten(2).transpose(1, 2).reshape(bs * heads, dim, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)          # (bs*heads, Q, P, 2)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros',
                                     align_corners=False))               # (bs*heads, D, Q, P)
    wts = attention_weights.reshape(bs, nq, heads, nl * npnt).transpose(1, 2)
    wts = wts.reshape(bs * heads, 1, nq, nl * npnt)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * wts).sum(-1)
    return out.view(bs, heads * dim, nq).transpose(1, 2).contiguous()


def _msda_stub(value, spatial_shapes, sampling_locations, attention_weights):
    """Called from the reference's (dead) CPU branch with sampling_locations=None."""
    caller = inspect.currentframe().f_back
    loc = caller.f_locals
    CAPTURED.clear()
    for k, v in loc.items():
        if torch.is_tensor(v):
            CAPTURED[k] = v.detach().clone()
    if 'reference_points_neighbor_cam' in loc and 'attention_weights_neighbor' in loc:
        # Second call site of Deform3DCrossAttnMP.forward (deform3d_cross_attn_multi_point.py:408-417).  Its dead CPU
        # branch assigns the result to `output` (clobbering the first pass) with the FIRST pass's arguments and then
        # reads an `output_neighbor` it never defined.  What the CUDA branch computes is unambiguous (:403-406); the
        # stub evaluates that from the caller's locals, plants it into the caller's `output_neighbor` local and hands
        # the first-pass `output` back so the clobbering assignment is a no-op.  The reference file runs unmodified.
        import ctypes
        out_n = msda_pytorch(value, spatial_shapes, loc['reference_points_neighbor_cam'],
                             loc['attention_weights_neighbor'])
        CAPTURED['msda_output_neighbor'] = out_n.detach().clone()
        caller.f_locals['output_neighbor'] = out_n
        ctypes.pythonapi.PyFrame_LocalsToFast(ctypes.py_object(caller), ctypes.c_int(0))
        return loc['output']
    if sampling_locations is None:
        sampling_locations = loc['reference_points_cam']
    out = msda_pytorch(value, spatial_shapes, sampling_locations, attention_weights)
    CAPTURED['msda_output'] = out.detach().clone()
    return out


class _MSDAFunction:
    """Stand-in for the CUDA autograd Function (reference only calls .apply on CUDA)."""

    @staticmethod
    def apply(value, spatial_shapes, level_start_index, sampling_locations,
              attention_weights, im2col_step):
        return msda_pytorch(value, spatial_shapes, sampling_locations, attention_weights)


class MultiScaleDeformableAttention(BaseModule):
    """Name only: the reference isinstance()-checks against it in init_weights."""


# --------------------------------------------------------------------------------------
# third-party mmcv transformer bricks, restated from their public semantics (mmcv 1.x)
# --------------------------------------------------------------------------------------
class MultiheadAttention(BaseModule):
    """mmcv 1.x wrapper around nn.MultiheadAttention: q,k get the positional encodings,
    v does not; returns identity + dropout(attn_out)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=None,
                 init_cfg=None, batch_first=False, dropout=None, **kwargs):
        super().__init__(init_cfg)
        if dropout is not None:           # deprecated alias used by the reference configs
            attn_drop = dropout
            dropout_layer = dict(type='Dropout', drop_prob=dropout)
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        p = dropout_layer.get('drop_prob', 0.) if dropout_layer else 0.
        self.dropout_layer = nn.Dropout(p) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_pos=None, attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


class FFN(BaseModule):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(cin, feedforward_channels),
                                        nn.ReLU(inplace=True), nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


class BaseTransformerLayer(BaseModule):
    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None,
                 norm_cfg=dict(type='LN'), init_cfg=None, batch_first=False, **kwargs):
        super().__init__(init_cfg)
        ffn_cfgs = dict(ffn_cfgs or dict(type='FFN', embed_dims=256, feedforward_channels=1024,
                                         num_fcs=2, ffn_drop=0.))
        for old, new in (('feedforward_channels', 'feedforward_channels'),
                         ('ffn_dropout', 'ffn_drop'), ('ffn_num_fcs', 'num_fcs')):
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        self.operation_order = operation_order
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        idx = 0
        for op in operation_order:
            if op in ('self_attn', 'cross_attn'):
                cfg = copy.deepcopy(attn_cfgs[idx])
                cfg['batch_first'] = batch_first
                attn = ATTENTION.build(cfg)
                attn.operation_name = op
                self.attentions.append(attn)
                idx += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = ModuleList()
        for _ in range(operation_order.count('ffn')):
            cfg = dict(ffn_cfgs)
            cfg.setdefault('embed_dims', self.embed_dims)
            cfg.pop('type', None)
            self.ffns.append(FFN(**cfg))
        self.norms = ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(nn.LayerNorm(self.embed_dims))

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        ni = ai = fi = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None] * len(self.attentions)
        elif torch.is_tensor(attn_masks):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(len(self.attentions))]
        for op in self.operation_order:
            if op == 'self_attn':
                query = self.attentions[ai](query, query, query,
                                            identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=query_pos,
                                            attn_mask=attn_masks[ai],
                                            key_padding_mask=query_key_padding_mask, **kwargs)
                ai += 1
                identity = query
            elif op == 'norm':
                query = self.norms[ni](query)
                ni += 1
            elif op == 'cross_attn':
                query = self.attentions[ai](query, key, value,
                                            identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=key_pos,
                                            attn_mask=attn_masks[ai],
                                            key_padding_mask=key_padding_mask, **kwargs)
                ai += 1
                identity = query
            elif op == 'ffn':
                query = self.ffns[fi](query, identity if self.pre_norm else None)
                fi += 1
        return query


class DetrTransformerDecoderLayer(BaseTransformerLayer):
    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN'),
                 ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order,
                         ffn_num_fcs=ffn_num_fcs, **kwargs)


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        self.num_layers = num_layers
        self.layers = ModuleList()
        for i in range(num_layers):
            self.layers.append(TRANSFORMER_LAYER.build(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


def build_transformer_layer_sequence(cfg, default_args=None):
    return TRANSFORMER_LAYER_SEQUENCE.build(cfg, **(default_args or {}))


ATTENTION.register_module(module=MultiheadAttention)
TRANSFORMER_LAYER.register_module(module=BaseTransformerLayer)
TRANSFORMER_LAYER.register_module(module=DetrTransformerDecoderLayer)


# --------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_gd4d_stub', False):
        return
    _mod('mmcv', _gd4d_stub=True)
    _mod('mmcv.cnn', xavier_init=xavier_init, constant_init=constant_init)
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.registry', ATTENTION=ATTENTION,
         TRANSFORMER_LAYER_SEQUENCE=TRANSFORMER_LAYER_SEQUENCE,
         TRANSFORMER_LAYER=TRANSFORMER_LAYER, FEEDFORWARD_NETWORK=FEEDFORWARD_NETWORK,
         POSITIONAL_ENCODING=POSITIONAL_ENCODING)
    _mod('mmcv.cnn.bricks.transformer',
         MultiScaleDeformableAttention=MultiScaleDeformableAttention,
         TransformerLayerSequence=TransformerLayerSequence,
         build_transformer_layer_sequence=build_transformer_layer_sequence,
         BaseTransformerLayer=BaseTransformerLayer, MultiheadAttention=MultiheadAttention,
         FFN=FFN)
    _mod('mmcv.runner')
    _mod('mmcv.runner.base_module', BaseModule=BaseModule, ModuleList=ModuleList)
    _mod('mmcv.ops')
    _mod('mmcv.ops.multi_scale_deform_attn',
         MultiScaleDeformableAttnFunction=_MSDAFunction,
         multi_scale_deformable_attn_pytorch=_msda_stub)
    _mod('mmdet')
    _mod('mmdet.models')
    _mod('mmdet.models.utils')
    _mod('mmdet.models.utils.builder', TRANSFORMER=TRANSFORMER)
    # (h_detr3d_transformer.py:26 imports DetrTransformerDecoder by name and never uses it: a name-only stub)
    _mod('mmdet.models.utils.transformer', DetrTransformerDecoderLayer=DetrTransformerDecoderLayer,
         DetrTransformerDecoder=type('DetrTransformerDecoder', (), {}))
    _mod('mmdet3d')
    _mod('mmdet3d.core')
    _mod('mmdet3d.core.bbox')
    _mod('mmdet3d.core.bbox.structures')
    _mod('mmdet3d.core.bbox.structures.utils', rotation_3d_in_axis=None)


_PKG = '_gd4d_refutils'


def load_reference(names=('deform3d_cross_attn', 'detr3d_transformer')):
    """Load reference hot-path files by path into a synthetic package; returns dict of modules."""
    install_stubs()
    base = os.path.join(REFERENCE_ROOT, _UTILS)
    if not os.path.isdir(base):
        raise FileNotFoundError(f'reference not present at {base} (build container only)')
    if _PKG not in sys.modules:
        pkg = types.ModuleType(_PKG)
        pkg.__path__ = [base]
        sys.modules[_PKG] = pkg
    out = {}
    for n in names:
        full = f'{_PKG}.{n}'
        if full not in sys.modules:
            spec = importlib.util.spec_from_file_location(full, os.path.join(base, n + '.py'))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[full] = mod
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                spec.loader.exec_module(mod)
            if n.startswith('deform3d_cross_attn'):
                mod.sampling_locations = None      # the undefined name of the dead CPU branch
        out[n] = sys.modules[full]
    return out


# --------------------------------------------------------------------------------------
# bbox coder: the step right after the decoder
# --------------------------------------------------------------------------------------
BBOX_CODERS = Registry('bbox_coder')


class BaseBBoxCoder:
    """mmdet.core.bbox.BaseBBoxCoder: an interface with encode/decode, nothing else."""

    def __init__(self, **kwargs):
        pass


def load_coder():
    """Import the reference's NMSFreeCoder (and the util.py it uses) unmodified.

    `projects.mmdet3d_plugin.core.bbox` is installed as a chain of empty namespace modules
    whose __path__ points into the reference tree, so the package __init__ files (which pull
    in all of mmdet3d) do not run, while `util.py` / `array_converter.py` / the coder file
    are found and executed as they lie.
    """
    install_stubs()
    _mod('mmdet.core')
    _mod('mmdet.core.bbox', BaseBBoxCoder=BaseBBoxCoder)
    _mod('mmdet.core.bbox.builder', BBOX_CODERS=BBOX_CODERS)
    rel = ''
    for part in ('projects', 'mmdet3d_plugin', 'core', 'bbox', 'coders'):
        rel = f'{rel}.{part}' if rel else part
        if rel not in sys.modules:
            m = types.ModuleType(rel)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *rel.split('.'))]
            sys.modules[rel] = m
    return importlib.import_module('projects.mmdet3d_plugin.core.bbox.coders.nms_free_coder')


# --------------------------------------------------------------------------------------
# head position embedding: the step feeding the path
# --------------------------------------------------------------------------------------
def load_head_pe():
    """Import the reference's Detr3DHeadPE file and positional_encoding.py unmodified.

    Only `Detr3DHeadPE.position_embeding` (a method that needs nothing of DETRHead), `SELayer` and
    `SinePositionalEncoding3D` are used by tools/gen_golden.py; the class hierarchy, losses, assigners and
    torchvision are name-only stubs.  `inverse_sigmoid` (third-party mmdet.models.utils.transformer) is bound to the
    reference's own copy of that function (utils/detr3d_transformer.py:28-43).
    Returns (head module, positional_encoding module).
    """
    install_stubs()
    load_coder()                                          # projects.* namespace + mmdet.core.bbox stubs
    ref = load_reference(('detr3d_transformer',))['detr3d_transformer']
    sys.modules['mmcv.cnn'].Linear = nn.Linear
    sys.modules['mmcv.cnn'].bias_init_with_prob = lambda p: float(-math.log((1 - p) / p))
    sys.modules['mmcv.runner'].force_fp32 = lambda *a, **k: (lambda f: f)
    sys.modules['mmcv.runner'].BaseModule = BaseModule
    sys.modules['mmcv.cnn.bricks.transformer'].POSITIONAL_ENCODING = POSITIONAL_ENCODING
    _mod('torchvision', transforms=None)
    core = sys.modules['mmdet.core']
    core.multi_apply = core.reduce_mean = core.build_sampler = core.build_assigner = None
    _mod('mmdet.models.utils.transformer', inverse_sigmoid=ref.inverse_sigmoid,
         DetrTransformerDecoderLayer=DetrTransformerDecoderLayer)
    heads = Registry('head')
    sys.modules['mmdet.models'].HEADS = heads
    _mod('mmdet.models.dense_heads', DETRHead=type('DETRHead', (nn.Module,), {}))
    _mod('mmdet3d.core.bbox.coders', build_bbox_coder=None)
    _mod('mmdet3d.models')
    _mod('mmdet3d.models.builder', build_loss=None)
    rel = ''
    for part in ('projects', 'mmdet3d_plugin', 'models'):
        rel = f'{rel}.{part}' if rel else part
        if rel not in sys.modules:
            m = types.ModuleType(rel)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *rel.split('.'))]
            sys.modules[rel] = m
    for sub in ('dense_heads', 'utils'):
        full = f'projects.mmdet3d_plugin.models.{sub}'
        if full not in sys.modules:
            m = types.ModuleType(full)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *full.split('.'))]
            sys.modules[full] = m
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        head = importlib.import_module('projects.mmdet3d_plugin.models.dense_heads.detr3d_head_pe')
        pe = importlib.import_module('projects.mmdet3d_plugin.models.utils.positional_encoding')
    return head, pe


# --------------------------------------------------------------------------------------
# head loss: Hungarian assignment + the per-layer loss with its normalisers
# --------------------------------------------------------------------------------------
MATCH_COST = Registry('match cost')
BBOX_ASSIGNERS = Registry('bbox assigner')


class FocalLossCost:
    """mmdet 2.x `FocalLossCost` (third-party, not under /root/reference), restated from its published definition:
    cost[q, g] = (pos(p[q, label_g]) - neg(p[q, label_g])) * weight with p = sigmoid(logit),
    neg = -log(1 - p + eps) (1 - alpha) p^gamma, pos = -log(p + eps) alpha (1 - p)^gamma."""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.sigmoid()
        neg_cost = -(1 - cls_pred + self.eps).log() * (1 - self.alpha) * cls_pred.pow(self.gamma)
        pos_cost = -(cls_pred + self.eps).log() * self.alpha * (1 - cls_pred).pow(self.gamma)
        cls_cost = pos_cost[:, gt_labels] - neg_cost[:, gt_labels]
        return cls_cost * self.weight


class IoUCost:
    """Placeholder of the config's `IoUCost(weight=0.0)` ("fake cost", never called by HungarianAssigner3D.assign)."""

    def __init__(self, iou_mode='giou', weight=1.):
        self.weight = weight


class AssignResult:
    """mmdet `AssignResult`: a record of (num_gts, gt_inds, max_overlaps, labels)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class PseudoSampler:
    """mmdet `PseudoSampler`: positives = assigned (> 0), negatives = background (== 0), no sampling."""

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        res = types.SimpleNamespace()
        res.pos_inds, res.neg_inds = pos_inds, neg_inds
        res.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        res.pos_gt_bboxes = gt_bboxes.view(-1, gt_bboxes.shape[-1])[res.pos_assigned_gt_inds.long(), :]
        return res


def _weight_reduce_mean(loss, weight, avg_factor):
    """mmdet `weight_reduce_loss(..., reduction='mean', avg_factor)`: sum(loss * weight) / avg_factor."""
    if weight is not None:
        loss = loss * weight
    return loss.sum() / avg_factor


class FocalLoss(nn.Module):
    """mmdet 2.x `FocalLoss(use_sigmoid=True)`, restated from its published python path (`py_sigmoid_focal_loss`):
    BCE-with-logits(pred, one-hot) * (alpha t + (1 - alpha)(1 - t)) * pt^gamma, pt = (1 - p) t + p (1 - t);
    labels == num_classes are background (all-zero target row)."""

    def __init__(self, gamma=2.0, alpha=0.25, loss_weight=1.0):
        super().__init__()
        self.gamma, self.alpha, self.loss_weight = gamma, alpha, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        num_classes = pred.size(1)
        t = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        focal_weight = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * focal_weight
        if weight is not None:
            weight = weight.view(-1, 1)
        return self.loss_weight * _weight_reduce_mean(loss, weight, avg_factor)


class L1Loss(nn.Module):
    """mmdet `L1Loss`: loss_weight * sum(|pred - target| * weight) / avg_factor (zero when there is no target)."""

    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        if target.numel() == 0:
            return pred.sum() * 0
        return self.loss_weight * _weight_reduce_mean((pred - target).abs(), weight, avg_factor)


def multi_apply(func, *args, **kwargs):
    """mmdet `multi_apply`: map, then transpose the tuple of results."""
    results = map(lambda *a: func(*a, **kwargs), *args)
    return tuple(map(list, zip(*results)))


def load_head_loss():
    """Import the reference's Detr3DHeadPE (for `loss`, `loss_single`, `get_targets`, `_get_target_single`), its
    `HungarianAssigner3D` and `BBox3DL1Cost` unmodified.  The mmdet pieces they call (FocalLossCost, AssignResult,
    PseudoSampler, FocalLoss, L1Loss, multi_apply, reduce_mean) are third-party and absent from /root/reference: they are
    restated above from their published definitions (single process: reduce_mean is the identity).
    Returns (head module, assigner module)."""
    head, _ = load_head_pe()
    head.multi_apply = multi_apply
    head.reduce_mean = lambda t: t
    MATCH_COST.register_module(module=FocalLossCost)
    MATCH_COST.register_module(module=IoUCost)
    _mod('mmdet.core.bbox.builder', BBOX_CODERS=BBOX_CODERS, BBOX_ASSIGNERS=BBOX_ASSIGNERS)
    _mod('mmdet.core.bbox.assigners', AssignResult=AssignResult, BaseAssigner=object)
    _mod('mmdet.core.bbox.match_costs', build_match_cost=lambda cfg: MATCH_COST.build(cfg))
    _mod('mmdet.core.bbox.match_costs.builder', MATCH_COST=MATCH_COST)
    _mod('mmdet.core.bbox.iou_calculators', bbox_overlaps=None)
    for sub in ('assigners', 'match_costs'):
        full = f'projects.mmdet3d_plugin.core.bbox.{sub}'
        if full not in sys.modules:
            m = types.ModuleType(full)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *full.split('.'))]
            sys.modules[full] = m
    importlib.import_module('projects.mmdet3d_plugin.core.bbox.match_costs.match_cost')     # registers BBox3DL1Cost
    asg = importlib.import_module('projects.mmdet3d_plugin.core.bbox.assigners.hungarian_assigner_3d')
    return head, asg


# --------------------------------------------------------------------------------------
# distiller (BASELINE configs[4]): the instance distillation loss of the teacher - student step
# --------------------------------------------------------------------------------------
def load_distiller():
    """Import the reference's distillation/distillers/mix_distill.py unmodified.  Only the unbound
    `MixDistill.get_instance_distill_loss` (and `get_feat_distill_loss`'s 'vanilla' arm) is used by tools/gen_golden.py;
    detector building, checkpoint loading and the registry are name-only stubs.  Returns the module."""
    install_stubs()
    load_coder()                                          # projects.* namespace; core/bbox/util.py for denormalize_bbox
    _mod('mmdet.models.detectors')
    _mod('mmdet.models.detectors.base', BaseDetector=nn.Module)
    sys.modules['mmdet.models'].build_detector = None
    run = sys.modules['mmcv.runner']
    run.load_checkpoint = run._load_checkpoint = run.load_state_dict = None
    rel = 'projects.mmdet3d_plugin.distillation'
    for full in (rel, rel + '.distillers'):
        if full not in sys.modules:
            m = types.ModuleType(full)
            m.__path__ = [os.path.join(REFERENCE_ROOT, *full.split('.'))]
            sys.modules[full] = m
    _mod(rel + '.builder', DISTILLER=Registry('distiller'), build_distill_loss=None)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return importlib.import_module(rel + '.distillers.mix_distill')

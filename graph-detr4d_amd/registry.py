"""Registry shim with mmcv 1.x semantics (`@X.register_module()`, `build_from_cfg`).

The reference selects its attention modules purely by `type='...'` strings in config dicts
(projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:71-89) resolved through
mmcv.cnn.bricks.registry.ATTENTION etc.  mmcv cannot be installed on the ROCm box, so the package
carries this shim; when a real mmcv IS importable every class is additionally registered there
(force=True), which is what makes the package a drop-in for an existing mmdet3d checkout.
"""
import copy


class Registry:
    def __init__(self, name, mirror=None):
        self.name = name
        self.module_dict = {}
        self._mirror = mirror            # real mmcv registry, if present

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force and self.module_dict[key] is not cls:
                raise KeyError(f'{key} is already registered in {self.name}')
            self.module_dict[key] = cls
            if self._mirror is not None:
                self._mirror.register_module(name=key, force=True, module=cls)
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self.module_dict.get(key)

    def __contains__(self, key):
        return key in self.module_dict

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise KeyError(f'cfg must be a dict with a "type" key, got {cfg!r}')
    args = copy.deepcopy(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f'{typ} is not in the {registry.name} registry')
    return cls(**args)


def _mmcv(attr_path):
    try:
        import importlib
        mod_name, attr = attr_path.rsplit('.', 1)
        return getattr(importlib.import_module(mod_name), attr)
    except Exception:
        return None


ATTENTION = Registry('attention', _mmcv('mmcv.cnn.bricks.registry.ATTENTION'))
FEEDFORWARD_NETWORK = Registry('feed-forward network', _mmcv('mmcv.cnn.bricks.registry.FEEDFORWARD_NETWORK'))
TRANSFORMER_LAYER = Registry('transformer layer', _mmcv('mmcv.cnn.bricks.registry.TRANSFORMER_LAYER'))
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence',
                                      _mmcv('mmcv.cnn.bricks.registry.TRANSFORMER_LAYER_SEQUENCE'))
TRANSFORMER = Registry('transformer', _mmcv('mmdet.models.utils.builder.TRANSFORMER'))
BBOX_CODERS = Registry('bbox coder', _mmcv('mmdet.core.bbox.builder.BBOX_CODERS'))
BBOX_ASSIGNERS = Registry('bbox assigner', _mmcv('mmdet.core.bbox.builder.BBOX_ASSIGNERS'))


def build_attention(cfg, default_args=None):
    return build_from_cfg(cfg, ATTENTION, default_args)


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


def build_bbox_coder(cfg, default_args=None):
    return build_from_cfg(cfg, BBOX_CODERS, default_args)


def build_assigner(cfg, default_args=None):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)

"""graph-detr4d_amd: MI355X-native decoder hot path of Graph-DETR4D.

Importing the package registers the attention / transformer modules under the reference's type
names (`Deform3DCrossAttn`, `Detr3DCrossAtten`, `Detr3DTransformer`, ...) in the registry shim and,
when mmcv/mmdet are importable, in their real registries.
"""
__version__ = '0.1.0'

from .registry import (ATTENTION, BBOX_ASSIGNERS, BBOX_CODERS, TRANSFORMER, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE,  # noqa: F401
                       build_assigner, build_attention, build_bbox_coder, build_transformer, build_transformer_layer,
                       build_transformer_layer_sequence)
from .transformer_layers import (FFN, BaseTransformerLayer, DetrTransformerDecoderLayer,  # noqa: F401
                                 MultiheadAttention, TransformerLayerSequence)
from .deform3d_cross_attn import Deform3DCrossAttn  # noqa: F401
from .deform3d_cross_attn_mp import Deform3DCrossAttnMP  # noqa: F401
from .bbox_coder import NMSFreeCoder  # noqa: F401
from .criterion import Detr3DCriterion, HungarianAssigner3D  # noqa: F401
from .dgcnn_attn import DGCNNAttn  # noqa: F401
from .head_pe import FeaturePositionEmbedding  # noqa: F401
from . import functional, plumbing  # noqa: F401
from .detr3d_transformer import (Detr3DCrossAtten, Detr3DCrossAttenV2, Detr3DTransformer, Detr3DTransformerDecoder,  # noqa: F401
                                 HDetr3DTransformer, feature_sampling, inverse_sigmoid)

__all__ = ['Deform3DCrossAttn', 'Deform3DCrossAttnMP', 'DGCNNAttn', 'Detr3DCrossAtten', 'Detr3DCrossAttenV2', 'feature_sampling', 'Detr3DTransformer',
           'Detr3DTransformerDecoder', 'HDetr3DTransformer', 'MultiheadAttention', 'FFN', 'BaseTransformerLayer',
           'DetrTransformerDecoderLayer', 'TransformerLayerSequence', 'inverse_sigmoid',
           'NMSFreeCoder', 'HungarianAssigner3D', 'Detr3DCriterion', 'FeaturePositionEmbedding', 'BBOX_CODERS', 'BBOX_ASSIGNERS', 'ATTENTION', 'TRANSFORMER', 'TRANSFORMER_LAYER', 'TRANSFORMER_LAYER_SEQUENCE']

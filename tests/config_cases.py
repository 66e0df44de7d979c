"""Builders for the BASELINE.json configurations used by several test files (seeded, synthetic)."""
import torch

import graph_detr4d_amd as G
from graph_detr4d_amd import plumbing, synthetic


def decoder_cfg(cross, layers):
    return dict(type='Detr3DTransformerDecoder', num_layers=layers, return_intermediate=True,
                transformerlayers=dict(
                    type='DetrTransformerDecoderLayer',
                    attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1), cross],
                    feedforward_channels=512, ffn_dropout=0.1,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))


def reg_branches(layers, seed):
    nn = torch.nn
    torch.manual_seed(seed)
    regs = nn.ModuleList([nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                        nn.Linear(256, 10)) for _ in range(layers)])
    for r in regs:
        nn.init.normal_(r[-1].weight, std=0.02)
        nn.init.zeros_(r[-1].bias)
    return regs.eval()


def oracle_params(tr):
    sd = {k: v.detach().cpu() for k, v in tr.state_dict().items()}
    n = len(tr.decoder.layers)
    layers = [{k[len(f'decoder.layers.{i}.'):]: v for k, v in sd.items() if k.startswith(f'decoder.layers.{i}.')}
              for i in range(n)]
    return sd, layers


def config0(seed=1000):
    """configs[0]: DETR3D, 1 decoder layer, 100 queries, 6 x (3 x 256 x 256) synthetic images through a ResNet18 + FPN
    stand-in on the CPU (detectors/detr3d.py:39-66) -> 4 levels 32^2, 16^2, 8^2, 4^2 x 256 channels."""
    torch.manual_seed(seed)
    backbone = plumbing.ResNet18FPN(256).eval()
    extractor = plumbing.ImageFeatureExtractor(backbone)
    img = torch.randn(1, 6, 3, 256, 256, generator=torch.Generator().manual_seed(seed + 1))
    rig = synthetic.camera_rig(1, img_hw=(256, 256))
    metas = synthetic.make_img_metas(rig, img_shape=(256, 256, 3), batch=1)
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=6,
        decoder=decoder_cfg(dict(type='Detr3DCrossAtten', num_cams=6, pc_range=synthetic.PC_RANGE, num_points=1,
                                 embed_dims=256), 1)))
    tr.init_weights()
    synthetic.randomise_cross_attn_(tr.decoder.layers[0].attentions[1], seed=seed)
    regs = reg_branches(1, seed + 2)
    qe = torch.randn(100, 512, generator=torch.Generator().manual_seed(seed + 3))
    with torch.no_grad():
        feats = extractor(img, metas)
    return dict(extractor=extractor, img=img, metas=metas, tr=tr.eval(), regs=regs, query_embed=qe, feats=feats)

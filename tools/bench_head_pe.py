#!/usr/bin/env python
"""GPU time of the head's feature position embedding (graph_detr4d_amd.FeaturePositionEmbedding) at the headline
size (24 cameras, R50 pyramid), with a per-piece breakdown, beside the reference's op sequence (the oracle's
restatement) run with torch on the same GPU.  Dev tool."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graph_detr4d_amd as G  # noqa: E402
from graph_detr4d_amd import ops, synthetic  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--hip-only', action='store_true', help='only the library\'s own route (what a rocprofv3 --stats run should see)')
    a = ap.parse_args()
    n = 6 * a.frames
    torch.manual_seed(0)
    mod = G.FeaturePositionEmbedding(pc_range=synthetic.PC_RANGE).cuda().eval()
    feats = synthetic.feature_pyramid(n, device='cuda')
    rig = synthetic.camera_rig(a.frames)
    metas = synthetic.make_img_metas(rig)
    pixels = sum(f.shape[-2] * f.shape[-1] for f in feats) * n
    with torch.no_grad():
        first = timed(lambda: (setattr(mod, '_sine_cache', None), setattr(mod, '_mask_cache', None), mod(feats, metas)), 3)
        # the position embedding is kept per camera (keyed by the camera's matrix): time the three cases
        mod.cache_position_embedding = False
        steady = timed(lambda: mod(feats, metas))                      # every camera recomputed (what rounds 1 - 4 measured)
        mod.cache_position_embedding = True
        static = timed(lambda: mod(feats, metas))                      # a static rig: nothing recomputed
        import copy
        moving = [copy.deepcopy(metas) for _ in range(4)]
        for k, mm in enumerate(moving):                                # the past frames' cameras (all but the first 6) move each sample
            for cam in range(6, n):
                m = np.array(mm[0]['lidar2img'][cam], dtype=np.float64)
                m[:3, 3] += 0.05 * (k + 1)
                mm[0]['lidar2img'][cam] = m
        turn = [0]

        def temporal():
            turn[0] += 1
            return mod(feats, moving[turn[0] % 4])
        temporal_ms = timed(temporal)
        print(f'position embedding kept per camera: static rig {static:.2f} ms, current frame static + {n - 6} past-frame cameras moving '
              f'{temporal_ms:.2f} ms, every camera recomputed {steady:.2f} ms')
        mod.channels_last_out = True                                    # gate + fuse in ONE kernel (gd4d_mlp2_se_fuse_fwd), channels-last levels out
        static_cl = timed(lambda: mod(feats, metas))
        temporal_cl = timed(temporal)
        mod.cache_position_embedding = False
        steady_cl = timed(lambda: mod(feats, metas))
        mod.channels_last_out = False
        print(f'channels_last_out=True (the decoder then reads the levels in place: no slice-planar copy): static rig {static_cl:.2f} ms, '
              f'{n - 6} past-frame cameras moving {temporal_cl:.2f} ms, every camera recomputed {steady_cl:.2f} ms')
        if a.hip_only:
            print(f'cameras={n} pixels={pixels}  steady-state {steady:.2f} ms (first call, sine branch not cached: {first:.2f} ms)')
            return
        os.environ['GD4D_TORCH_OPS'] = '1'
        steady_conv = timed(lambda: mod(feats, metas))
        os.environ.pop('GD4D_TORCH_OPS')
        masks, pad_hw = mod.padding_masks(metas, feats)
        i2l = torch.from_numpy(np.linalg.inv(np.asarray(rig, dtype=np.float64))).float().cuda()
        x0 = {}

        def frustum():
            for lvl, f in enumerate(feats):
                x0[lvl] = ops.frustum_pe_input_fwd(i2l, f.shape[-2:], pad_hw, 64, 1, synthetic.PC_RANGE)[0]
        t_fr = timed(frustum)
        t_pe = timed(lambda: [mod.position_encoder(x0[l]) for l in x0])
        t_se = timed(lambda: [mod.fpe.gate_logits(f.flatten(0, 1)) for f in feats])
        fl = [f.flatten(0, 1) for f in feats]
        t_fuse = timed(lambda: [ops.se_fuse_fwd(f, f, f, f) for f in fl])
        t_sine = timed(lambda: (setattr(mod, '_sine_cache', None), mod._sine_branch(masks)))
    with torch.no_grad():
        r = n
        s_tot = pixels // n
        xcl = torch.empty(r, s_tot, 192, device='cuda')
        st = [0]
        for f in feats[:-1]:
            st.append(st[-1] + f.shape[-2] * f.shape[-1])
        t_frc = timed(lambda: [ops.frustum_pe_input_fwd(i2l, f.shape[-2:], pad_hw, 64, 1, synthetic.PC_RANGE, out=xcl,
                                                        row_start=s0) for f, s0 in zip(feats, st)])
        sw = mod._split_weights()
        hid = torch.empty(r * s_tot, 1024, device='cuda')
        pe_o = torch.empty(r * s_tot, 256, device='cuda')
        t_g1 = timed(lambda: ops.gemm_bf16x3_fwd(xcl.view(r * s_tot, 192), *sw['pe0'], mod.position_encoder[0].bias,
                                                 relu=True, out=hid))
        t_g2 = timed(lambda: ops.gemm_bf16x3_fwd(hid, *sw['pe2'], mod.position_encoder[2].bias, out=pe_o))
        crw = mod.fpe.conv_reduce.weight.view(256, 256).contiguous()
        t_vp = timed(lambda: ops.value_proj_fwd([f for f in feats], crw, mod.fpe.conv_reduce.bias))
        t_g3 = timed(lambda: ops.gemm_bf16x3_fwd(pe_o, *sw['se1'], mod.fpe.conv_expand.bias, relu_in=True))
        t_fz = timed(lambda: [ops.se_fuse_chlast_fwd(f.flatten(0, 1), pe_o.view(r, s_tot, 256), pe_o.view(r, s_tot, 256),
                                                     f.flatten(0, 1), s0) for f, s0 in zip(feats, st)])
        t_mlp = timed(lambda: ops.mlp2_bf16x3_fwd(xcl.view(r * s_tot, 192), sw['pe_mlp'], mod.position_encoder[2].bias, out=pe_o))
        t_fr_mlp = t_se_k = t_one = float('nan')
        if sw.get('pe_mlp_fr') is not None and sw.get('se_mlp') is not None:
            shp = [tuple(f.shape[-2:]) for f in feats]
            sine = mod._sine_branch(masks, chlast=True).view(r, s_tot, 256)
            fl4 = [f.flatten(0, 1).contiguous() for f in feats]
            outs = [torch.empty(r, h, w, 256, device='cuda') for h, w in shp]
            b2, ce = mod.position_encoder[2].bias, mod.fpe.conv_expand
            t_fr_mlp = timed(lambda: ops.mlp2_frustum_fwd(i2l, shp, pad_hw, 64, 1, synthetic.PC_RANGE, sw['pe_mlp_fr'], b2, out=pe_o))
            t_se_k = timed(lambda: ops.mlp2_se_fuse_fwd(fl4, sw['se_mlp'], ce.bias, pe_o.view(r, s_tot, 256), sine, outs=outs))
            t_one = timed(lambda: ops.mlp2_pe_se_fwd(i2l, fl4, pad_hw, 64, 1, synthetic.PC_RANGE, sw['pe_mlp_fr'], b2, sw['se_mlp'], ce.bias,
                                                     sine, outs=outs))
    print(f'  position MLP 192->1024->256 reading its rows (gd4d_mlp2_bf16x3_fwd) {t_mlp:.2f} ms, generating them (gd4d_mlp2_frustum_fwd) '
          f'{t_fr_mlp:.2f} ms; SE gate + fuse (gd4d_mlp2_se_fuse_fwd) {t_se_k:.2f} ms; both MLPs + fuse as one kernel (gd4d_mlp2_pe_se_fwd) '
          f'{t_one:.2f} ms')
    print(f'  GEMM path pieces: frustum (channels-last) {t_frc:.2f}  GEMM 192->1024 {t_g1:.2f}  GEMM 1024->256 {t_g2:.2f}  '
          f'conv_reduce (value_proj kernel) {t_vp:.2f}  GEMM 256->256 {t_g3:.2f}  transposing fuse {t_fz:.2f} ms')
    flops = pixels * 2 * (192 * 1024 + 1024 * 256 + 2 * 256 * 256)
    print(f'cameras={n} pixels={pixels}  steady-state {steady:.2f} ms with gd4d_gemm_bf16x3_fwd, {steady_conv:.2f} ms with library '
          f'1x1 convolutions (first call, sine branch not cached: {first:.2f} ms)')
    print(f'  frustum geometry kernel {t_fr:.2f} ms ({pixels * 192 * 4 / t_fr / 1e9:.2f} TB/s written)   '
          f'position_encoder convs {t_pe:.2f} ms   SE convs {t_se:.2f} ms   fuse kernel {t_fuse:.2f} ms '
          f'({pixels * 256 * 4 * 5 / t_fuse / 1e9:.2f} TB/s)   sine branch {t_sine:.2f} ms (cached afterwards)')
    print(f'  dense work per call: {flops / 1e12:.2f} TFLOP fp32 -> {flops / steady / 1e9:.1f} TFLOP/s overall')


if __name__ == '__main__':
    main()

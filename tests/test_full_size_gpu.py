"""Parity at the BASELINE size: every layer of the 6-layer decoder (900 queries, 24 cameras = 6 x T=4, the
full 116x200..15x25 pyramid) against the CPU oracle, plus the reference-point refinement.

Layers are checked with TEACHER FORCING (each layer gets the oracle's input state).  Chaining all six is
ill-conditioned on synthetic data: the feature maps are i.i.d. N(0,1) per pixel, so a 1e-6 rounding difference
in a refined reference point moves the bilinear sample by ~1e-3 and the difference triples per layer (measured:
0 / 0.7 / 2.6 / 7.9 / 17 / 55 % of query rows off by > 1e-3 after layers 1..6 when chained).  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('project', ['late', 'late-rows', 'early'])
def test_each_decoder_layer_900q_24cams_matches_oracle(project, monkeypatch):
    """project = 'late': the default aggregate-then-project value path (slice-planar copy, gd4d_cross_attn_plan_fwd,
    gd4d_cross_attn_agg_sliced_fwd, gd4d_value_proj_heads_fwd); 'late-rows': its one-workgroup-per-query form
    (gd4d_cross_attn_agg_fwd); 'early': value_proj over the pyramid + gd4d_cross_attn_fwd (the reference's order)."""
    monkeypatch.setenv('GD4D_PROJECT', 'early' if project == 'early' else 'late')
    monkeypatch.setenv('GD4D_AGG', 'rows' if project == 'late-rows' else 'sliced')
    import bench
    import graph_detr4d_amd as G
    from graph_detr4d_amd import functional as Fn
    from graph_detr4d_amd import ops, synthetic
    from oracle import torch_oracle as O
    torch.set_num_threads(16)
    frames, queries, layers = 4, 900, 6
    n = 6 * frames
    tr, regs = bench.build_decoder(G, n, layers, 'fp32', 1002)
    feats = synthetic.feature_pyramid(n, synthetic.R50_LEVELS, seed=77)
    qe = torch.randn(queries, 512, generator=torch.Generator().manual_seed(5))
    metas = synthetic.make_img_metas(synthetic.camera_rig(frames), batch=1)
    sd, layer_params = bench.state_as_oracle_params(tr)
    pc = synthetic.PC_RANGE
    dev = 'cuda'
    query_pos, query = (t.unsqueeze(1).contiguous() for t in torch.split(qe, 256, dim=1))   # (Q, 1, C)
    with torch.no_grad():
        ref = torch.nn.functional.linear(query_pos.permute(1, 0, 2), sd['reference_points.weight'],
                                         sd['reference_points.bias']).sigmoid()
        import copy
        regs_cpu = copy.deepcopy(regs)
        tr_d, regs_d = tr.to(dev), regs.to(dev)
        feats_d = [f.to(dev) for f in feats]
        x = query
        worst = []
        for lid in range(layers):
            # oracle: one layer + refinement from the oracle's own state
            y_ref, parts = O.decoder_layer(layer_params[lid], x, feats, query_pos, ref, metas, pc,
                                           cross='Deform3DCrossAttn', num_heads=8, num_points=4, return_parts=True)
            tmp = regs_cpu[lid](y_ref.permute(1, 0, 2))
            new = torch.zeros_like(ref)
            new[..., :2] = tmp[..., :2] + O.inverse_sigmoid(ref[..., :2])
            new[..., 2:3] = tmp[..., 4:5] + O.inverse_sigmoid(ref[..., 2:3])
            ref_next = new.sigmoid()
            # HIP: the same layer on the same inputs
            captured = {}
            orig = Fn.sample_aggregate

            def spy(value, shapes, ref_, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, order=None):
                hm = value.shape[2] == sum(h * w for h, w in shapes) and value.shape[1] != value.shape[2]
                _, captured['mask'] = ops.cross_attn_fwd(value, shapes, ref_.contiguous(), offsets.contiguous(),
                                                         attn_logits.contiguous(), cam_logits.contiguous(), lidar2img,
                                                         pc_range, img_h, img_w, head_major=hm, want_mask=True)
                return orig(value, shapes, ref_, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w,
                            order=order)
            orig_late = Fn.LateValues.aggregate

            def spy_late(self, module, ref_, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=None, **vp):
                if self.mode == 'sliced':
                    captured['mask'] = ops.cross_attn_plan_fwd(self.pyramid, ref_.contiguous(), offsets.contiguous(),
                                                               attn_logits.contiguous(), cam_logits.contiguous(), lidar2img,
                                                               module.pc_range, img_h, img_w, module.num_heads, want_mask=True)[1]
                else:
                    captured['mask'] = ops.cross_attn_agg_fwd(self.cl, self.shapes, ref_.contiguous(), offsets.contiguous(),
                                                              attn_logits.contiguous(), cam_logits.contiguous(), lidar2img,
                                                              module.pc_range, img_h, img_w, module.num_heads, want_mask=True)[2]
                return orig_late(self, module, ref_, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=order, **vp)
            Fn.sample_aggregate, Fn.LateValues.aggregate = spy, spy_late
            try:
                y = tr_d.decoder.layers[lid](x.to(dev), key=None, value=feats_d, query_pos=query_pos.to(dev),
                                             reference_points=ref.to(dev), img_metas=metas)
            finally:
                Fn.sample_aggregate, Fn.LateValues.aggregate = orig, orig_late
            tmp_d = Fn.run_branch(regs_d[lid], y.permute(1, 0, 2).contiguous())
            ref_d = Fn.refine_reference(tmp_d, ref.to(dev))
            # The visibility mask is the path's only discontinuity: a point within an ulp of a threshold can flip
            # between the GPU's and the CPU's GEMM rounding of the offsets and move ONE query row.  Both masks are
            # captured, exactly the rows with a flipped bit are excluded (their number is bounded), and every other
            # row must meet the path's tolerance (north_star: 1e-3) strictly.
            mism = captured['mask'].cpu().bool() != parts['mask']                     # (B, N, Q, Hh, P)
            flipped = mism.any(dim=4).any(dim=3).any(dim=1)[0]                        # (Q,)
            err = (y.cpu() - y_ref).abs().amax(dim=(1, 2))          # per query row
            worst.append((int(flipped.sum()), float(err[~flipped].max()), float(err.median())))
            assert flipped.sum().item() <= 8, (lid, worst)
            assert err[~flipped].max().item() < 1e-3, (lid, worst)
            assert err.median().item() < 2e-4, (lid, worst)
            assert (ref_d.cpu() - ref_next).abs().max().item() < 1e-3
            x, ref = y_ref, ref_next                                  # teacher forcing
    print('per-layer (rows with a flipped mask bit, max error of the other rows, median row error):', worst)

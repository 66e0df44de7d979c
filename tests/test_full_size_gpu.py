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
            assert flipped.sum().item() <= 2, (lid, worst)           # observed: 0 in every layer (profiles/r05_flipped_rows_offsets_exact_ab.txt)
            assert err[~flipped].max().item() < 1e-3, (lid, worst)
            assert err.median().item() < 2e-4, (lid, worst)
            assert (ref_d.cpu() - ref_next).abs().max().item() < 1e-3
            x, ref = y_ref, ref_next                                  # teacher forcing
    print('per-layer (rows with a flipped mask bit, max error of the other rows, median row error):', worst)


def test_fused_decoder_900q_24cams_matches_oracle():
    """The path bench.py times, at the size it times: Detr3DTransformer.forward -> fused_decoder.run (row chains with
    split-bf16 x3 products, HEADGEMM value_proj of the aggregates, plan + channel-sliced gather) on 900 queries x 24
    cameras.  Layer 0 of the whole-transformer call against the oracle (detr3d_transformer.py:130-147, :166-225), then every
    further layer through the same fused entry with TEACHER FORCING (the oracle's state and reference points as input; see
    the module docstring for why chaining is ill-conditioned), reference-point refinement included.  Rows whose visibility
    mask differs in a bit between GPU and CPU GEMM rounding are excluded, exactly those (their number is bounded)."""
    import copy
    import types
    import bench
    import graph_detr4d_amd as G
    from graph_detr4d_amd import functional as Fn
    from graph_detr4d_amd import fused_decoder, ops, synthetic
    from oracle import torch_oracle as O
    torch.set_num_threads(16)
    frames, queries, layers = 4, 900, 6
    n = 6 * frames
    tr, regs = bench.build_decoder(G, n, layers, 'fp32', 1002)
    feats = synthetic.feature_pyramid(n, synthetic.R50_LEVELS, seed=78)
    qe = torch.randn(queries, 512, generator=torch.Generator().manual_seed(6))
    metas = synthetic.make_img_metas(synthetic.camera_rig(frames), batch=1)
    sd, layer_params = bench.state_as_oracle_params(tr)
    pc = synthetic.PC_RANGE
    dev = 'cuda'
    regs_cpu = copy.deepcopy(regs)
    query_pos, query = (t.unsqueeze(1).contiguous() for t in torch.split(qe, 256, dim=1))   # (Q, 1, C)
    masks, calls = [], []
    orig_late, orig_single = Fn.LateValues.aggregate, fused_decoder.run_single

    def spy_late(self, module, ref_, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=None, **vp):
        assert self.mode == 'sliced'
        masks.append(ops.cross_attn_plan_fwd(self.pyramid, ref_.contiguous(), offsets.contiguous(), attn_logits.contiguous(),
                                             cam_logits.contiguous(), lidar2img, module.pc_range, img_h, img_w, module.num_heads,
                                             want_mask=True)[1])
        return orig_late(self, module, ref_, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=order, **vp)

    def spy_single(*a, **k):
        calls.append(1)
        return orig_single(*a, **k)

    def oracle_layer(lid, x, ref):
        y_ref, parts = O.decoder_layer(layer_params[lid], x, feats, query_pos, ref, metas, pc,
                                       cross='Deform3DCrossAttn', num_heads=8, num_points=4, return_parts=True)
        tmp = regs_cpu[lid](y_ref.permute(1, 0, 2))
        new = torch.zeros_like(ref)
        new[..., :2] = tmp[..., :2] + O.inverse_sigmoid(ref[..., :2])
        new[..., 2:3] = tmp[..., 4:5] + O.inverse_sigmoid(ref[..., 2:3])
        return y_ref, new.sigmoid(), parts['mask']

    def check(lid, y, ref_d, y_ref, ref_next, mask_d, mask_ref, worst):
        mism = mask_d.cpu().bool() != mask_ref                                        # (B, N, Q, Hh, P)
        flipped = mism.any(dim=4).any(dim=3).any(dim=1)[0]                            # (Q,)
        err = (y.cpu() - y_ref).abs().amax(dim=(1, 2))                                # per query row
        rerr = (ref_d.cpu() - ref_next).abs().amax(dim=(0, 2))
        worst.append((int(flipped.sum()), float(err[~flipped].max()), float(err.median()), float(rerr[~flipped].max())))
        assert flipped.sum().item() <= 2, (lid, worst)               # observed: 0 in every layer (profiles/r05_flipped_rows_offsets_exact_ab.txt)
        assert err[~flipped].max().item() < 1e-3, (lid, worst)                        # north_star: 1e-3 fp32
        assert err.median().item() < 2e-4, (lid, worst)
        assert rerr[~flipped].max().item() < 1e-3, (lid, worst)

    Fn.LateValues.aggregate, fused_decoder.run_single = spy_late, spy_single
    try:
        with torch.no_grad():
            tr_d, regs_d = tr.to(dev), regs.to(dev)
            feats_d = [f.to(dev) for f in feats]
            states, init_ref, refs = tr_d(feats_d, qe.to(dev), reg_branches=regs_d, img_metas=metas)
            assert calls == [1] and len(masks) == layers, 'the whole-transformer call must take the fused single-stream loop'
            assert states.shape == (layers, queries, 1, 256) and refs.shape == (layers, 1, queries, 3)
            ref0 = torch.nn.functional.linear(query_pos.permute(1, 0, 2), sd['reference_points.weight'],
                                              sd['reference_points.bias']).sigmoid()
            torch.testing.assert_close(init_ref.cpu(), ref0, rtol=1e-5, atol=1e-5)
            worst = []
            y_ref, ref_next, mask_ref = oracle_layer(0, query, ref0)
            check(0, states[0], refs[0], y_ref, ref_next, masks[0], mask_ref, worst)
            # layers 1 .. 5: the fused loop on ONE layer, fed the oracle's state (teacher forcing)
            x, ref = y_ref, ref_next
            pos_d = query_pos.to(dev)
            for lid in range(1, layers):
                y_ref, ref_next, mask_ref = oracle_layer(lid, x, ref)
                one = types.SimpleNamespace(layers=[tr_d.decoder.layers[lid]])
                late = Fn.LateValues(feats_d)
                del masks[:]
                out_all, ref_all = fused_decoder.run(one, x.to(dev), pos_d, feats_d, ref.to(dev), [regs_d[lid]], metas, None, None,
                                                     None, None, pc, True, late=late)
                late.finish()
                assert len(calls) == lid + 1 and len(masks) == 1
                check(lid, out_all[0], ref_all[0], y_ref, ref_next, masks[0], mask_ref, worst)
                x, ref = y_ref, ref_next
    finally:
        Fn.LateValues.aggregate, fused_decoder.run_single = orig_late, orig_single
    print('per-layer (rows with a flipped mask bit, max error of the other rows, median row error, max ref error):', worst)

#!/usr/bin/env python
"""Benchmark of the Graph-DETR4D decoder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one sample (6 cameras x T frames, pre-computed FPN pyramid resident in HBM) pushed
through the 6-layer decoder (self-attention + Deform3DCrossAttn + FFN + reference-point refinement),
eval mode, batch 1 per GPU - BASELINE.json's metric "6-cam samples/sec through decoder at
900q x T=4".  N > 1 runs one independent replica per GPU (the path shards by sample; inference has
no data-path collective), so scaling is "weak".  Rank 0 prints ONE JSON line.

The JSON line also carries
  roofline      - the step's fused sample-aggregate kernel (gd4d_cross_attn_agg_fwd in the default
                  aggregate-then-project form, gd4d_cross_attn_fwd with GD4D_PROJECT=early): algorithmic
                  bytes per launch (SURVEY.md §8d, V counted from the kernel's own visibility mask) / mean
                  launch duration measured with HIP events on the launch stream, vs 8 TB/s
  cpu_baseline  - the CPU oracle (oracle/torch_oracle.py, a port of the reference path) timed on this
                  box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--frames', type=int, default=4, help='T temporal frames (cameras = 6*T)')
    ap.add_argument('--queries', type=int, default=900)
    ap.add_argument('--layers', type=int, default=6)
    ap.add_argument('--levels', default='r50', choices=['r50', 'vov'])
    ap.add_argument('--value-dtype', default='fp32', choices=['fp32', 'bf16'],
                    help='storage of the projected value tensor (fp32 = the reference-parity mode)')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train', 'distill'],
                    help="'train': forward + backward + one flat RCCL gradient all-reduce + SGD per step (secondary metric); "
                         "'distill': BASELINE configs[4] - teacher pass (no grad), student pass and teacher-query-guided "
                         "student pass over one pyramid, instance distillation loss, backward, all-reduce, SGD")
    ap.add_argument('--input-layout', default='nchw', choices=['nchw', 'nhwc'],
                    help='memory layout of the resident feature levels: nchw = as the reference backbone hands them (the headline), '
                         'nhwc = channels-last levels, gathered in place without the per-sample copy')
    ap.add_argument('--no-nhwc-figure', action='store_true', help='skip the second figure (channels-last levels gathered in place)')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of hipGraph replay')
    ap.add_argument('--no-exact-figure', action='store_true', help='skip the figure with every chain GEMM on six products')
    ap.add_argument('--no-f2b-figure', action='store_true', help='skip the features -> boxes figure (head position embedding -> decoder -> box epilogue -> decode)')
    ap.add_argument('--only-f2b', action='store_true', help='dev: run ONLY the features -> boxes figure (what tools/prof_f2b.sh profiles)')
    ap.add_argument('--rotate', type=int, default=3,
                    help='K resident samples (own pyramid, queries and camera rig each) served round-robin by K hipGraphs, every graph '
                         'replaying a sample it was NOT captured on (inputs written into its static buffers in place, lidar2img '
                         'refreshed; checked against eager bit for bit before the timed region); 1 = one sample replayed')
    ap.add_argument('--inflight', type=int, default=2,
                    help="--mode infer: independent samples in flight per GPU, each on its own HIP stream with its own hipGraph "
                         "(a step = one sample on every stream); 1 = one sample at a time")
    ap.add_argument('--criterion', action='store_true',
                    help='--mode train: the head (cls / reg branches, box epilogue) and its real loss (Hungarian '
                         'assignment + focal / L1 terms over all layers) instead of a synthetic loss; eager launch')
    ap.add_argument('--gts', type=int, default=40, help='--criterion: ground-truth boxes per sample')
    ap.add_argument('--overlap-comm', action='store_true',
                    help='--mode train/distill with several ranks: eager backward, every bucket\'s all-reduce starts from a '
                         'post-accumulate hook while the backward of the layers below is still running')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-preflight', action='store_true', help='N > 1: skip the rank / device / all-reduce-alone check before the timed steps')
    ap.add_argument('--dropout', action='store_true', help='--mode train: modules in train() mode (dropout 0.1 as the reference trains); default: eval mode, autograd on')
    ap.add_argument('--no-fuse-wgrad', action='store_true', help='--mode train: leave the accumulation of parameter gradients to autograd')
    ap.add_argument('--optimizer', default='adamw', choices=['adamw', 'sgd', 'torch-sgd'],
                    help="--mode train: 'adamw' = the reference's recipe, gradient-norm clipping at 35 + AdamW lr 2e-4 wd 0.01 "
                         "(...ceph.py:205-213), two launches over the flat buffers (dist.FlatGradAllReducer.adamw_step); 'sgd' = plain SGD "
                         "as one launch (round 4's line); 'torch-sgd' = torch.optim.SGD.step()")
    ap.add_argument('--check', action='store_true', help='dev: print gradient / parameter checksums (the launch modes must train identically)')
    ap.add_argument('--split', action='store_true', help='dev, --mode train: serialised wall-clock split of the step')
    ap.add_argument('--min-seconds', type=float, default=1.0,
                    help='--mode infer: the K-step window is repeated until this much time has been measured; the line carries min / median / max')
    ap.add_argument('--no-roofline', action='store_true', help='dev: skip the kernel-level roofline section (roofline = null)')
    ap.add_argument('--no-stress', action='store_true',
                    help='skip the all-visible stress launches of the gather kernels (SURVEY 8d): a rocprofv3 --stats run then '
                         'averages only launches on the real workload')
    ap.add_argument('--cpu-threads', type=int, default=None, help='torch threads of the CPU baseline (default: min(cores, 16), the fastest measured)')
    ap.add_argument('--cpu-layers', type=int, default=None,
                    help='decoder layers of the CPU-baseline sample (default: bounded automatically)')
    return ap.parse_args()


def build_decoder(G, num_cams, layers, value_dtype, seed):
    from graph_detr4d_amd import synthetic
    torch.manual_seed(seed)
    tr = G.build_transformer(dict(
        type='Detr3DTransformer', num_feature_levels=4, num_cams=num_cams,
        decoder=dict(type='Detr3DTransformerDecoder', num_layers=layers, return_intermediate=True,
                     transformerlayers=dict(
                         type='DetrTransformerDecoderLayer',
                         attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1),
                                    dict(type='Deform3DCrossAttn', num_cams=num_cams,
                                         pc_range=synthetic.PC_RANGE, num_points=4, embed_dims=256,
                                         value_dtype=value_dtype)],
                         feedforward_channels=512, ffn_dropout=0.1,
                         operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))))
    tr.init_weights()
    for i, layer in enumerate(tr.decoder.layers):
        synthetic.randomise_cross_attn_(layer.attentions[1], seed=seed + i)
    nn = torch.nn
    regs = nn.ModuleList([nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                        nn.Linear(256, 10)) for _ in range(layers)])
    for r in regs:                                   # small refinements, like a trained head's deltas
        nn.init.normal_(r[-1].weight, std=0.02)
        nn.init.zeros_(r[-1].bias)
    return tr.eval(), regs.eval()


def state_as_oracle_params(tr):
    sd = {k: v.detach().cpu() for k, v in tr.state_dict().items()}
    n = len(tr.decoder.layers)
    layers = [{k[len(f'decoder.layers.{i}.'):]: v for k, v in sd.items()
               if k.startswith(f'decoder.layers.{i}.')} for i in range(n)]
    return sd, layers


def cpu_baseline(tr, regs, feats_cpu, query_embed, metas, pc_range, layers_to_time, threads=None, repeats=3):
    """Oracle (CPU port of the reference path) on this host's cores: the same sample through
    `layers_to_time` decoder layers, median of `repeats` runs (about 10 s of CPU work)."""
    from oracle import torch_oracle as O
    cores = os.cpu_count() or 1
    # measured on the 2 x 64-core GPU box: 16 threads 0.60 s/layer, 32: 0.81, 64: 0.93, 256: 15.7
    torch.set_num_threads(threads or min(cores, 16))
    sd, layer_params = state_as_oracle_params(tr)
    regs_cpu = [r.cpu() for r in regs]
    small = [f[:, :, :, :8, :8].contiguous() for f in feats_cpu]      # warm-up: page in code paths
    times = []
    with torch.no_grad():
        O.transformer(sd, layer_params[:1], small, query_embed[:64], metas, pc_range,
                      reg_branches=regs_cpu[:1], cross='Deform3DCrossAttn', num_points=4)
        for _ in range(repeats):
            t0 = time.perf_counter()
            O.transformer(sd, layer_params[:layers_to_time], feats_cpu, query_embed, metas, pc_range,
                          reg_branches=regs_cpu[:layers_to_time], cross='Deform3DCrossAttn', num_points=4)
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    per_layer = dt / layers_to_time
    full = per_layer * len(layer_params)
    return dict(value=1.0 / full, unit='samples/s', cores=torch.get_num_threads(), threads=torch.get_num_threads(), host_cores=cores,
                cores_note='cores = threads = the torch intra-op threads actually used (the fastest setting measured on this host class: '
                           '16: 0.60 s/layer, 32: 0.81, 64: 0.93, all: 15.7); host_cores = os.cpu_count() of the box',
                kind='port',
                sample=f'one sample (same inputs as the GPU run) through {layers_to_time} of '
                       f'{len(layer_params)} decoder layers, median of {repeats} runs = {dt:.2f} s'
                       + ('' if layers_to_time == len(layer_params) else ', scaled to all layers'),
                ms_per_layer=per_layer * 1e3)


def main():
    a = parse()
    from graph_detr4d_amd import dist as D
    rank, local_rank, world = D.env()
    if world != a.gpus and world > 1:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    if a.gpus > 1 and world == 1:
        raise SystemExit('for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    # (GD4D_DIST_BACKEND=gloo with every rank on one GPU is a dev path to exercise the N > 1 code on a 1-GPU box)
    backend = os.environ.get('GD4D_DIST_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    D.init(backend=backend, device=dev)              # "nccl" = RCCL; no-op for a single process
    # N > 1: before anything is timed - the rank count, who sits on which device, one all-reduce alone at the gradient sizes
    # of the decoder / an R50 model / a VoVNet-99 model (so that a scaling record separates transport from compute)
    a.preflight = None
    if world > 1 and not a.no_preflight:
        mb = os.environ.get('GD4D_PREFLIGHT_MB', '22,140,330' if backend == 'nccl' else '2,8')
        a.preflight = D.preflight(a.gpus, dev, sizes=[int(float(x) * (1 << 20)) for x in mb.split(',') if x])
        if rank == 0:
            print('[bench] preflight ' + json.dumps(a.preflight), file=sys.stderr)

    import graph_detr4d_amd as G
    from graph_detr4d_amd import _lib, ops, synthetic
    _lib.load()                                       # fail loudly if the HIP library is missing

    n_cams = 6 * a.frames
    levels = synthetic.R50_LEVELS if a.levels == 'r50' else synthetic.VOV_LEVELS
    seed = D.sample_seed(1000 + 2, rank)              # 1000 + config + rank
    tr, regs = build_decoder(G, n_cams, a.layers, a.value_dtype, 1000 + 2)   # same model on every rank
    feats_cpu = synthetic.feature_pyramid(n_cams, levels, seed=seed)
    g = torch.Generator().manual_seed(seed + 3)
    query_embed_cpu = torch.randn(a.queries, 512, generator=g)
    rig = synthetic.camera_rig(a.frames)
    metas = synthetic.make_img_metas(rig, batch=1)

    tr, regs = tr.to(dev), regs.to(dev)
    feats = [f.to(dev) for f in feats_cpu]            # inputs resident in HBM before timing
    if a.input_layout == 'nhwc':                      # same logical (B, N, C, H, W) tensors, stored (B, N, H, W, C)
        feats = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in feats]
    query_embed = query_embed_cpu.to(dev)

    def step():
        return tr(feats, query_embed, reg_branches=regs, img_metas=metas)

    if a.mode in ('train', 'distill'):
        train_bench(a, D, tr, regs, feats, query_embed, metas, dev, rank, n_cams, levels, G)
        return
    if a.only_f2b:
        print(json.dumps({'features_to_boxes': features_to_boxes(G, tr, regs, feats, query_embed, rig, a, D, dev, torch.cuda.Stream(dev), a.steps)}))
        D.shutdown()
        return

    # S samples in flight: a decoder layer alternates between an HBM-bound phase (the aggregate launch) and a
    # latency-bound one (attention core + row chains on a few dozen workgroups); with one sample on the device the memory
    # system idles during the second.  Request i has its own pyramid, queries, HIP stream and hipGraph; the streams run
    # free (no join between steps; the closing synchronise of the timed region waits for all of them).  One graph per
    # request, not one graph with S branches: a capture in which a forked stream forks again (request stream -> its
    # copy stream) sends this runtime's hipStreamEndCapture into an endless recursion.
    from graph_detr4d_amd import functional as Fn
    n_req = max(1, a.inflight)
    reqs = [(feats, query_embed)]
    for i in range(1, n_req):
        gi = torch.Generator().manual_seed(seed + 3 + 7919 * i)
        fi = [f.to(dev) for f in synthetic.feature_pyramid(n_cams, levels, seed=seed + 7919 * i)]
        if a.input_layout == 'nhwc':
            fi = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in fi]
        reqs.append((fi, torch.randn(a.queries, 512, generator=gi).to(dev)))
    streams = [torch.cuda.Stream(dev) for _ in range(n_req)]

    def request(i):
        return tr(reqs[i][0], reqs[i][1], reg_branches=regs, img_metas=metas)

    launch = 'eager'
    graphs = []
    single_ms = None
    with torch.no_grad():
        out = step()
        torch.cuda.synchronize()
        eager_outs = []
        for i in range(n_req):                        # every request once eagerly on its own stream (allocations, caches)
            with torch.cuda.stream(streams[i]), Fn.request_slot(i):
                eager_outs.append(request(i))
        torch.cuda.synchronize()
        torch.testing.assert_close(eager_outs[0][0], out[0], rtol=0, atol=0)      # the stream does not change a result
        if not a.no_graph:
            try:
                for i in range(n_req):
                    g_i = torch.cuda.CUDAGraph()
                    # thread_local: with a process group up, the RCCL watchdog thread polls events while we
                    # capture; in the default 'global' mode that would invalidate the capture
                    with torch.cuda.graph(g_i, stream=streams[i], capture_error_mode='thread_local'), Fn.request_slot(i):
                        static_out = request(i)
                    if i == 0:
                        static_out0 = static_out
                    g_i.replay()
                    torch.cuda.synchronize()
                    torch.testing.assert_close(static_out[0], eager_outs[i][0], rtol=1e-5, atol=1e-5)
                    graphs.append(g_i)
                launch = 'hipgraph'
            except Exception as e:                    # report, never hide
                print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly',
                      file=sys.stderr)
                graphs, launch = [], 'eager'

        def run():
            for i in range(n_req):
                with torch.cuda.stream(streams[i]), Fn.request_slot(i):
                    if graphs:
                        graphs[i].replay()
                    else:
                        request(i)

        # --rotate K: the serving pattern the figure stands for.  K samples stay resident (own pyramid, queries, rig); graph k is
        # captured on sample k's static buffers and then serves sample (k + 1) % K - written into those buffers in place, its
        # lidar2img matrices refreshed through Fn.lidar2img_device OUTSIDE the graph; each is checked against the eager result of
        # the sample it now holds, bit for bit.  The timed steps replay the K graphs round-robin: consecutive steps read different
        # pyramids.
        rotate = None
        rot_graphs = [graphs[0]] if graphs else []
        if graphs and a.rotate > 1:
            import numpy as np
            K = a.rotate
            resident = [(reqs[0][0], reqs[0][1], metas)]
            for k in range(1, K):
                gk = torch.Generator().manual_seed(seed + 3 + 104729 * k)
                fk = [f.to(dev) for f in synthetic.feature_pyramid(n_cams, levels, seed=seed + 104729 * k)]
                if a.input_layout == 'nhwc':
                    fk = [f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in fk]
                rig_k = rig.copy()
                rig_k[:, :3, 3] += (np.random.RandomState(k).randn(n_cams, 3) * np.array([20.0, 20.0, 0.02])).astype(np.float32)
                resident.append((fk, torch.randn(a.queries, 512, generator=gk).to(dev), synthetic.make_img_metas(rig_k, batch=1)))
            with Fn.request_slot(9000):                                # eager truth, on lidar2img buffers no graph reads
                truth = [tr(f, qe, reg_branches=regs, img_metas=mt) for f, qe, mt in resident]
            torch.cuda.synchronize()
            # static buffers: graph 0 keeps request 0's (reqs[0]); graphs 1 .. K-1 get their own, captured on sample k
            static = [(reqs[0][0], reqs[0][1])]
            rot_outs = [static_out0]
            for k in range(1, K):
                fk, qk, mk = resident[k]
                static.append(([torch.empty_like(f).copy_(f) for f in fk], qk.clone()))
                torch.cuda.synchronize()                               # (made on the default stream, read on the request's stream)
                with torch.cuda.stream(streams[0]), Fn.request_slot(1000 + k):
                    tr(static[k][0], static[k][1], reg_branches=regs, img_metas=mk)
                torch.cuda.synchronize()
                g_k = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_k, stream=streams[0], capture_error_mode='thread_local'), Fn.request_slot(1000 + k):
                    rot_outs.append(tr(static[k][0], static[k][1], reg_branches=regs, img_metas=mk))
                rot_graphs.append(g_k)
            resident[0] = ([f.clone() for f in reqs[0][0]], reqs[0][1].clone(), metas)      # (sample 0 itself, apart from graph 0's buffers)
            for k in range(K):                                         # graph k now serves sample (k + 1) % K
                fs, qs, ms = resident[(k + 1) % K]
                with torch.cuda.stream(streams[0]), Fn.request_slot(0 if k == 0 else 1000 + k):
                    for dst, src in zip(static[k][0], fs):
                        dst.copy_(src)
                    static[k][1].copy_(qs)
                    Fn.lidar2img_device(ms, static[k][1])
                    for t in rot_outs[k]:
                        t.fill_(float('nan'))
                    rot_graphs[k].replay()
                torch.cuda.synchronize()
                for got, want in zip(rot_outs[k], truth[(k + 1) % K]):
                    if not torch.equal(got, want):
                        raise SystemExit(f'[bench] --rotate: graph {k} replayed on sample {(k + 1) % K} differs from the eager result')
            ops.check_handoff()
            rotate = {'samples': K, 'graphs': K,
                      'check': 'every graph replayed on a sample it was not captured on (features, queries written into its static buffers '
                               'in place, lidar2img refreshed outside the graph): equal to the eager result bit for bit'}
        rot_step = [0]

        def run_one():
            with torch.cuda.stream(streams[0]), Fn.request_slot(0):
                if rot_graphs:
                    rot_graphs[rot_step[0] % len(rot_graphs)].replay()
                    rot_step[0] += 1
                else:
                    request(0)

        stats = {}
        torch.cuda.synchronize()
        # THE figure: one request at a time, batch 1.  A window = exactly `steps` samples, barrier + synchronise on
        # both sides, MAX over ranks; windows are repeated until --min-seconds of them have been measured (a 20-step window is
        # 32 ms: one window is at the mercy of a clock ramp) - value = the MEDIAN window, min / max beside it.
        windows, measured = [], 0.0
        while True:
            el = D.timed_steps(run_one, a.steps, a.warmup if not windows else 0, dev, stats if not windows else {})
            windows.append(el)
            measured += el
            if measured >= a.min_seconds or len(windows) >= 200:
                break
        ops.check_handoff()                           # a SIGNAL / WAIT hand-off that timed out inside a replayed graph raises here
        windows_ms = sorted(w / a.steps * 1e3 for w in windows)
        elapsed = sorted(windows)[len(windows) // 2]
        single_ms = elapsed / a.steps * 1e3
        if rotate is not None:                        # graph 0's buffers hold sample 0 again (what the sections below and the CPU baseline read)
            with torch.cuda.stream(streams[0]), Fn.request_slot(0):
                for dst, src in zip(static[0][0], resident[0][0]):
                    dst.copy_(src)
                static[0][1].copy_(resident[0][1])
                Fn.lidar2img_device(metas, static[0][1])
            torch.cuda.synchronize()
            del resident, static, truth
        inflight = None
        if n_req > 1:                                 # secondary: n_req requests in flight (a step = one sample on every stream)
            el2 = sorted(D.timed_steps(run, a.steps, a.warmup if i == 0 else 0, dev, {}) for i in range(max(3, min(len(windows), 15))))
            ops.check_handoff()
            inflight = {'requests': n_req, 'value': D.aggregate_throughput(n_req, a.steps, a.gpus, el2[len(el2) // 2]), 'unit': 'samples/s',
                        'ms_per_step': el2[len(el2) // 2] / a.steps * 1e3, 'ms_per_step_min': el2[0] / a.steps * 1e3,
                        'ms_per_step_max': el2[-1] / a.steps * 1e3, 'windows': len(el2),
                        'note': f'{n_req} independent requests in flight, one HIP stream and one hipGraph each; a step = one sample on every stream'}

        def run_eager():                              # no hipGraph: the modules dropped in without the capture recipe
            with torch.cuda.stream(streams[0]), Fn.request_slot(0):
                request(0)
        eager_steps = max(1, min(a.steps, 10))
        eager_ms = D.timed_steps(run_eager, eager_steps, 2, dev, {}) / eager_steps * 1e3

        # ---- what fp32-class arithmetic on the query side costs: the same request with EVERY chain GEMM on six bf16 products
        # (GD4D_CHAIN_EXACT, ~2^-24 per product) instead of three (~2^-16); value_proj of the aggregates and the attention core's
        # two products stay on three.  Secondary figure beside the x3 headline.
        exact = None
        if graphs and not a.no_exact_figure:
            try:
                with ops.all_exact():
                    with torch.cuda.stream(streams[0]), Fn.request_slot(0):
                        o_ex = request(0)
                    torch.cuda.synchronize()
                    g_ex = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_ex, stream=streams[0], capture_error_mode='thread_local'), Fn.request_slot(0):
                        request(0)

                def run_ex():
                    with torch.cuda.stream(streams[0]), Fn.request_slot(0):
                        g_ex.replay()
                el_ex = sorted(D.timed_steps(run_ex, a.steps, a.warmup if i == 0 else 0, dev, {}) for i in range(7))[3]
                ops.check_handoff()
                exact = {'value': D.aggregate_throughput(1, a.steps, a.gpus, el_ex), 'unit': 'samples/s', 'ms_per_step': el_ex / a.steps * 1e3,
                         'max_abs_difference_to_x3_last_layer': float((o_ex[0][-1] - eager_outs[0][0][-1]).abs().max()),
                         'max_abs_difference_to_x3_first_layer': float((o_ex[0][0] - eager_outs[0][0][0]).abs().max()),
                         'note': 'every GEMM operation of the row chains on six split-bf16 products (in/out-proj, the cross-attention Linears, '
                                 'output_proj, FFN, reg branches); value_proj of the aggregates and the attention core stay on three'}
                del g_ex, o_ex
            except Exception as e:                    # secondary figure: report, never fail the bench line
                exact = {'error': f'{type(e).__name__}: {e}'}

        # ---- second figure: the same requests with the feature levels STORED channels-last (the FPN's output
        # layout) - the cross-attention gathers them in place, the per-sample copy is not launched.  Same logical tensors,
        # same results; the NCHW figure above stays the headline.
        nhwc = None
        if a.input_layout == 'nchw' and not a.no_nhwc_figure:
            try:
                reqs_cl = [([f.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for f in fi], qi) for fi, qi in reqs]
                torch.cuda.synchronize()              # (made on the default stream, read on the requests' streams)
                graphs_cl = []
                for i in range(n_req):
                    with torch.cuda.stream(streams[i]), Fn.request_slot(i):
                        o_cl = tr(reqs_cl[i][0], reqs_cl[i][1], reg_branches=regs, img_metas=metas)
                    torch.cuda.synchronize()
                    torch.testing.assert_close(o_cl[0], eager_outs[i][0], rtol=0, atol=0)     # the layout does not change a result
                    if graphs:
                        g_i = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_i, stream=streams[i], capture_error_mode='thread_local'), Fn.request_slot(i):
                            tr(reqs_cl[i][0], reqs_cl[i][1], reg_branches=regs, img_metas=metas)
                        graphs_cl.append(g_i)

                def run_cl(count):
                    for i in range(count):
                        with torch.cuda.stream(streams[i]), Fn.request_slot(i):
                            if graphs_cl:
                                graphs_cl[i].replay()
                            else:
                                tr(reqs_cl[i][0], reqs_cl[i][1], reg_branches=regs, img_metas=metas)
                el_cl = D.timed_steps(lambda: run_cl(n_req), a.steps, a.warmup, dev, {})
                one_cl = D.timed_steps(lambda: run_cl(1), a.steps, a.warmup, dev, {}) / a.steps * 1e3 if n_req > 1 else el_cl / a.steps * 1e3
                ops.check_handoff()
                nhwc = {'value': a.gpus * 1e3 / one_cl, 'unit': 'samples/s', 'ms_per_step': one_cl,
                        'value_inflight': D.aggregate_throughput(n_req, a.steps, a.gpus, el_cl), 'requests_in_flight': n_req,
                        'ms_per_step_inflight': el_cl / a.steps * 1e3, 'value_batch1': a.gpus * 1e3 / one_cl,
                        'ms_per_sample_batch1': one_cl, 'launch': 'hipgraph' if graphs_cl else 'eager',
                        'note': 'levels stored (B, N, H, W, C): gathered in place by gd4d_cross_attn_agg_sliced_fwd, no per-sample copy; '
                                'outputs bit-identical to the NCHW run (checked)'}
                del reqs_cl, graphs_cl
            except Exception as e:                    # secondary figure: report, never fail the bench line
                nhwc = {'error': f'{type(e).__name__}: {e}'}

    # ---------------- features -> boxes as one request (secondary figure) ----------------
    f2b = None
    # (single-GPU runs only: its timing windows use the job's barrier - a figure one rank computes alone would leave the others behind)
    if world == 1 and not a.no_f2b_figure and a.input_layout == 'nchw' and a.value_dtype == 'fp32':
        try:
            f2b = features_to_boxes(G, tr, regs, feats, query_embed, rig, a, D, dev, streams[0], a.steps)
        except Exception as e:                        # secondary figure: report, never fail the bench line
            f2b = {'error': f'{type(e).__name__}: {e}'}

    # ---------------- kernel-level roofline of the fused sample-aggregate kernel ----------------
    roofline, kernels = None, {}
    if rank == 0 and not a.no_roofline:
        roofline, kernels = fused_kernel_roofline(tr, regs, feats, query_embed, metas, a, ops, synthetic)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        n_layers = a.cpu_layers or a.layers
        cpu = cpu_baseline(tr, regs, feats_cpu, query_embed_cpu, metas, synthetic.PC_RANGE,
                           min(n_layers, a.layers), a.cpu_threads)

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        line = {
            'metric': f'decoder_samples_per_sec_{a.queries}q_T{a.frames}',
            'value': D.aggregate_throughput(1, a.steps, a.gpus, elapsed), 'unit': 'samples/s', 'n_gpus': a.gpus,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'ms_per_step_min': windows_ms[0], 'ms_per_step_median': windows_ms[len(windows_ms) // 2], 'ms_per_step_max': windows_ms[-1],
            'windows': len(windows_ms), 'seconds_measured': measured,
            'dtype': 'f32 (bf16x3 GEMMs)' if a.value_dtype == 'fp32' else 'bf16-storage/f32-accumulate (bf16x3 GEMMs)',
            'dtype_detail': 'features, gather, aggregation, softmax, LayerNorm: fp32; the query-side GEMMs of the row chains (in/out-proj, '
                            'the Linears of the cross-attention, value_proj of the aggregates, FFN) and the two products of the attention '
                            'core are split-bf16 x3 products on the bf16 MFMA with fp32 accumulation (~2^-16 relative per product, '
                            'inside the 1e-3 contract); the GEMMs whose outputs become reference points or sampling offsets (initial '
                            'reference, reg branch, deform_sampling_offsets) use six products (~2^-24)',
            'value_batch1': a.gpus * 1e3 / single_ms, 'ms_per_sample_batch1': single_ms,
            'value_inflight2': None if inflight is None else inflight['value'],
            'requests_in_flight': inflight,
            'eager_ms_per_sample': eager_ms,
            'channels_last_input': nhwc,
            'rotate': rotate,
            'exact_gemms': exact,
            'features_to_boxes': f2b,
            'data': 'synthetic',
            'config': {'workload': f'Graph-DETR4D decoder, {a.layers} layers, {a.queries} queries, '
                                   f'{n_cams} cameras (6 x T={a.frames}), 4 FPN levels '
                                   f'{"x".join(str(h) + "*" + str(w) for h, w in levels)}, 256 ch, '
                                   f'batch 1 per GPU, ONE request at a time (SURVEY 8(d)\'s definition; a step = one sample through the '
                                   f'decoder, one HIP stream, one hipGraph), pyramid resident in HBM; value = the median of '
                                   f'{len(windows_ms)} windows of {a.steps} steps'
                                   + ('' if rotate is None else f'; {rotate["samples"]} resident samples (own pyramid, queries, camera rig) served '
                                      f'round-robin, every graph on a sample it was not captured on'),
                       'rotate': 1 if rotate is None else rotate['samples'],
                       'metric_8d': 'value',
                       'baseline_config': 'configs[2]', 'launch': launch, 'inflight': 1, 'input_layout': a.input_layout, 'global_batch': a.gpus,
                       'samples_per_step': a.gpus,
                       'parallelism': f'replicas x{a.gpus}' if a.gpus > 1 else 'single GPU'},
            'preflight': a.preflight,
            'ranks': stats['ranks'], 'ms_per_step_rank_min': stats['rank_seconds_min'] / a.steps * 1e3,
            'ms_per_step_rank_max': stats['rank_seconds_max'] / a.steps * 1e3, 'allreduce_bytes_per_step': 0,
            'roofline': roofline, 'cpu_baseline': cpu, 'kernels': kernels,
        }
        print(json.dumps(line))
    D.shutdown()


def train_bench(a, D, tr, regs, feats, query_embed, metas, dev, rank, n_cams, levels, G=None):
    """Secondary mode: one training step of the decoder per sample - forward, a synthetic loss, backward
    (gd4d_cross_attn_bwd, gd4d_value_proj_bwd_*, gd4d_linear_bwd_weight), ONE flat gradient all-reduce over RCCL, SGD.
    The feature pyramid requires grad (it comes from the backbone in the reference's training).

    --mode distill (BASELINE configs[4]; distillation/distillers/mix_distill.py:92-106): per step the teacher's
    decoder + head run under no_grad on the teacher's pyramid and queries; the student's decoder runs on its own
    queries and again on the teacher's queries (dense_heads/detr3d_head_pe.py:560-566, :617-625) over ONE pyramid whose
    value tensors are projected once (Detr3DTransformer.forward_shared); the loss is the student's loss plus the
    instance distillation terms of get_instance_distill_loss (mix_distill.py:140-168: BCE-with-logits against the
    teacher's sigmoid scores and L1 on the boxes, both re-weighted by the teacher's max score)."""
    world_size = max(a.gpus, 1)
    for f in feats:
        f.requires_grad_(True)
    distill = a.mode == 'distill'
    teacher = None
    if distill:
        from graph_detr4d_amd import functional as Fn, synthetic
        nn = torch.nn
        t_tr, t_regs = build_decoder(G, n_cams, a.layers, a.value_dtype, 2000 + 4)       # the teacher: its own weights
        t_tr, t_regs = t_tr.to(dev), t_regs.to(dev)

        def branches(seed):
            torch.manual_seed(seed)
            return nn.ModuleList([nn.Sequential(nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                                                nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                                                nn.Linear(256, 10)) for _ in range(a.layers)]).to(dev)
        t_cls, s_cls = branches(81).eval(), branches(82)
        t_feats = [f.to(dev) for f in synthetic.feature_pyramid(n_cams, levels, seed=2000 + 4 + rank)]
        t_queries = torch.randn(a.queries, 512, generator=torch.Generator().manual_seed(83)).to(dev)   # teacher.query_embedding
        for p_ in list(t_tr.parameters()) + list(t_regs.parameters()) + list(t_cls.parameters()):
            p_.requires_grad_(False)
        teacher = (t_tr, t_regs, t_cls, t_feats, t_queries, s_cls)
    crit = cls_branches = None
    if a.criterion:
        # the reference's training loss (dense_heads/detr3d_head_pe.py:568-612 + :1014-1094): class branches as in
        # Detr3DHeadPE._init_layers, synthetic ground truth inside the point-cloud range
        from graph_detr4d_amd import Detr3DCriterion, synthetic
        from graph_detr4d_amd import functional as Fn
        nn = torch.nn
        torch.manual_seed(77)
        cls_branches = nn.ModuleList([nn.Sequential(nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                                                    nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True),
                                                    nn.Linear(256, 10)) for _ in range(a.layers)]).to(dev)
        crit = Detr3DCriterion(pc_range=synthetic.PC_RANGE).to(dev)
        g = torch.Generator().manual_seed(78 + rank)
        gt = torch.randn(a.gts, 9, generator=g)
        gt[:, 0:2] *= 25.
        gt[:, 3:6] = gt[:, 3:6].abs() * 2 + 0.3
        gt_boxes, gt_labels = [gt.to(dev)], [torch.randint(0, 10, (a.gts,), generator=g).to(dev)]
    params = list(tr.parameters()) + list(regs.parameters()) + (list(cls_branches.parameters()) if a.criterion else []) \
        + (list(teacher[5].parameters()) if distill else [])
    opt = torch.optim.SGD(params, lr=1e-4)
    # plain SGD as ONE launch over the flat parameter / gradient buffers (dist.FlatGradAllReducer.sgd_step: the same update, bit
    # for bit) instead of torch.optim.SGD's four multi-tensor launches (~110 us of the step); --torch-sgd keeps the optimizer object
    flat_sgd = a.optimizer != 'torch-sgd'

    def sgd_step():
        if a.optimizer == 'adamw':                    # the reference's recipe: clip_grad_norm_(35) + AdamW(lr 2e-4, wd 0.01)
            reducer.adamw_step(lr=2e-4, weight_decay=0.01, max_norm=35.0)
        elif flat_sgd:
            reducer.sgd_step(1e-4)
        else:
            opt.step()
    # gradient buckets in the order backward finalises them (head branches, then the decoder layers last to first): with
    # several ranks the slices are all-reduced back to back after the captured backward (reduce_buckets), or from
    # post-accumulate hooks while the backward is still running (--overlap-comm, eager launch)
    reducer = D.FlatGradAllReducer(params, buckets=D.decoder_buckets(tr, regs, cls_branches if a.criterion else None,
                                                                     teacher[5] if distill else None),
                                   align=4 if flat_sgd else 1)      # (16-byte slices: the flat parameter buffer's layout too)
    # .grad = views of one flat buffer from the start (graph captures); unless the all-reduce is driven by autograd hooks
    # (--overlap-comm) the weight-gradient kernels add into those views themselves (no accumulation launch per parameter)
    fuse = not (a.overlap_comm and world_size > 1) and not a.no_fuse_wgrad
    reducer.bind(fuse_weight_grads=fuse)
    if flat_sgd:
        reducer.flatten_params()                      # before anything captures a parameter's address
    if a.dropout:
        tr.train()                                    # the reference's training mode: dropout 0.1 in the attentions and the FFN
    else:
        tr.eval()                                     # dropout off keeps the step deterministic; autograd stays on

    class DecoderAndHead(torch.nn.Module):
        """query_embed + pyramid -> (all_cls_scores, all_bbox_preds): the part of the step in front of the assignment."""

        def __init__(self):
            super().__init__()
            self.tr, self.regs, self.cls_branches = tr, regs, cls_branches

        def forward(self, query_embed, *pyramid):
            states, init_ref, refs = self.tr(list(pyramid), query_embed, reg_branches=self.regs, img_metas=metas)
            outs = Fn.head_outputs(states, init_ref, refs, self.cls_branches, self.regs, synthetic.PC_RANGE)
            return outs['all_cls_scores'], outs['all_bbox_preds']

    front = DecoderAndHead() if a.criterion else None
    front_launch = 'eager'
    prepared = crit.prepare_ground_truth(gt_boxes, gt_labels, a.queries, dev) if a.criterion else None   # (resident ground truth: outside any capture)

    def criterion_loss(all_cls, all_box):
        # cost matrices, the Hungarian assignment (gd4d_hungarian_assign_fwd: on the device since round 5 - no host round trip, so
        # the step below is ONE hipGraph) and the focal / L1 terms of every layer
        return sum(crit.loss(gt_boxes, gt_labels, dict(all_cls_scores=all_cls, all_bbox_preds=all_box), prepared=prepared).values())

    from graph_detr4d_amd.criterion import instance_distill_loss

    def distill_loss():
        t_tr, t_regs, t_cls, t_feats, t_queries, s_cls = teacher
        with torch.no_grad():                                           # mix_distill.py:92-96
            t_states, t_init, t_refs = t_tr(t_feats, t_queries, reg_branches=t_regs, img_metas=metas)
            t_out = Fn.head_outputs(t_states, t_init, t_refs, t_cls, t_regs, synthetic.PC_RANGE)
        (s_states, s_init, s_refs), (g_states, g_init, g_refs) = tr.forward_shared(
            feats, [query_embed, t_queries], reg_branches=regs, img_metas=metas)
        guided = Fn.head_outputs(g_states, g_init, g_refs, s_cls, regs, synthetic.PC_RANGE)
        loss = (s_states ** 2).mean()                                   # stand-in for the student's own loss
        # get_instance_distill_loss (mix_distill.py:140-168), reweight_score=True, both loss weights 1
        terms = instance_distill_loss(t_out, dict(guided_cls_scores=guided['all_cls_scores'], guided_bbox_preds=guided['all_bbox_preds']))
        return loss + sum(terms.values())

    def step():
        reducer.zero_grad()
        for f in feats:
            f.grad = None
        if a.split:                                  # dev: serialised wall-clock split of the step
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if a.criterion:
                all_cls, all_box = front(query_embed, *feats)
            else:
                states, init_ref, refs = tr(feats, query_embed, reg_branches=regs, img_metas=metas)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            loss = criterion_loss(all_cls, all_box) if a.criterion else (states ** 2).mean()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            loss.backward()
            t3c = time.perf_counter()
            torch.cuda.synchronize(); t3 = time.perf_counter()
            print(f'[split] backward: host returned after {1e3 * (t3c - t2):.2f} ms', file=sys.stderr)
            reducer.reduce()
            sgd_step()
            torch.cuda.synchronize(); t4 = time.perf_counter()
            print(f'[split] forward {1e3 * (t1 - t0):.2f}  loss {1e3 * (t2 - t1):.2f}  backward {1e3 * (t3 - t2):.2f}  '
                  f'reduce + SGD {1e3 * (t4 - t3):.2f} ms', file=sys.stderr)
            return
        if distill:
            loss = distill_loss()
        elif a.criterion:
            loss = criterion_loss(*front(query_embed, *feats))
        else:
            states, init_ref, refs = tr(feats, query_embed, reg_branches=regs, img_metas=metas)
            loss = (states ** 2).mean()
        if overlap:
            reducer.begin_step()
        loss.backward()
        if finish:
            if overlap:
                reducer.finish()
            elif world_size > 1:
                reducer.reduce_buckets()
            else:
                reducer.reduce()
            sgd_step()

    # The step is ~2500 launches, most of them small: eagerly it is bound by the host's launch rate, not by the GPU.
    # One process: forward + backward + SGD are captured into one hipGraph (warm-up on a side stream first so that
    # autograd and the allocator have seen every shape) and replayed per step.  Several ranks: the capture ends before
    # the gradient all-reduce - the collective and the (few, fused) optimizer launches stay eager, so the step does not
    # depend on the communication backend being capturable.
    finish = True
    overlap = bool(a.overlap_comm) and world_size > 1
    if overlap:
        reducer.install_hooks()
        a.no_graph = True                              # hooks run on the host during an eager backward
    run, launch = step, front_launch
    for _ in range(2):
        step()
        if rank == 0 and a.check:
            print(f'[check] warm step: |grad| = {float(reducer.flat.double().norm()):.9e}', file=sys.stderr)
    torch.cuda.synchronize()
    if not a.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
            torch.cuda.current_stream().wait_stream(s)
            finish = world_size == 1                  # (read by step(): captured with or without all-reduce + SGD)
            with torch.cuda.graph(graph, stream=s, capture_error_mode='thread_local'):   # same stream as the warm-up
                step()
            if world_size == 1:
                run, launch = graph.replay, 'hipgraph'
            else:
                def run():
                    graph.replay()
                    reducer.reduce_buckets()
                    sgd_step()
                launch = 'hipgraph (forward + backward), eager bucketed all-reduce + ' + ('clip + AdamW' if a.optimizer == 'adamw' else 'SGD')
            run()
            torch.cuda.synchronize()
        except Exception as e:                        # report, never hide
            print(f'[bench] hipGraph capture of the training step failed ({type(e).__name__}: {e}); running eagerly',
                  file=sys.stderr)
            run, launch = step, 'eager'
        finish = True
    stats = {}
    elapsed = D.timed_steps(run, a.steps, a.warmup, dev, stats)
    from graph_detr4d_amd import ops as _ops
    _ops.check_handoff()                              # a hand-off that timed out inside the replayed step raises here
    if a.criterion:
        crit.assigner.check_status()                  # ... and so does an assignment that met a label out of range
    if rank == 0 and a.check:                                 # dev: the launch modes must train identically
        flat = reducer.flat
        psum = sum(float(p.detach().double().sum()) for p in params)
        print(f'[check] |grad| = {float(flat.double().norm()):.9e}  sum(params) = {psum:.9e}  '
              f'|d pyramid| = {float(sum(f.grad.double().norm() ** 2 for f in feats if f.grad is not None) ** 0.5):.6e}',
              file=sys.stderr)
    if rank == 0:
        line = {
            'metric': f'decoder_{"distill" if distill else "train"}_samples_per_sec_{a.queries}q_T{a.frames}',
            'value': D.aggregate_throughput(1, a.steps, a.gpus, elapsed), 'unit': 'samples/s', 'n_gpus': a.gpus,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': elapsed / a.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (bf16x3 GEMMs)' if _chain_calls() else 'f32', 'data': 'synthetic',
            'config': {'workload': ('distillation step (teacher pass under no_grad, student pass + teacher-query-guided student '
                                    'pass sharing one value projection, instance distillation loss); ' if distill else '') +
                                   ('training step of the decoder + head with the reference\'s loss (Hungarian assignment, '
                                    f'{a.gts} boxes; ' if a.criterion else '') +
                                   f'training step of the {a.layers}-layer decoder (forward + backward + flat gradient '
                                   f'all-reduce of {reducer.bytes_per_step() / 1e6:.1f} MB + '
                                   f'{"clip + AdamW" if a.optimizer == "adamw" else "SGD"}), {a.queries} queries, '
                                   f'{n_cams} cameras, batch 1 per GPU, pyramid (requires grad) resident in HBM',
                       'baseline_config': 'configs[4]' if distill else ('configs[3]' if a.levels == 'vov' else 'configs[2] + backward'),
                       'launch': launch + (', all-reduce overlapped with backward (hooks)' if overlap else ''),
                       'overlap_comm': overlap, 'weight_grads_accumulated_by_kernels': fuse, 'input_layout': a.input_layout,
                       'dropout': 'on (train mode)' if a.dropout else 'off (modules in eval mode, autograd on)',
                       'optimizer': ('clip_grad_norm_(35, L2) + AdamW lr 2e-4 wd 0.01 (the reference\'s recipe, ...ceph.py:205-213), two launches over '
                                     'the flat parameter / gradient buffers (dist.FlatGradAllReducer.adamw_step, gd4d_adamw_flat)' if a.optimizer == 'adamw'
                                     else 'SGD lr 1e-4, ' + ('one launch over the flat parameter / gradient buffers (dist.FlatGradAllReducer.sgd_step)'
                                                             if flat_sgd else 'torch.optim.SGD.step()')),
                       'query_side': ('row chains forward and backward, one autograd node for the decoder (graph_detr4d_amd/fused_train.py)'
                                      if _chain_calls() else
                                      'one autograd node per Linear / LayerNorm / attention core (the generic path)'),
                       'parallelism': f'dp{a.gpus}' if a.gpus > 1 else 'single GPU'},
            'preflight': a.preflight,
            'ranks': stats['ranks'], 'ms_per_step_rank_min': stats['rank_seconds_min'] / a.steps * 1e3,
            'ms_per_step_rank_max': stats['rank_seconds_max'] / a.steps * 1e3,
            'allreduce_bytes_per_step': reducer.describe()['allreduce_bytes'] if world_size > 1 else 0,
            'param_bytes': sum(p.numel() * 4 for p in params),
            'allreduce_buckets': reducer.describe()['buckets'],
            'roofline': None, 'cpu_baseline': None,
        }
        if not a.no_roofline and not distill and n_cams <= 64 and len(levels) <= 4:
            line['roofline'], line['kernels'] = _train_kernel_figures(a, n_cams, levels, dev)
        print(json.dumps(line))
    D.shutdown()


def _chain_calls():
    from graph_detr4d_amd import fused_train
    return fused_train.CALLS[0]


def _train_kernel_figures(a, n_cams, levels, dev):
    """The pyramid side of the training step, kernel by kernel, on synthetic inputs of the step's shapes (HIP events on the
    launch stream, each kernel replayed from a hipGraph of `layers` launches on `layers` different query sets).  roofline =
    the gather-dot kernel of the backward pass (the longest single kernel type of the step): SURVEY 8(d)'s algorithmic bytes
    (V visible (query, head, camera, point, level) tuples x 4 corners x Dh x 4 B, V from the plan kernel's own mask) over
    its duration, against 8 TB/s - the same recipe as the inference line's roofline; the kernel itself reads 8x that
    (256 raw channels per corner instead of Dh = 32 projected ones: what removes value_proj over the pyramid)."""
    from graph_detr4d_amd import ops, synthetic
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
    from bench_late import timed
    nl, q, hh = a.layers, a.queries, 8
    g = torch.Generator().manual_seed(11)
    feats = [torch.randn(1, n_cams, 256, h, w, generator=g).to(dev) for h, w in levels]
    l2i = torch.from_numpy(synthetic.camera_rig(max(1, n_cams // 6))[:n_cams]).unsqueeze(0).to(dev)
    sp, shapes = ops.pyramid_slice_planar_fwd(feats)
    pyr = ops.PyramidView.slice_planar(sp, shapes)
    del feats
    w_v, b_v = (torch.randn(256, 256, generator=g) / 16).to(dev), torch.randn(256, generator=g).to(dev)
    lay, vis = [], 0
    for _ in range(nl):
        ref = torch.rand(1, q, 3, generator=g).to(dev)
        off = (torch.randn(1, q, hh, 4, 3, generator=g) * 1.5).to(dev)
        attn = torch.randn(1, q, hh, len(levels), 4, generator=g).to(dev)
        cam = torch.randn(1, q, n_cams, generator=g).to(dev)
        order = ops.query_order_fwd(ref, synthetic.PC_RANGE)
        plan, mask = ops.cross_attn_plan_fwd(pyr, ref, off, attn, cam, l2i, synthetic.PC_RANGE, 900, 1600, hh, query_order=order,
                                             want_mask=True)
        vis += int(mask.sum().item())
        lay.append((ref, off, attn, cam, plan, torch.randn(1, q, 256, generator=g).to(dev)))
    vis /= nl                                                        # visible (camera, query, head, point) tuples per launch
    sink = ops.PyramidGrad(pyr, nl, 1, q, hh)
    dpart = torch.empty(ops.cross_attn_dot_bytes(1, n_cams, q, hh), device=dev, dtype=torch.uint8)
    beta = torch.empty(1, q, hh, device=dev)
    agg = ops.cross_attn_agg_sliced_fwd(lay[0][4])
    grads = [torch.empty(n_cams, 256, h, w, device=dev) for h, w in levels]

    def counts():
        sink.count.zero_()
        sink.plans = []
        for i, L in enumerate(lay):
            sink.add_layer(i, L[4])

    def prepare():
        counts()
        sink.prepare()

    def finish():
        prepare()
        sink.reduce(grads)
    it = 5
    t = {
        'cross_attn_plan': timed(lambda: [ops.cross_attn_plan_fwd(pyr, L[0], L[1], L[2], L[3], l2i, synthetic.PC_RANGE, 900, 1600, hh,
                                                                  plan=L[4], query_order=L[4].order) for L in lay], it, nl),
        'cross_attn_agg_sliced': timed(lambda: [ops.cross_attn_agg_sliced_fwd(L[4], agg=agg) for L in lay], it, nl),
        'value_proj_heads_bwd': timed(lambda: [ops.value_proj_heads_bwd(L[5], w_v, b_v, hh, grad_agg=sink.grad_agg_rows(i), beta=beta)
                                               for i, L in enumerate(lay)], it, nl),
        'cross_attn_dot_sliced': timed(lambda: [ops.cross_attn_dot_sliced(L[4], sink.grad_agg_rows(i), dpart=dpart)
                                                for i, L in enumerate(lay)], it, nl),
        'cross_attn_plan_bwd': timed(lambda: [ops.cross_attn_plan_bwd(L[4], dpart, beta, L[0], L[1], L[2], L[3], l2i, synthetic.PC_RANGE,
                                                                      900, 1600) for L in lay], it, nl),
    }
    t_counts, t_prep, t_all = timed(counts, it, 1), timed(prepare, it, 1), timed(finish, it, 1)
    records = int(sink.count.sum().item()) if sink.count.sum().item() else 0
    counts()
    records = int(sink.count.sum().item())
    sink.plans = []
    es = 4
    alg = vis * len(levels) * 4 * (256 // hh) * es + q * (3 + hh * 12 + hh * len(levels) * 4 + n_cams) * 4 + n_cams * 64 + q * hh * 256 * 4
    dot_us = t['cross_attn_dot_sliced']
    kernels = {k: {'us_per_launch': v, 'launches_per_step': nl} for k, v in t.items()}
    kernels['cross_attn_dot_sliced'].update({'gathered_bytes_per_launch': vis * len(levels) * 4 * 256 * es,
                                             'gathered_GBps': vis * len(levels) * 4 * 256 * es / dot_us / 1e3})
    kernels['pyramid_grad_count'] = {'us_per_launch': (t_counts) / nl, 'launches_per_step': nl, 'records_per_step': records,
                                     'note': 'one slot atomic per group of lanes that share a chunk'}
    kernels['pyramid_grad_scan_fill_sort'] = {'us_per_step': t_prep - t_counts}
    kernels['pyramid_grad_reduce'] = {'us_per_step': t_all - t_prep, 'table_bytes_read': records * 1024,
                                      'table_GBps': records * 1024 / (t_all - t_prep) / 1e3,
                                      'gradient_bytes_written': sum(n_cams * 256 * h * w * 4 for h, w in levels),
                                      'note': 'd(pyramid) of all layers in one pass: every pixel written once, no atomics on feature data'}
    # (counter bytes: the committed PMC pass of tools/bench_raw_bwd.py on this workload, while its source hash matches)
    traffic, traffic_source = _pmc_traffic(a, 'gd4d_cross_attn_sliced_bwd.hip', 'r*_pmc_cross_attn_dot_sliced.json') \
        if getattr(a, 'value_dtype', 'fp32') == 'fp32' else (None, None)
    roofline = {'kernel': 'gd4d_cross_attn_dot_sliced (backward gather, per decoder layer)', 'bound': 'hbm',
                'achieved': alg / dot_us / 1e3, 'peak': 8000.0, 'unit': 'GB/s', 'frac': alg / dot_us / 1e3 / 8000.0,
                'traffic': traffic, 'traffic_source': traffic_source,
                'algorithmic_bytes_per_launch': alg, 'us_per_launch': dot_us,
                'note': 'algorithmic bytes as SURVEY 8(d) defines them for the forward gather (Dh = 32 projected channels per corner); '
                        'the kernel gathers 256 raw channels per corner (kernels.cross_attn_dot_sliced.gathered_GBps)'}
    return roofline, kernels


def _events():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def _time_rounds(calls, rounds):
    """HIP events on the launch stream (torch's current stream IS the stream the C ABI launches on) around
    `rounds` x (every call once).  Returns ms per round."""
    for c in calls:
        c()
    e0, e1 = _events()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(rounds):
        for c in calls:
            c()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rounds


def _all_visible_inputs(c0, ops):
    """SURVEY.md 8(d) stress case: every camera is camera 0 and every reference point sits in front of it."""
    b_, q_ = c0['ref'].shape[0], c0['ref'].shape[1]
    n_ = c0['l2i'].shape[1]
    g = torch.Generator().manual_seed(4242)
    ref_av = torch.rand(b_, q_, 3, generator=g)
    ref_av[..., 0] = 0.6 + 0.3 * ref_av[..., 0]
    ref_av[..., 1] = 0.45 + 0.1 * ref_av[..., 1]
    ref_av[..., 2] = 0.6 + 0.1 * ref_av[..., 2]
    ref_av = ref_av.to(c0['ref'].device)
    l2i_av = c0['l2i'][:, :1].expand(-1, n_, -1, -1).contiguous()
    return ref_av, l2i_av, ops.query_order_fwd(ref_av, c0['pc_range'])


def _unique_pixels(mask, uv, shapes):
    """Distinct (camera row, level, pixel) a launch touches: the in-bounds bilinear corners of every visible sample, from the
    kernel's own mask (B, N, Q, Hh, P) and uv (B, N, Q, Hh, P, 2) - mmcv's sampling grid (x = u W - 0.5, corners floor / +1)."""
    b, n = mask.shape[0], mask.shape[1]
    vis = mask.bool()
    rows = torch.arange(b * n, device=mask.device).view(b, n, 1, 1, 1).expand_as(mask)[vis].long()
    u, v = uv[..., 0][vis], uv[..., 1][vis]
    total = 0
    for (h, w) in shapes:
        x, y = u * w - 0.5, v * h - 0.5
        x0, y0 = torch.floor(x).long(), torch.floor(y).long()
        keys = []
        for dx in (0, 1):
            for dy in (0, 1):
                xi, yi = x0 + dx, y0 + dy
                ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
                keys.append(((rows * h + yi) * w + xi)[ok])
        total += int(torch.unique(torch.cat(keys)).numel())
    return total


def _pmc_traffic(a, source, pattern):
    """PMC cannot be read from inside the bench: `traffic` comes from a committed rocprofv3 --pmc pass of the kernel on this
    workload (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction; tools/prof_pmc.sh).  The record carries
    the sha256 of the kernel source it was measured on; if the source has changed since, the figure is stale and is NOT
    reported (traffic = null)."""
    if not (a.queries == 900 and a.frames == 4 and a.levels == 'r50' and a.value_dtype == 'fp32'):
        return None, None
    import glob
    import hashlib
    src_hash = hashlib.sha256(open(os.path.join(ROOT, 'graph-detr4d_amd', 'csrc', source), 'rb').read()).hexdigest()
    for pmc_path in sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)), reverse=True):
        pm = json.load(open(pmc_path))
        if pm.get('kernel_source_sha256') == src_hash:
            return pm['hbm_read_bytes_per_launch'] + pm['hbm_write_bytes_per_launch'], os.path.relpath(pmc_path, ROOT)
    return None, f'no committed PMC pass matches the current {source} (sha256 {src_hash[:12]})'


POST_RANGE = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]


def features_to_boxes(G, tr, regs, feats, query_embed, rig, a, D, dev, stream, steps, windows=5):
    """Features -> boxes as ONE measured request (SURVEY 8(f1) + 8(a) + 8(f2)): the head's feature position embedding
    (dense_heads/detr3d_head_pe.py:495-557; channels-last output, which the decoder gathers in place) -> the 6-layer decoder ->
    the cls / reg epilogue of every layer (:568-612) -> gd4d_nms_free_decode_fwd (core/bbox/coders/nms_free_coder.py:47-118, top 300)
    as one hipGraph per request.  Before every replay the NEXT sample's camera matrices are written into the persistent device
    buffers the graph reads (FeaturePositionEmbedding.refresh_matrices, functional.lidar2img_device) - 1.5 KB, host work of a
    request, inside the timed steps.  Two patterns: `temporal` - the shipped T = 4 pattern: the current frame's 6 cameras keep their
    calibration (their embedding is kept), the 18 past-frame cameras carry the ego motion and are recomputed every sample;
    `every_camera` - nothing kept."""
    import copy
    import numpy as np
    from graph_detr4d_amd import functional as Fn
    from graph_detr4d_amd import ops, synthetic
    nl = len(tr.decoder.layers)
    nn = torch.nn
    cls_b = nn.ModuleList(nn.Sequential(nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU(inplace=True), nn.Linear(256, 256),
                                        nn.LayerNorm(256), nn.ReLU(inplace=True), nn.Linear(256, 10)) for _ in range(nl))
    synthetic.randomise_all_(cls_b, seed=21, std=0.04)
    cls_b = cls_b.to(dev).eval()
    n = feats[0].shape[1]
    metas = []
    for k in range(4):                                   # the past frames' cameras (all but the first 6) move with every sample
        rk = rig.copy()
        rk[6:, :3, 3] += (np.random.RandomState(50 + k).randn(max(n - 6, 0), 3) * np.array([20.0, 20.0, 0.02])).astype(np.float32)
        metas.append(synthetic.make_img_metas(rk, batch=1))
    res = {}
    with torch.no_grad():
        for name, keep in (('temporal', True), ('every_camera', False)):
            pe = G.FeaturePositionEmbedding(pc_range=synthetic.PC_RANGE, channels_last_out=True)
            synthetic.randomise_all_(pe, seed=11, std=0.04)
            pe = pe.to(dev).eval()
            pe.cache_position_embedding = keep

            def request(mt):
                gf = pe(feats, mt)
                states, init, refs = tr(gf, query_embed, reg_branches=regs, img_metas=mt)
                outs = Fn.head_outputs(states, init, refs, cls_b, regs, synthetic.PC_RANGE)
                return ops.nms_free_decode_fwd(outs['all_cls_scores'][-1].contiguous().float(), outs['all_bbox_preds'][-1].contiguous().float(),
                                               POST_RANGE, 300)

            def refresh(mt):
                pe.refresh_matrices(mt, dev)
                Fn.lidar2img_device(mt, query_embed)
            with torch.cuda.stream(stream), Fn.request_slot(7000 + (0 if keep else 1)):
                request(metas[0])                                        # eager: allocations, caches, the kept embedding of sample 0
                want = [t.clone() for t in request(metas[1])]            # (eager result of the sample the graph is captured on)
                torch.cuda.synchronize()
                request(metas[0])
                refresh(metas[1])
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream, capture_error_mode='thread_local'):
                    out = request(metas[1])
                graph.replay()
                torch.cuda.synchronize()
                same = all(torch.equal(x, y) for x, y in zip(out, want))
                turn = [0]

                def step():
                    turn[0] += 1
                    with torch.cuda.stream(stream), Fn.request_slot(7000 + (0 if keep else 1)):
                        refresh(metas[turn[0] % 4])
                        graph.replay()
                els = sorted(D.timed_steps(step, steps, 3 if i == 0 else 0, dev, {}) for i in range(windows))
                ops.check_handoff()
                kept_boxes = int(out[3].sum().item())
            ms = els[len(els) // 2] / steps * 1e3
            res[name] = {'ms_per_sample': ms, 'samples_per_s': 1e3 / ms, 'ms_min': els[0] / steps * 1e3, 'ms_max': els[-1] / steps * 1e3,
                         'graph_equals_eager': bool(same), 'boxes_kept_of_300': kept_boxes}
            del graph, pe, out, want
            torch.cuda.empty_cache()
    res['note'] = ('head position embedding (channels-last out) -> decoder gathering in place -> cls / reg epilogue of every layer -> '
                   'gd4d_nms_free_decode_fwd (top 300), one hipGraph per request; the next sample\'s camera matrices are refreshed (host, '
                   '1.5 KB) before every replay, inside the timed steps; temporal = the 6 current-frame cameras keep their embedding, the 18 '
                   'past-frame cameras are recomputed; every_camera = nothing kept')
    return res


def fused_kernel_roofline(tr, regs, feats, query_embed, metas, a, ops, synthetic):
    """Re-run the decoder once, intercepting the sample-aggregate inputs of every layer; then time each layer's launch of
    the step's dominant kernel with HIP events on the launch stream and count V from the kernel's own mask.

    The default step (GD4D_PROJECT=late) aggregates the raw features per head and projects afterwards: its dominant kernel
    is gd4d_cross_attn_agg_fwd.  With GD4D_PROJECT=early it is gd4d_cross_attn_fwd on projected values.  Whichever runs
    in the step is `roofline`; the other form is measured on the same inputs and reported under `kernels`."""
    from graph_detr4d_amd import functional as Fn
    early_cap, late_cap = [], []
    orig_early, orig_late = Fn.sample_aggregate, Fn.LateValues.aggregate

    def spy_early(value, shapes, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, order=None):
        nl_pix = sum(h * w for h, w in shapes)
        early_cap.append(dict(head_major=(value.shape[2] == nl_pix and value.shape[1] != nl_pix), value=value, shapes=shapes,
                              ref=ref.contiguous(), offsets=offsets.contiguous(), attn=attn_logits.contiguous(),
                              cam=cam_logits.contiguous(), l2i=lidar2img, pc_range=pc_range, img_h=img_h, img_w=img_w, order=order))
        return orig_early(value, shapes, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, order=order)

    def spy_late(self, module, ref, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=None, **vp):
        late_cap.append(dict(late=self, cl=self.cl, shapes=self.shapes, module=module, ref=ref.contiguous(), offsets=offsets.contiguous(),
                             attn=attn_logits.contiguous(), cam=cam_logits.contiguous(), l2i=lidar2img, pc_range=module.pc_range,
                             img_h=img_h, img_w=img_w, order=order, vp=vp))
        return orig_late(self, module, ref, offsets, attn_logits, cam_logits, lidar2img, img_h, img_w, order=order, **vp)
    Fn.sample_aggregate, Fn.LateValues.aggregate = spy_early, spy_late
    try:
        with torch.no_grad():
            tr(feats, query_embed, reg_branches=regs, img_metas=metas)
    finally:
        Fn.sample_aggregate, Fn.LateValues.aggregate = orig_early, orig_late
    torch.cuda.synchronize()
    late_mode = bool(late_cap)
    captured = late_cap if late_mode else early_cap
    mods = [m for layer in tr.decoder.layers for m in layer.attentions if hasattr(m, 'value_proj')]
    hh = mods[0].num_heads
    kernels = {}
    rounds = 20

    def side_bytes(q, n, nl, p):
        return q * (3 + hh * p * 3 + hh * nl * p + n) * 4 + n * 64 + q * 256 * 4

    # ---------------- projected-value gather (gd4d_cross_attn_fwd), SURVEY.md 8(d) formula ----------------
    def measure_early(caps, tag):
        """caps: per-layer dicts with value / shapes / query-side inputs.  Rotating through the layers' value tensors
        (L x 757 MB >> 256 MB Infinity Cache) keeps every timed launch cache-cold like in a decoder step."""
        calls, per_layer, tot = [], [], 0.0
        for c in caps:
            call = (lambda c: (lambda **kw: ops.cross_attn_fwd(c['value'], c['shapes'], c['ref'], c['offsets'], c['attn'], c['cam'],
                                                               c['l2i'], c['pc_range'], c['img_h'], c['img_w'],
                                                               head_major=c['head_major'], query_order=c['order'], **kw)))(c)
            out, mask = call(want_mask=True)
            b, n, q, _, p = mask.shape
            nl = len(c['shapes'])
            dh, es = c['value'].shape[-1], c['value'].element_size()
            v = int(mask.sum().item()) * nl                  # visible (cam, query, head, level, point) tuples
            alg = min(v * 4 * dh * es, c['value'].numel() * es) + side_bytes(q, n, nl, p)
            calls.append((lambda call, out: (lambda: call(out=out)))(call, out))
            per_layer.append(dict(visible_tuples=v, visible_frac=v / (mask.numel() * nl), alg_bytes=alg))
            tot += alg
        ms = _time_rounds(calls, rounds)
        for d in per_layer:
            d['us_mean'] = ms / len(calls) * 1e3
        return dict(per_layer=per_layer, alg_bytes_per_launch=tot / len(calls), us_per_launch=ms / len(calls) * 1e3,
                    gbs=tot / ms / 1e6, frac=tot / ms / 1e6 / HBM_PEAK_GBS, launches=len(calls))

    def early_all_visible(caps):
        c0 = caps[0]
        ref_av, l2i_av, order_av = _all_visible_inputs(c0, ops)
        run_av = lambda c, **kw: ops.cross_attn_fwd(c['value'], c['shapes'], ref_av, c['offsets'], c['attn'], c['cam'],   # noqa: E731
                                                    l2i_av, c['pc_range'], c['img_h'], c['img_w'], head_major=c['head_major'],
                                                    query_order=order_av, **kw)
        out_av, mask_av = run_av(c0, want_mask=True)
        nl_ = len(c0['shapes'])
        v_av = int(mask_av.sum().item()) * nl_
        es_ = c0['value'].element_size()
        alg_av = min(v_av * 4 * c0['value'].shape[-1] * es_, c0['value'].numel() * es_) + \
            side_bytes(mask_av.shape[2], mask_av.shape[1], nl_, mask_av.shape[4])
        ms = _time_rounds([(lambda c: (lambda: run_av(c, out=out_av)))(c) for c in caps], 5)
        us_av = ms / len(caps) * 1e3
        return dict(visible_frac=v_av / (mask_av.numel() * nl_), alg_bytes=alg_av, us_per_launch=us_av,
                    gbs=alg_av / us_av / 1e3, frac=alg_av / us_av / 1e3 / HBM_PEAK_GBS)

    if not late_mode:
        e = measure_early(early_cap, 'early')
        traffic, traffic_source = _pmc_traffic(a, 'gd4d_cross_attn.hip', 'r*_pmc_cross_attn.json')
        roofline = dict(kernel='gd4d::cross_attn_fwd_block (fused project+sample+aggregate on projected values)', bound='hbm',
                        achieved=e['gbs'], peak=HBM_PEAK_GBS, unit='GB/s', frac=e['frac'], traffic=traffic,
                        traffic_source=traffic_source, alg_bytes_per_launch=e['alg_bytes_per_launch'],
                        us_per_launch=e['us_per_launch'], launches_per_step=e['launches'])
        kernels['cross_attn_fwd_per_layer'] = e['per_layer']
        try:
            if not a.no_stress:
                kernels['cross_attn_fwd_all_visible'] = early_all_visible(early_cap)
        except Exception as ex:                               # secondary figure: report, never fail the bench line
            kernels['cross_attn_fwd_all_visible'] = {'error': f'{type(ex).__name__}: {ex}'}
    else:
        # ---------------- aggregate-then-project: the step's gather of RAW features ----------------
        # roofline.frac is SURVEY.md 8(d) verbatim: V * 4 corners * Dh * e + the query-side terms + the output, V counted from
        # the kernel's own bit-exact mask - the bytes the reference's formulation (gather of PROJECTED values) would have
        # to move.  This formulation deliberately gathers all C = 8 Dh channels of a corner (so that value_proj over the
        # pyramid disappears from the step); what it moves is reported beside it: gathered_bytes (V * 4 * C * e, uncapped),
        # unique_bytes (every touched (pixel, 128-byte line) once - the compulsory HBM reads of a launch) and the counter
        # traffic of the committed PMC pass.
        sliced = late_cap[0]['late'].mode == 'sliced'
        items = os.environ.get('GD4D_PLAN', 'items') != 'pairs'                 # the plan form the step uses (Fn.LateValues.aggregate)
        calls, per_layer, tot8d, tot_c, tot_u = [], [], 0.0, 0.0, 0.0
        plan_calls = []
        for c in late_cap:
            if sliced:
                coarse_on = bool(c['vp'].get('coarse'))                          # the step gathers the coarse levels from projected rows
                plan, mask, uv = ops.cross_attn_plan_fwd(c['late'].pyramid, c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'], c['pc_range'],
                                                         c['img_h'], c['img_w'], hh, query_order=c['order'], want_mask=True, want_uv=True,
                                                         items=items or coarse_on)
                if coarse_on:
                    # this layer's projected coarse rows (its own buffer here; the step re-uses one) - made once, outside the timing
                    late_c = c['late']
                    rows_c = torch.empty_like(late_c.coarse.rows)
                    ops.value_proj_guest_fwd(ops.chain_guest(late_c.coarse_src, ops.value_proj_image(c['module'].value_proj.weight,
                                                                                                     c['module'].value_proj.bias), rows_c))
                    cv = ops.CoarseValues(rows_c, late_c.shapes[2:])
                    agg_buf, pagg_buf = ops.cross_attn_agg_coarse_fwd(plan, cv)
                    calls.append((lambda plan, cv, agg_buf, pagg_buf: (lambda: ops.cross_attn_agg_coarse_fwd(plan, cv, agg=agg_buf, pagg=pagg_buf)))(
                        plan, cv, agg_buf, pagg_buf))
                else:
                    agg_buf = ops.cross_attn_agg_sliced_fwd(plan)
                    calls.append((lambda plan, agg_buf: (lambda: ops.cross_attn_agg_sliced_fwd(plan, agg=agg_buf)))(plan, agg_buf))
                plan_calls.append((lambda c, plan: (lambda: ops.cross_attn_plan_fwd(
                    c['late'].pyramid, c['ref'], c['offsets'], c['attn'], c['cam'], c['l2i'], c['pc_range'], c['img_h'], c['img_w'], hh,
                    query_order=c['order'], plan=plan, items=plan.items)))(c, plan))
                es = c['late'].pyramid.tensors[0].element_size()
            else:
                # (as the step launches it: with value_proj of the aggregates in the epilogue when the step does that)
                run = (lambda c: (lambda **kw: ops.cross_attn_agg_fwd(c['cl'], c['shapes'], c['ref'], c['offsets'], c['attn'], c['cam'],
                                                                      c['l2i'], c['pc_range'], c['img_h'], c['img_w'], hh,
                                                                      query_order=c['order'], **c['vp'], **kw)))(c)
                res = run(want_mask=True, want_uv=True)
                mask, uv = res[-2], res[-1]
                calls.append(run)
                es = c['cl'].element_size()                                     # 4, or 2 with value_dtype='bf16' (bf16 storage)
            b, n, q, _, p = mask.shape
            nl = len(c['shapes'])
            v = int(mask.sum().item()) * nl
            pyramid_bytes = b * n * sum(h * w for h, w in c['shapes']) * 256 * es
            side = side_bytes(q, n, nl, p)
            alg8d = min(v * 4 * (256 // hh) * es, pyramid_bytes) + side
            uniq = _unique_pixels(mask, uv, c['shapes']) * 256 * es
            # bytes through the L1s: a raw corner is all 256 channels (8 slices x 128 B); with the coarse levels projected first a
            # corner of levels 2-3 is the head's own 32 fp32 channels
            gathered = v * 4 * 256 * es
            if sliced and c['vp'].get('coarse'):
                gathered = (v // nl) * 2 * 4 * 256 * es + (v // nl) * 2 * 4 * (256 // hh) * 4
            per_layer.append(dict(visible_tuples=v, visible_frac=v / (mask.numel() * nl), alg_bytes=alg8d,
                                  gathered_bytes=gathered, unique_bytes=uniq))
            tot8d += alg8d
            tot_c += gathered
            tot_u += uniq
        ms = _time_rounds(calls, rounds)
        launches = len(calls)
        for d in per_layer:
            d['us_mean'] = ms / launches * 1e3
        if sliced:
            traffic, traffic_source = _pmc_traffic(a, 'gd4d_cross_attn_sliced.hip', 'r*_pmc_cross_attn_sliced.json')
            coarse_step = bool(late_cap[0]['vp'].get('coarse'))
            kname = ('gd4d::cross_attn_agg_items_coarse_kernel (levels 0-1: raw features, one workgroup per (query, 32-channel slice); levels 2-3: '
                     'rows the layer\'s value_proj was applied to by guest workgroups of the previous row-chain launch, a ninth workgroup per query; '
                     if coarse_step else
                     ('gd4d::cross_attn_agg_items_kernel' if items else 'gd4d::cross_attn_agg_sliced_kernel') +
                     ' (gather of raw features, one workgroup per (query, 32-channel slice); ') + \
                    'projection / mask / weights come from gd4d::cross_attn_plan_kernel, timed beside it)'
            ms_plan = _time_rounds(plan_calls, rounds)
            kernels['cross_attn_plan'] = dict(us_per_launch=ms_plan / launches * 1e3, launches_per_step=launches,
                                              note='projection + mask + softmax + bilinear corners, once per (layer, query)')
        else:
            traffic, traffic_source = _pmc_traffic(a, 'gd4d_cross_attn_late.hip', 'r*_pmc_cross_attn_agg.json')
            kname = 'gd4d::cross_attn_agg_kernel (fused project + sample + per-head aggregate of raw features)'
        us = ms / launches * 1e3
        roofline = dict(kernel=kname, bound='hbm', achieved=tot8d / ms / 1e6, peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=tot8d / ms / 1e6 / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_source,
                        alg_bytes_per_launch=tot8d / launches, us_per_launch=us, launches_per_step=launches,
                        alg_bytes_rule='SURVEY 8(d) verbatim: V * 4 * Dh * e (V from the bit-exact mask, capped at the pyramid) + '
                                       'Q * (3 + Hh*P*3 + Hh*L*P + N) * 4 + N * 64 + Q * C * 4',
                        gathered_bytes_per_launch=tot_c / launches,
                        frac_on_gathered_bytes=tot_c / ms / 1e6 / HBM_PEAK_GBS,
                        unique_bytes_per_launch=tot_u / launches, frac_on_unique_bytes=tot_u / ms / 1e6 / HBM_PEAK_GBS,
                        frac_on_counter_bytes=None if traffic is None else traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        note=('levels 0-1: this kernel gathers C = 8 Dh channels per corner (8 x the 8(d) bytes) so that value_proj over the '
                              'pyramid (97 GFLOP + 757 MB written per layer) leaves the step; levels 2-3 (6 % of the pixels): rows value_proj was '
                              'applied to by guest workgroups of the previous row-chain launch (43 800 rows, 5.7 GFLOP per layer), 8(d)\'s own '
                              'bytes; frac prices the launch against the 8(d) bytes of all four levels'
                              if sliced and late_cap[0]['vp'].get('coarse') else
                              'this kernel gathers C = 8 Dh channels per corner (8 x the 8(d) bytes) so that value_proj over the '
                              'pyramid (97 GFLOP + 757 MB written per layer) leaves the step; frac prices its time against the '
                              '8(d) bytes all the same'))
        kernels['cross_attn_agg_per_layer'] = per_layer
        with torch.no_grad():
            # the two small kernels around it: the per-sample copy of the pyramid and value_proj of the aggregates
            try:
                late0 = late_cap[0]['late']
                cl = late0.cl
                if cl is None:
                    kernels['pyramid_copy'] = dict(us=0.0, launches_per_step=0, note='caller-owned channels-last levels are gathered in place: no copy')
                else:
                    vals = [f.contiguous() for f in feats][:2 if late0.partial else None]   # (coarse levels projected: the copy leaves them out)
                    cus = torch.cuda.get_device_properties(cl.device).multi_processor_count
                    copy_cus = int(os.environ.get('GD4D_COPY_CUS') or max(8, (cus * 7 // 8) // 8 * 8))      # as Fn.LateValues launches it
                    copy = ops.pyramid_slice_planar_fwd if sliced else ops.pyramid_channels_last_fwd
                    ms_cl = _time_rounds([lambda: copy(vals, out=cl, max_cus=copy_cus, out_dtype=cl.dtype)], 5)
                    cl_bytes = cl.numel() * (4 + cl.element_size())             # fp32 NCHW read, channels-last copy written
                    kernels['pyramid_copy'] = dict(kernel='gd4d_pyramid_slice_planar_fwd' if sliced else 'gd4d_pyramid_channels_last_fwd',
                                                   us=ms_cl * 1e3, bytes=cl_bytes, gbs=cl_bytes / ms_cl / 1e6,
                                                   frac=cl_bytes / ms_cl / 1e6 / HBM_PEAK_GBS, launches_per_step=1, compute_units=copy_cus,
                                                   note='alone on the device; in the step the first layer\'s query side runs on the other CUs')
            except Exception as ex:
                kernels['pyramid_copy'] = {'error': f'{type(ex).__name__}: {ex}'}
            # all-visible stress case of this kernel (8 x the bytes of the projected-value form per corner: its worst case)
            if not a.no_stress:
                try:
                    c0 = late_cap[0]
                    ref_av, l2i_av, order_av = _all_visible_inputs(c0, ops)
                    if sliced:
                        plan_av, mask_av = ops.cross_attn_plan_fwd(c0['late'].pyramid, ref_av, c0['offsets'], c0['attn'], c0['cam'], l2i_av,
                                                                   c0['pc_range'], c0['img_h'], c0['img_w'], hh, query_order=order_av,
                                                                   want_mask=True, items=items or bool(late_cap[0]['vp'].get('coarse')))
                        if late_cap[0]['vp'].get('coarse'):
                            cv0 = c0['late'].coarse                                 # (rows of the last layer's projection: any will do for timing)
                            agg_av, pagg_av = ops.cross_attn_agg_coarse_fwd(plan_av, cv0)
                            av_calls = [lambda: ops.cross_attn_agg_coarse_fwd(plan_av, cv0, agg=agg_av, pagg=pagg_av)]
                        else:
                            agg_av = ops.cross_attn_agg_sliced_fwd(plan_av)
                            av_calls = [lambda: ops.cross_attn_agg_sliced_fwd(plan_av, agg=agg_av)]
                        es_ = c0['late'].pyramid.tensors[0].element_size()
                    else:
                        run_av = lambda c, **kw: ops.cross_attn_agg_fwd(c['cl'], c['shapes'], ref_av, c['offsets'], c['attn'], c['cam'], l2i_av,   # noqa: E731
                                                                        c['pc_range'], c['img_h'], c['img_w'], hh, query_order=order_av,
                                                                        **c['vp'], **kw)
                        mask_av = run_av(c0, want_mask=True)[-1]
                        av_calls = [(lambda c: (lambda: run_av(c)))(c) for c in late_cap]
                        es_ = c0['cl'].element_size()
                    nl_ = len(c0['shapes'])
                    v_av = int(mask_av.sum().item()) * nl_
                    alg_av = v_av * 4 * (256 // hh) * es_ + side_bytes(mask_av.shape[2], mask_av.shape[1], nl_, mask_av.shape[4])
                    ms_av = _time_rounds(av_calls, 3)
                    us_av = ms_av / len(av_calls) * 1e3
                    kernels['cross_attn_agg_all_visible'] = dict(visible_frac=v_av / (mask_av.numel() * nl_), alg_bytes=alg_av,
                                                                 gathered_bytes=v_av * 4 * 256 * es_, us_per_launch=us_av,
                                                                 gbs=alg_av / us_av / 1e3, frac=alg_av / us_av / 1e3 / HBM_PEAK_GBS,
                                                                 l2_level_gbs=v_av * 4 * 256 * es_ / us_av / 1e3)
                except Exception as ex:
                    kernels['cross_attn_agg_all_visible'] = {'error': f'{type(ex).__name__}: {ex}'}
            # the projected-value form on the same query-side inputs (not part of the step): value_proj for every layer,
            # then gd4d_cross_attn_fwd - SURVEY 8(d)'s own formula
            try:
                hm = Fn.use_head_major(mods[0].value_dtype)
                vals = [f.contiguous() for f in feats]
                proj = ops.value_proj_multi_fwd(vals, [m.value_proj.weight for m in mods], [m.value_proj.bias for m in mods],
                                                mods[0].value_dtype, num_heads=hh, head_major=hm)
                caps = []
                for c, v in zip(late_cap, proj):
                    bn = v.shape[0]
                    caps.append(dict(c, value=(v if hm else v.view(bn, -1, hh, 256 // hh)), head_major=hm))
                e = measure_early(caps, 'early')
                kernels['cross_attn_fwd_projected_values'] = dict(
                    note='gd4d_cross_attn_fwd on projected values (GD4D_PROJECT=early), same query-side inputs; not in the step',
                    alg_bytes_per_launch=e['alg_bytes_per_launch'], us_per_launch=e['us_per_launch'], gbs=e['gbs'], frac=e['frac'])
                if not a.no_stress:
                    kernels['cross_attn_fwd_projected_values_all_visible'] = early_all_visible(caps)
                del proj, caps
            except Exception as ex:
                kernels['cross_attn_fwd_projected_values'] = {'error': f'{type(ex).__name__}: {ex}'}
    # value_proj over the pyramid (the early path's other large kernel; the training step uses it): launched as the decoder
    # launches it (one multi-layer launch per group of layers, all CUs), HIP events on the launch stream.  Algorithmic bytes
    # of a launch: the pyramid read once + one value tensor written per layer; flops: 2 * rows * 256 * 256 per layer (x3 on
    # the matrix pipe: split-bf16).
    try:
        vals = [f.contiguous() for f in feats]
        groups = Fn.pipeline_groups(os.environ.get('GD4D_PREPROJECT', 'auto'), len(mods)) or (len(mods),)
        hm = Fn.use_head_major(mods[0].value_dtype)

        def run(ms):
            return ops.value_proj_multi_fwd(vals, [m.value_proj.weight for m in ms], [m.value_proj.bias for m in ms],
                                            ms[0].value_dtype, num_heads=ms[0].num_heads, head_major=hm,
                                            bf16_math=ms[0].value_dtype == torch.bfloat16)
        bounds, lo = [], 0
        for g_ in groups:
            bounds.append((lo, lo + g_))
            lo += g_
        with torch.no_grad():
            outs0 = run(mods[bounds[0][0]:bounds[0][1]])
            ms_all = _time_rounds([(lambda a_, b_: (lambda: run(mods[a_:b_])))(a_, b_) for a_, b_ in bounds], 3)
        us_all = ms_all * 1e3                                  # all layers of the decoder
        rd = sum(f.numel() * f.element_size() for f in vals)
        wr = outs0[0].numel() * outs0[0].element_size()
        rows = rd // (256 * 4)
        fl = 2.0 * rows * 256 * 256
        alg = len(bounds) * rd + len(mods) * wr
        kernels['value_proj_fwd'] = dict(in_step=not late_mode, layer_groups=list(groups), us_all_layers=us_all,
                                         us_per_layer=us_all / len(mods),
                                         alg_bytes_all_layers=alg, hbm_gbs=alg / us_all / 1e3,
                                         hbm_frac=alg / us_all / 1e3 / HBM_PEAK_GBS,
                                         gflop_per_layer=fl / 1e9, mfma_tflops_x3=3 * fl * len(mods) / us_all / 1e6,
                                         mfma_frac_of_2500=3 * fl * len(mods) / us_all / 1e6 / 2500.0,
                                         launches_per_step=len(bounds) if not late_mode else 0,
                                         note='bounded by the sustained (power-limited) matrix rate and the HBM write rate '
                                              'together: profiles/r02_value_proj.md')
        del outs0
    except Exception as e:                                    # secondary figure: report, never fail the bench line
        kernels['value_proj_fwd'] = {'error': f'{type(e).__name__}: {e}'}
    if roofline is not None:
        # what a reader should see without opening `kernels`: the north_star target, how much more the counters saw than the
        # 8(d) rule prices, and - when the step runs the raw-feature form - the projected-value kernel (the formulation the
        # target is priced on, not in the step) as a named sibling
        roofline['target'] = 0.70
        roofline['counter_over_alg'] = None if not roofline.get('traffic') else roofline['traffic'] / roofline['alg_bytes_per_launch']
        pv = kernels.get('cross_attn_fwd_projected_values') if late_mode else None
        if isinstance(pv, dict) and 'frac' in pv:
            roofline['projected_value_form'] = dict(kernel='gd4d::cross_attn_fwd_kernel on projected values (needs value_proj over the '
                                                           'pyramid per layer: not in the step)', frac=pv['frac'],
                                                    us_per_launch=pv.get('us_per_launch'), in_step=False)
    return roofline, kernels


if __name__ == '__main__':
    main()

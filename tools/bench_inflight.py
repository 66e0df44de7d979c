"""Dev: S independent samples in flight on S HIP streams - does the latency-bound query side of one sample hide under the
aggregate launches of another?  python tools/bench_inflight.py [S ...]
MODE=graphs (default): one hipGraph per request, replayed on its own stream (what bench.py --inflight does).
MODE=fork / first_on_cur: ONE graph with a branch per request - the request's own fork to its copy stream makes
hipStreamEndCapture recurse without end on this runtime (MODE=nested reproduces it with plain torch ops; MODE=nested2,
where the inner stream is forked from the origin and only waited for, captures fine)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402
import graph_detr4d_amd as G                           # noqa: E402
from graph_detr4d_amd import functional as Fn, synthetic                 # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 3]
    vd = os.environ.get('VALUE_DTYPE', 'fp32')
    tr, regs = bench.build_decoder(G, 24, 6, vd, 1002)
    tr, regs = tr.to(dev), regs.to(dev)
    rig = synthetic.camera_rig(4)
    metas = synthetic.make_img_metas(rig, batch=1)
    smax = max(sizes)
    feats = [[f.to(dev) for f in synthetic.feature_pyramid(24, synthetic.R50_LEVELS, seed=1002 + i)] for i in range(smax)]
    qe = [torch.randn(900, 512, generator=torch.Generator().manual_seed(5 + i)).to(dev) for i in range(smax)]
    streams = [torch.cuda.Stream(dev) for _ in range(smax)]
    dummy = torch.zeros(64, device=dev)
    with torch.no_grad():
        ref = [tr(feats[i], qe[i], reg_branches=regs, img_metas=metas) for i in range(smax)]
        torch.cuda.synchronize()
        for s_n in sizes:
            def step():
                if os.environ.get('MODE') == 'plain':
                    return [tr(feats[0], qe[0], reg_branches=regs, img_metas=metas)]
                cur = torch.cuda.current_stream()
                mode = os.environ.get('MODE', 'first_on_cur')
                outs = []
                if mode == 'dummy':
                    dummy.add_(1.0)
                lo = 1 if mode == 'first_on_cur' else 0
                for i in range(lo, s_n):
                    streams[i].wait_stream(cur)
                    if os.environ.get('PREFORK', '1') == '1':      # the runtime's EndCapture recurses forever on nested forks
                        with torch.cuda.stream(streams[i]):
                            Fn._companion_stream(Fn._SIDE_STREAMS, dev).wait_stream(cur)
                if lo:
                    with Fn.request_slot(0):
                        outs.append(tr(feats[0], qe[0], reg_branches=regs, img_metas=metas))
                for i in range(lo, s_n):
                    with torch.cuda.stream(streams[i]), Fn.request_slot(i):
                        if mode == 'simple':
                            outs.append((qe[i] @ qe[i].t(), qe[i] * 2))
                        elif mode == 'nested2':          # inner is a child of the ORIGIN; the request stream only waits for it
                            a_ = qe[i] * 3
                            inner = Fn._companion_stream(Fn._SIDE_STREAMS, dev)
                            with torch.cuda.stream(inner):
                                b_ = qe[i] @ qe[i].t()
                                ev = torch.cuda.Event()
                                ev.record(inner)
                            c_ = a_ + 1
                            streams[i].wait_event(ev)
                            d_ = b_ * 2
                            streams[i].wait_stream(inner)
                            outs.append((d_, c_))
                        elif mode == 'nested':
                            a_ = qe[i] * 3
                            inner = Fn._companion_stream(Fn._SIDE_STREAMS, dev)
                            inner.wait_stream(streams[i])
                            with torch.cuda.stream(inner):
                                b_ = qe[i] @ qe[i].t()
                                ev = torch.cuda.Event()
                                ev.record(inner)
                            c_ = a_ + 1
                            streams[i].wait_event(ev)
                            d_ = b_ * 2
                            streams[i].wait_stream(inner)
                            outs.append((d_, c_))
                        else:
                            outs.append(tr(feats[i], qe[i], reg_branches=regs, img_metas=metas))
                for i in range(lo, s_n):
                    cur.wait_stream(streams[i])
                if mode == 'dummy':
                    dummy.add_(1.0)
                return outs
            if os.environ.get('MODE', 'graphs') == 'graphs':
                # one hipGraph per sample, replayed on its own stream (a single graph with nested forks - request
                # stream -> its copy stream - sends this runtime's EndCapture into an endless recursion)
                graphs, outs = [], []
                for i in range(s_n):
                    with torch.cuda.stream(streams[i]), Fn.request_slot(i):
                        tr(feats[i], qe[i], reg_branches=regs, img_metas=metas)
                    torch.cuda.synchronize()
                    gi = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gi, stream=streams[i]), Fn.request_slot(i):
                        outs.append(tr(feats[i], qe[i], reg_branches=regs, img_metas=metas))
                    graphs.append(gi)

                class _G:
                    @staticmethod
                    def replay():
                        cur = torch.cuda.current_stream()
                        for i in range(s_n):
                            streams[i].wait_stream(cur)
                            with torch.cuda.stream(streams[i]):
                                graphs[i].replay()
                        for i in range(s_n):
                            cur.wait_stream(streams[i])
                g = _G
            else:
                w = torch.cuda.Stream(dev)
                w.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(w):
                    step()
                torch.cuda.current_stream().wait_stream(w)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs = step()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            ok = os.environ.get('MODE') in ('simple', 'nested', 'nested2') or all(torch.equal(o[0], r[0]) and torch.equal(o[1], r[1]) for o, r in zip(outs, ref))
            t0 = time.perf_counter()
            n = int(os.environ.get('REPLAYS', '50'))
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print(f'inflight {s_n}: {ms:.3f} ms per step of {s_n} samples = {ms / s_n:.3f} ms per sample, '
                  f'{s_n / ms * 1e3:.1f} samples/s, identical to one-at-a-time: {ok}', flush=True)


main()

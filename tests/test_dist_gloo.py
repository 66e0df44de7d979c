"""N > 1 path of the bench (replicas sharded by sample, barrier, MAX-reduced timing) with two gloo
processes on CPU."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from graph_detr4d_amd import dist as D
    r, w = D.init(backend='gloo')
    assert (r, w) == (rank, world)
    idx = D.sample_indices(7, r, w)
    seed = D.sample_seed(1002, r)
    calls = []

    def run():                      # rank 1 is the slow rank
        calls.append(1)
        time.sleep(0.02 if r == 1 else 0.001)
    elapsed = D.timed_steps(run, steps=5, warmup=2, device=None)
    mx = D.max_over_ranks(float(r + 1))
    thr = D.aggregate_throughput(1, 5, w, elapsed)
    q.put((r, idx, seed, len(calls), elapsed, mx, thr))
    D.shutdown()


def test_two_rank_replicas_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, idx0, seed0, n0, e0, mx0, thr0), (r1, idx1, seed1, n1, e1, mx1, thr1) = res
    assert sorted(idx0 + idx1) == list(range(7)) and not set(idx0) & set(idx1)      # disjoint, complete
    assert (seed0, seed1) == (1002, 1003)
    assert n0 == n1 == 7                                   # 2 warm-up + exactly 5 timed steps
    assert abs(e0 - e1) < 1e-9 and e0 >= 5 * 0.02 * 0.9    # both ranks report the slow rank's time
    assert mx0 == mx1 == 2.0
    assert abs(thr0 - 2 * 5 / e0) < 1e-9                   # whole-job aggregate


def test_single_process_helpers():
    from graph_detr4d_amd import dist as D
    assert D.sample_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.max_over_ranks(3.5) == 3.5
    n = []
    e = D.timed_steps(lambda: n.append(1), steps=3, warmup=1)
    assert len(n) == 4 and e >= 0


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from graph_detr4d_amd import dist as D
    D.init(backend='gloo')
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    net[2].bias.requires_grad_(False)                       # frozen parameter is skipped
    x = torch.full((3, 8), float(rank + 1))
    net(x).sum().backward()
    net[0].bias.grad = None                                  # a parameter without grad on this rank
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    red = D.FlatGradAllReducer(net.parameters(), max_extras=2)
    red.bind()                                                # the address a hipGraph capture would record ...
    ptr = red.flat.data_ptr()
    extras = red.reduce(extras=torch.tensor([float(rank + 1), 10.0]))
    assert red.flat.data_ptr() == ptr                         # ... never moves (extras used to reallocate)
    try:
        red.reduce(extras=torch.ones(3))
        raise AssertionError('more extras than max_extras must be refused')
    except ValueError:
        pass
    first = [None if p.grad is None else p.grad.numpy().copy() for p in net.parameters()]
    # second step: .grad is now a view of the flat buffer - zero_grad is one fill, autograd accumulates in place
    bound = [p.grad for p in net.parameters() if p.requires_grad]
    red.zero_grad()
    assert all(float(g.abs().sum()) == 0.0 for g in bound)
    net(x).sum().backward()
    assert all(p.grad is g for p, g in zip((p for p in net.parameters() if p.requires_grad), bound))
    red.reduce()
    second = [None if p.grad is None else p.grad.numpy().copy() for p in net.parameters()]
    # the head loss's normalisers: one 2-element all-reduce (mean over the ranks)
    from graph_detr4d_amd.criterion import Detr3DCriterion
    avg = Detr3DCriterion().normalisers([3 + 4 * rank], 900, torch.device('cpu'))
    q.put((rank, [None if g is None else g.numpy() for g in local], first, extras.numpy(), red.bytes_per_step(),
           second, avg.numpy()))
    D.shutdown()


def test_flat_gradient_allreduce_two_ranks():
    import numpy as np
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, loc0, avg0, ex0, nbytes, sec0, norm0), (_, loc1, avg1, ex1, _, sec1, norm1) = res
    np.testing.assert_allclose(norm0, [5.0, 5.0])             # (3 + 7) / 2 positives, bg_cls_weight = 0
    np.testing.assert_allclose(norm1, [5.0, 5.0])
    for s0, s1, l0, l1 in zip(sec0, sec1, loc0, loc1):
        if s0 is None:
            continue
        np.testing.assert_array_equal(s0, s1)
        if l0 is not None and l1 is not None:                 # second step: every parameter has a gradient again
            np.testing.assert_allclose(s0, (l0 + l1) / 2, rtol=1e-6)
    assert nbytes == (8 * 16 + 16 + 16 * 4) * 4             # frozen bias excluded
    for a0, a1, l0, l1 in zip(avg0, avg1, loc0, loc1):
        if a0 is None:                                        # the frozen parameter
            assert a1 is None
            continue
        np.testing.assert_array_equal(a0, a1)                 # both ranks hold the same averaged gradient
        z = np.zeros_like(a0)
        np.testing.assert_allclose(a0, ((z if l0 is None else l0) + (z if l1 is None else l1)) / 2, rtol=1e-6)
    np.testing.assert_allclose(ex0, [3.0, 20.0])              # extras are summed, not averaged
    np.testing.assert_allclose(ex1, [3.0, 20.0])


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import copy
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from graph_detr4d_amd import dist as D
    D.init(backend='gloo')
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                              torch.nn.Linear(16, 4))
    net2 = copy.deepcopy(net)
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(rank + 1))
    # single-shot reducer
    one = D.FlatGradAllReducer(net.parameters(), max_extras=1)
    one.bind()
    net(x).square().sum().backward()
    ex1 = one.reduce(extras=torch.tensor([float(rank + 1)]))
    # bucketed reducer, last layer first, all-reduces launched from post-accumulate hooks during backward
    buckets = [list(net2[4].parameters()), list(net2[2].parameters())]      # net2[0] is left to the tail bucket
    two = D.FlatGradAllReducer(net2.parameters(), max_extras=1, buckets=buckets)
    two.install_hooks()
    ptr = two.flat.data_ptr()
    launched = []
    for _ in range(2):                                                       # two steps: hooks re-arm
        two.zero_grad()
        two.begin_step()
        net2(x).square().sum().backward()
        launched.append(list(two._launched))                                 # buckets started by hooks, before finish()
        ex2 = two.finish(extras=torch.tensor([float(rank + 1)]))
    assert two.flat.data_ptr() == ptr
    g1 = {n: p.grad.clone() for n, p in net.named_parameters()}
    g2 = {n: p.grad.clone() for n, p in net2.named_parameters()}
    same = all(torch.equal(g1[n], g2[n]) for n in g1)
    # the back-to-back form (after a captured backward) on the same gradients
    two.zero_grad()
    net2(x).square().sum().backward()
    two.remove_hooks()
    two.reduce_buckets()
    g3 = {n: p.grad.clone() for n, p in net2.named_parameters()}
    same3 = all(torch.equal(g1[n], g3[n]) for n in g1)
    q.put((rank, same, same3, launched, float(ex1), float(ex2), two.describe()))
    D.shutdown()


def test_bucketed_overlapped_allreduce_equals_single_shot():
    """the reducer chunked by layer in reverse order (hooks start a bucket's all-reduce during
    backward) gives bit-identical gradients to the single-shot reducer, on 2 gloo ranks."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, same3, launched, ex1, ex2, desc in res:
        assert same and same3
        assert launched == [[0, 1, 2], [0, 1, 2]]          # every bucket was started by its hook, last layer first
        assert ex1 == ex2 == 3.0
        assert desc['buckets'] == [(16 * 4 + 4) * 4, (16 * 16 + 16) * 4, (8 * 16 + 16) * 4]
        assert desc['allreduce_bytes'] == sum(desc['buckets']) + 4


def _preflight_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from graph_detr4d_amd import dist as D
    D.init(backend='gloo')
    try:
        D.preflight(world + 1)
        wrong = 'accepted'
    except RuntimeError as e:
        wrong = str(e)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    red = D.FlatGradAllReducer(net.parameters(), max_extras=2, buckets=[list(net[2].parameters()), list(net[0].parameters())])
    pre = D.preflight(world, None, sizes=(1 << 16, 1 << 20), iters=3, warmup=1, bucket_plan=red.describe())
    q.put((rank, wrong, pre, sum(p.numel() * 4 for p in net.parameters())))
    D.shutdown()


def test_preflight_two_ranks_gloo():
    """bench.py --gpus N runs dist.preflight before the timed steps (make the first RCCL run uneventful): the
    rank count is asserted, every rank is listed, all-reduces of the given sizes are timed alone and verified, the bucket
    plan is carried; the bytes a training step all-reduces equal the parameters' bytes (+ the loss normalisers' extras)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_preflight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, wrong, pre, param_bytes in res:
        assert 'asked for [3]' in wrong
        assert pre['world'] == 2 and pre['backend'] == 'gloo' and [r['rank'] for r in pre['ranks']] == [0, 1]
        assert [x['bytes'] for x in pre['allreduce']] == [1 << 16, 1 << 20]
        assert all(x['sum_ok'] and x['ms'] > 0 and abs(x['busbw_GBps'] - x['algbw_GBps']) < 1e-9 for x in pre['allreduce'])   # N = 2: factor 1
        plan = pre['bucket_plan']
        assert plan['allreduce_bytes'] == param_bytes + 4 * 2 and sum(plan['buckets']) == param_bytes
    assert res[0][2]['allreduce'][0]['ms'] == res[1][2]['allreduce'][0]['ms']        # MAX over ranks: one figure for the job


def _uneven_worker(rank, world, port, q, samples):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from graph_detr4d_amd import dist as D
    D.init(backend='gloo')
    # (1) one rank was launched for another job size: EVERY rank's preflight raises (nobody is left waiting in a collective)
    try:
        D.preflight(8 if rank == 2 else world, None, sizes=(1 << 12,), iters=1, warmup=0)
        failed = 'accepted'
    except RuntimeError as e:
        failed = str(e)
    pre = D.preflight(world, None, sizes=(1 << 12,), iters=2, warmup=1)
    # (2) 10 samples over 4 ranks: shards of 3, 3, 2, 2 - in the last step ranks 2 and 3 have no sample, take part in the
    # all-reduce with a zero gradient, and every rank ends every step with the same parameters
    mine = D.sample_indices(samples, rank, world)
    steps = -(-samples // world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    red = D.FlatGradAllReducer(net.parameters())
    red.bind()
    sums = []
    for step in range(steps):
        red.zero_grad()
        if step < len(mine):
            x = torch.randn(4, 6, generator=torch.Generator().manual_seed(D.sample_seed(100, mine[step])))
            net(x).square().mean().backward()
        red.reduce()
        with torch.no_grad():
            for p in net.parameters():
                p.add_(p.grad, alpha=-0.1)
        sums.append([float(p.detach().double().sum()) for p in net.parameters()])
    q.put((rank, failed, pre['world'], mine, sums))
    D.shutdown()


def test_four_ranks_uneven_shards_and_a_rank_that_fails_preflight():
    """world size 4 on gloo with a sample count that does not divide (10 = 3 + 3 + 2 + 2) - ranks without a sample in
    the last step still take part in the all-reduce (zero gradient) and all ranks hold the same parameters after every step, equal
    to a single process averaging the same per-sample gradients over 4; and a rank asked for another world size fails the job
    on EVERY rank instead of leaving three ranks in a collective."""
    world, samples = 4, 10
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q, samples)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    for rank, failed, w, mine, sums in res:
        assert 'rank(s) [2] were asked for [8]' in failed, failed
        assert w == 4
        assert sums == res[0][4]                                   # the same parameters on every rank after every step
    # the single-process reference: per step the mean over 4 ranks of the per-sample gradients (absent samples count as zero)
    from graph_detr4d_amd import dist as D
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    for step in range(3):
        acc = [torch.zeros_like(p) for p in net.parameters()]
        for r in range(world):
            mine = D.sample_indices(samples, r, world)
            if step < len(mine):
                x = torch.randn(4, 6, generator=torch.Generator().manual_seed(D.sample_seed(100, mine[step])))
                gs = torch.autograd.grad(net(x).square().mean(), list(net.parameters()))
                acc = [a + g for a, g in zip(acc, gs)]
        with torch.no_grad():
            for p, a in zip(net.parameters(), acc):
                p.add_(a / world, alpha=-0.1)
        want = [float(p.detach().double().sum()) for p in net.parameters()]
        np.testing.assert_allclose(res[0][4][step], want, rtol=1e-5)


def test_flat_sgd_step_equals_torch_sgd():
    """FlatGradAllReducer.flatten_params / sgd_step: every parameter becomes a view of one flat buffer (values kept, modules
    still see them), and the one-launch update equals torch.optim.SGD(lr) bit for bit over three steps; version counters move."""
    import copy
    import torch
    from graph_detr4d_amd import dist as D
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))
    ref = copy.deepcopy(net)
    with pytest.raises(RuntimeError):
        D.FlatGradAllReducer(list(net.parameters())).flatten_params()      # slices of 16 bytes only
    red = D.FlatGradAllReducer(list(net.parameters()), align=4)
    red.bind()
    before = [p.detach().clone() for p in net.parameters()]
    fp = red.flatten_params()
    assert fp.numel() == red.numel >= sum(p.numel() for p in net.parameters()) and all(p.data_ptr() % 16 == 0 for p in net.parameters())
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p, b) and p.data_ptr() >= fp.data_ptr() and p.data_ptr() < fp.data_ptr() + 4 * fp.numel()
    opt = torch.optim.SGD(ref.parameters(), lr=0.05)
    for step in range(3):
        x = torch.randn(11, 7)
        red.zero_grad()
        opt.zero_grad()
        net(x).square().mean().backward()
        ref(x).square().mean().backward()
        v0 = [p._version for p in net.parameters()]
        red.sgd_step(0.05)
        opt.step()
        assert all(p._version > v for p, v in zip(net.parameters(), v0))
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.equal(p, q)

"""gd4d_row_chain_fwd programs (the row-local work of a decoder layer as one launch) against fp64 references.  GPU only."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ln(n, seed):
    torch.manual_seed(seed)
    m = torch.nn.LayerNorm(n)
    with torch.no_grad():
        m.weight.copy_(torch.randn(n) * 0.3 + 1.0)
        m.bias.copy_(torch.randn(n) * 0.2)
    return m.to(DEV)


@pytest.mark.parametrize('m,k,n', [(900, 256, 256), (900, 256, 768), (900, 512, 256), (37, 256, 10), (1, 128, 24), (130, 256, 96)])
def test_chain_gemm_matches_fp64(m, k, n):
    """split-bf16 x3 products, fp32 accumulation: fp32-class (2^-16 relative per product)."""
    from graph_detr4d_amd import ops
    torch.manual_seed(m + k + n)
    x, w, b = torch.randn(m, k), torch.randn(n, k) * 0.08, torch.randn(n)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    out = torch.empty(m, n, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, xd), ops.chain_gemm(0, wd, bd, out=out)], m)
    ref = F.linear(x.double(), w.double(), b.double())
    assert (out.cpu().double() - ref).abs().max().item() < 1e-4
    out2 = torch.empty(m, n, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, xd), ops.chain_gemm(0, wd, bd, out=out2)], m)
    assert torch.equal(out, out2)                                # run-to-run identical


@pytest.mark.parametrize('heads,m', [(8, 900), (4, 45), (8, 7)])
def test_chain_headgemm_matches_fp64_and_the_stand_alone_kernel(heads, m):
    """HEADGEMM = gd4d_value_proj_heads_fwd as a chain operation: value_proj of the per-head aggregates."""
    from graph_detr4d_amd import ops
    torch.manual_seed(heads + m)
    dh = 256 // heads
    agg, wsum = torch.randn(m, heads, 256), torch.rand(m, heads)
    w, b = torch.randn(256, 256) * 0.06, torch.randn(256)
    res = torch.randn(m, 256)
    d = [t.to(DEV) for t in (agg, wsum, w, b, res)]
    out = torch.empty(m, 256, device=DEV)
    ops.row_chain_fwd([ops.chain_load(1, d[4]), ops.chain_headgemm(d[0], d[1], d[2], d[3], dst=0, res=1, out=out)], m)
    want = torch.einsum('mhc,hdc->mhd', agg.double(), w.double().view(heads, dh, 256)) + \
        b.double().view(heads, dh) * wsum.double().unsqueeze(-1)
    assert (out.cpu().double() - (want.reshape(m, 256) + res.double())).abs().max().item() < 1e-4
    alone = ops.value_proj_heads_fwd(d[0], d[1], d[2], d[3])
    torch.testing.assert_close(out - d[4], alone, rtol=1e-4, atol=1e-4)
    nob = torch.empty(m, 256, device=DEV)
    ops.row_chain_fwd([ops.chain_headgemm(d[0], d[1], d[2], None, out=nob)], m)
    assert (nob.cpu().double() - torch.einsum('mhc,hdc->mhd', agg.double(), w.double().view(heads, dh, 256)).reshape(m, 256)).abs().max().item() < 1e-4


def test_chain_b_shaped_program_matches_torch():
    """output_proj + two residuals, LayerNorm, FFN + residual, LayerNorm, next in-projection, reg branch, refinement - the
    program the fused decoder loop runs after the gather (fused_decoder.run), against plain torch in fp64."""
    from graph_detr4d_amd import ops
    torch.manual_seed(2)
    q, c = 900, 256
    g = lambda *s: torch.randn(*s) * 0.06        # noqa: E731
    agg, wsum = torch.randn(q, 8, c), torch.rand(q, 8)
    x1, posf, pos = torch.randn(q, c), torch.randn(q, c), torch.randn(q, c)
    wv, bv, wo, bo = g(c, c), g(c), g(c, c), g(c)
    w1, b1, w2, b2 = g(512, c), g(512), g(c, 512), g(c)
    win, bin_ = g(3 * c, c), g(3 * c)
    r1, rb1, r2, rb2, r3, rb3 = g(c, c), g(c), g(c, c), g(c), g(10, c), g(10)
    ref = torch.rand(q, 3)
    n1, n2 = _ln(c, 1), _ln(c, 2)
    D = lambda t: t.to(DEV)                        # noqa: E731
    dd = {k: D(v) for k, v in dict(agg=agg, wsum=wsum, x1=x1, posf=posf, pos=pos, wv=wv, bv=bv, wo=wo, bo=bo, w1=w1, b1=b1, w2=w2,
                                   b2=b2, win=win, bin=bin_, r1=r1, rb1=rb1, r2=r2, rb2=rb2, r3=r3, rb3=rb3, ref=ref).items()}
    x3 = torch.empty(q, c, device=DEV)
    qkv = torch.empty(q, 3 * c, device=DEV)
    new_ref = torch.empty(q, 3, device=DEV)
    prog = [ops.chain_headgemm(dd['agg'], dd['wsum'], dd['wv'], dd['bv'], dst=0),
            ops.chain_load(3, dd['x1'], dd['posf']),
            ops.chain_gemm(0, dd['wo'], dd['bo'], dst=1, res=3),
            ops.chain_layernorm(1, n1, dst=2),
            ops.chain_gemm(2, dd['w1'], dd['b1'], dst=0, relu=True),
            ops.chain_gemm(0, dd['w2'], dd['b2'], dst=1, res=2),
            ops.chain_layernorm(1, n2, dst=3, out=x3),
            ops.chain_add(0, 3, c, add=dd['pos']),
            ops.chain_gemm(0, dd['win'][:2 * c], dd['bin'][:2 * c], out=qkv[:, :2 * c]),
            ops.chain_gemm(3, dd['win'][2 * c:], dd['bin'][2 * c:], out=qkv[:, 2 * c:]),
            ops.chain_gemm(3, dd['r1'], dd['rb1'], dst=1, relu=True),
            ops.chain_gemm(1, dd['r2'], dd['rb2'], dst=2, relu=True),
            ops.chain_gemm(2, dd['r3'], dd['rb3'], dst=1),
            ops.chain_refine(1, dd['ref'], new_ref)]
    ops.row_chain_fwd(prog, q)
    f = lambda t: t.double()                        # noqa: E731
    val = (torch.einsum('mhc,hdc->mhd', f(agg), f(wv).view(8, 32, c)) + f(bv).view(8, 32) * f(wsum).unsqueeze(-1)).reshape(q, c)
    y1 = F.linear(val, f(wo), f(bo)) + f(x1) + f(posf)
    x2 = F.layer_norm(y1, (c,), f(n1.weight.cpu()), f(n1.bias.cpu()), n1.eps)
    y2 = F.linear(F.linear(x2, f(w1), f(b1)).relu(), f(w2), f(b2)) + x2
    want_x3 = F.layer_norm(y2, (c,), f(n2.weight.cpu()), f(n2.bias.cpu()), n2.eps)
    want_qkv = torch.cat([F.linear(want_x3 + f(pos), f(win[:2 * c]), f(bin_[:2 * c])), F.linear(want_x3, f(win[2 * c:]), f(bin_[2 * c:]))], -1)
    tmp = F.linear(F.linear(F.linear(want_x3, f(r1), f(rb1)).relu(), f(r2), f(rb2)).relu(), f(r3), f(rb3))
    inv = lambda t: torch.log(t.clamp(min=1e-5) / (1 - t).clamp(min=1e-5))      # noqa: E731
    want_ref = torch.stack([tmp[:, 0] + inv(f(ref)[:, 0]), tmp[:, 1] + inv(f(ref)[:, 1]), tmp[:, 4] + inv(f(ref)[:, 2])], -1).sigmoid()
    assert (x3.cpu().double() - want_x3).abs().max().item() < 2e-4
    assert (qkv.cpu().double() - want_qkv).abs().max().item() < 5e-4
    assert (new_ref.cpu().double() - want_ref).abs().max().item() < 1e-4


def test_two_programs_in_one_launch_equal_two_launches():
    """gd4d_row_chain2_fwd: chain A of a decoder layer next to [reg branch, refinement parked in LDS, position_encoder with
    inverse_sigmoid inside SMALL_LINEAR] - bit-identical to the same programs launched one after the other (the second
    one reading the refined points back from global memory), for row counts with and without a partial last block."""
    from graph_detr4d_amd import ops
    torch.manual_seed(4)
    c = 256
    g = lambda *s: (torch.randn(*s) * 0.06).to(DEV)        # noqa: E731
    w = {k: g(*s) for k, s in dict(o=(c, c), cam=(24, c), off=(96, c), att=(128, c), r1=(c, c), r2=(c, c), r3=(10, c),
                                   p0=(c, 3), p3=(c, c)).items()}
    b = {k: g(v.shape[0]) for k, v in w.items()}
    n0, l1, l4 = _ln(c, 1), _ln(c, 2), _ln(c, 3)
    for q in (900, 37):
        o, x, pos, xprev = g(q, c), g(q, c), g(q, c), g(q, c)
        ref = torch.rand(q, 3, device=DEV)

        def prog_a(x1, cam, off, att):
            return [ops.chain_load(0, o), ops.chain_load(3, x), ops.chain_gemm(0, w['o'], b['o'], dst=1, res=3),
                    ops.chain_layernorm(1, n0, dst=2, out=x1), ops.chain_add(0, 2, c, add=pos),
                    ops.chain_gemm(0, w['cam'], b['cam'], out=cam), ops.chain_gemm(0, w['off'], b['off'], out=off),
                    ops.chain_gemm(0, w['att'], b['att'], out=att)]

        def reg_ops(new_ref, park):
            return [ops.chain_load(3, xprev), ops.chain_gemm(3, w['r1'], b['r1'], dst=1, relu=True),
                    ops.chain_gemm(1, w['r2'], b['r2'], dst=2, relu=True), ops.chain_gemm(2, w['r3'], b['r3'], dst=1),
                    ops.chain_refine(1, ref, new_ref, dst=park)]

        def pos_ops(out, from_global=None):
            head = [ops.chain_load(0, from_global, inv_sigmoid=True)] if from_global is not None else []
            return head + [ops.chain_small_linear(0, w['p0'], b['p0'], 1, inv_sigmoid=from_global is None),
                           ops.chain_layernorm(1, l1, dst=2, relu=True), ops.chain_gemm(2, w['p3'], b['p3'], dst=1),
                           ops.chain_layernorm(1, l4, relu=True, out=out)]
        e = lambda *s: torch.empty(*s, device=DEV)          # noqa: E731
        got = dict(x1=e(q, c), cam=e(q, 24), off=e(q, 96), att=e(q, 128), ref=e(q, 3), pf=e(q, c))
        ops.row_chain2_fwd(prog_a(got['x1'], got['cam'], got['off'], got['att']),
                           reg_ops(got['ref'], 0) + pos_ops(got['pf']), q)
        want = dict(x1=e(q, c), cam=e(q, 24), off=e(q, 96), att=e(q, 128), ref=e(q, 3), pf=e(q, c))
        ops.row_chain_fwd(prog_a(want['x1'], want['cam'], want['off'], want['att']), q)
        ops.row_chain_fwd(reg_ops(want['ref'], -1), q)
        ops.row_chain_fwd(pos_ops(want['pf'], from_global=want['ref']), q)
        for k in got:
            assert torch.equal(got[k], want[k]), (q, k)
        assert torch.isfinite(got['pf']).all() and got['ref'].min() > 0 and got['ref'].max() < 1


def test_chain_rejects_bad_programs():
    from graph_detr4d_amd import ops
    from graph_detr4d_amd._lib import Gd4dError
    x = torch.randn(20, 256, device=DEV)
    w = torch.randn(256, 256, device=DEV)
    with pytest.raises(Gd4dError):                                   # GEMM writing its own source buffer
        ops.row_chain_fwd([ops.chain_load(0, x), ops.chain_gemm(0, w, dst=0)], 20)
    with pytest.raises(Gd4dError):                                   # GEMM without any destination
        ops.row_chain_fwd([ops.chain_load(0, x), ops.chain_gemm(0, w)], 20)
    with pytest.raises(ValueError):                                  # heads that do not divide into 32-column groups
        ops.chain_headgemm(torch.randn(20, 16, 256, device=DEV), torch.rand(20, 16, device=DEV), w)


@pytest.mark.parametrize('m,k,n', [(900, 256, 256), (900, 256, 3), (37, 256, 10), (130, 512, 96)])
def test_chain_gemm_exact_matches_fp64_to_fp32_class(m, k, n):
    """GD4D_CHAIN_EXACT: three bf16 pieces per operand, six products - fp32-class (what a reference point needs: its error
    is multiplied by the 102-m range and the focal length); two orders of magnitude below the split-bf16 x3 form."""
    from graph_detr4d_amd import ops
    torch.manual_seed(m + k + n)
    x, w, b = torch.randn(m, k), torch.randn(n, k) * 0.08, torch.randn(n)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    out, out3 = torch.empty(m, n, device=DEV), torch.empty(m, n, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, xd), ops.chain_gemm(0, wd, bd, out=out, exact=True)], m)
    ops.row_chain_fwd([ops.chain_load(0, xd), ops.chain_gemm(0, wd, bd, out=out3)], m)
    ref = F.linear(x.double(), w.double(), b.double())
    err, err3 = (out.cpu().double() - ref).abs().max().item(), (out3.cpu().double() - ref).abs().max().item()
    fp32 = (F.linear(x, w, b).double() - ref).abs().max().item()           # what a plain fp32 GEMM on the host does
    assert err < max(3e-6, 4 * fp32), (err, fp32)
    assert err < err3 or err3 < 3e-6
    # sigmoid epilogue (the initial reference points, detr3d_transformer.py:133-134)
    sig = torch.empty(m, n, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, xd), ops.chain_gemm(0, wd, bd, out=sig, sigmoid=True, exact=True)], m)
    assert (sig.cpu().double() - torch.sigmoid(ref)).abs().max().item() < max(1e-6, fp32)


@pytest.mark.parametrize('m', [900, 37, 1, 2700])
def test_signal_wait_hand_off_between_the_two_programs(m):
    """SIGNAL / WAIT: the second program writes rows to global memory and signals per row block; the first program, after
    a GEMM of its own, waits and LOADs them - what run_single does with position_encoder next to chain B.  Equal, bit for
    bit, to the two programs run one launch after the other; repeated 50 times back to back on ONE flag buffer (a missed hand-off
    would read the poison the row buffer is refilled with before every launch; a flag left up would let the next launch's WAIT
    through too early); the give-up counter stays 0; without a second program the operations are refused."""
    from graph_detr4d_amd import _lib, ops
    torch.manual_seed(m)
    x, y = torch.randn(m, 256, device=DEV), torch.randn(m, 3, device=DEV).sigmoid()
    w1, b1 = torch.randn(256, 256, device=DEV) * 0.06, torch.randn(256, device=DEV)
    w0, b0 = torch.randn(256, 3, device=DEV), torch.randn(256, device=DEV)
    w2, b2 = torch.randn(256, 256, device=DEV) * 0.06, torch.randn(256, device=DEV)
    ln = _ln(256, 3)
    blocks = (m + 15) // 16

    def producer(out, flags=None):
        return [ops.chain_load(0, y, inv_sigmoid=True), ops.chain_small_linear(0, w0, b0, 1), ops.chain_layernorm(1, ln, dst=0, relu=True),
                ops.chain_gemm(0, w2, b2, dst=1), ops.chain_layernorm(1, ln, relu=True, out=out)] + \
            ([ops.chain_signal(flags)] if flags is not None else [])

    def consumer(pos, out, flags=None, errors=None):
        return [ops.chain_load(0, x), ops.chain_gemm(0, w1, b1, dst=1)] + \
            ([ops.chain_wait(flags, errors)] if flags is not None else []) + \
            [ops.chain_load(3, x, pos), ops.chain_gemm(1, w2, b2, dst=0, res=3, out=out)]
    pos_ref, out_ref = torch.empty(m, 256, device=DEV), torch.empty(m, 256, device=DEV)
    ops.row_chain_fwd(producer(pos_ref), m)
    ops.row_chain_fwd(consumer(pos_ref, out_ref), m)
    errors = torch.zeros(1, device=DEV, dtype=torch.int32)
    pos, out = torch.empty(m, 256, device=DEV), torch.empty(m, 256, device=DEV)
    # ONE flag buffer for all 50 launches: the WAITing program takes every flag down again after it has seen it (what lets a
    # request slot keep one persistent buffer and a replayed graph hold no fill)
    flags = torch.zeros((blocks + 7) // 8 * 8, device=DEV, dtype=torch.int32)
    for _ in range(50):
        pos.fill_(float('nan'))
        out.fill_(float('nan'))
        ops.row_chain2_fwd(producer(pos, flags), consumer(pos, out, flags, errors), m)      # the SIGNALling program first
        assert torch.equal(pos, pos_ref) and torch.equal(out, out_ref)
        assert int(flags.sum().item()) == 0
    assert int(errors.item()) == 0
    with pytest.raises(_lib.Gd4dError):
        ops.row_chain_fwd(producer(pos, flags), m)
    with pytest.raises(_lib.Gd4dError):                            # the waiting program first: its producers would be dispatched
        ops.row_chain2_fwd(consumer(pos, out, flags, errors), producer(pos, flags), m)      # after it - refused (deadlock-prone)


@pytest.mark.parametrize('m', [900, 45])
def test_fused_gemm_forms_equal_their_separate_operations(m):
    """GD4D_CHAIN_SRC2 (the packed in-projection as one operation whose last 256 columns read another buffer) and
    GD4D_CHAIN_SPLIT_OUT (three Linears of one input as one GEMM over the stacked weights, three outputs): a column's sum does
    not depend on the operation it is part of - bit-identical to the separate operations; also after a weight update."""
    from graph_detr4d_amd import ops
    torch.manual_seed(m + 1)
    x, pos = torch.randn(m, 256, device=DEV), torch.randn(m, 256, device=DEV)
    w, b = torch.randn(768, 256, device=DEV) * 0.06, torch.randn(768, device=DEV)
    a, bq = torch.empty(m, 768, device=DEV), torch.empty(m, 768, device=DEV)
    ops.row_chain_fwd([ops.chain_load(0, x, pos), ops.chain_load(3, x), ops.chain_gemm(0, w[:512], b[:512], out=a[:, :512]),
                       ops.chain_gemm(3, w[512:], b[512:], out=a[:, 512:])], m)
    ops.row_chain_fwd([ops.chain_load(0, x, pos), ops.chain_load(3, x), ops.chain_gemm_two_sources(0, 3, 512, w, b, bq)], m)
    assert torch.equal(a, bq)
    ref = torch.cat([F.linear((x + pos).double(), w[:512].double(), b[:512].double()), F.linear(x.double(), w[512:].double(), b[512:].double())], 1)
    assert (bq.double() - ref).abs().max().item() < 1e-4
    lins = [torch.nn.Linear(256, n).to(DEV) for n in (24, 96, 128)]
    for _ in range(2):
        sep = [torch.empty(m, l.out_features, device=DEV) for l in lins]
        one = [torch.full((m, l.out_features), float('nan'), device=DEV) for l in lins]
        ops.row_chain_fwd([ops.chain_load(0, x)] + [ops.chain_gemm(0, l.weight, l.bias, out=o) for l, o in zip(lins, sep)], m)
        ops.row_chain_fwd([ops.chain_load(0, x), ops.chain_gemm_three_outputs(0, lins, one)], m)
        for s_, o_ in zip(sep, one):
            assert torch.equal(s_, o_)
        with torch.no_grad():                                   # the stacked weight follows an in-place update (version counter)
            lins[1].weight.mul_(1.5)
            lins[2].bias.add_(0.25)


def test_a_hand_off_that_times_out_is_loud():
    """VERDICT r4 weak #2 / ADVICE r4: a WAIT nobody answers (here: the first program never SIGNALs on the flags the second one
    waits for) gives up after ~0.2 s - and then the rows its program LOADs are NaN, every output behind them is NaN, the device's
    error word counts the give-ups, and ops.check_handoff() raises (once: the word is cleared).  A WAIT that is not followed by
    the LOAD of what was handed over is refused."""
    from graph_detr4d_amd import _lib, ops
    torch.manual_seed(3)
    m, c = 100, 256
    blocks = (m + 15) // 16
    x = torch.randn(m, c, device=DEV)
    w, b = (torch.randn(c, c) * 0.06).to(DEV), torch.randn(c).to(DEV)
    ops.check_handoff()                                          # nothing pending from earlier tests
    err = ops.handoff_error_word(DEV)
    flags_a = torch.zeros((blocks + 7) // 8 * 8, device=DEV, dtype=torch.int32)
    flags_b = torch.zeros_like(flags_a)                          # nobody signals on these
    side, out = torch.empty(m, c, device=DEV), torch.zeros(m, c, device=DEV)
    producer = [ops.chain_load(0, x), ops.chain_gemm(0, w, b, out=side), ops.chain_signal(flags_a)]
    consumer = [ops.chain_wait(flags_b, err), ops.chain_load(1, x), ops.chain_gemm(1, w, b, out=out)]
    ops.row_chain2_fwd(producer, consumer, m)
    torch.cuda.synchronize()
    assert torch.isnan(out).all(), 'a hand-off that timed out must poison what it hands on'
    assert torch.isfinite(side).all()
    assert int(err.item()) == blocks
    with pytest.raises(_lib.Gd4dError, match='timed out'):
        ops.check_handoff()
    ops.check_handoff()                                          # cleared
    # a time-out is evidence that the placement the hand-offs rely on did not hold: the device builds its steps without them from
    # here on (until someone probes again - this test forced the time-out, so it restores the state)
    assert ops.handoff_enabled(DEV, 'GD4D_POS_ENCODER') is False
    ops._handoff_state(DEV)['placement'] = None
    assert ops.handoff_enabled(DEV, 'GD4D_POS_ENCODER') is True
    # the same programs with the flags that ARE raised: finite, no error
    consumer = [ops.chain_wait(flags_a, err), ops.chain_load(1, x), ops.chain_gemm(1, w, b, out=out)]
    flags_a.zero_()
    ops.row_chain2_fwd(producer, consumer, m)
    assert torch.equal(out, side)
    ops.check_handoff()
    with pytest.raises(_lib.Gd4dError):                          # what was handed over must enter through a LOAD
        ops.row_chain2_fwd(producer, [ops.chain_load(1, x), ops.chain_wait(flags_a, err), ops.chain_gemm(1, w, b, out=out)], m)
    with pytest.raises(_lib.Gd4dError):                          # ... and a WAIT needs its error word
        ops.row_chain2_fwd(producer, [ops.chain_wait(flags_a), ops.chain_load(1, x), ops.chain_gemm(1, w, b, out=out)], m)


def test_xcd_placement_self_test():
    """The hand-offs go through ONE XCD's L2: workgroups j and j + 8 k of a launch must share an XCD.  The probe kernel reports every
    workgroup's XCC id; the package checks the pattern once per device and builds its steps without hand-offs where it fails."""
    from graph_detr4d_amd import _lib, ops
    out = torch.full((512,), -1, device=DEV, dtype=torch.int32)
    _lib.check(_lib.load().gd4d_xcd_placement_probe(out.data_ptr(), 512, torch.cuda.current_stream().cuda_stream), 'probe')
    ids = out.cpu()
    assert (ids >= 0).all() and (ids < 16).all()
    assert ops.handoff_placement_ok(DEV) == bool((ids == ids[:8].repeat(64)).all())
    print('XCC ids of workgroups 0..15:', ids[:16].tolist(), 'placement ok:', ops.handoff_placement_ok(DEV))


@pytest.mark.parametrize('m', [900, 37, 16, 929])
def test_in_projection_writes_the_attention_cores_split_operands(m):
    """GD4D_CHAIN_SPLIT_KV + gd4d_mha_core_presplit_fwd: the packed in-projection also writes K (row-major) and V^T as bf16 hi / lo
    planes - hi = bf16(x), lo = bf16(x - hi), bit for bit what torch's round-to-nearest-even conversion gives on the fp32 output -
    and the attention core on those planes equals gd4d_mha_core_fwd on the fp32 rows bit for bit (poisoned planes first: every
    element a tile reads was written, incl. the rows past M of the last block)."""
    from graph_detr4d_amd import _lib, ops
    torch.manual_seed(m)
    c, heads = 256, 8
    x, pos = torch.randn(m, c, device=DEV), torch.randn(m, c, device=DEV)
    w, b = torch.randn(3 * c, c, device=DEV) * 0.08, torch.randn(3 * c, device=DEV) * 0.2
    qkv_ref = torch.empty(m, 1, 3 * c, device=DEV)
    prog = lambda out, kv: [ops.chain_load(0, x, pos), ops.chain_load(1, x), ops.chain_gemm_two_sources(0, 1, 2 * c, w, b, out.view(m, -1), kv=kv)]  # noqa: E731
    ops.row_chain_fwd(prog(qkv_ref, None), m)
    kv = ops.KVPlanes(m, c, DEV)
    kv.k.view(torch.int16).fill_(0x7fc0)                    # bf16 NaN
    kv.v.view(torch.int16).fill_(0x7fc0)
    qkv = torch.full((m, 1, 3 * c), 7.0, device=DEV)
    ops.row_chain_fwd(prog(qkv, kv), m)
    assert torch.equal(qkv[..., :c], qkv_ref[..., :c]) and bool((qkv[..., c:] == 7.0).all())    # fp32: the Q columns only
    kf, vf = qkv_ref[:, 0, c:2 * c], qkv_ref[:, 0, 2 * c:]
    for planes, ref in ((kv.k_rows(), kf), (kv.v_rows(), vf)):
        hi = ref.to(torch.bfloat16)
        assert torch.equal(planes[0, :m], hi) and torch.equal(planes[1, :m], (ref - hi.float()).to(torch.bfloat16))
        assert bool(torch.isfinite(planes.float()).all())    # the filler keys of the last block / step
    qh, kh, vh = qkv_ref.split(c, dim=-1)
    want = ops.mha_core_fwd(qh, kh, vh, heads)
    got = ops.mha_core_presplit_fwd(qkv[..., :c], kv, heads)
    assert torch.equal(got, want)
    blocked = torch.rand(m, m, device=DEV) < 0.3                 # H-DETR's kind of mask (bool), and an additive one
    blocked.fill_diagonal_(False)
    for mask in (blocked, torch.randn(m, m, device=DEV)):
        assert torch.equal(ops.mha_core_presplit_fwd(qkv[..., :c], kv, heads, mask), ops.mha_core_fwd(qh, kh, vh, heads, mask))
    # a training step: the fp32 K / V rows stay (keep_fp32), the core drops probabilities as the fp32 kernel does and saves the
    # log-sum-exp: the same elements dropped, values within the split-bf16 products' 2^-16
    kv2 = ops.KVPlanes(m, c, DEV)
    qkv2 = torch.empty(m, 1, 3 * c, device=DEV)
    op = ops.chain_gemm_two_sources(0, 1, 2 * c, w, b, qkv2.view(m, -1), kv=kv2, keep_fp32=True)
    ops.row_chain_fwd([ops.chain_load(0, x, pos), ops.chain_load(1, x), op], m)
    assert torch.equal(qkv2, qkv_ref) and torch.equal(kv2.k, kv.k) and torch.equal(kv2.v, kv.v)
    seed = ops.mha_dropout_seed(DEV)
    for pdrop in (0., 0.1):
        want, want_lse = ops.mha_core_fwd(qh, kh, vh, heads, want_lse=True, dropout_p=pdrop, seed=seed if pdrop else None)
        got, got_lse = ops.mha_core_presplit_fwd(qkv2[..., :c], kv2, heads, want_lse=True, dropout_p=pdrop, seed=seed if pdrop else None)
        torch.testing.assert_close(got, want, rtol=1e-3, atol=2e-4)      # (without dropout the two are bit-identical: above)
        torch.testing.assert_close(got_lse, want_lse, rtol=1e-5, atol=1e-5)
    with pytest.raises(_lib.Gd4dError):                          # planes only beside the whole 768-column projection
        op = ops.chain_gemm(0, w[:c], b[:c], out=qkv_ref.view(m, -1)[:, :c])
        op.flags |= ops.CHAIN_SPLIT_KV
        op.p2, op.p3, op.ld1, op.ld2 = kv.k.data_ptr(), kv.v.data_ptr(), kv.v[0].numel(), kv.k[0].numel()
        ops.row_chain_fwd([ops.chain_load(0, x), op], m)


@pytest.mark.parametrize('two_programs,layout,workgroups', [(False, 'nchw', 0), (True, 'nchw', 0), (True, 'nhwc', 0), (False, 'nhwc', 3),
                                                            (True, 'nchw', 500)])
def test_guest_workgroups_project_pyramid_levels_beside_the_chain(two_programs, layout, workgroups):
    """gd4d_row_chain_guest_fwd: the chain's results are those of the plain launch, bit for bit, and the guests' rows are
    gd4d_value_proj_fwd's (same body: bit-identical from NCHW levels; channels-last levels hold the same values)."""
    from graph_detr4d_amd import ops
    torch.manual_seed(11)
    m, r = 900, 5
    x, w, b = torch.randn(m, 256, device=DEV), torch.randn(256, 256, device=DEV) * 0.08, torch.randn(256, device=DEV)
    levels = [torch.randn(1, r, 256, h, ww, device=DEV) for h, ww in [(29, 50), (15, 25)]]       # partial tiles in both
    wv, bv = torch.randn(256, 256, device=DEV) * 0.06, torch.randn(256, device=DEV)
    want_rows = ops.value_proj_fwd(levels, wv, bv)
    src = levels if layout == 'nchw' else [t.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3) for t in levels]
    out = torch.full_like(want_rows, float('nan'))
    guest = ops.chain_guest(src, ops.value_proj_image(wv, bv), out, workgroups=workgroups)
    prog_a = lambda o: [ops.chain_load(0, x), ops.chain_gemm(0, w, b, out=o)]
    prog_b = lambda o: [ops.chain_load(1, x), ops.chain_gemm(1, w, None, out=o, relu=True)]
    plain_a, plain_b = torch.empty(m, 256, device=DEV), torch.empty(m, 256, device=DEV)
    got_a, got_b = torch.empty(m, 256, device=DEV), torch.empty(m, 256, device=DEV)
    if two_programs:
        ops.row_chain2_fwd(prog_a(plain_a), prog_b(plain_b), m)
        ops.row_chain2_fwd(prog_a(got_a), prog_b(got_b), m, guest=guest)
        assert torch.equal(got_b, plain_b)
    else:
        ops.row_chain_fwd(prog_a(plain_a), m)
        ops.row_chain_fwd(prog_a(got_a), m, guest=guest)
    assert torch.equal(got_a, plain_a)
    assert torch.equal(out, want_rows)
    # the cached image follows the weights
    assert ops.value_proj_image(wv, bv) is ops.value_proj_image(wv, bv)
    img = ops.value_proj_image(wv, bv)
    bv.add_(1.0)
    assert ops.value_proj_image(wv, bv) is not img


def test_headgemm_adds_a_global_addend():
    """GD4D_CHAIN_ADD_GOUT: HEADGEMM + pagg (the coarse levels' already projected part) = the two-step sum, bit for bit."""
    from graph_detr4d_amd import ops
    torch.manual_seed(5)
    m = 900
    agg, wsum = torch.randn(m, 8, 256, device=DEV), torch.rand(m, 8, device=DEV)
    w, b, pagg = torch.randn(256, 256, device=DEV) * 0.06, torch.randn(256, device=DEV), torch.randn(m, 256, device=DEV)
    plain, got = torch.empty(m, 256, device=DEV), torch.empty(m, 256, device=DEV)
    ops.row_chain_fwd([ops.chain_headgemm(agg, wsum, w, b, dst=0), ops.chain_load(1, pagg), ops.chain_add(2, 0, 256, res=1, out=plain)], m)
    ops.row_chain_fwd([ops.chain_headgemm(agg, wsum, w, b, dst=0, addend=pagg), ops.chain_add(2, 0, 256, out=got)], m)
    assert torch.equal(got, plain)

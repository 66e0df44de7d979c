"""The C-ABI library loads and exports every symbol include/gd4d.h declares (no compute, no GPU)."""
import os
import re


def test_library_exports_every_declared_symbol(repo_root):
    from graph_detr4d_amd import _lib
    hdr = open(os.path.join(repo_root, 'include', 'gd4d.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(gd4d_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f'libgd4d.so does not export {name}'
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.gd4d_abi_version() == _lib.ABI_VERSION
    assert b'not supported' in lib.gd4d_error_string(-2)


def test_product_path_never_imports_oracle(repo_root):
    """The package must not route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(repo_root, 'graph-detr4d_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'c_oracle' not in src and 'torch_oracle' not in src, f


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from graph_detr4d_amd import _lib, ops
    z = torch.zeros
    with pytest.raises(_lib.Gd4dError):
        ops.cross_attn_fwd(z(6, 4, 8, 32), [(2, 2)], z(1, 3, 3), z(1, 3, 8, 4, 3), z(1, 3, 8, 1, 4),
                           z(1, 3, 6), z(1, 6, 4, 4), [0, 0, 0, 1, 1, 1], 8, 8)


def test_host_linear_sum_assignment_matches_scipy():
    """gd4d_linear_sum_assignment_batch is host code: checked here without a GPU against scipy (what the reference
    calls, hungarian_assigner_3d.py:130), including tie-heavy and constant-column matrices and the threaded path."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    from graph_detr4d_amd import ops
    rng = np.random.default_rng(0)
    mats = []
    for trial in range(120):
        r, c = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        if trial % 3 == 0:
            m = rng.standard_normal((r, c))
        elif trial % 3 == 1:
            m = rng.integers(0, 4, (r, c)).astype(np.float64)              # many ties
        else:
            m = rng.standard_normal((r, c))
            m[:, rng.integers(0, c)] = 100.0                               # a nan_to_num column
        mats.append(m.astype(np.float32))
    mats += [np.zeros((0, 5), np.float32), np.zeros((4, 0), np.float32)]
    flat = np.concatenate([m.ravel() for m in mats])
    offs = np.concatenate([[0], np.cumsum([m.size for m in mats])])
    problems = [(int(offs[i]), m.shape[0], m.shape[1]) for i, m in enumerate(mats)]
    for threads in (1, 4):
        got = ops.linear_sum_assignment_batch(flat, problems, threads)
        for m, a in zip(mats, got):
            want = np.full(m.shape[0], -1, np.int32)
            if m.size:
                rows, cols = linear_sum_assignment(m)
                want[rows] = cols
            np.testing.assert_array_equal(a, want)


def test_argument_validation_returns_before_any_gpu_work():
    """Every entry point checks its arguments first: bad calls come back with an error code on a box without a GPU."""
    import ctypes
    from graph_detr4d_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    one = (ctypes.c_float * 4)()
    ptr = ctypes.cast(one, ctypes.c_void_p)
    EINVAL, EUNSUPPORTED = -1, -2
    # null pointers
    assert lib.gd4d_value_proj_bwd_input(null, ptr, ptr, ptr, 1, 256, 1, 0, null) == EINVAL
    assert lib.gd4d_value_proj_bwd_weight(ptr, ptr, ptr, null, null, ptr, 0, 1, 256, 1, null) == EINVAL
    assert lib.gd4d_linear_bwd_weight(null, ptr, ptr, null, 4, 4, 4, 4, 4, 0, null) == EINVAL
    assert lib.gd4d_match_cost_fwd(ptr, ptr, ptr, ptr, null, ptr, 1, 1, 4, 10, 10, 9, 1, 1, 2.0, 0.25, 0.25, null) == EINVAL
    assert lib.gd4d_head_loss_fwd_bwd(ptr, ptr, ptr, ptr, ptr, ptr, null, ptr, ptr, ptr, 1, 1, 4, 10, 10, 9, 1,
                                      0.25, 2.0, 0.25, null) == EINVAL
    assert lib.gd4d_linear_sum_assignment_batch(null, ptr, ptr, ptr, 1, ptr, ptr, 1) == EINVAL
    # shapes the kernels are not built for
    lv = (ctypes.c_int32 * 2)(4, 4)
    lvp = ctypes.cast(lv, ctypes.c_void_p)
    assert lib.gd4d_value_proj_bwd_input(ptr, ptr, ptr, lvp, 1, 128, 1, 0, null) == EUNSUPPORTED          # C != 256
    assert lib.gd4d_match_cost_fwd(ptr, ptr, ptr, ptr, ptr, ptr, 1, 1, 4, 10, 10, 9, 5000, 5000,
                                   2.0, 0.25, 0.25, null) == EUNSUPPORTED                                  # > 1024 boxes per sample
    assert lib.gd4d_head_loss_fwd_bwd(ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, 1, 1, 4, 10, 4, 9, 1,
                                      0.25, 2.0, 0.25, null) == EUNSUPPORTED                               # box code < 8
    # workspace too small for the weight gradient
    assert lib.gd4d_value_proj_bwd_weight(ptr, ptr, lvp, ptr, null, ptr, 16, 1, 256, 1, null) == EINVAL
    assert lib.gd4d_value_proj_bwd_weight_workspace_bytes() >= 256 * (256 * 256 + 256) * 4


def test_training_entry_points_validate_before_any_gpu_work():
    """The raw-pyramid training backward (gd4d_cross_attn_sliced_bwd.hip): sizes, workspaces and unsupported shapes are
    refused with an error code on a box without a GPU; the size helpers agree with the layouts include/gd4d.h states."""
    import ctypes
    from graph_detr4d_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 96)()
    ptr = ctypes.c_void_p((ctypes.addressof(buf) + 63) & ~63)                  # 64-byte aligned: alignment checks pass
    EINVAL, EUNSUPPORTED, EWORKSPACE = -1, -2, -5
    b, n, q, hh = 1, 24, 900, 8
    cap_t = (n * 4 + 3) // 4
    assert lib.gd4d_cross_attn_dot_bytes(b, n, q, hh, 4) == 8 * b * q * hh * cap_t * 64 * 4
    assert lib.gd4d_pyramid_grad_slots_bytes(b, n, q, hh, 4) == b * q * hh * cap_t * 64 * 8
    assert lib.gd4d_cross_attn_dot_bytes(0, n, q, hh, 4) == 0
    lv = (ctypes.c_int32 * 8)(116, 200, 58, 100, 29, 50, 15, 25)
    lvp = ctypes.cast(lv, ctypes.c_void_p)
    chunks = lib.gd4d_pyramid_grad_chunks(lvp, 24, 4)
    geo = (ctypes.c_int32 * 20)()
    assert lib.gd4d_pyramid_grad_chunk_geometry(lvp, 24, 4, ctypes.cast(geo, ctypes.c_void_p)) == 0
    total = 0
    for l, (h, w) in enumerate([(116, 200), (58, 100), (29, 50), (15, 25)]):
        cws, chs, cw_n, ch_n, base = geo[5 * l:5 * l + 5]
        assert base == total and (1 << cws) * (1 << chs) in (8, 16, 32, 64)
        assert cw_n == -(-w >> cws) and ch_n == -(-h >> chs)                    # ceil
        total += 24 * cw_n * ch_n
    assert chunks == total
    assert lib.gd4d_pyramid_grad_chunks(lvp, 0, 4) == 0
    # null pointers / bad sizes
    assert lib.gd4d_value_proj_heads_bwd(null, ptr, ptr, ptr, ptr, 4, 8, 256, null) == EINVAL
    assert lib.gd4d_value_proj_heads_bwd(ptr, ptr, ptr, ptr, ptr, 4, 8, 128, null) == EUNSUPPORTED            # C != 256
    assert lib.gd4d_value_proj_heads_bwd(ptr, ptr, ptr, ptr, ptr, 4, 3, 256, null) == EUNSUPPORTED            # heads
    assert lib.gd4d_value_proj_heads_bwd_weight(ptr, ptr, ptr, ptr, ptr, ptr, 16, 4, 8, 256, 0, null) == EWORKSPACE
    assert lib.gd4d_value_proj_heads_bwd_weight(ptr, ptr, null, ptr, ptr, ptr, 1 << 30, 4, 8, 256, 0, null) == EINVAL   # bias without wsum
    ptrs = (ctypes.c_void_p * 4)(ptr.value, ptr.value, ptr.value, ptr.value)
    assert lib.gd4d_cross_attn_dot_sliced(ptrs, 128, ptr, ptr, ptr, 16, 1, 6, 4, 8, 256, 4, 4, _lib.F32, null, null) == EWORKSPACE
    assert lib.gd4d_cross_attn_dot_sliced(ptrs, 128, ptr, ptr, ptr, 1 << 40, 1, 6, 4, 8, 256, 4, 4, 7, null, null) == EUNSUPPORTED          # dtype
    assert lib.gd4d_cross_attn_dot_sliced(ptrs, 128, null, ptr, ptr, 1 << 40, 1, 6, 4, 8, 256, 4, 4, _lib.F32, null, null) == EINVAL
    rng = ctypes.cast((ctypes.c_double * 6)(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), ctypes.c_void_p)
    assert lib.gd4d_cross_attn_plan_bwd(ptr, ptr, ptr, ptr, ptr, rng, 900.0, 1600.0, lvp, ptr, null, null, ptr, ptr, ptr, ptr,
                                        null, 0, null, 1, 6, 4, 8, 4, 4, 0, null, null) == EINVAL            # no dpart
    assert lib.gd4d_cross_attn_plan_bwd(ptr, ptr, ptr, ptr, ptr, rng, 900.0, 1600.0, lvp, ptr, ptr, null, ptr, ptr, ptr, ptr,
                                        null, 0, null, 2, 6, 4, 8, 4, 4, 0, null, null) == EWORKSPACE        # B > 1 needs the workspace
    cs = ctypes.cast((ctypes.c_int64 * 4)(4096, 4096, 4096, 4096), ctypes.c_void_p)
    assert lib.gd4d_pyramid_grad_count(ptr, lvp, cs, 128, ptr, ptr, 16, 1, 6, 4, 8, 4, 4, null) == EWORKSPACE
    assert lib.gd4d_pyramid_grad_count(ptr, lvp, cs, 128, ptr, ptr, 1 << 40, 1, 6, 4, 8, 4, 3, null) == EUNSUPPORTED      # P != 4
    assert lib.gd4d_pyramid_grad_scan(ptr, ptr, ptr, 0, 100000, null) == EWORKSPACE
    assert lib.gd4d_pyramid_grad_fill(ptr, ptr, ptr, ptr, (1 << 26) - 8, null, 1, 6, 4, 8, 4, null) == EUNSUPPORTED      # row ids need 26 bits
    assert lib.gd4d_pyramid_grad_sort(ptr, ptr, ptr, ptr, null, 10, null) == EINVAL
    assert lib.gd4d_pyramid_grad_reduce(ptr, ptr, ptr, ptr, ptrs, lvp, null, 24, 128, 4, 0, null) == EUNSUPPORTED
    # dropout of the self-attention probabilities: a rate without a seed, a rate outside [0, 1), more elements than ids
    f = ctypes.c_float
    assert lib.gd4d_mha_core_fwd(ptr, ptr, ptr, null, ptr, 16, 16, 1, 8, 32, 256, 256, 256, 256, 0, f(0.17), null, f(0.1), null, null) == EINVAL
    assert lib.gd4d_mha_core_fwd(ptr, ptr, ptr, null, ptr, 16, 16, 1, 8, 32, 256, 256, 256, 256, 0, f(0.17), null, f(1.0), ptr, null) == EINVAL
    assert lib.gd4d_mha_core_fwd(ptr, ptr, ptr, null, ptr, 30000, 30000, 1, 8, 32, 256, 256, 256, 256, 0, f(0.17), null, f(0.1), ptr, null) == EUNSUPPORTED
    assert lib.gd4d_mha_core_bwd(ptr, ptr, ptr, ptr, ptr, null, ptr, ptr, ptr, ptr, ptr, 16, 16, 1, 8, 32, 256, 256, 256, 256, 256,
                                 256, 256, 256, 0, f(0.17), f(0.1), null, null) == EINVAL


def test_round6_entry_points_validate_before_any_gpu_work():
    """The coarse-projected gather, the guest launch and the value_proj image: null pointers, unsupported shapes and misuse come back
    as error codes on a box without a GPU; the guest struct has the size the header declares."""
    import ctypes
    from graph_detr4d_amd import _lib, ops
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 96)()
    ptr = ctypes.c_void_p((ctypes.addressof(buf) + 63) & ~63)
    EINVAL, EUNSUPPORTED = -1, -2
    assert lib.gd4d_chain_guest_bytes() == ctypes.sizeof(ops.ChainGuest)
    assert lib.gd4d_value_proj_image_bytes() == 8 * 32768 + 1024               # 8 chunks of split-bf16 fragments + the bias table
    assert lib.gd4d_value_proj_image(null, null, ptr, null) == EINVAL
    # the coarse gather: two fine + two coarse levels, 8 heads, every pointer there
    lv = (ctypes.c_int32 * 8)(16, 28, 8, 14, 4, 7, 2, 4)
    cs = (ctypes.c_int64 * 4)(16 * 28 * 1024, 8 * 14 * 1024, 4 * 7 * 1024, 2 * 4 * 1024)
    lp = (ctypes.c_void_p * 4)(ptr.value, ptr.value, ptr.value, ptr.value)
    pp = (ctypes.c_void_p * 2)(ptr.value, ptr.value)
    pcs = (ctypes.c_int64 * 2)(4 * 7 * 1024, 2 * 4 * 1024)
    args = lambda L=4, hh=8, pagg=ptr, proj=pp: (lp, lv, cs, 1024, 128, proj, pcs, ptr, ptr, ptr, pagg, 1, 6, 9, hh, 256, L, 4, 0, null, null)
    assert lib.gd4d_cross_attn_agg_items_coarse_fwd(*args(L=3)) == EUNSUPPORTED
    assert lib.gd4d_cross_attn_agg_items_coarse_fwd(*args(hh=4)) == EUNSUPPORTED
    assert lib.gd4d_cross_attn_agg_items_coarse_fwd(*args(pagg=null)) == EINVAL
    assert lib.gd4d_cross_attn_agg_items_coarse_fwd(*args(proj=null)) == EINVAL
    short = (ctypes.c_int64 * 2)(4 * 7 * 1024 - 16, 2 * 4 * 1024)              # a camera stride shorter than the level's rows
    a = list(args())
    a[6] = short
    assert lib.gd4d_cross_attn_agg_items_coarse_fwd(*a) == EINVAL
    # the guest launch: a guest is required, its job must be complete, training operations are refused
    op = ops.ChainOp(kind=ops.CHAIN_LOAD, src=-1, dst=0, res=-1, N=256, p0=ptr.value)
    prog = (ops.ChainOp * 1)(op)
    assert lib.gd4d_row_chain_guest_fwd(prog, 1, null, 0, 16, null, null) == EINVAL
    g = ops.ChainGuest()
    assert lib.gd4d_row_chain_guest_fwd(prog, 1, null, 0, 16, ctypes.byref(g), null) == EINVAL      # no image / out / levels
    g.image, g.out, g.L, g.R = ptr.value, ptr.value, 1, 2
    g.feats[0] = ptr.value
    g.level_hw[0], g.level_hw[1] = 4, 7
    store = ops.ChainOp(kind=ops.CHAIN_LOAD, src=-1, dst=0, res=-1, N=256, p0=ptr.value, gout=ptr.value, ldg=256)   # a store from LOAD: a training program
    assert lib.gd4d_row_chain_guest_fwd((ops.ChainOp * 1)(store), 1, null, 0, 16, ctypes.byref(g), null) == EUNSUPPORTED
    g.L = 9
    assert lib.gd4d_row_chain_guest_fwd(prog, 1, null, 0, 16, ctypes.byref(g), null) == EINVAL
    assert lib.gd4d_value_proj_guest_fwd(null, 0, null) == EINVAL


def test_position_embedding_one_kernel_entry_points_validate_before_any_gpu_work():
    """gd4d_mlp2_frustum_fwd / gd4d_mlp2_pe_se_fwd (ABI 54): null pointers, depth counts other than 64, hidden widths the images are not
    built for and more than 4 levels come back as error codes on a box without a GPU."""
    import ctypes
    from graph_detr4d_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 96)()
    ptr = ctypes.c_void_p((ctypes.addressof(buf) + 63) & ~63)
    odd = ctypes.c_void_p(ptr.value + 4)
    EINVAL, EUNSUPPORTED, EALIGN = -1, -2, -3
    lv = (ctypes.c_int32 * 10)(16, 28, 8, 14, 4, 7, 2, 4, 1, 2)
    rng = (ctypes.c_double * 6)(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)
    fr = lambda i2l=ptr, L=4, D=64, img=ptr, out=ptr, H=1024, ldo=256: (i2l, lv, L, 6, 928.0, 1600.0, D, 1.0, rng, img, null, out, H, ldo, null)
    assert lib.gd4d_mlp2_frustum_fwd(*fr(i2l=null)) == EINVAL
    assert lib.gd4d_mlp2_frustum_fwd(*fr(out=null)) == EINVAL
    assert lib.gd4d_mlp2_frustum_fwd(*fr(ldo=128)) == EINVAL
    assert lib.gd4d_mlp2_frustum_fwd(*fr(L=5)) == EUNSUPPORTED
    assert lib.gd4d_mlp2_frustum_fwd(*fr(D=32)) == EUNSUPPORTED
    assert lib.gd4d_mlp2_frustum_fwd(*fr(H=1000)) == EUNSUPPORTED
    assert lib.gd4d_mlp2_frustum_fwd(*fr(img=odd)) == EALIGN
    fp = (ctypes.c_void_p * 4)(ptr.value, ptr.value, ptr.value, ptr.value)
    ps = lambda feats=fp, L=4, D=64, pe_h=1024, se_h=256, sine=ptr, outs=fp, se_img=ptr: (
        ptr, feats, lv, L, 6, 928.0, 1600.0, D, 1.0, rng, ptr, null, pe_h, se_img, null, se_h, sine, outs, null, null)
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(feats=null)) == EINVAL
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(sine=null)) == EINVAL
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(outs=null)) == EINVAL
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(L=5)) == EUNSUPPORTED
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(D=48)) == EUNSUPPORTED
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(se_h=100)) == EUNSUPPORTED
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(se_img=odd)) == EALIGN
    holes = (ctypes.c_void_p * 4)(ptr.value, 0, ptr.value, ptr.value)
    assert lib.gd4d_mlp2_pe_se_fwd(*ps(feats=holes)) == EINVAL


def test_fill_carrying_chain_validates_before_any_gpu_work():
    """gd4d_row_chain_fill_fwd (ABI 55): jobs, start and records must be there, at most two jobs, a plan's table rows must fit 26 bits."""
    import ctypes
    from graph_detr4d_amd import _lib, ops
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 96)()
    ptr = ctypes.c_void_p((ctypes.addressof(buf) + 63) & ~63)
    EINVAL, EUNSUPPORTED = -1, -2
    prog = (ops.ChainOp * 1)(ops.ChainOp(kind=ops.CHAIN_LOAD, src=-1, dst=0, res=-1, N=256, p0=ptr.value))
    job = ops.FillJob(ptr.value, ptr.value, None, 0, 9)
    jobs = (ops.FillJob * 1)(job)
    call = lambda j=jobs, n=1, start=ptr, rec=ptr, hh=8, p=4, wg=0: lib.gd4d_row_chain_fill_fwd(prog, 1, null, 0, 16, j, n, start, rec, 1, 6, hh, p, wg, null)
    assert call(j=null) == EINVAL
    assert call(n=3) == EINVAL
    assert call(start=null) == EINVAL
    assert call(rec=null) == EINVAL
    assert call(wg=-1) == EINVAL
    assert call(p=5) == EUNSUPPORTED
    assert call(hh=32) == EUNSUPPORTED
    far = (ops.FillJob * 1)(ops.FillJob(ptr.value, ptr.value, None, (1 << 26) - 8, 9))
    assert call(j=far) == EUNSUPPORTED
    hole = (ops.FillJob * 1)(ops.FillJob(ptr.value, None, None, 0, 9))
    assert call(j=hole) == EINVAL


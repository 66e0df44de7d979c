// Backward kernels of the decoder's query side (training): LayerNorm and the self-attention core.
//
// The reference trains through ATen autograd: layer_norm_backward (three kernels per LayerNorm), softmax / bmm
// backward inside nn.MultiheadAttention (config ...ceph.py:74-78).  Round 1 of this build left those to ATen; here they
// are hand-written so that a training step's kernel table is gd4d:: only (plus optimizer and collectives).
// Everything fp32; reductions in a fixed order (run-to-run identical results).
#include "gd4d_common.h"
#include "gd4d_mha_dropout.h"
#include "gd4d_pyramid_fill.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(4))) float t4;

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = [ReLU] LN(x [+ res]) * gamma + beta  (gd4d_layernorm_fwd); given dy:
//   g = dy * gamma (dy masked by y > 0 with ReLU), xhat = (x - mean) * rstd,
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),   dgamma = sum_rows dy * xhat,   dbeta = sum_rows dy.
// mean / rstd are recomputed from x (two-pass, as the forward), nothing is saved by the forward.
// Kernel 1: one workgroup = 16 rows (16 waves x 1 row for C <= 256, else 4 waves x 4 rows), C <= 1024; writes dx and the workgroup's partial dgamma / dbeta.
// Kernel 2: adds the partials in workgroup order.
struct LnBwdParams {
  const float* x; const float* res; const float* gamma; const float* beta; const float* dy;
  float* dx; float* part;      // part: [workgroups][2][C]
  int M, C, relu;
  float eps;
};

__device__ __forceinline__ float t_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// VPL = float4s of a row per lane (C <= 256 VPL), WAVES x RPW = 16 rows.  C <= 256 runs one row per wave (16 waves): the four
// dependent wave reductions of a row (mean, variance, the two sums of the gradient) are a latency chain of ~2 us behind a
// cold load, and a wave that walks four rows pays it four times (10.7 us per launch in the step, 30 launches).
template <int VPL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void layernorm_bwd_kernel(const LnBwdParams p) {
  constexpr int RPW = 16 / WAVES;
  __shared__ float s_part[WAVES][2][256 * VPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = p.C / 4;
  float4 ag[VPL], ab[VPL];                                   // this wave's partial dgamma / dbeta for the lane's channels
#pragma unroll
  for (int i = 0; i < VPL; ++i) { ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i]; }
  float4 gm[VPL], bt[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = min(lane + 64 * i, nv - 1);
    gm[i] = reinterpret_cast<const float4*>(p.gamma)[c];
    bt[i] = p.relu ? reinterpret_cast<const float4*>(p.beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = blockIdx.x * 16 + wave * RPW + rr;
    if (row >= p.M) break;                               // wave-uniform
    const float4* x = reinterpret_cast<const float4*>(p.x + (size_t)row * p.C);
    const float4* r = p.res ? reinterpret_cast<const float4*>(p.res + (size_t)row * p.C) : nullptr;
    const float4* dy = reinterpret_cast<const float4*>(p.dy + (size_t)row * p.C);
    float4 v[VPL], d[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f); d[i] = v[i];
      if (c < nv) {
        v[i] = x[c];
        if (r) { const float4 t = r[c]; v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w; }
        d[i] = dy[c];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = t_wave_sum(s) / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (lane + 64 * i < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, c2 = v[i].z - mean, e = v[i].w - mean;
        q += (a * a + b * b) + (c2 * c2 + e * e);
      }
    const float rstd = 1.0f / sqrtf(t_wave_sum(q) / (float)p.C + p.eps);
    float s1 = 0.f, s2 = 0.f;
    float4 xh[VPL], g[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      xh[i] = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
      if (p.relu) {                                      // the forward clamped y at 0: those outputs pass no gradient
        if (xh[i].x * gm[i].x + bt[i].x <= 0.f) d[i].x = 0.f;
        if (xh[i].y * gm[i].y + bt[i].y <= 0.f) d[i].y = 0.f;
        if (xh[i].z * gm[i].z + bt[i].z <= 0.f) d[i].z = 0.f;
        if (xh[i].w * gm[i].w + bt[i].w <= 0.f) d[i].w = 0.f;
      }
      g[i] = make_float4(d[i].x * gm[i].x, d[i].y * gm[i].y, d[i].z * gm[i].z, d[i].w * gm[i].w);
      if (lane + 64 * i < nv) {
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
        ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
      }
    }
    const float m1 = t_wave_sum(s1) / (float)p.C, m2 = t_wave_sum(s2) / (float)p.C;
    float4* dx = reinterpret_cast<float4*>(p.dx + (size_t)row * p.C);
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (lane + 64 * i < nv)
        dx[lane + 64 * i] = make_float4(rstd * (g[i].x - m1 - xh[i].x * m2), rstd * (g[i].y - m1 - xh[i].y * m2),
                                        rstd * (g[i].z - m1 - xh[i].z * m2), rstd * (g[i].w - m1 - xh[i].w * m2));
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i)
    if (lane + 64 * i < nv) {
      reinterpret_cast<float4*>(s_part[wave][0])[lane + 64 * i] = ag[i];
      reinterpret_cast<float4*>(s_part[wave][1])[lane + 64 * i] = ab[i];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * p.C; e += 64 * WAVES) {
    const int k = e / p.C, c = e - k * p.C;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; w += 4) t += (s_part[w][k][c] + s_part[w + 1][k][c]) + (s_part[w + 2][k][c] + s_part[w + 3][k][c]);
    p.part[((size_t)blockIdx.x * 2 + k) * p.C + c] = t;
  }
}

// 64 columns per workgroup, four threads per column over the partial rows (independent loads), added in a fixed order
// (one thread per column walking all partial rows was a chain of dependent loads: 14 us)
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ part, float* dgamma, float* dbeta,
                                                                  int parts, int C, int accumulate) {
  __shared__ float s_p[4][64];
  const int col = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  const int k = e / C, c = e - k * C;
  float s = 0.f;
  if (e < 2 * C) {
#pragma unroll 4
    for (int w = pg; w < parts; w += 4) s += part[((size_t)w * 2 + k) * C + c];
  }
  s_p[pg][col] = s;
  __syncthreads();
  if (pg == 0 && e < 2 * C) {
    const float t = ((s_p[0][col] + s_p[1][col]) + s_p[2][col]) + s_p[3][col];
    float* d = (k == 0 ? dgamma : dbeta) + c;
    *d = accumulate ? *d + t : t;
  }
}

// The column reduces of up to LN_GROUP LayerNorm backward calls in ONE launch (a training step queues them: the parameter
// gradients are needed by nobody before the optimizer, and thirty 4-us launches are thirty launches).
constexpr int LN_GROUP = 32;

struct LnReduceGroup {
  const float* part[LN_GROUP];
  float* dgamma[LN_GROUP];
  float* dbeta[LN_GROUP];
  int parts[LN_GROUP];
  int C[LN_GROUP];
  int blk0[LN_GROUP + 1];       // first workgroup of problem i (64 columns of its 2 C per workgroup)
  int count, accumulate;
};

__global__ __launch_bounds__(256) void layernorm_bwd_reduce_group_kernel(const LnReduceGroup g) {
  __shared__ float s_p[4][64];
  int i = 0;
#pragma unroll
  for (int j = 1; j < LN_GROUP; ++j)
    if (j < g.count && (int)blockIdx.x >= g.blk0[j]) i = j;
  const float* part = g.part[0];
  float* dgamma = g.dgamma[0];
  float* dbeta = g.dbeta[0];
  int parts = g.parts[0], C = g.C[0], b0 = g.blk0[0];
#pragma unroll
  for (int j = 1; j < LN_GROUP; ++j)
    if (j == i) { part = g.part[j]; dgamma = g.dgamma[j]; dbeta = g.dbeta[j]; parts = g.parts[j]; C = g.C[j]; b0 = g.blk0[j]; }
  const int col = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int e = ((int)blockIdx.x - b0) * 64 + col;
  const int k = e / C, c = e - k * C;
  float s = 0.f;
  if (e < 2 * C) {
#pragma unroll 4
    for (int w = pg; w < parts; w += 4) s += part[((size_t)w * 2 + k) * C + c];
  }
  s_p[pg][col] = s;
  __syncthreads();
  if (pg == 0 && e < 2 * C) {
    const float t = ((s_p[0][col] + s_p[1][col]) + s_p[2][col]) + s_p[3][col];
    float* d = (k == 0 ? dgamma : dbeta) + c;
    *d = g.accumulate ? *d + t : t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Self-attention core backward.  Forward (gd4d_mha_core_fwd): P = softmax(scale q k^T + mask), o = P v, per head
// (D = 32), with lse[q] = log sum_k exp(scale q k^T + mask) saved per (query, batch, head).  Given do:
//   Dq = sum_d do[q][d] o[q][d];  dP = do v^T;  dS = P o (dP - Dq);  dq = scale dS k;  dk = scale dS^T q;  dv = P^T do.
// With dropout of the probabilities (o = (P o M) v, M = keep / (1 - p), gd4d_mha_dropout.h): dv = (P o M)^T do and
// dS = P o (M o dP - Dq); Dq = sum_k P M dP is still sum_d do o.  M is regenerated from the seed, not stored.
// Two kernels with the forward's register layout (transposed score tiles on v_mfma_f32_16x16x4_f32: the C/D layout of
// the score tile is the B-operand layout of the product that consumes it, so probabilities never move between lanes):
//   dq kernel:    workgroup = 16 queries, waves split the key tiles; also writes Dq for the second kernel
//   dk/dv kernel: workgroup = 16 keys, waves split the query tiles
struct MhaBwdParams {
  const float* q; const float* k; const float* v; const float* o; const float* dout;   // rows (l*B + b), head h at +32 h
  const void* mask;
  const float* lse;       // (Lq, B, H)
  float* dsum;            // (Lq, B, H): Dq (written by the dq kernel, read by the dk/dv kernel)
  float* dq; float* dk; float* dv;
  int Lq, Lk, B, H, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, mask_kind;
  float scale;
  const uint32_t* seed; uint32_t drop_thresh; float inv_keep;
};

#ifndef TB_WAVES_N
#define TB_WAVES_N 8
#endif
constexpr int TB_D = 32, TB_WAVES = TB_WAVES_N;

// A 16 x 16 score-like tile, transposed: T^T[row][col] = sum_d R[row][d] C[col][d], rows permuted so that lane
// (col = lane & 15, g = lane >> 4) register r holds row g + 4 r.  rowv = the row operand's 8 values of this lane (row
// rho = (lane & 15 >> 2) + 4 (lane & 3), dims 8 g ..), colv = the column operand's 8 values (col lane & 15, dims 8 g ..).
__device__ __forceinline__ t4 tb_tile(const float* rowv, const float* colv) {
  t4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) s = __builtin_amdgcn_mfma_f32_16x16x4f32(rowv[i], colv[i], s, 0, 0, 0);
  return s;
}

// The same tile on the 32-deep bf16 MFMA with split operands (x = hi + lo; hi hi + lo hi + hi lo: ~2^-16 relative per product, the
// arithmetic of the forward kernel): a lane's eight values ARE its k-group of the 16 x 16 x 32 instruction (dims 8 g .. 8 g + 7) and
// the C/D rows come out in the same order, so the operands and the row permutation above carry over unchanged - 3 instructions
// instead of 8.  -DTB_SCORES_FP32=1 keeps the fp32 MFMA.
typedef __attribute__((ext_vector_type(8))) __bf16 tb_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned tb_u4;
struct TbSplit { tb_u4 h, l; };
__device__ __forceinline__ TbSplit tb_split8(const float* v) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hh[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
    const float r0 = v[2 * i] - __uint_as_float(hh[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ll[i]) : "v"(r0), "v"(r1));
  }
  return TbSplit{tb_u4{hh[0], hh[1], hh[2], hh[3]}, tb_u4{ll[0], ll[1], ll[2], ll[3]}};
}
__device__ __forceinline__ t4 tb_tile3(const TbSplit& row, const TbSplit& col) {
  t4 s = {0.f, 0.f, 0.f, 0.f};
  s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tb_bf16x8, row.l), __builtin_bit_cast(tb_bf16x8, col.h), s, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tb_bf16x8, row.h), __builtin_bit_cast(tb_bf16x8, col.l), s, 0, 0, 0);
  s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tb_bf16x8, row.h), __builtin_bit_cast(tb_bf16x8, col.h), s, 0, 0, 0);
  return s;
}
#ifndef TB_SCORES_FP32
#define TB_SCORES_FP32 0
#endif

__device__ __forceinline__ void tb_load8(const float* src, float* dst, float mul) {
  const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
  dst[0] = a.x * mul; dst[1] = a.y * mul; dst[2] = a.z * mul; dst[3] = a.w * mul;
  dst[4] = c.x * mul; dst[5] = c.y * mul; dst[6] = c.z * mul; dst[7] = c.w * mul;
}

// SIDE = 0: the workgroup owns 16 queries (columns of the transposed tiles), loops over key tiles (rows): dq.
// SIDE = 1: the workgroup owns 16 keys (columns), loops over query tiles (rows): dk and dv.
template <int SIDE, bool DROP>
__device__ __forceinline__ void mha_bwd_body(const MhaBwdParams& p, const int bx, const int by, const int bz,
                                             float (&s_acc)[TB_WAVES][2][TB_D][17]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci = lane & 15, g = lane >> 4;
  const int c0 = bx * 16;
  const int h = by, b = bz;
  const size_t hoff = (size_t)h * TB_D;
  const int Lc = SIDE == 0 ? p.Lq : p.Lk;               // extent of the column (owned) side
  const int Lr = SIDE == 0 ? p.Lk : p.Lq;               // extent of the row (looped) side
  const int rho = (ci >> 2) + 4 * (ci & 3);             // row (within a tile) this lane feeds as MFMA A row

  // column-side operands of this lane (column ci, dims 8 g .. 8 g + 7)
  const int crow = min(c0 + ci, Lc - 1);
  float cq[8], cd[8];                                    // SIDE 0: scaled q, do;  SIDE 1: k, v
  float c_lse = 0.f, c_dsum = 0.f;
  if (SIDE == 0) {
    tb_load8(p.q + ((size_t)crow * p.B + b) * p.ldq + hoff + 8 * g, cq, p.scale);
    tb_load8(p.dout + ((size_t)crow * p.B + b) * p.lddo + hoff + 8 * g, cd, 1.f);
    float ov[8];
    tb_load8(p.o + ((size_t)crow * p.B + b) * p.ldo + hoff + 8 * g, ov, 1.f);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += cd[i] * ov[i];
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
    c_dsum = d;
    c_lse = p.lse[((size_t)crow * p.B + b) * p.H + h];
    if (wave == 0 && g == 0 && c0 + ci < Lc) p.dsum[((size_t)crow * p.B + b) * p.H + h] = d;
  } else {
    tb_load8(p.k + ((size_t)crow * p.B + b) * p.ldk + hoff + 8 * g, cq, 1.f);
    tb_load8(p.v + ((size_t)crow * p.B + b) * p.ldv + hoff + 8 * g, cd, 1.f);
  }

  const TbSplit cq3 = tb_split8(cq), cd3 = tb_split8(cd);          // the column side's operands: split once
  uint32_t seed_lo = 0, seed_hi = 0;
  if (DROP) { seed_lo = p.seed[0]; seed_hi = p.seed[1]; }
  const uint32_t drop_base = DROP ? mha_drop_row(b, h, 0, p.H, p.Lq, p.Lk) : 0u;
  t4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;   // SIDE 0: dq^T (a);  SIDE 1: dk^T (a), dv^T (b)
  const int ntiles = (Lr + 15) / 16;
  // Everything a tile reads from memory, requested one tile ahead of its use (a wave has ~2 neighbours on its SIMD: without
  // the look-ahead every tile pays a full round trip to the L2).  Both kernels at 900 x 900 x 8 heads, back to back from a
  // graph (tools/bench_train_small.py): 75.5 us without look-ahead (4 waves), 60.9 / 62.6 / 81.3 us with it and 4 / 8 / 16
  // waves per workgroup; with the H-DETR bool mask 100.5 -> 82.9 / 79.1 / 98.6 us.  2.9 GFLOP (S and dP are formed in both
  // kernels): 47 TFLOP/s of fp32 MFMA, the forward kernel's rate (0.83 GFLOP in 21 us).
  struct Tile {
    float rk[8], rv[8];          // SIDE 0: k, v rows;  SIDE 1: scaled q, do rows (row rho of the tile, dims 8 g ..)
    float x0[4], x1[4];          // SIDE 0: k^T[d = ci (+16)][row g + 4 st];  SIDE 1: q^T
    float y0[4], y1[4];          // SIDE 1: do^T
    float lse[4], dsum[4];       // SIDE 1: statistics of query row g + 4 r
  };
  auto load_tile = [&](int rt, Tile& t) {
    const int rbase = rt * 16;
    const int rrow = min(rbase + rho, Lr - 1);
    if (SIDE == 0) {
      tb_load8(p.k + ((size_t)rrow * p.B + b) * p.ldk + hoff + 8 * g, t.rk, 1.f);
      tb_load8(p.v + ((size_t)rrow * p.B + b) * p.ldv + hoff + 8 * g, t.rv, 1.f);
    } else {                                             // (scale q in the score; the plain q below for dk)
      tb_load8(p.q + ((size_t)rrow * p.B + b) * p.ldq + hoff + 8 * g, t.rk, p.scale);
      tb_load8(p.dout + ((size_t)rrow * p.B + b) * p.lddo + hoff + 8 * g, t.rv, 1.f);
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int xr = min(rbase + g + 4 * st, Lr - 1);    // rows past the end carry pr = ds = 0
      if (SIDE == 0) {
        const float* ks = p.k + ((size_t)xr * p.B + b) * p.ldk + hoff + ci;
        t.x0[st] = ks[0]; t.x1[st] = ks[16];
      } else {
        const float* qs = p.q + ((size_t)xr * p.B + b) * p.ldq + hoff + ci;
        const float* os = p.dout + ((size_t)xr * p.B + b) * p.lddo + hoff + ci;
        t.x0[st] = qs[0]; t.x1[st] = qs[16]; t.y0[st] = os[0]; t.y1[st] = os[16];
        const size_t li = ((size_t)xr * p.B + b) * p.H + h;
        t.lse[st] = p.lse[li]; t.dsum[st] = p.dsum[li];
      }
    }
  };
  Tile cur;
  if (wave < ntiles) load_tile(wave, cur);
  for (int rt = wave; rt < ntiles; rt += TB_WAVES) {
    const int rbase = rt * 16;
    Tile nxt;
    load_tile(min(rt + TB_WAVES, ntiles - 1), nxt);
    // S^T and dP^T tiles: lane (col ci, g) register r <-> row rbase + g + 4 r
    const t4 s = TB_SCORES_FP32 ? tb_tile(cur.rk, cq) : tb_tile3(tb_split8(cur.rk), cq3);
    const t4 dp = TB_SCORES_FP32 ? tb_tile(cur.rv, cd) : tb_tile3(tb_split8(cur.rv), cd3);
    float pr[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rowi = rbase + g + 4 * r;
      const int qi = SIDE == 0 ? c0 + ci : rowi, ki = SIDE == 0 ? rowi : c0 + ci;
      float val = s[r];
      bool dead = rowi >= Lr;
      if (!dead && p.mask_kind) {
        const size_t mi = (size_t)min(qi, p.Lq - 1) * p.Lk + min(ki, p.Lk - 1);
        if (p.mask_kind == 1) dead = static_cast<const uint8_t*>(p.mask)[mi] != 0;
        else val += static_cast<const float*>(p.mask)[mi];
      }
      const float lse = SIDE == 0 ? c_lse : cur.lse[r], dsum = SIDE == 0 ? c_dsum : cur.dsum[r];
      const float pv = dead ? 0.f : __builtin_amdgcn_exp2f((val - lse) * 1.4426950408889634f);   // one v_exp_f32 (as the forward kernels), not expf's range reduction
      float mk = 1.f;
      if (DROP) {
        const uint32_t id = drop_base + (uint32_t)min(qi, p.Lq - 1) * (uint32_t)p.Lk + (uint32_t)min(ki, p.Lk - 1);
        mk = mha_drop_keep(seed_lo, seed_hi, id, p.drop_thresh) ? p.inv_keep : 0.f;
      }
      pr[r] = pv * mk;
      ds[r] = pv * (dp[r] * mk - dsum);
    }
    // products with the row side as reduction index: A = X^T[d = ci (+16)][row = rbase + g + 4 st], B = pr / ds
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x0[st], ds[st], a0, 0, 0, 0);   // dq^T += K^T dS^T  |  dk^T += Q^T dS
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x1[st], ds[st], a1, 0, 0, 0);
      if (SIDE == 1) {                                                              // dv^T += dO^T P
        b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y0[st], pr[st], b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y1[st], pr[st], b1, 0, 0, 0);
      }
    }
    cur = nxt;
  }
  // merge the waves (fixed order); accumulators: d = 4 g + r (a0 / b0), 16 + 4 g + r (a1 / b1), column ci
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s_acc[wave][0][4 * g + r][ci] = a0[r];
    s_acc[wave][0][16 + 4 * g + r][ci] = a1[r];
    if (SIDE == 1) { s_acc[wave][1][4 * g + r][ci] = b0[r]; s_acc[wave][1][16 + 4 * g + r][ci] = b1[r]; }
  }
  __syncthreads();
  for (int e = tid; e < 16 * TB_D; e += 64 * TB_WAVES) {
    const int i = e / TB_D, d = e % TB_D;
    if (c0 + i >= Lc) continue;
    const size_t rowo = (size_t)(c0 + i) * p.B + b;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int w = 0; w < TB_WAVES; ++w) { s0 += s_acc[w][0][d][i]; if (SIDE == 1) s1 += s_acc[w][1][d][i]; }
    if (SIDE == 0) p.dq[rowo * p.lddq + hoff + d] = s0 * p.scale;       // (q was scaled going in: d(scale q)/dq)
    else { p.dk[rowo * p.lddk + hoff + d] = s0 * p.scale; p.dv[rowo * p.lddv + hoff + d] = s1; }   // rows were q, not scale q
  }
}

template <int SIDE, bool DROP>
__global__ __launch_bounds__(64 * TB_WAVES) void mha_bwd_kernel(const MhaBwdParams p) {
  __shared__ float s_acc[TB_WAVES][2][TB_D][17];
  mha_bwd_body<SIDE, DROP>(p, blockIdx.x, blockIdx.y, blockIdx.z, s_acc);
}

// The dk / dv kernel with up to two layers' record fills of the pyramid gradient as guest workgroups (gd4d_mha_core_bwd_fill;
// gd4d_pyramid_fill.h).  The fills of a training step need the scan over ALL layers' counts and nothing from the backward pass:
// six launches of 22 us sat between the passes.  They stream plans and scatter 8-byte records - memory work; this kernel is
// bound by its matrix and vector pipes.  (Beside the backward gather-dot, which lives on the fabric too, they hid nothing.)
struct MhaBwdFillArgs {
  MhaBwdParams p;
  FillGuest fg;
  int nx;                      // key tiles (the host grid's x extent)
};

template <bool DROP>
__global__ __launch_bounds__(64 * TB_WAVES) void mha_bwd_fill_kernel(const MhaBwdFillArgs a) {
  __shared__ float s_acc[TB_WAVES][2][TB_D][17];
  int host;
  if (fill_guest_or_host(a.fg, host)) return;
  const int bx = host % a.nx, rest = host / a.nx;
  if (rest >= a.p.H * a.p.B) return;
  mha_bwd_body<1, DROP>(a.p, bx, rest % a.p.H, rest / a.p.H, s_acc);
}

}  // namespace gd4d

extern "C" size_t gd4d_layernorm_bwd_workspace_bytes(int M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)((M + 15) / 16) * 2 * C * sizeof(float);
}

extern "C" int gd4d_layernorm_bwd(const float* x, const float* res, const float* gamma, const float* beta, const float* dy,
                                  float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int M,
                                  int C, float eps, int flags, void* stream) {
  using namespace gd4d;
  if (!x || !gamma || !dy || !dx || !workspace || M <= 0 || C <= 0) return GD4D_EINVAL;
  if (!(flags & GD4D_LN_DEFER_REDUCE) && (!dgamma || !dbeta)) return GD4D_EINVAL;
  const int relu = flags & GD4D_LN_RELU;
  if (relu && !beta) return GD4D_EINVAL;
  if (C % 4 != 0 || C > 1024) return GD4D_EUNSUPPORTED;
  if (workspace_bytes < gd4d_layernorm_bwd_workspace_bytes(M, C)) return GD4D_EINVAL;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !aligned16(gamma) || (res && !aligned16(res)) ||
      (beta && !aligned16(beta)))
    return GD4D_EALIGN;
  LnBwdParams p{x, res, gamma, beta, dy, dx, static_cast<float*>(workspace), M, C, relu ? 1 : 0, eps};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int parts = (M + 15) / 16;
  if (C <= 256) hipLaunchKernelGGL((layernorm_bwd_kernel<1, 16>), dim3(parts), dim3(1024), 0, st, p);
  else hipLaunchKernelGGL((layernorm_bwd_kernel<4, 4>), dim3(parts), dim3(256), 0, st, p);
  if (int rc = check_launch()) return rc;
  if (flags & GD4D_LN_DEFER_REDUCE) return GD4D_OK;       // the caller reduces the partials later (gd4d_layernorm_bwd_reduce_group)
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * C + 63) / 64), dim3(256), 0, st, p.part, dgamma, dbeta, parts, C,
                     (flags & GD4D_LN_ACCUMULATE) ? 1 : 0);
  return check_launch();
}

static int mha_core_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* dout,
                             const void* mask, const float* lse, float* dsum, float* dq, float* dk, float* dv, int Lq,
                             int Lk, int B, int H, int D, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq,
                             int lddk, int lddv, int mask_kind, float scale, float drop_p, const void* seed, void* stream,
                             gd4d::FillGuest* fg) {
  using namespace gd4d;
  if (!q || !k || !v || !o || !dout || !lse || !dsum || !dq || !dk || !dv || Lq <= 0 || Lk <= 0 || B <= 0 || H <= 0)
    return GD4D_EINVAL;
  if (!(drop_p >= 0.f && drop_p < 1.f) || (drop_p > 0.f && !seed)) return GD4D_EINVAL;
  if (drop_p > 0.f && (double)B * H * Lq * Lk >= 4294967296.0) return GD4D_EUNSUPPORTED;
  if (D != TB_D || mask_kind < 0 || mask_kind > 2 || (mask_kind && !mask)) return GD4D_EUNSUPPORTED;
  const int ld_min = H * D;
  if (ldq < ld_min || ldk < ld_min || ldv < ld_min || ldo < ld_min || lddo < ld_min || lddq < ld_min || lddk < ld_min ||
      lddv < ld_min)
    return GD4D_EINVAL;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o) || !aligned16(dout) || (ldq % 4) || (ldk % 4) ||
      (ldv % 4) || (ldo % 4) || (lddo % 4))
    return GD4D_EALIGN;
  MhaBwdParams p{q, k, v, o, dout, mask, lse, dsum, dq, dk, dv, Lq, Lk, B, H, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv,
                 mask_kind, scale, static_cast<const uint32_t*>(seed), mha_drop_thresh(drop_p), 1.f / (1.f - drop_p)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 gq((Lq + 15) / 16, H, B), gk((Lk + 15) / 16, H, B), block(64 * TB_WAVES);
  if (drop_p > 0.f) hipLaunchKernelGGL((mha_bwd_kernel<0, true>), gq, block, 0, st, p);
  else hipLaunchKernelGGL((mha_bwd_kernel<0, false>), gq, block, 0, st, p);
  if (int rc = check_launch()) return rc;
  if (fg) {                                              // the dk / dv kernel carries the fills
    MhaBwdFillArgs a{};
    a.p = p; a.fg = *fg; a.nx = (int)gk.x;
    const int hosts = (int)(gk.x * gk.y * gk.z);
    const int wgs = fg->wgs0 + (fg->hdr[1] ? (fg->BQ[1] * fg->HH + 7) / 8 : 0);
    a.fg.guest_groups = (wgs + 7) / 8;
    a.fg.total_groups = (hosts + 7) / 8 + a.fg.guest_groups;
    const dim3 grid(8 * a.fg.total_groups);
    if (drop_p > 0.f) hipLaunchKernelGGL((mha_bwd_fill_kernel<true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((mha_bwd_fill_kernel<false>), grid, block, 0, st, a);
    return check_launch();
  }
  if (drop_p > 0.f) hipLaunchKernelGGL((mha_bwd_kernel<1, true>), gk, block, 0, st, p);
  else hipLaunchKernelGGL((mha_bwd_kernel<1, false>), gk, block, 0, st, p);
  return check_launch();
}

extern "C" int gd4d_mha_core_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                 const void* mask, const float* lse, float* dsum, float* dq, float* dk, float* dv, int Lq,
                                 int Lk, int B, int H, int D, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq,
                                 int lddk, int lddv, int mask_kind, float scale, float drop_p, const void* seed, void* stream) {
  return mha_core_bwd_impl(q, k, v, o, dout, mask, lse, dsum, dq, dk, dv, Lq, Lk, B, H, D, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv,
                           mask_kind, scale, drop_p, seed, stream, nullptr);
}

extern "C" int gd4d_mha_core_bwd_fill(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                      const void* mask, const float* lse, float* dsum, float* dq, float* dk, float* dv, int Lq,
                                      int Lk, int B, int H, int D, int ldq, int ldk, int ldv, int ldo, int lddo, int lddq,
                                      int lddk, int lddv, int mask_kind, float scale, float drop_p, const void* seed,
                                      const gd4d_fill_job* jobs, int njobs, const int32_t* start, void* records, int fill_B,
                                      int fill_N, int fill_Hh, int fill_P, void* stream) {
  using namespace gd4d;
  if (!jobs || njobs < 1 || njobs > 2 || !start || !records || fill_B <= 0 || fill_N <= 0 || fill_Hh <= 0) return GD4D_EINVAL;
  if ((fill_P != kPoints && fill_P != 8) || fill_N > 64 || fill_B > 16 || fill_Hh > kPlanHdr) return GD4D_EUNSUPPORTED;
  FillGuest fg{};
  for (int j = 0; j < njobs; ++j) {
    const gd4d_fill_job& jb = jobs[j];
    if (!jb.plan || !jb.slots || jb.Q <= 0) return GD4D_EINVAL;
    if ((unsigned long long)jb.id_base + (unsigned long long)fill_B * jb.Q * fill_Hh > (1ull << 26)) return GD4D_EUNSUPPORTED;
    fg.hdr[j] = static_cast<const int*>(jb.plan);
    fg.pair[j] = reinterpret_cast<const uint2*>(static_cast<const char*>(jb.plan) + plan_hdr_bytes(fill_B, jb.Q));
    fg.slots[j] = static_cast<const uint2*>(jb.slots);
    fg.order[j] = jb.query_order;
    fg.id_base[j] = jb.id_base;
    fg.BQ[j] = fill_B * jb.Q;
  }
  fg.HH = fill_Hh;
  fg.cap_t = plan_cap_t(fill_N, fill_P);
  fg.wgs0 = (fg.BQ[0] * fill_Hh + 7) / 8;
  fg.start = start;
  fg.rec = static_cast<uint2*>(records);
  return mha_core_bwd_impl(q, k, v, o, dout, mask, lse, dsum, dq, dk, dv, Lq, Lk, B, H, D, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv,
                           mask_kind, scale, drop_p, seed, stream, &fg);
}

extern "C" int gd4d_layernorm_bwd_reduce_group(const void* const* workspaces, void* const* dgamma, void* const* dbeta, const int32_t* dims,
                                               int count, int accumulate, void* stream) {
  using namespace gd4d;
  if (!workspaces || !dgamma || !dbeta || !dims || count <= 0) return GD4D_EINVAL;
  if (count > LN_GROUP) return GD4D_EUNSUPPORTED;
  LnReduceGroup g{};
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const int M = dims[2 * i], C = dims[2 * i + 1];
    if (!workspaces[i] || !dgamma[i] || !dbeta[i] || M <= 0 || C <= 0) return GD4D_EINVAL;
    g.part[i] = static_cast<const float*>(workspaces[i]);
    g.dgamma[i] = static_cast<float*>(dgamma[i]); g.dbeta[i] = static_cast<float*>(dbeta[i]);
    g.parts[i] = (M + 15) / 16; g.C[i] = C; g.blk0[i] = blocks;
    blocks += (2 * C + 63) / 64;
  }
  for (int i = count; i <= LN_GROUP; ++i) g.blk0[i] = blocks;
  for (int i = count; i < LN_GROUP; ++i) { g.part[i] = g.part[0]; g.dgamma[i] = g.dgamma[0]; g.dbeta[i] = g.dbeta[0]; g.parts[i] = 0; g.C[i] = 1; }
  g.count = count; g.accumulate = accumulate ? 1 : 0;
  hipLaunchKernelGGL(layernorm_bwd_reduce_group_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  return check_launch();
}

// gd4d_linear_ln_fwd: Linear (+bias, ReLU, two residuals) with an optional LayerNorm in the epilogue, one workgroup
// per block of 16 complete rows.
//
// The reference runs nn.Linear followed by nn.LayerNorm as separate ATen kernels at four places of every decoder layer:
//   nn.MultiheadAttention out_proj -> norms[0], Deform3DCrossAttn.output_proj (+ residual + pos_feat) -> norms[1]
//   (deform3d_cross_attn.py:326-336), mmcv FFN layers[1] (+ residual) -> norms[2], position_encoder[3] -> [4] -> ReLU
//   (deform3d_cross_attn.py:108-110).
// gd4d_linear_fwd tiles the output 32 x 32 over many small workgroups: lowest latency on an idle GPU, but LayerNorm
// needs whole rows, and when value_proj of the next layer holds three quarters of the CUs (docs/design_notes_r01_r03.md 4.5) hundreds of
// small workgroups queue up in rounds.  Here a workgroup owns 16 rows x up to 256 columns (4 waves x 64 columns, each
// wave the full K), so ceil(M / 16) = 57 workgroups do the whole layer, the row statistics are two LDS exchanges, and
// LayerNorm costs no extra launch.
//
// MFMA: v_mfma_f32_16x16x4_f32, exact fp32 products (same numerics class as gd4d_linear_fwd).  Operands go from global
// memory (L2) straight into registers in MFMA layout - lane (i = l & 15, g = l >> 4) loads one float4 of row i at
// k = 16 j + 4 g per 16-k step, for the x row block once and for each of the wave's four 16-column W tiles - through a
// 4-deep register ring so four steps of loads are always in flight behind the 16 MFMAs of the current step.
#include "gd4d_common.h"

#ifndef GD4D_RB_DBG
#define GD4D_RB_DBG 0     // dev ablation: 1 = no loads inside the K loop, 2 = no MFMAs
#endif

namespace gd4d {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct RowBlockParams {
  const float* x;      // (M, K), row stride ldx
  const float* x2;     // optional addend on the input rows for output columns < n_split
  const float* w;      // (N, K) row-major
  const float* bias;   // (N) or null
  const float* r1;     // optional residual (M, N), row stride ldr1
  const float* r2;     // optional second residual, row stride ldr2
  const float* gamma;  // LayerNorm weight (N) or null: no LayerNorm
  const float* beta;
  float* y;            // (M, N), row stride ldy
  int M, K, N, n_split, flags;   // flags: bit0 ReLU before the residuals, bit2 ReLU after LayerNorm
  int ldx, ldy, ldr1, ldr2;
  float eps;
};

constexpr int RB_M = 16, RB_WAVES = 4, RB_TILES = 4, RB_N = 16 * RB_TILES * RB_WAVES, RB_DEPTH = 4;

// X2 / NRES / LN are compile-time: a runtime null-pointer test inside the load ring costs a branch and a full
// s_waitcnt vmcnt(0) per step (seen in the ISA), which serialises every load behind the MFMAs.
template <bool X2, int NRES, bool LN>
__global__ __launch_bounds__(64 * RB_WAVES) void rowblock_linear_kernel(const RowBlockParams p) {
  __shared__ float s_red[2][RB_WAVES][RB_M];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * RB_M;
  const int n0 = blockIdx.y * RB_N + wave * (16 * RB_TILES);        // first column of this wave
  // block-uniform (n_split % 256 == 0): column blocks past n_split still load x2 but scale it by 0 (branch-free)
  const float add2 = (X2 && (int)(blockIdx.y * RB_N) < p.n_split) ? 1.f : 0.f;

  const float* xr = p.x + (size_t)min(m0 + i16, p.M - 1) * p.ldx + 4 * g;
  const float* x2r = X2 ? p.x2 + (size_t)min(m0 + i16, p.M - 1) * p.ldx + 4 * g : nullptr;
  const float* wr[RB_TILES];
#pragma unroll
  for (int c = 0; c < RB_TILES; ++c) wr[c] = p.w + (size_t)min(n0 + 16 * c + i16, p.N - 1) * p.K + 4 * g;

  f32x4 acc[RB_TILES];
#pragma unroll
  for (int c = 0; c < RB_TILES; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra[RB_DEPTH], rb[RB_DEPTH][RB_TILES];
  f32x4 ra2[X2 ? RB_DEPTH : 1];
  const int steps = p.K / 16;                       // host guarantees K % 64 == 0: steps is a multiple of the ring depth
  auto issue = [&](int slot, int j) {
    const float4 t = *reinterpret_cast<const float4*>(xr + 16 * j);
    ra[slot] = f32x4{t.x, t.y, t.z, t.w};
    if (X2) {                                      // the add happens when the slot is consumed
      const float4 u = *reinterpret_cast<const float4*>(x2r + 16 * j);
      ra2[X2 ? slot : 0] = f32x4{u.x, u.y, u.z, u.w};
    }
#pragma unroll
    for (int c = 0; c < RB_TILES; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(wr[c] + 16 * j);
      rb[slot][c] = f32x4{v.x, v.y, v.z, v.w};
    }
  };
#pragma unroll
  for (int d = 0; d < RB_DEPTH; ++d) issue(d, d);

  // Epilogue operands are requested before the K loop so their latency hides behind it.  Unconditional loads from
  // clamped (always valid) addresses: a load under a lane predicate becomes its own basic block and the compiler
  // closes it with s_waitcnt vmcnt(0) - a dozen serialised L2 round trips before the first MFMA (seen in the ISA).
  float e_bias[RB_TILES], e_res[RB_TILES][4], e_gamma[RB_TILES], e_beta[RB_TILES];
  const float* bias_p = p.bias ? p.bias : p.w;      // any readable address; scaled by 0 below
  const float bias_on = p.bias ? 1.f : 0.f;
#pragma unroll
  for (int c = 0; c < RB_TILES; ++c) {
    const int n = min(n0 + 16 * c + i16, p.N - 1);
    e_bias[c] = bias_p[n] * bias_on;
    e_gamma[c] = LN ? p.gamma[n] : 0.f;
    e_beta[c] = LN ? p.beta[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min(m0 + 4 * g + r, p.M - 1);
      float t = 0.f;
      if (NRES > 0) t = p.r1[(size_t)m * p.ldr1 + n];
      if (NRES > 1) t += p.r2[(size_t)m * p.ldr2 + n];
      e_res[c][r] = t;
    }
  }

  auto consume = [&](int d) {
    f32x4 a = ra[d];
    if (X2) a += ra2[X2 ? d : 0] * add2;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < RB_TILES; ++c)
#if GD4D_RB_DBG & 2
        asm volatile("" ::"v"(a[e]), "v"(rb[d][c][e]));
#else
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], rb[d][c][e], acc[c], 0, 0, 0);
#endif
  };
  for (int j0 = 0; j0 + RB_DEPTH < steps; j0 += RB_DEPTH) {
#pragma unroll
    for (int d = 0; d < RB_DEPTH; ++d) {
      consume(d);
#if !(GD4D_RB_DBG & 1)
      issue(d, j0 + d + RB_DEPTH);
#endif
    }
  }
#pragma unroll
  for (int d = 0; d < RB_DEPTH; ++d) consume(d);    // the last ring's worth: nothing left to load

  // C/D of 16x16x4: col = lane & 15 (+ 16 c), row = 4 * (lane >> 4) + r
  float v[RB_TILES][4];
#pragma unroll
  for (int c = 0; c < RB_TILES; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = acc[c][r] + e_bias[c];
      if (p.flags & 1) t = fmaxf(t, 0.f);
      v[c][r] = t + e_res[c][r];
    }

  if (LN) {                                         // LayerNorm over the N columns of each row (gridDim.y == 1)
    // two-pass statistics like ATen: mean, then the centred sum of squares, biased variance, eps inside the sqrt
    float mean[4], rstd[4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float s[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < RB_TILES; ++c) {
          const bool nv = n0 + 16 * c + i16 < p.N;
          const float d = pass == 0 ? v[c][r] : v[c][r] - mean[r];
          t += nv ? (pass == 0 ? d : d * d) : 0.f;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o);      // over the 16 lanes of a row group
        s[r] = t;
      }
      if (i16 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[pass][wave][4 * g + r] = s[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < RB_WAVES; ++w) t += s_red[pass][w][4 * g + r];
        if (pass == 0) mean[r] = t / (float)p.N;
        else rstd[r] = 1.0f / sqrtf(t / (float)p.N + p.eps);
      }
    }
#pragma unroll
    for (int c = 0; c < RB_TILES; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = (v[c][r] - mean[r]) * rstd[r] * e_gamma[c] + e_beta[c];
        if (p.flags & 4) t = fmaxf(t, 0.f);
        v[c][r] = t;
      }
  }

#pragma unroll
  for (int c = 0; c < RB_TILES; ++c) {
    const int n = n0 + 16 * c + i16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * g + r;
      if (m < p.M && n < p.N) p.y[(size_t)m * p.ldy + n] = v[c][r];
    }
  }
}

}  // namespace gd4d

extern "C" int gd4d_linear_ln_fwd(const float* x, const float* x2, const float* w, const float* bias, const float* r1,
                                  const float* r2, const float* gamma, const float* beta, float* y, int M, int K, int N,
                                  int n_split, int flags, float eps, int ldx, int ldy, int ldr1, int ldr2,
                                  void* stream) {
  using namespace gd4d;
  if (!x || !w || !y) return GD4D_EINVAL;
  if (M <= 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) return GD4D_EINVAL;
  if ((r1 && ldr1 < N) || (r2 && ldr2 < N) || (gamma && !beta)) return GD4D_EINVAL;
  if (K % (16 * RB_DEPTH) != 0 || ldx % 4 != 0) return GD4D_EUNSUPPORTED;
  if (gamma && N > RB_N) return GD4D_EUNSUPPORTED;                   // LayerNorm needs the row in one workgroup
  if (x2 && n_split < N && (n_split % RB_N) != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(w) || (x2 && !aligned16(x2))) return GD4D_EALIGN;
  RowBlockParams p{};
  p.x = x; p.x2 = x2; p.w = w; p.bias = bias; p.r1 = r1; p.r2 = r2; p.gamma = gamma; p.beta = beta; p.y = y;
  p.M = M; p.K = K; p.N = N; p.n_split = x2 ? n_split : 0; p.flags = flags; p.eps = eps;
  p.ldx = ldx; p.ldy = ldy; p.ldr1 = ldr1; p.ldr2 = ldr2;
  const dim3 grid((M + RB_M - 1) / RB_M, (N + RB_N - 1) / RB_N);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nres = (r1 ? 1 : 0) + (r2 ? 1 : 0);
  if (nres == 1 && !r1) { p.r1 = r2; p.ldr1 = ldr2; p.r2 = nullptr; }
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(64 * RB_WAVES), 0, st, p); };
#define GD4D_RB_CASE(X2_, NRES_, LN_) go(rowblock_linear_kernel<X2_, NRES_, LN_>)
  const bool has_ln = gamma != nullptr, has_x2 = x2 != nullptr;
  if (has_x2) {
    if (nres == 0) { if (has_ln) GD4D_RB_CASE(true, 0, true); else GD4D_RB_CASE(true, 0, false); }
    else if (nres == 1) { if (has_ln) GD4D_RB_CASE(true, 1, true); else GD4D_RB_CASE(true, 1, false); }
    else { if (has_ln) GD4D_RB_CASE(true, 2, true); else GD4D_RB_CASE(true, 2, false); }
  } else {
    if (nres == 0) { if (has_ln) GD4D_RB_CASE(false, 0, true); else GD4D_RB_CASE(false, 0, false); }
    else if (nres == 1) { if (has_ln) GD4D_RB_CASE(false, 1, true); else GD4D_RB_CASE(false, 1, false); }
    else { if (has_ln) GD4D_RB_CASE(false, 2, true); else GD4D_RB_CASE(false, 2, false); }
  }
#undef GD4D_RB_CASE
  return check_launch();
}

// gd4d_linear_fwd / gd4d_layernorm_fwd: the decoder's query-side dense layers on the fp32 MFMA.
//
// The reference runs these as nn.Linear / nn.LayerNorm (ATen addmm + elementwise kernels):
//   deform3d_cross_attn.py:211,227,281  camera logits / metre offsets / attention logits of (query+pos)
//   deform3d_cross_attn.py:326,334      output_proj, position_encoder (Linear-LN-ReLU-Linear-LN-ReLU)
//   nn.MultiheadAttention in_proj / out_proj (config ...ceph.py:74-78), mmcv FFN (:86), LayerNorm x3
// Sizes are tiny (M = 900 rows, K <= 512, N <= 768): what matters is launch count and fusing the
// elementwise neighbours, not peak TFLOP/s.  One kernel computes
//     y[m, n] = act( sum_k (x[m,k] + (n < n_split ? x2[m,k] : 0)) * W[n,k] + b[n] ) + r1[m,n] + r2[m,n]
// with v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: exact fp32 products, the same numerics
// class as the reference's fp32 GEMM).
#include "gd4d_common.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct LinearParams {
  const float* x;      // (M, K), row stride ldx
  const float* x2;     // optional addend on the input rows (query_pos), same layout as x
  const float* w;      // (N, K) row-major
  const float* bias;   // (N) or null
  const float* r1;     // optional residual (M, N), row stride ldr1
  const float* r2;     // optional second residual, row stride ldr2
  float* y;            // (M, N), row stride ldy
  float* xsum;         // optional (M, K) contiguous: receives x + x2 (what a training step keeps for the weight gradient)
  int M, K, N, n_split, flags;      // flags: bit0 ReLU on the output, bit1 inverse_sigmoid on the input,
                                    // bit3 the weight is given TRANSPOSED, (K, N) row-major (input gradient of a Linear)
  int ldx, ldy, ldr1, ldr2;
  // grouped mode (gd4d_linear_group_fwd): up to 4 (W, bias, y, N) sets sharing the input; blockIdx.y walks the
  // groups' column tiles back to back.  groups == 0: the plain single-output form above.
  int groups;
  const float* gw[4];
  const float* gb[4];
  float* gy[4];
  int gn[4];
};

constexpr int LN_TM = 32, LN_TN = 32, LN_WAVES = 4, LN_KC = 64;   // k per wave-chunk

typedef __attribute__((ext_vector_type(4))) float f32x4;

// One workgroup = one 32 x 32 output tile; its 4 waves split K in 64-wide chunks (wave w takes
// chunks w, w+4, ..) and the partial tiles are summed through LDS once.
//
// MFMA operands go straight from global memory into registers with v_mfma_f32_16x16x4_f32:
// lane (i = l&15, g = l>>4) supplies A[i][k] / B[k][i] for ONE k per instruction, and WHICH k the
// four lane groups pair up is free as long as A and B agree.  So group g owns k = 16j + 4g + e:
// it loads one float4 (e = 0..3) per 16-k step j, and the four groups of a row read 64 contiguous
// bytes - 16 cache lines per load instruction instead of 64 with a row-per-lane layout (the L1 tag
// rate, not the MFMA, bounds these tiny GEMMs).
// RES: the launch has residual operands.  Without them the sixteen prefetch registers are not allocated: 143 -> under 128
// registers, four workgroups per CU instead of three - a 900 x 1024 output (928 workgroups) then runs in one round.
// WAVES: how many waves split K (64 columns of it at a time each).  4 for K <= 512 (one round trip per wave up to K = 256);
// LN_DEEP_WAVES for the deep contractions of the FFN (K = 1024: four dependent round trips per wave with 4 waves).
// MODE: 0 = every form (bounds checks, scalar tails, inverse_sigmoid, an addend for some of the column tiles only: loads
// behind selects and branches, which the compiler issues and waits for one at a time); LN_FAST [| LN_FAST_WKN] [| LN_FAST_X2] =
// the shapes the decoder actually runs (K % 64 == 0, 16-byte rows, the addend - if any - for every column): a chunk's
// sixteen to twenty-four loads are issued back to back and pinned before the first use.
constexpr int LN_FAST = 1, LN_FAST_WKN = 2, LN_FAST_X2 = 4;
template <bool RES, int WAVES, int MODE>
__device__ __forceinline__ void linear_body(const LinearParams& p, float (*s_part)[16][64]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * LN_TM;
  int n0 = blockIdx.y * LN_TN;
  const float* W = p.w;
  const float* Bv = p.bias;
  float* Y = p.y;
  int N = p.N, ldy = p.ldy;
  if (p.groups > 0) {                      // block-uniform walk over the groups (static indices: no kernarg copy)
    int t = blockIdx.y;
    W = p.gw[0]; Bv = p.gb[0]; Y = p.gy[0]; N = p.gn[0];
#pragma unroll
    for (int gi = 1; gi < 4; ++gi) {
      const int tiles = (N + LN_TN - 1) / LN_TN;
      if (gi < p.groups && t >= tiles) { t -= tiles; W = p.gw[gi]; Bv = p.gb[gi]; Y = p.gy[gi]; N = p.gn[gi]; }
      else break;
    }
    n0 = t * LN_TN;
    ldy = N;
  }
  const bool add2 = p.x2 != nullptr && n0 < p.n_split;     // block-uniform (n_split % 32 == 0)
  const bool in_isig = (p.flags & 2) != 0;
  const bool vec = (p.K % 4 == 0) && (p.ldx % 4 == 0);
  const bool store_sum = p.xsum != nullptr && blockIdx.y == 0;    // block-uniform; the entry points require vec and x2 for it

  // rows of the two 16-row A blocks and the two 16-row W blocks this lane feeds (clamped rows are never stored)
  const float* xr[2];
  const float* x2r[2];
  const float* wr[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int gm = min(m0 + 16 * t + i16, p.M - 1);
    const int gn = min(n0 + 16 * t + i16, N - 1);
    xr[t] = p.x + (size_t)gm * p.ldx;
    x2r[t] = add2 ? p.x2 + (size_t)gm * p.ldx : nullptr;
    wr[t] = (p.flags & 8) ? W + gn : W + (size_t)gn * p.K;
  }
  const bool wkn = (p.flags & 8) != 0;             // B[k][n] = W[k * N + n]: four row-strided dwords per lane and step

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load4 = [&](const float* row, const float* row2, int k, bool is_x) {
    float v[4];
    if (vec && k + 4 <= p.K) {
      const float4 t = *reinterpret_cast<const float4*>(row + k);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      if (row2) {
        const float4 u = *reinterpret_cast<const float4*>(row2 + k);
        v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (k + e < p.K) ? row[k + e] + (row2 ? row2[k + e] : 0.f) : 0.f;
    }
    if (is_x && in_isig) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (k + e < p.K) ? inv_sigmoid(v[e]) : 0.f;
    }
    return f32x4{v[0], v[1], v[2], v[3]};
  };

  // Epilogue operands (bias, residuals) are requested NOW by the wave that will need them, so their
  // memory latency overlaps the K loop instead of following the reduction.
  float pre_bias[2] = {0.f, 0.f};
  constexpr int RA = RES ? 2 : 1;
  float pre_r1[RA][2][4], pre_r2[RA][2][4];           // the two residuals apart: each a batch of unconditional loads (clamped
#pragma unroll                                        // addresses) behind ONE uniform branch - summed as they were requested,
  for (int a = 0; a < RA; ++a)                        // `t += r1[..]; t += r2[..]` per element, every load waited for the one before
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { pre_r1[a][c][r] = 0.f; pre_r2[a][c][r] = 0.f; }
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = min(n0 + 16 * c + i16, N - 1);
      if (Bv) pre_bias[c] = Bv[n];
    }
    if (RES && p.r1) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pre_r1[a][c][r] = p.r1[(size_t)min(m0 + 16 * a + 4 * g + r, p.M - 1) * p.ldr1 + min(n0 + 16 * c + i16, N - 1)];
    }
    if (RES && p.r2) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pre_r2[a][c][r] = p.r2[(size_t)min(m0 + 16 * a + 4 * g + r, p.M - 1) * p.ldr2 + min(n0 + 16 * c + i16, N - 1)];
    }
  }

  if (MODE & LN_FAST) {
    for (int kc = wave * LN_KC; kc < p.K; kc += LN_KC * WAVES) {
      f32x4 av[4][2], bv[4][2], xv[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int k = kc + 16 * j + 4 * g;
          av[j][t] = *reinterpret_cast<const f32x4*>(xr[t] + k);
          if (MODE & LN_FAST_X2) xv[j][t] = *reinterpret_cast<const f32x4*>(p.x2 + (xr[t] - p.x) + k);
          if (MODE & LN_FAST_WKN) {
            bv[j][t] = f32x4{wr[t][(size_t)k * N], wr[t][(size_t)(k + 1) * N], wr[t][(size_t)(k + 2) * N], wr[t][(size_t)(k + 3) * N]};
          } else {
            bv[j][t] = *reinterpret_cast<const f32x4*>(wr[t] + k);
          }
        }
      asm volatile("" : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[1][0]), "+v"(av[1][1]), "+v"(av[2][0]), "+v"(av[2][1]),
                        "+v"(av[3][0]), "+v"(av[3][1]), "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[1][0]), "+v"(bv[1][1]),
                        "+v"(bv[2][0]), "+v"(bv[2][1]), "+v"(bv[3][0]), "+v"(bv[3][1]));
      if (MODE & LN_FAST_X2) {
        asm volatile("" : "+v"(xv[0][0]), "+v"(xv[0][1]), "+v"(xv[1][0]), "+v"(xv[1][1]), "+v"(xv[2][0]), "+v"(xv[2][1]),
                          "+v"(xv[3][0]), "+v"(xv[3][1]));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            av[j][t] += xv[j][t];
            if (store_sum && m0 + 16 * t + i16 < p.M)
              *reinterpret_cast<f32x4*>(p.xsum + (size_t)(m0 + 16 * t + i16) * p.K + kc + 16 * j + 4 * g) = av[j][t];
          }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][a][e], bv[j][c][e], acc[a][c], 0, 0, 0);
    }
  } else {
  for (int kc = wave * LN_KC; kc < p.K; kc += LN_KC * WAVES) {
    f32x4 av[4][2], bv[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // all 16 (+8) loads of the chunk in flight together
      const int k = kc + 16 * j + 4 * g;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        av[j][t] = load4(xr[t], x2r[t], k, true);
        if (store_sum && m0 + 16 * t + i16 < p.M && k + 4 <= p.K)   // (the first column tile's workgroup sees every (row, k) once)
          *reinterpret_cast<f32x4*>(p.xsum + (size_t)(m0 + 16 * t + i16) * p.K + k) = av[j][t];
        if (!wkn) {
          bv[j][t] = load4(wr[t], nullptr, k, false);
        } else {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = wr[t][(size_t)min(k + e, p.K - 1) * N] * (k + e < p.K ? 1.f : 0.f);
          bv[j][t] = f32x4{v[0], v[1], v[2], v[3]};
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][a][e], bv[j][c][e], acc[a][c], 0, 0, 0);
  }

  }

  // The epilogue operands requested before the K loop have arrived by now; say so HERE.  Left to the first use - inside the
  // epilogue's per-element branches, after earlier stores - the counter can only be waited down to zero, which also waits
  // for the previous store's acknowledgement: sixteen stores, one after the other.
  asm volatile("" : "+v"(pre_bias[0]), "+v"(pre_bias[1]));
  float pre_res[RA][2][4];
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (RES) {
        asm volatile("" : "+v"(pre_r1[a][c][0]), "+v"(pre_r1[a][c][1]), "+v"(pre_r1[a][c][2]), "+v"(pre_r1[a][c][3]),
                          "+v"(pre_r2[a][c][0]), "+v"(pre_r2[a][c][1]), "+v"(pre_r2[a][c][2]), "+v"(pre_r2[a][c][3]));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) pre_res[a][c][r] = pre_r1[a][c][r] + pre_r2[a][c][r];   // (0 + r1) + r2, as before
    }
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_part[wave - 1][(a * 2 + c) * 4 + r][lane] = acc[a][c][r];
  }
  __syncthreads();
  if (wave == 0) {
    // C/D of 16x16x4: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int n = n0 + 16 * c + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[a][c][r];
#pragma unroll
          for (int w = 0; w < WAVES - 1; ++w) v += s_part[w][(a * 2 + c) * 4 + r][lane];
          const int m = m0 + 16 * a + 4 * g + r;
          if (m < p.M && n < N) {
            v += pre_bias[c];
            if (p.flags & 1) v = fmaxf(v, 0.f);
            if (RES) v += pre_res[a][c][r];
            Y[(size_t)m * ldy + n] = v;
          }
        }
      }
  }
}

__global__ __launch_bounds__(64 * LN_WAVES) void linear_kernel_res(const LinearParams p) {
  __shared__ float s_part[LN_WAVES - 1][16][64];
  linear_body<true, LN_WAVES, 0>(p, s_part);
}

__global__ __launch_bounds__(64 * LN_WAVES) void linear_kernel(const LinearParams p) {
  __shared__ float s_part[LN_WAVES - 1][16][64];
  linear_body<false, LN_WAVES, 0>(p, s_part);
}

#ifndef LN_DEEP_WAVES_N
#define LN_DEEP_WAVES_N 8
#endif
constexpr int LN_DEEP_WAVES = LN_DEEP_WAVES_N, LN_DEEP_K = 768, LN_FAST_DEEP_WAVES = 8;

template <bool RES>
__global__ __launch_bounds__(64 * LN_DEEP_WAVES) void linear_kernel_deep(const LinearParams p) {
  __shared__ float s_part[LN_DEEP_WAVES - 1][16][64];
  linear_body<RES, LN_DEEP_WAVES, 0>(p, s_part);
}

template <bool RES, int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) void linear_kernel_fast(const LinearParams p) {
  __shared__ float s_part[WAVES - 1][16][64];
  linear_body<RES, WAVES, MODE>(p, s_part);
}

template <bool RES, int WAVES>
static void launch_linear_fast(const LinearParams& p, dim3 grid, hipStream_t st, int mode) {
  switch (mode) {
    case LN_FAST: hipLaunchKernelGGL((linear_kernel_fast<RES, WAVES, LN_FAST>), grid, dim3(64 * WAVES), 0, st, p); break;
    case LN_FAST | LN_FAST_WKN: hipLaunchKernelGGL((linear_kernel_fast<RES, WAVES, LN_FAST | LN_FAST_WKN>), grid, dim3(64 * WAVES), 0, st, p); break;
    case LN_FAST | LN_FAST_X2: hipLaunchKernelGGL((linear_kernel_fast<RES, WAVES, LN_FAST | LN_FAST_X2>), grid, dim3(64 * WAVES), 0, st, p); break;
    default: hipLaunchKernelGGL((linear_kernel_fast<RES, WAVES, LN_FAST | LN_FAST_WKN | LN_FAST_X2>), grid, dim3(64 * WAVES), 0, st, p); break;
  }
}

static void launch_linear(const LinearParams& p, dim3 grid, hipStream_t st) {
  const bool res = p.r1 || p.r2;
  const bool wkn = (p.flags & 8) != 0;
  // the branch-free form: whole 64-column chunks of K, 16-byte rows (and for the transposed weight: N known per launch),
  // no inverse_sigmoid on the way in, the addend - if any - for every column tile
  const bool x2_all = p.x2 != nullptr && (p.groups > 0 || p.n_split >= p.N);
  const bool fast = p.K % LN_KC == 0 && p.ldx % 4 == 0 && !(p.flags & 2) && (p.x2 == nullptr || x2_all) && !(wkn && p.groups > 0) &&
                    reinterpret_cast<uintptr_t>(p.groups > 0 ? (const void*)p.gw[0] : (const void*)p.w) % 16 == 0;
  if (fast) {
    const int mode = LN_FAST | (wkn ? LN_FAST_WKN : 0) | (x2_all ? LN_FAST_X2 : 0);
    if (p.K >= LN_DEEP_K) {                  // (8 waves: with 16 the 128-register budget of a 1024-thread workgroup spills)
      if (res) launch_linear_fast<true, LN_FAST_DEEP_WAVES>(p, grid, st, mode);
      else launch_linear_fast<false, LN_FAST_DEEP_WAVES>(p, grid, st, mode);
    } else {
      if (res) launch_linear_fast<true, LN_WAVES>(p, grid, st, mode);
      else launch_linear_fast<false, LN_WAVES>(p, grid, st, mode);
    }
    return;
  }
  if (p.K >= LN_DEEP_K) {
    if (res) hipLaunchKernelGGL(linear_kernel_deep<true>, grid, dim3(64 * LN_DEEP_WAVES), 0, st, p);
    else hipLaunchKernelGGL(linear_kernel_deep<false>, grid, dim3(64 * LN_DEEP_WAVES), 0, st, p);
  } else if (res) {
    hipLaunchKernelGGL(linear_kernel_res, grid, dim3(64 * LN_WAVES), 0, st, p);
  } else {
    hipLaunchKernelGGL(linear_kernel, grid, dim3(64 * LN_WAVES), 0, st, p);
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension (C <= 1024, C % 4 == 0): one wave per row.
//   y = LN(x + res) * gamma + beta, optional ReLU.  Statistics like ATen: biased variance, eps
//   inside the sqrt, two-pass (mean, then centred sum of squares) in fp32.
struct LayerNormParams {
  const float* x;
  const float* res;     // optional, added before normalisation
  const float* gamma;
  const float* beta;
  float* y;
  int M, C, relu;
  float eps;
  // optional prologue (gd4d_small_linear_layernorm_fwd): x = lin_in W^T + b with lin_k <= 4 inputs per row
  const float* lin_in;  // (M, lin_k)
  const float* lin_w;   // (C, lin_k)
  const float* lin_b;   // (C) or null
  int lin_k, lin_isig;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void layernorm_kernel(const LayerNormParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const float* x = p.x ? p.x + (size_t)row * p.C : nullptr;
  const float* r = p.res ? p.res + (size_t)row * p.C : nullptr;
  float4 v[4];                                   // up to 1024 channels: 4 float4 per lane
  const int nv = p.C / 4;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv) {
      if (p.lin_in) {                              // tiny Linear instead of a load: 4 channels x lin_k inputs
        float in[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < p.lin_k; ++k) {
          const float t = p.lin_in[(size_t)row * p.lin_k + k];
          in[k] = p.lin_isig ? inv_sigmoid(t) : t;
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wrow = p.lin_w + (size_t)(4 * c + e) * p.lin_k;
          float a = 0.f;
          for (int k = 0; k < p.lin_k; ++k) a = fmaf(in[k], wrow[k], a);
          o[e] = a + (p.lin_b ? p.lin_b[4 * c + e] : 0.f);
        }
        v[i] = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        v[i] = reinterpret_cast<const float4*>(x)[c];
      }
      if (r) {
        const float4 t = reinterpret_cast<const float4*>(r)[c];
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.C + p.eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4 g = reinterpret_cast<const float4*>(p.gamma)[c];
      const float4 b = reinterpret_cast<const float4*>(p.beta)[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      reinterpret_cast<float4*>(p.y + (size_t)row * p.C)[c] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Reference-point refinement of Detr3DTransformerDecoder.forward (detr3d_transformer.py:201-214):
//   new_xy = sigmoid(tmp[..., 0:2] + inverse_sigmoid(ref_xy)); new_z = sigmoid(tmp[..., 4] + inverse_sigmoid(ref_z))
// (about a dozen elementwise launches in the reference) as one kernel, one thread per query.
__global__ __launch_bounds__(256) void refine_reference_kernel(const float* __restrict__ tmp,
                                                               const float* __restrict__ ref,
                                                               float* __restrict__ out, int M, int ldt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float* t = tmp + (size_t)i * ldt;
  const float* r = ref + (size_t)i * 3;
  const float a = t[0] + inv_sigmoid(r[0]);
  const float b = t[1] + inv_sigmoid(r[1]);
  const float c = t[4] + inv_sigmoid(r[2]);
  out[(size_t)i * 3 + 0] = 1.0f / (1.0f + expf(-a));
  out[(size_t)i * 3 + 1] = 1.0f / (1.0f + expf(-b));
  out[(size_t)i * 3 + 2] = 1.0f / (1.0f + expf(-c));
}

}  // namespace gd4d

extern "C" int gd4d_linear_fwd(const float* x, const float* x2, const float* w, const float* bias,
                               const float* r1, const float* r2, float* y, int M, int K, int N,
                               int n_split, int flags, int ldx, int ldy, int ldr1, int ldr2,
                               float* xsum, void* stream) {
  using namespace gd4d;
  if (!x || !w || !y) return GD4D_EINVAL;
  if (xsum && (!x2 || n_split < LN_TN)) return GD4D_EINVAL;
  if (xsum && ((K % 4) || (ldx % 4) || (flags & 2))) return GD4D_EUNSUPPORTED;
  if (xsum && !aligned16(xsum)) return GD4D_EALIGN;
  if (M <= 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) return GD4D_EINVAL;
  if (r1 && ldr1 < N) return GD4D_EINVAL;
  if (r2 && ldr2 < N) return GD4D_EINVAL;
  if (x2 && (n_split % LN_TN) != 0 && n_split < N) return GD4D_EUNSUPPORTED;   // split must align to tiles
  if (!aligned16(x) || !aligned16(w) || (x2 && !aligned16(x2))) return GD4D_EALIGN;
  LinearParams p{};
  p.x = x; p.x2 = x2; p.w = w; p.bias = bias; p.r1 = r1; p.r2 = r2; p.y = y; p.xsum = xsum;
  p.M = M; p.K = K; p.N = N; p.n_split = x2 ? n_split : 0; p.flags = flags;
  p.ldx = ldx; p.ldy = ldy; p.ldr1 = ldr1; p.ldr2 = ldr2;
  const dim3 grid((M + LN_TM - 1) / LN_TM, (N + LN_TN - 1) / LN_TN);
  launch_linear(p, grid, static_cast<hipStream_t>(stream));
  return check_launch();
}

extern "C" int gd4d_linear_group_fwd(const float* x, const float* x2, const float* const* w, const float* const* bias,
                                     float* const* y, const int32_t* n_out, int G, int M, int K, int ldx,
                                     float* xsum, void* stream) {
  using namespace gd4d;
  if (!x || !w || !y || !n_out || G <= 0 || M <= 0 || K <= 0 || ldx < K) return GD4D_EINVAL;
  if (xsum && !x2) return GD4D_EINVAL;
  if (xsum && ((K % 4) || (ldx % 4))) return GD4D_EUNSUPPORTED;
  if (xsum && !aligned16(xsum)) return GD4D_EALIGN;
  if (G > 4) return GD4D_EUNSUPPORTED;
  if (!aligned16(x) || (x2 && !aligned16(x2))) return GD4D_EALIGN;
  LinearParams p{};
  p.x = x; p.x2 = x2; p.M = M; p.K = K; p.ldx = ldx; p.groups = G; p.xsum = xsum;
  p.n_split = 1 << 30;                    // the addend applies to every output column
  int tiles = 0;
  for (int g = 0; g < G; ++g) {
    if (!w[g] || !y[g] || n_out[g] <= 0) return GD4D_EINVAL;
    if (!aligned16(w[g])) return GD4D_EALIGN;
    p.gw[g] = w[g]; p.gb[g] = bias ? bias[g] : nullptr; p.gy[g] = y[g]; p.gn[g] = n_out[g];
    tiles += (n_out[g] + LN_TN - 1) / LN_TN;
  }
  const dim3 grid((M + LN_TM - 1) / LN_TM, tiles);
  launch_linear(p, grid, static_cast<hipStream_t>(stream));
  return check_launch();
}

extern "C" int gd4d_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                                  float* y, int M, int C, float eps, int relu, void* stream) {
  using namespace gd4d;
  if (!x || !gamma || !beta || !y || M <= 0 || C <= 0) return GD4D_EINVAL;
  if (C % 4 != 0 || C > 1024) return GD4D_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta) || (res && !aligned16(res)))
    return GD4D_EALIGN;
  LayerNormParams p{};
  p.x = x; p.res = res; p.gamma = gamma; p.beta = beta; p.y = y; p.M = M; p.C = C; p.relu = relu; p.eps = eps;
  hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_small_linear_layernorm_fwd(const float* in, const float* w, const float* bias, const float* gamma,
                                               const float* beta, float* y, int M, int Kin, int C, float eps,
                                               int flags, void* stream) {
  using namespace gd4d;
  if (!in || !w || !gamma || !beta || !y || M <= 0 || C <= 0 || Kin <= 0) return GD4D_EINVAL;
  if (Kin > 4 || C % 4 != 0 || C > 1024) return GD4D_EUNSUPPORTED;
  if (!aligned16(y) || !aligned16(gamma) || !aligned16(beta)) return GD4D_EALIGN;
  LayerNormParams p{};
  p.gamma = gamma; p.beta = beta; p.y = y; p.M = M; p.C = C; p.relu = flags & GD4D_LIN_RELU; p.eps = eps;
  p.lin_in = in; p.lin_w = w; p.lin_b = bias; p.lin_k = Kin; p.lin_isig = (flags & GD4D_LIN_INV_SIGMOID_IN) ? 1 : 0;
  hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

namespace gd4d {
__global__ __launch_bounds__(256) void inverse_sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = inv_sigmoid(x[i]);
}
// what autograd derives for the reference's inverse_sigmoid (clamp to [0, 1], clamp(x, eps), clamp(1 - x, eps), log of the ratio):
// d/dx = [x > eps] / x + [1 - x > eps] / (1 - x) inside [0, 1], 0 outside
__global__ __launch_bounds__(256) void inverse_sigmoid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                  const float* __restrict__ add, float* __restrict__ gx, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i], eps = 1e-5f;
  float d = 0.f;
  if (v >= 0.f && v <= 1.f) d = (v > eps ? 1.0f / v : 0.f) + (1.0f - v > eps ? 1.0f / (1.0f - v) : 0.f);
  gx[i] = gy[i] * d + (add ? add[i] : 0.f);
}
}  // namespace gd4d

extern "C" int gd4d_inverse_sigmoid_bwd(const float* x, const float* grad_y, const float* add, float* grad_x, int64_t n, void* stream) {
  using namespace gd4d;
  if (!x || !grad_y || !grad_x || n <= 0) return GD4D_EINVAL;
  hipLaunchKernelGGL(inverse_sigmoid_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     grad_y, add, grad_x, (long long)n);
  return check_launch();
}

extern "C" int gd4d_inverse_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream) {
  using namespace gd4d;
  if (!x || !y || n <= 0) return GD4D_EINVAL;
  hipLaunchKernelGGL(inverse_sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y,
                     (long long)n);
  return check_launch();
}

extern "C" int gd4d_refine_reference_fwd(const float* tmp, const float* ref, float* out, int M, int ldt,
                                         void* stream) {
  using namespace gd4d;
  if (!tmp || !ref || !out || M <= 0) return GD4D_EINVAL;
  if (ldt < 5) return GD4D_EUNSUPPORTED;
  hipLaunchKernelGGL(refine_reference_kernel, dim3((M + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), tmp, ref, out, M, ldt);
  return check_launch();
}

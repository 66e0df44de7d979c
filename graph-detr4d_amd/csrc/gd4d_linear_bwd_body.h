// The weight-gradient tile of the decoder's small dense layers as a device function: gd4d_linear_bwd.hip launches it as kernels of
// its own, gd4d_cross_attn_sliced_bwd.hip as guest workgroups of a backward gather-dot (gd4d_cross_attn_dot_sliced_wgrad).
#pragma once
#include "gd4d_common.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct LinBwdParams {
  const float* x;    // (M, K), row stride ldx
  const float* dy;   // (M, N), row stride ldy
  float* dw;         // (N, K) contiguous
  float* db;         // (N) or null
  int M, N, K, ldx, ldy;
  int accumulate;    // add to dw / db instead of overwriting them
};

constexpr int LB_TN = 16, LB_TK = 32, LB_WAVES = 16, LB_UNROLL = 4, LB_T = LB_TK / 16;
// LDS of one workgroup of WAVES waves: the waves' partial tiles and bias sums
template <int WAVES> struct LinBwdShared { float part[WAVES][LB_TN * LB_TK]; float bpart[WAVES][LB_TN]; };

// WAVES = 16: the kernels of gd4d_linear_bwd.hip; 8: the same tile as a guest workgroup of a backward gather-dot's launch
// (gd4d_cross_attn_dot_sliced_wgrad; the waves split the rows, so the two sum in different orders - each deterministic)
template <int WAVES>
__device__ __forceinline__ void linear_bwd_weight_tile(const LinBwdParams& p, int k0, int n0, bool first_k_tile, LinBwdShared<WAVES>& sh) {
  float (&part)[WAVES][LB_TN * LB_TK] = sh.part;
  float (&bpart)[WAVES][LB_TN] = sh.bpart;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, r4 = lane >> 4;

  // rows of this wave: its share of M rounded up to whole MFMA steps of 4 rows
  const int per = ((p.M + WAVES - 1) / WAVES + 3) & ~3;
  const int m_begin = wave * per, m_end = min(m_begin + per, p.M);

  const int n = n0 + c16;
  const bool n_ok = n < p.N;
  const float* dy_col = p.dy + (n_ok ? n : p.N - 1);
  const float* x_col[LB_T];
  bool k_ok[LB_T];
#pragma unroll
  for (int t = 0; t < LB_T; ++t) {
    const int k = k0 + 16 * t + c16;
    k_ok[t] = k < p.K;
    x_col[t] = p.x + (k_ok[t] ? k : p.K - 1);
  }

  f32x4 acc[LB_T];
#pragma unroll
  for (int t = 0; t < LB_T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  for (int m0 = m_begin; m0 < m_end; m0 += 4 * LB_UNROLL) {
    float a[LB_UNROLL], b[LB_UNROLL][LB_T];
    static_assert(LB_UNROLL == 4 && LB_T == 2, "the operand list of the asm below");
#pragma unroll
    for (int u = 0; u < LB_UNROLL; ++u) {            // all loads of the unrolled steps first (clamped: always in bounds)
      const int m = m0 + 4 * u + r4;
      const size_t mr = (size_t)min(m, m_end - 1);
      a[u] = dy_col[mr * p.ldy];
#pragma unroll
      for (int t = 0; t < LB_T; ++t) b[u][t] = x_col[t][mr * p.ldx];
    }
    // ... and all of them complete before anything is masked: with the selects next to the loads the compiler sank every load
    // into its select's branch and waited for each where it stood - twelve round trips in turn per step of sixteen rows
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]),
                      "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]));
#pragma unroll
    for (int u = 0; u < LB_UNROLL; ++u) {
      const bool m_ok = m0 + 4 * u + r4 < m_end;
      a[u] = (m_ok && n_ok) ? a[u] : 0.f;
#pragma unroll
      for (int t = 0; t < LB_T; ++t) b[u][t] = (m_ok && k_ok[t]) ? b[u][t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < LB_UNROLL; ++u) {
      bsum += a[u];
#pragma unroll
      for (int t = 0; t < LB_T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][t], acc[t], 0, 0, 0);
    }
  }

  // C/D of 16x16x4: column j = lane & 15 (k), rows i = 4 * (lane >> 4) + r (n)
#pragma unroll
  for (int t = 0; t < LB_T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][(4 * r4 + r) * LB_TK + 16 * t + c16] = acc[t][r];
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  if (lane < 16) bpart[wave][lane] = bsum;
  __syncthreads();
  for (int i = threadIdx.x; i < LB_TN * LB_TK; i += 64 * WAVES) {
    const int nn = n0 + i / LB_TK, kk = k0 + i % LB_TK;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += part[w][i];
    if (nn < p.N && kk < p.K) {
      float* d = p.dw + (size_t)nn * p.K + kk;
      *d = p.accumulate ? *d + s : s;
    }
  }
  if (p.db && first_k_tile && threadIdx.x < LB_TN && n0 + threadIdx.x < p.N) {
    const int i = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += bpart[w][i];
    p.db[n0 + i] = p.accumulate ? p.db[n0 + i] + s : s;
  }
}


// Several independent weight gradients in ONE launch (a decoder layer's backward produces eleven of these 13-us launches,
// each filling half the device for a chain of four dependent load rounds; nothing reads a weight gradient before the
// optimizer, so the training step queues them and issues them sixteen at a time): workgroup b works on tile b - tile0[i]
// of problem i.
constexpr int LB_GROUP = 16;

struct LinBwdGroup {
  LinBwdParams p[LB_GROUP];
  int tile0[LB_GROUP + 1];      // first workgroup of problem i
  int tiles_k[LB_GROUP];        // k tiles of problem i
  int count;
};

// workgroup `block` of a group launch: tile block - tile0[i] of problem i.  g points INTO THE KERNEL ARGUMENT SEGMENT (scalar
// loads): a reference to the by-value kernel parameter made the compiler copy the whole descriptor to scratch (1 KB per lane,
// the launch 40 % slower).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) LinBwdGroup* lin_group_ptr_t;
#else
typedef const LinBwdGroup* lin_group_ptr_t;
#endif
template <int WAVES>
__device__ __forceinline__ void linear_bwd_weight_group_tile(const lin_group_ptr_t g, const int block, LinBwdShared<WAVES>& sh) {
  int i = 0;
#pragma unroll
  for (int j = 1; j < LB_GROUP; ++j)
    if (j < g->count && block >= g->tile0[j]) i = j;
  // (indexed with a runtime index the descriptor would go through scratch: pick the problem with selects)
  LinBwdParams p = g->p[0];
  int tk = g->tiles_k[0], t0 = g->tile0[0];
#pragma unroll
  for (int j = 1; j < LB_GROUP; ++j)
    if (j == i) { p = g->p[j]; tk = g->tiles_k[j]; t0 = g->tile0[j]; }
  const int t = block - t0;
  const int kt = t % tk, nt = t / tk;
  linear_bwd_weight_tile<WAVES>(p, kt * LB_TK, nt * LB_TN, kt == 0, sh);
}

// host: the group descriptor from the C ABI's arrays (gd4d_linear_bwd_weight_group); returns the tile count in `tiles`
static inline int fill_lin_bwd_group(LinBwdGroup& g, int& tiles, const void* const* x, const void* const* grad_y, void* const* grad_w,
                                     void* const* grad_b, const int32_t* dims, int count, int accumulate) {
  if (!x || !grad_y || !grad_w || !grad_b || !dims || count <= 0) return GD4D_EINVAL;
  if (count > LB_GROUP) return GD4D_EUNSUPPORTED;
  tiles = 0;
  for (int i = 0; i < count; ++i) {
    const int M = dims[5 * i], K = dims[5 * i + 1], N = dims[5 * i + 2], ldx = dims[5 * i + 3], ldy = dims[5 * i + 4];
    if (!x[i] || !grad_y[i] || !grad_w[i] || M <= 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) return GD4D_EINVAL;
    g.p[i] = LinBwdParams{static_cast<const float*>(x[i]), static_cast<const float*>(grad_y[i]), static_cast<float*>(grad_w[i]),
                          static_cast<float*>(grad_b[i]), M, N, K, ldx, ldy, accumulate ? 1 : 0};
    g.tile0[i] = tiles;
    g.tiles_k[i] = (K + LB_TK - 1) / LB_TK;
    tiles += g.tiles_k[i] * ((N + LB_TN - 1) / LB_TN);
  }
  for (int i = count; i <= LB_GROUP; ++i) g.tile0[i] = tiles;
  for (int i = count; i < LB_GROUP; ++i) { g.p[i] = g.p[0]; g.tiles_k[i] = 1; }
  g.count = count;
  return GD4D_OK;
}

}  // namespace gd4d

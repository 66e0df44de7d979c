// gd4d_gemm_bf16x3_fwd: C = act(A W^T + b), fp32 in / fp32 out, on the bf16 matrix cores with split operands
// (a = a_hi + a_lo, w = w_hi + w_lo in bf16; a_hi w_hi + a_hi w_lo + a_lo w_hi accumulated in fp32: about 2^-16 relative
// per product, the same fp32-class scheme as gd4d_value_proj_fwd).  Row-major A (M, K), W (N, K) pre-split once by
// gd4d_split_bf16_fwd (weights are static), row-major C (M, N).
//
// Used for the dense part of the head's feature position embedding (1x1 convolutions 192 -> 1024 -> 256 and the SE
// layer's 256 -> 256 over 739 800 pixels: 0.87 TFLOP per sample), where a library fp32 GEMM runs on the fp32 MFMA
// (157 TFLOP/s peak) and this runs on the bf16 MFMA (2.5 PFLOP/s peak, three products per output).
//
// Workgroup = 256 x 256 output tile, 16 waves in a 4 x 4 arrangement, each wave 2 x 2 tiles of v_mfma_f32_32x32x16_bf16
// (with 128 x 128 tiles the kernel was bound by L2 -> CU traffic: 12 GB per GEMM at 9 TB/s; 256 x 256 halves it).
// K advances in steps of 32 through a double-buffered LDS stage holding the tile's A_hi / A_lo / W_hi / W_lo as
// [k-group of 8][row][8 x bf16] so that every MFMA fragment is one conflict-free ds_read_b128 per lane.  Global loads of
// step k+2 are issued into registers as soon as step k+1 has been converted / parked into the other buffer (before the
// barrier), so they are in flight across the barrier and the MFMAs of step k+1: one barrier per step.  128 KB of LDS, one workgroup of 16 waves per CU.
#include "gd4d_common.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int GM_BM = 256, GM_BN = 256, GM_BK = 32, GM_THREADS = 1024;   // 16 waves, 4 x 4, each 64 x 64
constexpr int GM_ARR = 4 * GM_BM * 16;               // bytes of one [4 k-groups][256 rows][16 B] array: 16 KB
constexpr int GM_STAGE = 4 * GM_ARR;                 // A_hi, A_lo, W_hi, W_lo

struct GemmParams {
  const float* a;
  const uint16_t* w_hi;
  const uint16_t* w_lo;
  const float* bias;
  float* c;
  int M, N, K, lda, ldc, relu, relu_in, mask_c;
};

__device__ __forceinline__ unsigned gm_cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}

// 8 consecutive floats -> 16 bytes of bf16 "hi" halves and 16 bytes of bf16 "lo" (residual) halves
__device__ __forceinline__ void gm_split8(const float4& p, const float4& q, u32x4& h, u32x4& l) {
  const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = gm_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    const float ra = v[2 * i] - __uint_as_float(hh[i] << 16);            // exact: hi is a rounding of the value
    const float rb = v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u);
    ll[i] = gm_cvt_pk_bf16(ra, rb);
  }
  h = u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}

__global__ __launch_bounds__(GM_THREADS) void gemm_bf16x3_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l32 = lane & 31, kg = lane >> 5;
  const int n0 = blockIdx.x * GM_BN;                   // n fastest: the workgroups sharing an A tile run together
  const int m0 = blockIdx.y * GM_BM;

  // staging role: 16-byte chunk (row = tid / 4, k-group = tid % 4) of the 256 x 32 tiles
  const int srow = tid >> 2, sk = tid & 3;
  constexpr int PASSES = GM_BM * 4 / GM_THREADS;      // 16-byte chunks per thread and operand
  constexpr int PROWS = GM_THREADS / 4;
  const float* a_src[PASSES];
  const uint16_t* wh_src[PASSES];
  const uint16_t* wl_src[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int r = PROWS * ps + srow;
    a_src[ps] = p.a + (size_t)min(m0 + r, p.M - 1) * p.lda + 8 * sk;
    wh_src[ps] = p.w_hi + (size_t)(n0 + r) * p.K + 8 * sk;
    wl_src[ps] = p.w_lo + (size_t)(n0 + r) * p.K + 8 * sk;
  }
  float4 ra[PASSES][2];
  u32x4 rwh[PASSES], rwl[PASSES];
  auto issue = [&](int k0) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      ra[ps][0] = *reinterpret_cast<const float4*>(a_src[ps] + k0);
      ra[ps][1] = *reinterpret_cast<const float4*>(a_src[ps] + k0 + 4);
      rwh[ps] = *reinterpret_cast<const u32x4*>(wh_src[ps] + k0);
      rwl[ps] = *reinterpret_cast<const u32x4*>(wl_src[ps] + k0);
    }
  };
  auto park = [&](int stage) {
    char* base = smem + stage * GM_STAGE;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int off = (sk * GM_BM + PROWS * ps + srow) * 16;
      u32x4 h, l;
      if (p.relu_in) {                                  // activation of the previous layer, applied on the way in
        float4& x = ra[ps][0];
        float4& y = ra[ps][1];
        x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
        y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
      }
      gm_split8(ra[ps][0], ra[ps][1], h, l);
      *reinterpret_cast<u32x4*>(base + off) = h;
      *reinterpret_cast<u32x4*>(base + GM_ARR + off) = l;
      *reinterpret_cast<u32x4*>(base + 2 * GM_ARR + off) = rwh[ps];
      *reinterpret_cast<u32x4*>(base + 3 * GM_ARR + off) = rwl[ps];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int steps = p.K / GM_BK;
  issue(0);
  park(0);
  if (steps > 1) issue(GM_BK);                          // step 1's operands: in flight across the barrier and step 0's MFMAs
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int cur = s & 1;
    const char* base = smem + cur * GM_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int aoff = ((2 * ks + kg) * GM_BM + 64 * wm + 32 * i + l32) * 16;
        const int boff = ((2 * ks + kg) * GM_BM + 64 * wn + 32 * i + l32) * 16;
        ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + aoff));
        al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + GM_ARR + aoff));
        bh[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + 2 * GM_ARR + boff));
        bl[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + 3 * GM_ARR + boff));
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    if (s + 1 < steps) {
      park(cur ^ 1);                                     // the other buffer: its readers finished before the last barrier
      if (s + 2 < steps) issue((s + 2) * GM_BK);         // registers are free again: next loads go out before the barrier
    }
    __syncthreads();
  }

  // C/D of 32x32x16: column n = lane & 31, rows 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + 64 * wn + 32 * ni + l32;
    float bv = p.bias ? p.bias[n] : 0.f;
    // (the bias has arrived before the first store is issued: said once here - first used inside the per-row branches below,
    // every use after a store could only wait the counter down to zero, i.e. for that store's acknowledgement: 64 in turn)
    asm volatile("" : "+v"(bv));
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 64 * wm + 32 * mi + 4 * kg + (r & 3) + 8 * (r >> 2);
        if (m < p.M) {
          float v = acc[mi][ni][r] + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          // backward through a ReLU: c holds the forward's activation and is replaced by the gradient where it was > 0
          if (p.mask_c && !(p.c[(size_t)m * p.ldc + n] > 0.f)) v = 0.f;
          p.c[(size_t)m * p.ldc + n] = v;
        }
      }
  }
}

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = w[i];
  const uint16_t h = f32_to_bf16(v);
  hi[i] = h;
  lo[i] = f32_to_bf16(v - bf16_to_f32(h));
}

// ---- gd4d_gemm_tn_bf16x3: C (M, N) = sum over rows r of A[r, :M]^T B[r, :N] ------------------------------------------
// The weight gradient of a Linear / 1x1 convolution over R rows (A = gradient of the output, B = the layer's input), with
// the column sums of A (the bias gradient) on the side.  Same split-bf16 x 3 arithmetic as the forward.
//
// Both operands are read straight from HBM in the MFMA's own fragment layout - no LDS, no barrier: the contraction index
// (rows) is the slow axis of both matrices, so a lane that owns output row i of an MFMA tile needs 8 consecutive ROWS of
// one column.  A 16-byte load of columns 4 l32 .. 4 l32 + 3 serves FOUR tiles at once when tile `mi` is defined as the
// columns {4 i + mi}: the tiles interleave in memory, every load is a dwordx4 (dwordx2 for B's two tiles), and the
// interleave is undone for free when the result is stored.  Wave = 128 (M) x 64 (N): 4 x 2 tiles of 32x32x16, 24 MFMAs
// per 16 rows against 16 loads; workgroup = 2 x 2 waves = 256 x 128.  Rows are cut into `splits` ranges (grid.y), each
// writing its partial product; gemm_tn_reduce_kernel adds the partials in split order (fixed summation order, no atomics).
struct GemmTnParams {
  const float* a;
  const float* b;
  float* part;      // (splits, M, N)
  float* colpart;   // (splits, M) or null
  long long R;
  int M, N, lda, ldb, rows_per_split, relu_b;
};

__device__ __forceinline__ void tn_split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = gm_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    const float ra = v[2 * i] - __uint_as_float(hh[i] << 16);
    const float rb = v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u);
    ll[i] = gm_cvt_pk_bf16(ra, rb);
  }
  h = __builtin_bit_cast(bf16x8, u32x4{hh[0], hh[1], hh[2], hh[3]});
  l = __builtin_bit_cast(bf16x8, u32x4{ll[0], ll[1], ll[2], ll[3]});
}

__global__ __launch_bounds__(256) void gemm_tn_bf16x3_kernel(const GemmTnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, kg = lane >> 5;
  const int tiles_n = (p.N + 127) / 128;
  const int m_base = ((int)blockIdx.x / tiles_n) * 256 + 128 * wm;
  const int n_base = ((int)blockIdx.x % tiles_n) * 128 + 64 * wn;
  if (m_base >= p.M || n_base >= p.N) return;          // (no barrier anywhere in this kernel)
  const int split = blockIdx.y;
  const long long r_begin = (long long)split * p.rows_per_split;
  const long long r_end = min(p.R, r_begin + p.rows_per_split);

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};

  const float* ap = p.a + (size_t)(r_begin + 8 * kg) * p.lda + m_base + 4 * l32;
  const float* bp = p.b + (size_t)(r_begin + 8 * kg) * p.ldb + n_base + 2 * l32;
  float4 ra[8];
  float2 rb[8];
  auto issue = [&](long long r0) {                      // rows r0 + 8 kg + j; past the end: row R - 1, zeroed when used
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long back = max(0ll, r0 + 8 * kg + j - (p.R - 1));
      ra[j] = *reinterpret_cast<const float4*>(ap + (ptrdiff_t)(r0 - r_begin + j - back) * p.lda);
      rb[j] = *reinterpret_cast<const float2*>(bp + (ptrdiff_t)(r0 - r_begin + j - back) * p.ldb);
    }
  };
  issue(r_begin);
  for (long long r0 = r_begin; r0 < r_end; r0 += 16) {
    bf16x8 ah[4], al[4], bh[2], bl[2];
    {
      float va[4][8], vb[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool live = r0 + 8 * kg + j < r_end;
        va[0][j] = live ? ra[j].x : 0.f; va[1][j] = live ? ra[j].y : 0.f;
        va[2][j] = live ? ra[j].z : 0.f; va[3][j] = live ? ra[j].w : 0.f;
        float bx = rb[j].x, by = rb[j].y;
        if (p.relu_b) { bx = fmaxf(bx, 0.f); by = fmaxf(by, 0.f); }
        vb[0][j] = bx; vb[1][j] = by;
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[mi] += va[mi][j];
        tn_split8(va[mi], ah[mi], al[mi]);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) tn_split8(vb[ni], bh[ni], bl[ni]);
    }
    if (r0 + 16 < r_end) issue(r0 + 16);                // the raw registers are free: next rows in flight under the MFMAs
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
      }
  }

  // tile (mi, ni), element (i, jn) is C[m_base + 4 i + mi][n_base + 2 jn + ni]; a lane holds jn = l32 and
  // i = 8 (r >> 2) + 4 kg + (r & 3): its two ni values are neighbours in memory
  float* out = p.part + (size_t)split * p.M * p.N;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r >> 2) + 4 * kg + (r & 3);
      *reinterpret_cast<float2*>(out + (size_t)(m_base + 4 * i + mi) * p.N + n_base + 2 * l32) =
          make_float2(acc[mi][0][r], acc[mi][1][r]);
    }
  if (p.colpart && n_base == 0) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) csum[mi] += __shfl_xor(csum[mi], 32);
    if (kg == 0)
      *reinterpret_cast<float4*>(p.colpart + (size_t)split * p.M + m_base + 4 * l32) =
          make_float4(csum[0], csum[1], csum[2], csum[3]);
  }
}

// Sum of the partial products in split order.  Four elements per thread (float4), the splits cut into four consecutive
// groups (threadIdx.y) whose sums meet in LDS and are added in group order: a fixed order whatever the launch.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ part, const float* __restrict__ colpart,
                                                             float* __restrict__ c, float* __restrict__ colsum, int mn, int m,
                                                             int splits) {
  __shared__ float4 sums[4][64];
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = 4 * (blockIdx.x * 64 + tx);
  const int per = (splits + 3) / 4;
  const int k0 = grp * per, k1 = min(splits, k0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < mn)
    for (int k = k0; k < k1; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * mn + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  sums[grp][tx] = s;
  __syncthreads();
  if (grp == 0 && i < mn) {
#pragma unroll
    for (int g = 1; g < 4; ++g) { s.x += sums[g][tx].x; s.y += sums[g][tx].y; s.z += sums[g][tx].z; s.w += sums[g][tx].w; }
    *reinterpret_cast<float4*>(c + i) = s;
  }
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (colsum && j < m) {
    float t = 0.f;
    for (int k = 0; k < splits; ++k) t += colpart[(size_t)k * m + j];
    colsum[j] = t;
  }
}

}  // namespace gd4d

extern "C" int gd4d_split_bf16_fwd(const float* w, uint16_t* hi, uint16_t* lo, size_t n, void* stream) {
  using namespace gd4d;
  if (!w || !hi || !lo || n == 0) return GD4D_EINVAL;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w, hi, lo, n);
  return check_launch();
}

extern "C" int gd4d_gemm_bf16x3_fwd(const float* a, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias, float* c,
                                    int M, int N, int K, int lda, int ldc, int flags, void* stream) {
  using namespace gd4d;
  if (!a || !w_hi || !w_lo || !c || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N) return GD4D_EINVAL;
  if (N % GM_BN != 0 || K % GM_BK != 0 || lda % 4 != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(a) || !aligned16(w_hi) || !aligned16(w_lo)) return GD4D_EALIGN;
  GemmParams p{a, w_hi, w_lo, bias, c, M, N, K, lda, ldc, (flags & GD4D_LIN_RELU) ? 1 : 0,
               (flags & GD4D_GEMM_RELU_IN) ? 1 : 0, (flags & GD4D_GEMM_MASK_C) ? 1 : 0};
  if (!allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16x3_kernel), 2 * GM_STAGE)) return GD4D_ELAUNCH;
  const dim3 grid(N / GM_BN, (M + GM_BM - 1) / GM_BM);
  hipLaunchKernelGGL(gemm_bf16x3_kernel, grid, dim3(GM_THREADS), 2 * GM_STAGE, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

namespace gd4d {
static int tn_splits(long long R, int M, int N) {
  const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
  long long want = (1024 + tiles - 1) / tiles;                      // about four workgroups per CU
  const long long most = (R + 255) / 256;                           // at least 256 rows per split
  if (want > most) want = most;
  return (int)(want < 1 ? 1 : want);
}
}  // namespace gd4d

extern "C" size_t gd4d_gemm_tn_bf16x3_workspace_bytes(long long R, int M, int N) {
  if (R <= 0 || M <= 0 || N <= 0) return 0;
  return (size_t)gd4d::tn_splits(R, M, N) * ((size_t)M * N + M) * sizeof(float);
}

extern "C" int gd4d_gemm_tn_bf16x3(const float* a, const float* b, float* c, float* colsum, void* workspace, long long R, int M,
                                   int N, int lda, int ldb, int flags, void* stream) {
  using namespace gd4d;
  if (!a || !b || !c || !workspace || R <= 0 || M <= 0 || N <= 0 || lda < M || ldb < N) return GD4D_EINVAL;
  if (M % 128 != 0 || N % 64 != 0 || lda % 4 != 0 || ldb % 2 != 0 || (size_t)M * N > (size_t)1 << 30) return GD4D_EUNSUPPORTED;
  if (!aligned16(a) || ((uintptr_t)b & 7) || !aligned16(workspace) || !aligned16(c)) return GD4D_EALIGN;
  const int splits = tn_splits(R, M, N);
  long long rows = (R + splits - 1) / splits;
  rows = (rows + 15) / 16 * 16;
  float* part = static_cast<float*>(workspace);
  float* colpart = part + (size_t)splits * M * N;
  GemmTnParams p{a, b, part, colsum ? colpart : nullptr, R, M, N, lda, ldb, (int)rows, (flags & GD4D_GEMM_RELU_IN) ? 1 : 0};
  const int tiles = ((M + 255) / 256) * ((N + 127) / 128);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gemm_tn_bf16x3_kernel, dim3(tiles, splits), dim3(256), 0, st, p);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((M * N / 4 + 63) / 64), dim3(256), 0, st, part, colpart, c, colsum, M * N, M,
                     splits);
  return check_launch();
}

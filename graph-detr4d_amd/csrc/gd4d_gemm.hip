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
  int M, N, K, lda, ldc, relu, relu_in;
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
               (flags & GD4D_GEMM_RELU_IN) ? 1 : 0};
  if (!allow_dynamic_lds(reinterpret_cast<const void*>(gemm_bf16x3_kernel), 2 * GM_STAGE)) return GD4D_ELAUNCH;
  const dim3 grid(N / GM_BN, (M + GM_BM - 1) / GM_BM);
  hipLaunchKernelGGL(gemm_bf16x3_kernel, grid, dim3(GM_THREADS), 2 * GM_STAGE, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

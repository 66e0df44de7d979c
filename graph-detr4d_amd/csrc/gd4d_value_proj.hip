// gd4d_value_proj_fwd: value_proj over the multi-camera FPN pyramid for gfx950.
//
// Reference: Deform3DCrossAttn.forward, deform3d_cross_attn.py:264-280 - every level
// (B*N, C, H, W) is flattened + transposed to channels-last, the levels are concatenated and a
// Linear(256 -> 256) is applied (a 739 800 x 256 x 256 GEMM at N = 24, the dominant dense
// contraction of a decoder layer), then viewed (B*N, S, Hh, Dh).
//
// This kernel reads the NCHW maps exactly as the caller holds them and writes the channels-last
// head-major value tensor in ONE pass (the reference makes a transposed copy, a concatenated copy
// and the GEMM output): HBM traffic = read pyramid once + write value once.
//
//   out[r, start_l + pix, co] = sum_ci in_l[r, ci, pix] * W[co, ci] + bias[co]
//
// Arithmetic: split-bf16 ("bf16x3") on the MFMA pipe.  x = hi + lo with hi = bf16(x),
// lo = bf16(x - hi); a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulate.  Dropped terms are
// <= ~2^-17 relative per product, i.e. fp32-class results (measured <= 2e-5 abs against an fp32
// GEMM on N(0,1) features) at 3/16 of the cost of the f32-input MFMA.  The f32-input MFMA would make
// this layer MFMA-bound at ~0.65 ms (157 TF peak); bf16x3 puts it under the HBM time (~0.28 ms).
//
// Structure (per CU: one persistent 512-thread workgroup = 8 waves, 2 per SIMD):
//   * wave w owns output channels [32w, 32w+32) and keeps its W_hi / W_lo MFMA B-fragments for
//     the whole K = 256 in registers (2 x 64 VGPRs), loaded once per launch;
//   * per tile of BM pixels: all 512 threads load the fp32 (ci, pix) block with pixel-contiguous
//     (coalesced) dword loads, split to hi/lo bf16 with v_cvt_pk_bf16_f32 and park it in LDS as
//     [pix][ci] rows (the transpose happens in registers: a lane gathers 8 ci of ONE pixel and
//     issues one ds_write_b128), 16-byte chunks XOR-swizzled by (pix & 15) so that both the
//     writes and the ds_read_b128 A-fragment reads are bank-conflict free;
//   * each wave then runs K/16 x 3 MFMA 32x32x16 per 32-pixel sub-tile and stores its 32 channels
//     (128 B per pixel row, one full L2 line) straight from the accumulators;
//   * next tile's global loads are issued before the MFMA phase (register prefetch) and LDS is
//     double buffered, so HBM reads overlap the matrix work.
#include <stdlib.h>

#include "gd4d_common.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int VP_C = 256;          // in = out channels (embed_dims of every reference config)
constexpr int VP_THREADS = 512;
constexpr int VP_KSTEPS = VP_C / 16;

struct ValueProjParams {
  const void* in[GD4D_MAX_LEVELS];   // level l: (R, C, HW_l)
  int hw[GD4D_MAX_LEVELS];
  int start[GD4D_MAX_LEVELS];        // pixel offset of level l inside a row of `out`
  int tiles[GD4D_MAX_LEVELS];        // tiles per camera-row at level l
  int tile_base[GD4D_MAX_LEVELS + 1];  // prefix over levels of R * tiles[l]
  const float* weight[GD4D_MAX_LAYERS];   // per decoder layer: (C, C) row-major [co][ci]
  const float* bias[GD4D_MAX_LAYERS];     // (C) or null
  void* out[GD4D_MAX_LAYERS];             // (R, S, C)
  int R, L, S, NL;
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}

// split two floats into packed (hi, hi) and (lo, lo) bf16 pairs
struct HiLo { unsigned hi, lo; };
__device__ __forceinline__ HiLo split2(float a, float b) {
  HiLo r;
  r.hi = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(r.hi << 16);          // exact: hi is a rounding of a
  const float rb = b - __uint_as_float(r.hi & 0xffff0000u);
  r.lo = cvt_pk_bf16(ra, rb);
  return r;
}

// 8 consecutive floats -> one 16-byte chunk of hi halves and one of lo halves
__device__ __forceinline__ void split8(const float* v, u32x4& h, u32x4& l) {
  const HiLo a = split2(v[0], v[1]), b = split2(v[2], v[3]), c = split2(v[4], v[5]), d = split2(v[6], v[7]);
  h = u32x4{a.hi, b.hi, c.hi, d.hi};
  l = u32x4{a.lo, b.lo, c.lo, d.lo};
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
  return __builtin_bit_cast(bf16x8, v);
}

// byte offset of 16-byte chunk `chunk` (8 bf16) of row `row` in a [rows][256] bf16 LDS image
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * (VP_C * 2) + ((chunk ^ (row & 15)) << 4);
}

template <int BM, bool OUT_BF16>
__global__ __launch_bounds__(VP_THREADS, 2) void value_proj_kernel(const ValueProjParams p) {
  constexpr int PIX_GROUPS = VP_THREADS / BM;          // threads sharing one pixel column
  constexpr int CPT = VP_C / PIX_GROUPS;               // input channels per thread per tile
  constexpr int SUB = BM / 32;                         // 32-pixel MFMA sub-tiles per tile
  constexpr int IMG = BM * VP_C * 2;                   // bytes of one bf16 [BM][256] image
  static_assert(CPT % 8 == 0, "a thread packs whole 16-byte chunks");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][hi, lo][BM][256] bf16

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31;           // MFMA column (output channel within the wave's block) / A row
  const int kg = lane >> 5;            // which 8 of the 16 k of a step this lane holds

  // Workgroup -> (layer, slot).  Consecutive workgroups serve different layers of the SAME tile
  // sequence, so the NL layer groups sweep the pyramid in step and all but the first reader of a
  // tile are served from L2 / Infinity Cache instead of HBM.
  const int layer = blockIdx.x % p.NL;
  const int slot = blockIdx.x / p.NL;
  const int slots = gridDim.x / p.NL;
  const float* __restrict__ weight = p.weight[layer];
  void* __restrict__ outp = p.out[layer];

  // ---- W fragments for this wave: co = 32*wave + col, k = 16*s + 8*kg .. +8 ----
  bf16x8 whi[VP_KSTEPS], wlo[VP_KSTEPS];
  {
    const float* wrow = weight + (size_t)(32 * wave + col) * VP_C + 8 * kg;
#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(wrow + 16 * s);
      const float4 b = *reinterpret_cast<const float4*>(wrow + 16 * s + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      u32x4 h, l;
      split8(v, h, l);
      whi[s] = as_bf16x8(h);
      wlo[s] = as_bf16x8(l);
    }
  }
  const float bias = p.bias[layer] ? p.bias[layer][32 * wave + col] : 0.f;

  // staging role of this thread: pixel column `spix` of the tile, channels [sc0, sc0 + CPT)
  const int spix = tid % BM;
  const int sc0 = (tid / BM) * CPT;
  float stage[CPT];

  const int total = p.tile_base[p.L];
  auto decode = [&](int t, int& lvl, int& row, int& pix0) {
    lvl = 0;
#pragma unroll
    for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
      if (l < p.L && t >= p.tile_base[l]) lvl = l;
    const int rel = t - p.tile_base[lvl];
    row = rel / p.tiles[lvl];
    pix0 = (rel - row * p.tiles[lvl]) * BM;
  };
  auto issue_loads = [&](int t) {
    int lvl, row, pix0;
    decode(t, lvl, row, pix0);
    const int hw = p.hw[lvl];
    const int pix = pix0 + spix;
    const float* src = static_cast<const float*>(p.in[lvl]) + ((size_t)row * VP_C + sc0) * hw + pix;
    if (pix < hw) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) stage[k] = src[(size_t)k * hw];
    } else {
#pragma unroll
      for (int k = 0; k < CPT; ++k) stage[k] = 0.f;
    }
  };
  auto park = [&](int buf) {
    char* hi_img = smem + buf * 2 * IMG;
    char* lo_img = hi_img + IMG;
#pragma unroll
    for (int c = 0; c < CPT / 8; ++c) {
      u32x4 h, l;
      split8(stage + 8 * c, h, l);
      const int off = lds_off(spix, sc0 / 8 + c);
      *reinterpret_cast<u32x4*>(hi_img + off) = h;
      *reinterpret_cast<u32x4*>(lo_img + off) = l;
    }
  };

  int t = slot;
  if (t >= total) return;
  issue_loads(t);
  park(0);
  __syncthreads();

  int buf = 0;
  for (; t < total; t += slots) {
    const int tn = t + slots;
    const bool has_next = tn < total;
    if (has_next) issue_loads(tn);            // register prefetch: in flight during the MFMA phase

    int lvl, row, pix0;
    decode(t, lvl, row, pix0);
    const char* hi_img = smem + buf * 2 * IMG;
    const char* lo_img = hi_img + IMG;

    f32x16 acc[SUB];
#pragma unroll
    for (int m = 0; m < SUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = bias;

#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
#pragma unroll
      for (int m = 0; m < SUB; ++m) {
        const int off = lds_off(32 * m + col, 2 * s + kg);
        const bf16x8 ahi = as_bf16x8(*reinterpret_cast<const u32x4*>(hi_img + off));
        const bf16x8 alo = as_bf16x8(*reinterpret_cast<const u32x4*>(lo_img + off));
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, whi[s], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wlo[s], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, whi[s], acc[m], 0, 0, 0);
      }
    }

    // ---- epilogue: C/D layout col = lane&31 (co), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pix) ----
    {
      const int hw = p.hw[lvl];
      const size_t obase = ((size_t)row * p.S + p.start[lvl] + pix0 + 4 * kg) * VP_C + 32 * wave + col;
      const bool full = pix0 + BM <= hw;                     // workgroup-uniform
#pragma unroll
      for (int m = 0; m < SUB; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dp = 32 * m + (r & 3) + 8 * (r >> 2);    // pixel offset inside the tile (minus 4*kg)
          if (full || pix0 + dp + 4 * kg < hw) {
            if (OUT_BF16)
              static_cast<uint16_t*>(outp)[obase + (size_t)dp * VP_C] = f32_to_bf16(acc[m][r]);
            else
              static_cast<float*>(outp)[obase + (size_t)dp * VP_C] = acc[m][r];
          }
        }
      }
    }

    if (has_next) park(buf ^ 1);              // other buffer: nobody reads it during this iteration
    __syncthreads();
    buf ^= 1;
  }
}

}  // namespace gd4d

extern "C" size_t gd4d_value_proj_workspace_bytes(void) { return 0; }

namespace gd4d {

static int vp_tile_pixels() {
  static int bm = 0;
  if (!bm) {
    const char* e = getenv("GD4D_VP_BM");            // dev A/B switch
    bm = (e && atoi(e) == 64) ? 64 : 32;
  }
  return bm;
}

template <int BM>
static int vp_launch(ValueProjParams& p, const int32_t* level_hw, int R, int L, int NL, int out_dtype,
                     hipStream_t st) {
  int s = 0, base = 0;
  for (int l = 0; l < L; ++l) {
    const int hw = level_hw[2 * l] * level_hw[2 * l + 1];
    p.hw[l] = hw;
    p.start[l] = s;
    p.tiles[l] = (hw + BM - 1) / BM;
    p.tile_base[l] = base;
    s += hw;
    base += R * p.tiles[l];
  }
  for (int l = L; l <= GD4D_MAX_LEVELS; ++l) p.tile_base[l] = base;
  p.S = s;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  int slots = cus / NL;                                // persistent: <= one workgroup per CU
  if (slots < 1) slots = 1;
  if (slots > base) slots = base;
  const int grid = slots * NL;
  const size_t lds = 2 * 2 * (size_t)BM * VP_C * 2;    // 2 buffers x (hi, lo) x [BM][256] bf16
  if (out_dtype == GD4D_BF16) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(value_proj_kernel<BM, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((value_proj_kernel<BM, true>), dim3(grid), dim3(VP_THREADS), lds, st, p);
  } else {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(value_proj_kernel<BM, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL((value_proj_kernel<BM, false>), dim3(grid), dim3(VP_THREADS), lds, st, p);
  }
  return check_launch();
}

}  // namespace gd4d

extern "C" int gd4d_value_proj_multi_fwd(const void* const* feats, const int32_t* level_hw,
                                         const float* const* weights, const float* const* biases,
                                         void* const* outs, int R, int C, int L, int NL, int in_dtype,
                                         int out_dtype, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !weights || !outs) return GD4D_EINVAL;
  if (R <= 0 || C <= 0 || L <= 0 || NL <= 0) return GD4D_EINVAL;
  if (C != VP_C || L > GD4D_MAX_LEVELS || NL > GD4D_MAX_LAYERS || in_dtype != GD4D_F32) return GD4D_EUNSUPPORTED;
  if (out_dtype != GD4D_F32 && out_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  ValueProjParams p{};
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    p.in[l] = feats[l];
  }
  for (int i = 0; i < NL; ++i) {
    if (!weights[i] || !outs[i]) return GD4D_EINVAL;
    if (!aligned16(weights[i])) return GD4D_EALIGN;
    p.weight[i] = weights[i];
    p.bias[i] = biases ? biases[i] : nullptr;
    p.out[i] = outs[i];
  }
  p.R = R; p.L = L; p.NL = NL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return vp_tile_pixels() == 64 ? vp_launch<64>(p, level_hw, R, L, NL, out_dtype, st)
                                : vp_launch<32>(p, level_hw, R, L, NL, out_dtype, st);
}

extern "C" int gd4d_value_proj_fwd(const void* const* feats, const int32_t* level_hw, const float* weight,
                                   const float* bias, void* out, int R, int C, int L, int in_dtype,
                                   int out_dtype, void* stream) {
  if (!weight || !out) return GD4D_EINVAL;
  const float* ws[1] = {weight};
  const float* bs[1] = {bias};
  void* os[1] = {out};
  return gd4d_value_proj_multi_fwd(feats, level_hw, ws, bs, os, R, C, L, 1, in_dtype, out_dtype, stream);
}

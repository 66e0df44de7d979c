// gd4d_value_proj (A-stationary form): value_proj over the multi-camera FPN pyramid for NL decoder layers in one launch.
//
// Reference: Deform3DCrossAttn.forward, deform3d_cross_attn.py:264-280 (flatten + transpose + cat + Linear 256 -> 256 +
// view (B*N, S, Hh, Dh)), once per decoder layer over the SAME pyramid (detr3d_transformer.py:192-198).
//
//   out_l[r, start + pix, co] = sum_ci in[r, ci, pix] * W_l[co, ci] + bias_l[co]        l = 0 .. NL-1
//
// Why this shape.  The first kernels of this file's family kept W in registers and streamed pixel tiles through LDS
// (DMA -> convert -> fragment reads): ~450 issued instructions per 48 MFMAs, phases that added instead of overlapping
// (profiles/r01f_value_proj_ablation.md).  Here the roles are swapped:
//   * a wave owns a tile of 32 pixels for a whole PASS: it loads the tile's 256 input channels straight from the NCHW
//     maps into registers (pixel-contiguous dword loads: a lane needs 8 channels of ONE pixel per MFMA k-step), splits
//     them once into bf16 hi / lo A-fragments (128 VGPRs) - no LDS staging, no transpose, no conversion in the loop;
//   * the weights of all NL layers are pre-split into bf16 hi / lo MFMA B-fragments in a workspace image (1.5 MB for 6
//     layers, L2-resident) and streamed through a 2-slot LDS ring in CHUNKS of (layer, 32 output channels): 32 KB of
//     fragments + the 32 bias values, fetched by LDS-DMA one chunk ahead;
//   * per chunk a wave runs 16 k-steps x 3 MFMA 32x32x16 (a_hi*w_hi, a_lo*w_hi, a_hi*w_lo: fp32-class split-bf16
//     arithmetic, fp32 accumulate) with 2 ds_read_b128 per k-step, computing the TRANSPOSED product (channels x pixels)
//     so that a lane holds 4 x 4 consecutive channels of its pixel and the 32 x 32 result leaves in four 16-byte stores
//     per lane (16 dword stores per lane made the kernel store-ISSUE bound: ablation in profiles/r02_value_proj.md).
// A pass is 8*NL chunks: the tile's load + split is amortised over 384*NL MFMAs per wave, the steady state issues ~3
// instructions per MFMA, and the pyramid is read from HBM once for all NL layers.  Traffic per pixel: 1 KB in, NL KB
// out, 1.5 MB of W per (waves per workgroup x 32) pixels from L2 into LDS.
//
// Synchronisation: one raw s_barrier per chunk (all waves of a workgroup consume the same chunk stream), preceded by
// "s_waitcnt vmcnt(0)": the wave's DMA pieces of this chunk were the LAST memory operations of the previous phase (its
// stores come first), issued half a phase ago.
#include <stdlib.h>

#include <type_traits>

#include "gd4d_common.h"
#include "gd4d_value_proj_body.h"

namespace gd4d {

// ---------------------------------------------------------------------------------------------------------------
// Weight image: chunk (layer, cb) = [s = 0..15][part = hi, lo][lane = 0..63][8 bf16], so that a 16-byte LDS-DMA per
// lane lands every fragment where lane `lane` reads it with one ds_read_b128 (conflict-free); behind the chunks the
// biases of all layers as one fp32 table (copied into LDS once per workgroup).
// Fragment of lane l at k-step s: W[co = 32*cb + (l & 31)][ci = 16*s + 8*(l >> 5) .. +8].
struct VpaWeights { const float* w[GD4D_MAX_LAYERS]; const float* b[GD4D_MAX_LAYERS]; };

template <bool SINGLE>
__global__ __launch_bounds__(256) void value_proj_wimg_kernel(const VpaWeights ptrs, char* wimg) {
  constexpr int PARTS = SINGLE ? 1 : 2;
  constexpr int CHUNK = va_chunk_bytes(SINGLE);
  const int chunk = blockIdx.x;                               // layer * 8 + cb
  const int layer = chunk >> 3, cb = chunk & 7;
  const float* w = ptrs.w[0];
  const float* bias = ptrs.b[0];
#pragma unroll
  for (int l = 1; l < GD4D_MAX_LAYERS; ++l)
    if (l == layer) { w = ptrs.w[l]; bias = ptrs.b[l]; }
  char* dst = wimg + (size_t)chunk * CHUNK;
  for (int e = threadIdx.x; e < VA_KSTEPS * 64; e += blockDim.x) {
    const int s = e >> 6, lane = e & 63;
    const float* src = w + (size_t)(32 * cb + (lane & 31)) * VA_C + 16 * s + 8 * (lane >> 5);
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32x4 h, l;
    va_split8(v, h, l);
    *reinterpret_cast<u32x4*>(dst + (s * PARTS) * 1024 + lane * 16) = h;
    if (!SINGLE) *reinterpret_cast<u32x4*>(dst + (s * PARTS + 1) * 1024 + lane * 16) = l;
  }
  if (threadIdx.x < 32) {
    float* table = reinterpret_cast<float*>(wimg + (size_t)gridDim.x * CHUNK);
    table[layer * VA_C + 32 * cb + threadIdx.x] = bias ? bias[32 * cb + threadIdx.x] : 0.f;
  }
}

template <int WAVES, bool OUT_BF16, bool HEAD_MAJOR, bool SINGLE, int DBG = 0, bool IN_CHLAST = false>
__global__ __launch_bounds__(64 * WAVES, 2) void value_proj_astat_kernel(const VpaParams p) {
  extern __shared__ __attribute__((aligned(16))) char va_smem[];
  value_proj_astat_body<WAVES, OUT_BF16, HEAD_MAJOR, SINGLE, DBG, IN_CHLAST>(p, (int)blockIdx.x, (int)gridDim.x, va_smem);
}

static int va_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return cus;
}

}  // namespace gd4d

extern "C" size_t gd4d_value_proj_workspace_bytes(int NL) {
  if (NL <= 0 || NL > GD4D_MAX_LAYERS) return 0;
  return gd4d::va_image_bytes(NL, false) + 4096;                // the weight image (the larger, split form) + dev trace
}

extern "C" size_t gd4d_value_proj_image_bytes(void) { return gd4d::va_image_bytes(1, false); }

extern "C" int gd4d_value_proj_image(const float* weight, const float* bias, void* image, void* stream) {
  using namespace gd4d;
  if (!weight || !image) return GD4D_EINVAL;
  if (!aligned16(weight) || !aligned16(image)) return GD4D_EALIGN;
  VpaWeights pack{};
  pack.w[0] = weight; pack.b[0] = bias;
  hipLaunchKernelGGL(value_proj_wimg_kernel<false>, dim3(8), dim3(256), 0, static_cast<hipStream_t>(stream), pack, static_cast<char*>(image));
  return check_launch();
}

extern "C" int gd4d_value_proj_guest_fwd(const gd4d_chain_guest* job, int max_cus, void* stream) {
  using namespace gd4d;
  VpaParams p{};
  if (int rc = va_guest_params(job, p)) return rc;
  int cus = va_cus();
  if (max_cus >= 8 && max_cus < cus) cus = max_cus - max_cus % 8;
  constexpr int waves = 8;
  int grid = job->workgroups > 0 ? job->workgroups : cus;
  const int need = (p.total + waves - 1) / waves;
  if (grid > need) grid = need;
  const size_t lds = va_lds_bytes(1, false, waves);
  auto go = [&](auto kern) -> int {
    if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), lds, static_cast<hipStream_t>(stream), p);
    return check_launch();
  };
  return p.in_chlast ? go(value_proj_astat_kernel<waves, false, false, false, 0, true>) : go(value_proj_astat_kernel<waves, false, false, false>);
}

extern "C" int gd4d_value_proj_multi_fwd(const void* const* feats, const int32_t* level_hw,
                                         const float* const* weights, const float* const* biases,
                                         void* const* outs, int R, int C, int L, int NL, int Hh, int in_dtype,
                                         int out_dtype, int out_layout, int precision, void* workspace,
                                         size_t workspace_bytes, int max_cus, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !weights || !outs || !workspace) return GD4D_EINVAL;
  if (R <= 0 || C <= 0 || L <= 0 || NL <= 0) return GD4D_EINVAL;
  if (C != VA_C || L > GD4D_MAX_LEVELS || NL > GD4D_MAX_LAYERS || in_dtype != GD4D_F32) return GD4D_EUNSUPPORTED;
  if (out_dtype != GD4D_F32 && out_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (out_layout != GD4D_LAYOUT_PIXEL_MAJOR && out_layout != GD4D_LAYOUT_HEAD_MAJOR) return GD4D_EUNSUPPORTED;
  if (Hh <= 0 || C % Hh != 0) return GD4D_EINVAL;
  if (precision != GD4D_VP_PRECISION_F32 && precision != GD4D_VP_PRECISION_BF16) return GD4D_EUNSUPPORTED;
  if (precision == GD4D_VP_PRECISION_BF16 && out_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (workspace_bytes < gd4d_value_proj_workspace_bytes(NL)) return GD4D_EINVAL;
  if (!aligned16(workspace)) return GD4D_EALIGN;
  const bool single = precision == GD4D_VP_PRECISION_BF16;
  VpaParams p{};
  int s = 0, base = 0;
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    const int hw = level_hw[2 * l] * level_hw[2 * l + 1];
    p.in[l] = feats[l]; p.hw[l] = hw; p.start[l] = s; p.tiles[l] = (hw + VA_TILE - 1) / VA_TILE; p.tile_base[l] = base;
    s += hw;
    base += R * p.tiles[l];
  }
  for (int l = L; l <= GD4D_MAX_LEVELS; ++l) p.tile_base[l] = base;
  for (int l = L; l < GD4D_MAX_LEVELS; ++l) { p.tiles[l] = 1; p.hw[l] = 1; }
  p.S = s; p.R = R; p.L = L; p.NL = NL; p.Hh = Hh; p.Dh = C / Hh; p.total = base;
  for (int i = 0; i < NL; ++i) {
    if (!weights[i] || !outs[i]) return GD4D_EINVAL;
    if (!aligned16(weights[i])) return GD4D_EALIGN;
    p.out[i] = outs[i];
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* wimg = static_cast<char*>(workspace);
  p.wimg = wimg;
  VpaWeights pack{};                                           // by value: kernel arguments are capturable, a copy is not
  for (int i = 0; i < NL; ++i) { pack.w[i] = weights[i]; pack.b[i] = biases ? biases[i] : nullptr; }
  if (single) hipLaunchKernelGGL(value_proj_wimg_kernel<true>, dim3(NL * 8), dim3(256), 0, st, pack, wimg);
  else hipLaunchKernelGGL(value_proj_wimg_kernel<false>, dim3(NL * 8), dim3(256), 0, st, pack, wimg);
  if (int rc = check_launch()) return rc;

  int cus = va_cus();
  if (max_cus >= 8 && max_cus < cus) cus = max_cus - max_cus % 8;   // leave CUs to kernels of another stream
  constexpr int waves = 8;                                     // one persistent 8-wave workgroup per CU
  int grid = cus;
  const int need = (base + waves - 1) / waves;
  if (grid > need) grid = need;
  const size_t lds = VA_RING * (size_t)va_chunk_bytes(single) + (size_t)NL * VA_C * 4 + 8 * VA_TILE * 144;   // ring, bias, patches
  const bool ob = out_dtype == GD4D_BF16, hm = out_layout == GD4D_LAYOUT_HEAD_MAJOR;
  auto go = [&](auto kern) {
    (void)allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), lds, st, p);
  };
#define GD4D_VA_DISPATCH(W)                                                                                         \
  if (single) { if (hm) go(value_proj_astat_kernel<W, true, true, true>); else go(value_proj_astat_kernel<W, true, false, true>); } \
  else if (ob) { if (hm) go(value_proj_astat_kernel<W, true, true, false>); else go(value_proj_astat_kernel<W, true, false, false>); } \
  else { if (hm) go(value_proj_astat_kernel<W, false, true, false>); else go(value_proj_astat_kernel<W, false, false, false>); }
  GD4D_VA_DISPATCH(8)
#undef GD4D_VA_DISPATCH
  return check_launch();
}

extern "C" int gd4d_value_proj_fwd(const void* const* feats, const int32_t* level_hw, const float* weight,
                                   const float* bias, void* out, int R, int C, int L, int Hh, int in_dtype,
                                   int out_dtype, int out_layout, int precision, void* workspace,
                                   size_t workspace_bytes, int max_cus, void* stream) {
  if (!weight || !out) return GD4D_EINVAL;
  const float* ws[1] = {weight};
  const float* bs[1] = {bias};
  void* os[1] = {out};
  return gd4d_value_proj_multi_fwd(feats, level_hw, ws, bs, os, R, C, L, 1, Hh, in_dtype, out_dtype, out_layout,
                                   precision, workspace, workspace_bytes, max_cus, stream);
}

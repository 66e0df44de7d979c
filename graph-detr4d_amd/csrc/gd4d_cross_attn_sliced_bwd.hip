// Training backward of the channel-sliced aggregate-then-project path (gfx950): the pyramid side of a training step
// WITHOUT a projected value tensor.
//
// Reference: the backward of Deform3DCrossAttn.forward is implicit autograd over deform3d_cross_attn.py:220-324 -
// value_proj over every pixel (:264-280), mmcv's ms_deformable_col2im (grad of value by atomicAdd, grad of sampling
// locations / attention weights), ~30 elementwise backward launches, and the gradient of the pyramid as a 97-GFLOP
// GEMM per layer.  The projected-value path of this library (gd4d_value_proj_multi_fwd, gd4d_cross_attn_bwd,
// gd4d_value_proj_bwd_input) keeps that order.  Here the forward is the inference step's (plan + sliced gather +
// value_proj of the aggregates):
//     out[q, h] = W_h A[q, h] + b_h s[q, h],   A[q, h] = sum_r w_r x_r,   s[q, h] = sum_r w_r
// (r = in-bounds bilinear corners of the visible samples of head h; x_r = RAW 256-channel pixel), and with
// g = dL/d out, dA = W_h^T g (256 channels), beta = <b_h, g>:
//     dL/d w_r   = <dA[q, h], x_r> + beta[q, h]          -> softmax / camera-weight / sampling-location gradients
//     dL/d x_r  += w_r dA[q, h]                          -> the pyramid's gradient
//     dL/d W_h   = sum_q g[q, h] (x) A[q, h],  dL/d b_h = sum_q g[q, h] s[q, h]        (host side: 900 x Hh rows)
//
//   gd4d_value_proj_heads_bwd    dA, beta from g                                   (transpose of gd4d_value_proj_heads_fwd)
//   gd4d_cross_attn_dot_sliced   D[pair] = <dA, x_r> per plan pair, one partial per channel slice: the forward's gather
//                                with the FMA replaced by a dot product (same plan, same walk, same traffic)
//   gd4d_cross_attn_plan_bwd     per query: the plan kernel's geometry again (bit-identical compaction), D summed over
//                                the slices -> grad of ref / offsets / attention logits / camera logits; every partial
//                                sum has its own LDS slot and is added in a fixed order (run-to-run identical)
//   gd4d_pyramid_grad_count / _scan / _fill / _reduce
//                                the pyramid's gradient WITHOUT atomics on feature data: the (pixel, weight, dA row)
//                                records of ALL decoder layers are bucketed by pixel (counting sort: one int atomic per
//                                record), then one pass over the pyramid sums each pixel's records (dA rows come from a
//                                table that lives in L2 / Infinity Cache) and writes the NCHW gradient once - instead of
//                                8x the fp32 atomics of the 32-channel form, a zero fill and NL accumulating GEMMs.
//
// Built with -ffp-contract=off: the visibility test must agree bit for bit with the plan kernel's (project_entry).
#include <stdlib.h>

#include "gd4d_common.h"
#include "gd4d_cross_attn_shared.h"
#include "gd4d_cross_attn_sliced.h"
#include "gd4d_pyramid_count.h"
#include "gd4d_pyramid_fill.h"
#include "gd4d_linear_bwd_body.h"

namespace gd4d {

// ---------------------------------------------------------------------------------------------------------------
// dA[m, h, :] = sum_d g[m, h Dh + d] W[h Dh + d, :],  beta[m, h] = sum_d g[m, h Dh + d] b[h Dh + d]
// One workgroup per (16 rows, head), thread = output channel: the head's Dh weight rows are requested up front (the
// first version walked all heads of 8 rows in one workgroup, a chain of Hh * Dh dependent loads: 101 us in the step).
constexpr int HB_ROWS = 16;

template <int DH>
__global__ __launch_bounds__(256) void value_proj_heads_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ da,
                                                                   float* __restrict__ beta, int M, int HH) {
  __shared__ float s_g[HB_ROWS][DH];
  const int tid = threadIdx.x;
  const int h = blockIdx.x % HH;
  const int m0 = (blockIdx.x / HH) * HB_ROWS;
  const int rows = min(HB_ROWS, M - m0);
  float wv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) wv[d] = w[(size_t)(h * DH + d) * kChannels + tid];
  for (int i = tid; i < HB_ROWS * DH; i += 256) {
    const int r = i / DH, d = i % DH;
    s_g[r][d] = r < rows ? g[(size_t)(m0 + r) * kChannels + h * DH + d] : 0.f;
  }
  __syncthreads();
  for (int r = 0; r < rows; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) acc = fmaf(s_g[r][d], wv[d], acc);
    da[((size_t)(m0 + r) * HH + h) * kChannels + tid] = acc;
  }
  if (beta && tid < rows) {
    float t = 0.f;
    if (bias)
      for (int d = 0; d < DH; ++d) t = fmaf(s_g[tid][d], bias[h * DH + d], t);
    beta[(size_t)(m0 + tid) * HH + h] = t;
  }
}

// dW[h Dh + d, c] = sum_m g[m, h Dh + d] agg[m, h, c],  db[h Dh + d] = sum_m g[m, h Dh + d] wsum[m, h]
// Split over the rows: workgroup (head, 64 input channels, slice ks of HW_KS) sums its rows into a partial (wave = 64
// channels x one group of Dh / 4 outputs: the g values it needs are wave-uniform - scalar loads); a second launch adds the
// HW_KS partials in a fixed order (run-to-run identical).  (One workgroup per (head, channels) over all 900 rows: 112 us;
// 8 slices of 113 rows, one workgroup per compute unit: 51 us - four loads in flight per wave.)
constexpr int HW_KS = 32;

template <int DH>
__device__ __forceinline__ void value_proj_heads_bwd_weight_body(const float* __restrict__ g, const float* __restrict__ agg,
                                                                 const float* __restrict__ wsum, float* __restrict__ part,
                                                                 float* __restrict__ part_b, int M, int HH, int block) {
  constexpr int DPT = DH / 4;                       // outputs per thread
  const int tid = threadIdx.x;
  const int c = tid & 63;
  const int dg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = kChannels / 64;
  const int ks = block % HW_KS, hc = block / HW_KS;
  const int h = hc / tiles, c0 = (hc % tiles) * 64;
  const int per = (M + HW_KS - 1) / HW_KS;
  const int m0 = ks * per, m1 = min(M, m0 + per);
  float acc[DPT];
#pragma unroll
  for (int i = 0; i < DPT; ++i) acc[i] = 0.f;
  float accb = 0.f;
  const bool do_bias = part_b && c0 == 0 && c < DPT;   // lane c < DPT of wave dg: db of output dg DPT + c over these rows
  const float* ap = agg + (size_t)h * kChannels + c0 + c;
  const float* gp = g + h * DH + dg * DPT;
#pragma unroll 8
  for (int m = m0; m < m1; ++m) {
    const float a = ap[(size_t)m * HH * kChannels];
#pragma unroll
    for (int i = 0; i < DPT; ++i) acc[i] = fmaf(gp[(size_t)m * kChannels + i], a, acc[i]);
  }
  if (do_bias) {                                     // (its own loop: inside the one above its branch split every batch of loads)
#pragma unroll 8
    for (int m = m0; m < m1; ++m) accb = fmaf(g[(size_t)m * kChannels + h * DH + dg * DPT + c], wsum[(size_t)m * HH + h], accb);
  }
#pragma unroll
  for (int i = 0; i < DPT; ++i) part[((size_t)ks * kChannels + h * DH + dg * DPT + i) * kChannels + c0 + c] = acc[i];
  if (do_bias) part_b[ks * kChannels + h * DH + dg * DPT + c] = accb;
}

template <int DH>
__global__ __launch_bounds__(256) void value_proj_heads_bwd_weight_kernel(const float* __restrict__ g, const float* __restrict__ agg,
                                                                           const float* __restrict__ wsum, float* __restrict__ part,
                                                                           float* __restrict__ part_b, int M, int HH) {
  value_proj_heads_bwd_weight_body<DH>(g, agg, wsum, part, part_b, M, HH, blockIdx.x);
}

// The same for up to HW_GROUP layers of a training step in ONE launch (blockIdx.y = layer): a layer's launch is one workgroup
// per compute unit waiting on its loads (12 us) followed by a 5-us sum - six layers side by side take hardly longer than one.
constexpr int HW_GROUP = 8;
struct HwGroup {
  const float* g[HW_GROUP]; const float* agg[HW_GROUP]; const float* wsum[HW_GROUP];
  float* gw[HW_GROUP]; float* gb[HW_GROUP];
  int M[HW_GROUP];
  float* part;                 // [layer][HW_KS * C * C + HW_KS * C]
  int count, accumulate;
};
__device__ __forceinline__ size_t hw_part_stride() { return (size_t)HW_KS * (kChannels + 1) * kChannels; }

template <int DH>
__global__ __launch_bounds__(256) void value_proj_heads_bwd_weight_group_kernel(const HwGroup q, int HH) {
  const int l = blockIdx.y;
  const float* g = q.g[0]; const float* agg = q.agg[0]; const float* wsum = q.wsum[0]; int M = q.M[0]; bool bias = q.gb[0] != nullptr;
#pragma unroll
  for (int j = 1; j < HW_GROUP; ++j)
    if (j == l) { g = q.g[j]; agg = q.agg[j]; wsum = q.wsum[j]; M = q.M[j]; bias = q.gb[j] != nullptr; }
  float* part = q.part + (size_t)l * hw_part_stride();
  value_proj_heads_bwd_weight_body<DH>(g, agg, wsum, part, bias ? part + (size_t)HW_KS * kChannels * kChannels : nullptr, M, HH, blockIdx.x);
}

__device__ __forceinline__ void value_proj_heads_bwd_weight_sum_body(const float* __restrict__ part, const float* __restrict__ part_b,
                                                                     float* __restrict__ gw, float* __restrict__ gb, int accumulate, int block);

__global__ __launch_bounds__(256) void value_proj_heads_bwd_weight_sum_group_kernel(const HwGroup q) {
  const int l = blockIdx.y;
  float* gw = q.gw[0]; float* gb = q.gb[0];
#pragma unroll
  for (int j = 1; j < HW_GROUP; ++j)
    if (j == l) { gw = q.gw[j]; gb = q.gb[j]; }
  const float* part = q.part + (size_t)l * hw_part_stride();
  value_proj_heads_bwd_weight_sum_body(part, part + (size_t)HW_KS * kChannels * kChannels, gw, gb, q.accumulate, blockIdx.x);
}

__global__ __launch_bounds__(256) void value_proj_heads_bwd_weight_sum_kernel(const float* __restrict__ part, const float* __restrict__ part_b,
                                                                               float* __restrict__ gw, float* __restrict__ gb, int accumulate) {
  value_proj_heads_bwd_weight_sum_body(part, part_b, gw, gb, accumulate, blockIdx.x);
}

__device__ __forceinline__ void value_proj_heads_bwd_weight_sum_body(const float* __restrict__ part, const float* __restrict__ part_b,
                                                                     float* __restrict__ gw, float* __restrict__ gb, int accumulate, int block) {
  const int i = block * 256 + threadIdx.x;           // kChannels * kChannels / 4 float4 results
  float4 t = reinterpret_cast<const float4*>(part)[i];
#pragma unroll
  for (int ks = 1; ks < HW_KS; ++ks) {
    const float4 v = reinterpret_cast<const float4*>(part)[(size_t)ks * (kChannels * kChannels / 4) + i];
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  float* o = gw + 4 * (size_t)i;                     // (a view of a flat gradient buffer need not be 16-byte aligned)
  if (accumulate) { t.x += o[0]; t.y += o[1]; t.z += o[2]; t.w += o[3]; }
  o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  if (gb && i < kChannels) {
    float b = part_b[i];
#pragma unroll
    for (int ks = 1; ks < HW_KS; ++ks) b += part_b[ks * kChannels + i];
    gb[i] = accumulate ? gb[i] + b : b;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Gather-dot: the walk of cross_attn_agg_sliced_kernel (grid = 8 XCDs x 8 slices x per_xcd, slice-major inside an XCD,
// wave = head, a pass = 8 loads of 8 corners x 128 B).  Per pass a lane ends up with the partial dot product of ONE
// pair of the pass (pair index = lane, the plan's [g][j] order) over this slice's 32 channels: the 8 loads leave lane
// (g, c) with 8 four-channel products (j = 0..7); a transposing butterfly over the 8 lanes of a corner slot (7 shuffles
// instead of 24) leaves lane (g, c) the sum for j = c.  One coalesced 256-byte store per pass and slice.
struct DotParams {
  const char* lvl_base[4];
  long long slice_stride;
  const int* hdr;
  const uint2* pair;
  const int32_t* order;
  const float* gagg;        // (BQ, HH, 256) dL/d agg
  float* dpart;             // [kSlices][BQ * HH * cap_t * 64]
  long long dslice;         // floats per slice of dpart
  int BQ, per_xcd, cap_t;
};

template <int HH, int LT, typename VT>
__device__ __forceinline__ void cross_attn_dot_sliced_body(const DotParams& p, const int bid, char* s_raw) {   // s_raw: [HH][CH][8][GP]
  constexpr int CH = 6, GP = 80, PASS = 8 * GP, ES = sizeof(VT);
  const int lane = threadIdx.x & 63;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = bid & 7, jb = bid >> 3;
  const int s = jb / p.per_xcd, qi = jb - s * p.per_xcd;
  const int pos = xcd * p.per_xcd + qi;
  if (pos >= p.BQ) return;
  const size_t prow = ((size_t)pos * HH + h) * p.cap_t * 64 + lane;
  const uint2* pp = p.pair + prow;
  float* dp = p.dpart + (size_t)s * p.dslice + prow;
  const uint2 first = pp[0];
  const int M = __builtin_amdgcn_readfirstlane(p.hdr[pos * kPlanHdr + h]);
  const int bq = p.order ? p.order[pos] : pos;
  const int T = (M + 3) >> 2;
  char* my = s_raw + h * (CH * PASS);
  const int g = lane >> 3, c = lane & 7;
  char* wr = my + g * GP + c * 8;
  const char* rd = my + g * GP;
  const char* base[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) base[l] = p.lvl_base[l] + (size_t)s * p.slice_stride;
  const unsigned lane_off = (unsigned)(c * 4 * ES);
  const float4 ga = *reinterpret_cast<const float4*>(p.gagg + ((size_t)bq * HH + h) * kChannels + s * kSlice + c * 4);
  const bool b4 = c & 4, b2 = c & 2, b1 = c & 1;

  for (int t0 = 0; t0 < T; t0 += CH) {
    const int nt = min(CH, T - t0);
    if (t0 > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (k >= nt) break;
      const uint2 v = (k == 0 && t0 == 0) ? first : pp[(size_t)(t0 + k) * 64];
      *reinterpret_cast<uint2*>(wr + k * PASS) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = 0; k < nt; ++k) {
      const uint4* row = reinterpret_cast<const uint4*>(rd + k * PASS);
      uint4 pr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = row[i];
      const bool second = (t0 + k) * 4 + 2 < M;
      float4 val[8];
      // all eight loads of the pass issued back to back, without a branch: when items 2, 3 of the pass do not exist their
      // loads repeat those of items 0, 1 (lines just requested) and their products are dropped below.  (Skipping them with
      // a `continue` per load made the compiler wait for each of the four in turn, a uniform branch around the four still
      // put the two groups one after the other: 181 us per launch.)
      unsigned off[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) off[j] = (j & 1) ? pr[j >> 1].z : pr[j >> 1].x;
      // (items 2, 3 absent: the offsets of items 0, 1 again - BEFORE the lane's share is added: adding it to an offset that
      //  already carries it read up to 112 bytes past the pixel, i.e. past the END of the pyramid for the last pixel of the last
      //  camera row of the last slice - a fault when the allocation ends where the mapped memory does)
#pragma unroll
      for (int j = 4; j < 8; ++j) off[j] = second ? off[j] : off[j - 4];
#pragma unroll
      for (int j = 0; j < 8; ++j) off[j] += lane_off;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        val[j] = (j & 3) < LT ? Quad<VT>::load(reinterpret_cast<const VT*>(base[j & 3] + off[j])) : make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_sched_barrier(0);
      float d[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        d[j] = (ga.x * val[j].x + ga.y * val[j].y) + (ga.z * val[j].z + ga.w * val[j].w);
        if (j >= 4) d[j] = second ? d[j] : 0.f;
      }
      float e[4], f[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = b4 ? d[i + 4] : d[i], send = b4 ? d[i] : d[i + 4];
        e[i] = keep + __shfl_xor(send, 4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? e[i + 2] : e[i], send = b2 ? e[i] : e[i + 2];
        f[i] = keep + __shfl_xor(send, 2);
      }
      const float keep = b1 ? f[1] : f[0], send = b1 ? f[0] : f[1];
      dp[(size_t)(t0 + k) * 64] = keep + __shfl_xor(send, 1);
    }
  }
}

template <int HH, int LT, typename VT, int OCC>
__global__ __launch_bounds__(64 * HH, OCC) void cross_attn_dot_sliced_kernel(const DotParams p) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  cross_attn_dot_sliced_body<HH, LT, VT>(p, (int)blockIdx.x, s_raw);
}

// A backward gather-dot and up to 16 queued weight gradients of the decoder's Linears (gd4d_linear_bwd_body.h) in ONE launch
// (gd4d_cross_attn_dot_sliced_wgrad).  Nothing reads a weight gradient before the optimizer; a training step queued them and issued
// them sixteen per launch when the backward pass ended: five launches of 36 - 47 us, each a chain of dependent load rounds on
// fp32 MFMA tiles.  Here the tiles of what is queued when a layer's gather-dot starts are guest workgroups of that launch (8
// waves each instead of 16), spread evenly among the gather-dot's, which live on the fabric's random reads; the gather-dot's are
// renumbered (a multiple of 8 leaves: XCD and walk unchanged).
struct WgradGuest {
  DotParams p;                                 // (one kernel argument: the descriptor below is read through the argument segment)
  LinBwdGroup g;
  int tiles, guest_groups, total_groups;       // groups of 8 workgroups: guests among all
};

template <int HH, int LT, typename VT, int OCC>
__global__ __launch_bounds__(64 * HH, OCC) void cross_attn_dot_sliced_wgrad_kernel(const WgradGuest wg) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  static_assert(sizeof(LinBwdShared<HH>) <= (size_t)HH * 6 * 8 * 80, "the guest's partial tiles fit the gather-dot's LDS");
  const long long grp = blockIdx.x >> 3;
  const int before = (int)(grp * wg.guest_groups / wg.total_groups);          // guest groups in front of this one
  if ((int)((grp + 1) * wg.guest_groups / wg.total_groups) > before) {
    const int tile = before * 8 + (int)(blockIdx.x & 7);
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) WgradGuest* args_ptr_t;
    const lin_group_ptr_t gp = &((args_ptr_t)__builtin_amdgcn_kernarg_segment_ptr())->g;
#else
    const lin_group_ptr_t gp = &wg.g;
#endif
    if (tile < wg.tiles) linear_bwd_weight_group_tile<HH>(gp, tile, *reinterpret_cast<LinBwdShared<HH>*>(s_raw));
    return;
  }
  cross_attn_dot_sliced_body<HH, LT, VT>(wg.p, (int)blockIdx.x - 8 * before, s_raw);
}

// ---------------------------------------------------------------------------------------------------------------
// Query-side gradients from D.  One workgroup per position of the locality order, phases as in cross_attn_plan_kernel
// (the projection and the compaction are the same code: item m of head h here IS item m of the plan).
struct PlanBwdParams {
  CrossAttnParams c;
  int lvl_w[4], lvl_h[4];
  const int* hdr;
  const float* dpart;
  long long dslice;
  int cap_t;
  const float* beta;        // (BQ, HH) or NULL
  float* grad_ref;          // (B, Q, 3)
  float* grad_offsets;      // (B, Q, HH, P, 3)
  float* grad_attn_logits;  // (B, Q, HH, L, P)
  float* grad_cam_logits;   // (B, Q, N) in the raw-view layout of cam_logits
  float* ga_part;           // B > 1: (B_geometry, B_class, Q, HH, L*P) partial dL/d a (before the softmax backward)
  int* status;              // optional: set to 1 when an item count disagrees with the plan (cannot happen: tested)
};

// PT = sampling points per head: 4 (every shipped config) or 8 (8 heads; what num_points 5 .. 8 are padded to - the reference's
// constructor default is 5, deform3d_cross_attn.py:56: padded points carry a NaN offset, so they are never visible, and a -inf
// logit, so their softmax weight is exactly 0)
template <int HH, int LT, int WAVES, bool BMULTI, int PT = kPoints>
__global__ __launch_bounds__(64 * WAVES) void cross_attn_plan_bwd_kernel(const PlanBwdParams pp) {
  const CrossAttnParams& p = pp.c;
  constexpr int E = HH * PT, LP = LT * PT, THREADS = 64 * WAVES;
  static_assert(LP <= GD4D_WAVE && PT * 3 <= GD4D_WAVE, "a lane per logit / per offset component");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int ncand = p.N * PT;
  float2* s_uv = reinterpret_cast<float2*>(smem_raw);                        // [N][E]; x < 0: not visible
  float* s_mat = reinterpret_cast<float*>(s_uv + p.N * E);                   // [N][12]
  float* s_cw = s_mat + p.N * 12;                                            // [N]
  float* s_aw = s_cw + ((p.N + 3) & ~3);                                     // [B][HH][LP]
  float* s_pt = s_aw + ((p.B * HH * LP + 3) & ~3);                           // [E][4] metre-space points
  float4* s_items = reinterpret_cast<float4*>(s_pt + E * 4);                 // [WAVES][ncand]
  float* s_ga = reinterpret_cast<float*>(s_items + WAVES * ncand);           // [WAVES][ncand][4]: cw T per (item, level)
  float* s_gpt = s_ga + WAVES * ncand * 4;                                   // [WAVES][ncand][4]: dL/d point per item
  float* s_camT = s_gpt + WAVES * ncand * 4;                                 // [HH][ncand]: sum_l a T per item
  float* s_ref = s_camT + HH * ncand;                                        // [HH][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pos = blockIdx.x;
  const int bq = p.order ? p.order[pos] : pos;
  const int b = bq / p.Q, q = bq - b * p.Q;

  for (int i = tid; i < p.N * 12; i += THREADS) s_mat[i] = p.lidar2img[((size_t)b * p.N + i / 12) * 16 + i % 12];
  if (tid < p.N) {
    const float cl = p.cam_logits[(size_t)b * p.Q * p.N + (size_t)tid * p.Q + q];
    s_cw[tid] = p.raw_cam ? cl : 1.0f / (1.0f + expf(-cl));
  }
  for (int i = tid; i < HH * ncand; i += THREADS) s_camT[i] = 0.f;
  if ((LP & (LP - 1)) == 0 && LP <= 32 && (HH * LP) % GD4D_WAVE == 0) {      // as the plan kernel (same weights bit for bit)
    for (int t = tid; t < p.B * HH * LP; t += THREADS) {
      const int bh = t / LP, i = t % LP;
      const int bb = bh / HH, hd = bh - bb * HH;
      const float x = p.attn_logits[(((size_t)bb * p.Q + q) * HH + hd) * LP + i];
      float mx = x;
#pragma unroll
      for (int o = 1; o < LP; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float e = expf(x - mx);
      float sum = e;
#pragma unroll
      for (int o = 1; o < LP; o <<= 1) sum += __shfl_xor(sum, o);
      s_aw[t] = e * (1.0f / sum);
    }
  } else {
    for (int t = tid; t < p.B * HH; t += THREADS) {
      const int bb = t / HH, hd = t - bb * HH;
      float w[LP];
      softmax_lp(p.attn_logits + (((size_t)bb * p.Q + q) * HH + hd) * LP, LP, w);
#pragma unroll
      for (int i = 0; i < LP; ++i) s_aw[t * LP + i] = w[i];
    }
  }
  {
    const int total = p.N * E;
    static_assert(E <= GD4D_WAVE && GD4D_WAVE % E == 0 && THREADS % E == 0, "a thread keeps its (head, point)");
    const int hp = tid % E;
    const float* rp = p.ref + (size_t)bq * 3;
    const float* offs = p.offsets + ((size_t)bq * E + hp) * 3;
    const float X = (rp[0] * p.rng_scale[0] + p.rng_lo[0]) + offs[0];
    const float Y = (rp[1] * p.rng_scale[1] + p.rng_lo[1]) + offs[1];
    const float Z = (rp[2] * p.rng_scale[2] + p.rng_lo[2]) + offs[2];
    if (tid < E) { s_pt[hp * 4] = X; s_pt[hp * 4 + 1] = Y; s_pt[hp * 4 + 2] = Z; }
    __syncthreads();
    for (int e0 = wave * GD4D_WAVE; e0 < total; e0 += THREADS) {
      const int e = e0 + lane;
      if (e < total) {
        float u, v;
        const bool vis = project_entry(p, s_mat + (e / E) * 12, X, Y, Z, u, v);
        s_uv[e] = vis ? make_float2(u, v) : make_float2(-1.f, -1.f);
      }
    }
  }
  __syncthreads();

  float4* items = s_items + wave * ncand;
  float* ga_w = s_ga + wave * ncand * 4;
  float* gpt_w = s_gpt + wave * ncand * 4;
  const int i_of = lane >> 2, l_of = lane & 3;
  int lw = pp.lvl_w[0], lh = pp.lvl_h[0];
#pragma unroll
  for (int l = 1; l < LT; ++l)
    if (l_of == l) { lw = pp.lvl_w[l]; lh = pp.lvl_h[l]; }
  const float flw = (float)lw, flh = (float)lh;
  const int slot0 = (((i_of & 1) << 2) << 3) | (((i_of >> 1) & 1) << 2) | l_of;
  for (int h = wave; h < HH; h += WAVES) {
    for (int i = lane; i < ncand * 4; i += GD4D_WAVE) { ga_w[i] = 0.f; gpt_w[i] = 0.f; }
    int M = 0;
    for (int c0 = 0; c0 < ncand; c0 += GD4D_WAVE) {
      const int cand = c0 + lane;
      const int n = min(cand, ncand - 1) / PT, k = cand % PT;
      const float2 uvc = s_uv[n * E + h * PT + k];
      const bool vis = cand < ncand && uvc.x >= 0.f;
      const unsigned long long bal = __ballot(vis);
      if (vis) items[M + __popcll(bal & ((1ull << lane) - 1ull))] = make_float4(uvc.x, uvc.y, __int_as_float((b * p.N + n) * PT + k), s_cw[n]);
      M += __popcll(bal);
    }
    if (pp.status && lane == 0 && M != pp.hdr[pos * kPlanHdr + h]) *pp.status = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const size_t prow = ((size_t)pos * HH + h) * pp.cap_t * 64;
    const float* aw_h = s_aw + h * LP + min(l_of, LT - 1) * PT;
    const float beta = pp.beta ? pp.beta[(size_t)bq * HH + h] : 0.f;
    for (int it0 = 0; it0 < M; it0 += 16) {
      const int item = it0 + i_of;
      const bool live = item < M && l_of < LT;
      const float4 rec = items[min(item, M - 1)];
      const int rk = __float_as_int(rec.z);
      const int row = rk / PT, k = rk % PT;
      const int n = row - b * p.N;
      const int cand = n * PT + k;
      const int lb = p.B == 1 ? 0 : row % p.B;
      const float a = aw_h[lb * HH * LP + k];
      const float cw = rec.w;
      const float x = fmaf(rec.x, flw, -0.5f);
      const float y = fmaf(rec.y, flh, -0.5f);
      const float xf = floorf(x), yf = floorf(y);
      const float dx = x - xf, dy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      // the 4 x 8 slice partials of this lane's corners, all requested before the first is used (a pass of the gather-dot
      // writes all 64 of its entries, so every address below holds a finite or at least harmless value: lanes past the list's
      // end read their slot of its last pass; what a dead lane or a corner outside the map reads is dropped by the select).
      // Loaded corner by corner inside `if (in)` the four groups were four round trips in turn.
      const float* dsrc = pp.dpart + prow + (size_t)(min(item, M - 1) >> 2) * 64 + slot0;
      float part[4][kSlices];
#pragma unroll
      for (int c_of = 0; c_of < 4; ++c_of)
#pragma unroll
        for (int s = 0; s < kSlices; ++s) part[c_of][s] = dsrc[(size_t)s * pp.dslice + (c_of << 3)];
      float d[4];
#pragma unroll
      for (int c_of = 0; c_of < 4; ++c_of) {
        const int xi = x0 + (c_of & 1), yi = y0 + (c_of >> 1);
        const bool in = live && xi >= 0 && xi < lw && yi >= 0 && yi < lh;
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < kSlices; ++s) t += part[c_of][s];
        t += beta;
        d[c_of] = in ? t : 0.f;                       // corners outside the map (zero padding) and dead lanes: 0
      }
      const float bw0 = (1.f - dx) * (1.f - dy), bw1 = dx * (1.f - dy), bw2 = (1.f - dx) * dy, bw3 = dx * dy;
      const float T = (bw0 * d[0] + bw1 * d[1]) + (bw2 * d[2] + bw3 * d[3]);
      const float dTdx = (1.f - dy) * (d[1] - d[0]) + dy * (d[3] - d[2]);
      const float dTdy = (1.f - dx) * (d[2] - d[0]) + dx * (d[3] - d[1]);
      if (live) ga_w[cand * 4 + l_of] = cw * T;
      float camv = a * T, gu = a * dTdx * flw, gv = a * dTdy * flh;       // dead lanes: T = 0
      camv += __shfl_xor(camv, 1); gu += __shfl_xor(gu, 1); gv += __shfl_xor(gv, 1);
      camv += __shfl_xor(camv, 2); gu += __shfl_xor(gu, 2); gv += __shfl_xor(gv, 2);
      if (l_of == 0 && item < M) {
        s_camT[h * ncand + cand] = camv;
        gu *= cw; gv *= cw;
        const float* m = s_mat + n * 12;
        const float X = s_pt[(h * PT + k) * 4], Y = s_pt[(h * PT + k) * 4 + 1], Z = s_pt[(h * PT + k) * 4 + 2];
        const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
        const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
        const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
        const float iz = 1.0f / cz;                    // visible: cz > eps
        const float gcx = gu * iz / p.img_w;
        const float gcy = gv * iz / p.img_h;
        const float gcz = -(gu * cx * iz * iz / p.img_w + gv * cy * iz * iz / p.img_h);
        gpt_w[cand * 4 + 0] = gcx * m[0] + gcy * m[4] + gcz * m[8];
        gpt_w[cand * 4 + 1] = gcx * m[1] + gcy * m[5] + gcz * m[9];
        gpt_w[cand * 4 + 2] = gcx * m[2] + gcy * m[6] + gcz * m[10];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // fixed-order sums over the cameras
    {
      const int i = lane;                                // logit (level, point) = (i / PT, i % PT)
      const int l = min(i, LP - 1) / PT, k = i % PT;
      if (!BMULTI) {
        float tot = 0.f;
        if (i < LP)
          for (int n = 0; n < p.N; ++n) tot += ga_w[(n * PT + k) * 4 + l];
        const float ai = i < LP ? s_aw[h * LP + i] : 0.f;
        float dot = ai * tot;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        if (i < LP) pp.grad_attn_logits[((size_t)bq * HH + h) * LP + i] = ai * (tot - dot);
      } else if (i < LP) {
        for (int cls = 0; cls < p.B; ++cls) {
          float tot = 0.f;
          for (int n = 0; n < p.N; ++n)
            if ((b * p.N + n) % p.B == cls) tot += ga_w[(n * PT + k) * 4 + l];
          pp.ga_part[((((size_t)b * p.B + cls) * p.Q + q) * HH + h) * LP + i] = tot;
        }
      }
      if (lane < PT * 3) {
        const int kk = lane / 3, dd = lane % 3;
        float tot = 0.f;
        for (int n = 0; n < p.N; ++n) tot += gpt_w[(n * PT + kk) * 4 + dd];
        pp.grad_offsets[((size_t)bq * HH + h) * PT * 3 + lane] = tot;
      }
      if (lane >= 61) {
        const int dd = lane - 61;
        float tot = 0.f;
        for (int kk = 0; kk < PT; ++kk) {
          float t = 0.f;
          for (int n = 0; n < p.N; ++n) t += gpt_w[(n * PT + kk) * 4 + dd];
          tot += t;
        }
        s_ref[h * 4 + dd] = tot;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  if (tid < p.N) {
    float cs = 0.f;
    for (int h = 0; h < HH; ++h)
      for (int k = 0; k < PT; ++k) cs += s_camT[h * ncand + tid * PT + k];
    const float cw = s_cw[tid];
    pp.grad_cam_logits[(size_t)b * p.Q * p.N + (size_t)tid * p.Q + q] = (p.raw_cam ? 1.f : cw * (1.f - cw)) * cs;
  }
  if (tid >= 64 && tid < 67) {
    const int dd = tid - 64;
    float t = 0.f;
    for (int h = 0; h < HH; ++h) t += s_ref[h * 4 + dd];
    pp.grad_ref[(size_t)bq * 3 + dd] = t * p.rng_scale[dd];
  }
}

// B > 1: grad_attn_logits[bb, q, h, :] from the partial rows of all geometry owners b (fixed order) - softmax backward.
template <int LP>
__global__ __launch_bounds__(64) void plan_bwd_logits_kernel(const float* __restrict__ ga_part, const float* __restrict__ attn_logits,
                                                             float* __restrict__ grad_attn_logits, int B, int Q, int HH) {
  const int bbq = blockIdx.x;                      // bb * Q + q
  const int bb = bbq / Q, q = bbq - bb * Q;
  const int hh = threadIdx.x;
  if (hh >= HH) return;
  float tot[LP];
#pragma unroll
  for (int i = 0; i < LP; ++i) tot[i] = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* src = ga_part + ((((size_t)b * B + bb) * Q + q) * HH + hh) * LP;
#pragma unroll
    for (int i = 0; i < LP; ++i) tot[i] += src[i];
  }
  float a2[LP];
  softmax_lp(attn_logits + ((size_t)bbq * HH + hh) * LP, LP, a2);
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) dot += a2[i] * tot[i];
  float* gl = grad_attn_logits + ((size_t)bbq * HH + hh) * LP;
#pragma unroll
  for (int i = 0; i < LP; ++i) gl[i] = a2[i] * (tot[i] - dot);
}

// ---------------------------------------------------------------------------------------------------------------
// The pyramid's gradient.  The pixels are grouped into CHUNKS of at most 64 (cw x ch pixels of one camera row and
// level, both powers of two: 32 x 2 on the fine levels, smaller on the coarse ones, where a pixel owns hundreds of
// records); a record is bucketed by CHUNK, not by pixel: the 4 corners of a sample - and at the coarse levels most
// samples of a head - share a chunk, so a wave adds them to the chunk's counter with ONE atomic (the L2 retires ~15 G
// atomic requests/s whatever their width: per-pixel buckets cost 1.8 ms per step in atomics alone).
// (PgChunks, the count step's body and fill_chunks: gd4d_pyramid_count.h - shared with the forward gather's launch)

__global__ __launch_bounds__(256) void pyramid_grad_count_kernel(const int* __restrict__ hdr, const uint2* __restrict__ pair,
                                                                 int cap_t, int HH, int BQ, PgChunks g, int* __restrict__ count,
                                                                 uint2* __restrict__ slots) {
  pyramid_grad_count_body(hdr, pair, cap_t, HH, BQ, g, count, slots, blockIdx.x * 4 + (threadIdx.x >> 6));   // gd4d_pyramid_count.h
}

// (the fill's body: gd4d_pyramid_fill.h - shared with the attention backward's launch)
__global__ __launch_bounds__(256) void pyramid_grad_fill_kernel(const int* __restrict__ hdr, const uint2* __restrict__ pair,
                                                                const uint2* __restrict__ slots, int cap_t, int HH, int BQ,
                                                                const int* __restrict__ start, uint2* __restrict__ rec,
                                                                const int32_t* __restrict__ order, unsigned id_base) {
  pyramid_grad_fill_body(hdr, pair, slots, cap_t, HH, BQ, start, rec, order, id_base, blockIdx.x * 4 + (threadIdx.x >> 6));
}

constexpr int SCAN_THREADS = 256, SCAN_PER = 16, SCAN_CHUNK = SCAN_THREADS * SCAN_PER;

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* s_w, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_w[w];
  total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void pg_scan_sums_kernel(const int* __restrict__ count, long long n, int* __restrict__ bsum) {
  __shared__ int s_w[4];
  const long long base = (long long)blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_PER;
  int v = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER; ++i) v += base + i < n ? count[base + i] : 0;
  int total;
  block_exclusive_scan_256(v, s_w, total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_THREADS) void pg_scan_bsums_kernel(int* __restrict__ bsum, int nb) {
  __shared__ int s_w[4];
  int carry = 0;
  for (int c0 = 0; c0 < nb; c0 += SCAN_THREADS) {
    const int i = c0 + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int total;
    const int ex = block_exclusive_scan_256(v, s_w, total);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void pg_scan_apply_kernel(const int* __restrict__ count, long long n,
                                                                    const int* __restrict__ bsum, int* __restrict__ cursor) {
  __shared__ int s_w[4];
  const long long base = (long long)blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_PER;
  int c[SCAN_PER];
  int v = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER; ++i) { c[i] = base + i < n ? count[base + i] : 0; v += c[i]; }
  int total;
  int run = bsum[blockIdx.x] + block_exclusive_scan_256(v, s_w, total);
#pragma unroll
  for (int i = 0; i < SCAN_PER; ++i) {
    if (base + i < n) cursor[base + i] = run;
    run += c[i];
  }
}

// Sort every chunk's records by pixel (6-bit keys, one 512-thread workgroup per chunk, two passes over the chunk: count
// per pixel - lanes of a wave that hold the same pixel are matched with 6 ballots and share one LDS atomic - a 64-entry
// scan, then placement).  Output: the records again, grouped by pixel (sorted[start[chunk] + ...]), and pxoff[chunk][65],
// the start of every pixel's run inside the chunk.  Needs nothing from the backward pass: it runs beside it.
struct PgMatch { int leader, rank, cnt; };

__device__ __forceinline__ PgMatch pg_match(bool valid, unsigned px, int lane) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < 6; ++bit) {
    const bool on = (px >> bit) & 1u;
    const unsigned long long bal = __ballot(on);
    peers &= on ? bal : ~bal;
  }
  PgMatch m;
  m.leader = valid ? __ffsll((long long)peers) - 1 : lane;
  m.rank = __popcll(peers & ((1ull << lane) - 1ull));
  m.cnt = __popcll(peers);
  return m;
}

constexpr int PG_THREADS = 512;

__global__ __launch_bounds__(PG_THREADS) void pyramid_grad_sort_kernel(const int* __restrict__ count, const int* __restrict__ start,
                                                                       const uint2* __restrict__ rec, uint2* __restrict__ sorted,
                                                                       int* __restrict__ pxoff) {
  __shared__ int s_hist[64], s_cur[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x;
  const int n = count[chunk];
  const size_t base = (size_t)start[chunk];
  if (tid < 64) s_hist[tid] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += PG_THREADS) {
    const int i = i0 + tid;
    const bool valid = i < n;
    const unsigned px = valid ? rec[base + i].y >> 26 : 0u;
    const PgMatch m = pg_match(valid, px, lane);
    if (valid && m.leader == lane) atomicAdd(&s_hist[px], m.cnt);
  }
  __syncthreads();
  if (wave == 0) {
    const int v = s_hist[lane];
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    s_cur[lane] = inc - v;
    pxoff[(size_t)chunk * 65 + lane] = inc - v;
    if (lane == 63) pxoff[(size_t)chunk * 65 + 64] = inc;
  }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += PG_THREADS) {
    const int i = i0 + tid;
    const bool valid = i < n;
    const uint2 r = valid ? rec[base + i] : make_uint2(0u, 0u);
    const unsigned px = r.y >> 26;
    const PgMatch m = pg_match(valid, px, lane);
    int at = 0;
    if (valid && m.leader == lane) at = atomicAdd(&s_cur[px], m.cnt);
    at = __shfl(at, m.leader);
    if (valid) sorted[base + at + m.rank] = r;
  }
}

// One 512-thread workgroup per chunk.  Per record: one 1-KB row of the dA table (lane c holds channels 4c .. 4c+3) and four
// FMAs with a scalar multiplier.  A finished pixel is one row of an LDS tile; the tile is then written NCHW, every pixel of
// the pyramid exactly once: no zero fill, no read-modify-write.  Grid = 8 XCDs x an eighth of the chunks (chunk_order: the
// walk; or level by level, coarse first).
//   * A wave takes an eighth of the chunk's RECORDS (whole groups of 8), whatever pixels they belong to; a pixel that starts
//     in an earlier wave's share is a partial sum in part[wave], added to the pixel's row after the barrier, in wave order
//     (a fixed order: run-to-run identical).  (Rounds 3 - 4 handed out PIXELS: the slowest of the 8 waves set the pace, and a
//     group of 8 loads never crossed a pixel - at level 0, 4.6 records per pixel, 8 loads were issued per 4.6 records.)
//   * The records arrive by VECTOR loads, 64 per instruction (lane = record), the next 64 requested a batch ahead; vector
//     loads return in order, so the wait for a group's table rows does not wait for anything issued after them.  (Rounds 3 - 4:
//     one scalar load per group of 8 - scalar loads return out of order, the wait for one group's records also waited for
//     the prefetched next one: an exposed latency per group.)  A record's weight / table row / pixel reach the scalar side by
//     v_readlane; pixel boundaries inside a group are found by comparing scalars (fast path: the group's last record has the
//     current pixel - records are sorted by pixel).
//   * Two groups of 8 table rows in flight per wave (16 KB; 16 waves per compute unit).
// What bounds it (docs/measurements_r05.md section 9): 0.73 ms, of which the write-out alone - 757 MB as 128-byte runs, one per
// (channel, row of 32 pixels) - takes 0.37 - 0.55 ms when nothing else runs (1-KB runs: 0.2 ms), and the table rows'
// 11.9 GB through the L1s (0.3 ms at 64 B/clk per compute unit).  A persistent form (workgroups that stay and prefetch
// their next chunk: docs/r05/persistent_walk_reduce_prototype.hip.txt) measured slower.
struct PgReduceParams {
  float* out[GD4D_MAX_LEVELS];
  PgChunks g;
  int per_xcd[GD4D_MAX_LEVELS];  // chunks of level l per XCD (walk without chunk_order)
  const int32_t* chunk_order;    // optional: the walk, a permutation of the chunks; XCD x takes entries [x per, (x + 1) per)
  int per;                       //   per = ceil(chunks / 8)
  const int* start;
  const int* pxoff;
  const uint2* rec;              // sorted by pixel inside every chunk
  const float* table;
  int R, L;
};

constexpr int PG_PX = 64, PG_PITCH = 260;
constexpr int PG_PART = 8 * kChannels;              // floats: one partial row per wave

__device__ __forceinline__ void pg_issue(const float* __restrict__ table, unsigned ry, int g, int lane, float4 (&v)[8]) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)ry, g * 8 + u) & 0x3ffffffu;
    v[u] = *reinterpret_cast<const float4*>(table + (size_t)id * kChannels + lane * 4);
  }
}

template <bool NHWC>
__global__ __launch_bounds__(PG_THREADS) void pyramid_grad_reduce_kernel(const PgReduceParams p) {
  extern __shared__ __attribute__((aligned(16))) float s_tp[];   // [PG_PX][PG_PITCH] + [8][kChannels]
  __shared__ int s_head[8];                                       // pixel whose partial sum wave w left in part[w], or -1
  float* s_part = s_tp + PG_PX * PG_PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7;
  int j = blockIdx.x >> 3, l = p.L - 1, idx, chunk;
  if (p.chunk_order) {
    const int e = xcd * p.per + j;
    if (j >= p.per || e >= p.g.total) return;
    chunk = p.chunk_order[e];
    l = 0;
    while (l + 1 < p.L && chunk >= p.g.chunk_base[l + 1]) ++l;
    idx = chunk - p.g.chunk_base[l];
  } else {
    while (l > 0 && j >= p.per_xcd[l]) { j -= p.per_xcd[l]; --l; }
    const int n_l = p.g.chunk_base[l + 1] - p.g.chunk_base[l];
    idx = xcd * p.per_xcd[l] + j;
    if (j >= p.per_xcd[l] || idx >= n_l) return;
    chunk = p.g.chunk_base[l] + idx;
  }
  const int CWl = p.g.CW[l], CHl = p.g.CH[l], cws = p.g.cws[l], chs = p.g.chs[l], W = p.g.lvl_w[l], H = p.g.lvl_h[l];
  const int row = idx / (CHl * CWl);
  const int rem = idx - row * (CHl * CWl);
  const int cy = rem / CWl, cx = rem - cy * CWl;
  const int* po = p.pxoff + (size_t)chunk * 65;
  const uint2* rec = p.rec + p.start[chunk];
  const int npx = 1 << (cws + chs);
  // lane = pixel: its run [x0, x1) of the chunk's records; n = all of them; this wave's share [s, e)
  const int x0 = po[lane], x1 = po[lane + 1];
  const int n = __builtin_amdgcn_readlane(x1, 63);
  const int G = (n + 7) >> 3;
  const int s = 8 * ((G * wave) >> 3), e = min(n, 8 * ((G * (wave + 1)) >> 3));
  auto fetch = [&](int at) -> uint2 {
    uint2 r = rec[min(at + lane, max(e - 1, 0))];
    if (at + lane >= e) r.x = 0u;                                  // past the share: the last record again, weight 0
    return r;
  };
  // the first batch of records, and the record in front of the share (its pixel says whether the share starts inside a run)
  uint2 rc = make_uint2(0u, 0u);
  unsigned prev_y = 0u;
  if (s < e) {
    rc = fetch(s);
    if (s > 0) prev_y = rec[s - 1].y;
  }
  // a pixel without records gets its row of zeros from the wave whose share its position falls into (the last wave: the end)
  {
    const bool mine = lane < npx && x0 == x1 && ((x0 >= s && x0 < e) || (wave == 7 && x0 == n));
    unsigned long long m = __ballot(mine);
    while (m) {
      const int px = __ffsll((long long)m) - 1;
      m &= m - 1;
      *reinterpret_cast<float4*>(&s_tp[px * PG_PITCH + lane * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  int head = -1;
  if (s < e) {
    int cur = __builtin_amdgcn_readfirstlane((int)(rc.y >> 26));
    bool open = s > 0 && (int)((unsigned)__builtin_amdgcn_readfirstlane((int)prev_y) >> 26) == cur;   // (wave-uniform)
    if (open) head = cur;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&]() {
      float* dst = open ? s_part + wave * kChannels + lane * 4 : &s_tp[cur * PG_PITCH + lane * 4];
      *reinterpret_cast<float4*>(dst) = acc;
      open = false;
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto fma_group = [&](unsigned rx, unsigned ry, int g, const float4 (&v)[8]) {
      const int last = (int)((unsigned)__builtin_amdgcn_readlane((int)ry, g * 8 + 7) >> 26);
      if (last == cur) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float w = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)rx, g * 8 + u));
          acc.x = fmaf(w, v[u].x, acc.x); acc.y = fmaf(w, v[u].y, acc.y);
          acc.z = fmaf(w, v[u].z, acc.z); acc.w = fmaf(w, v[u].w, acc.w);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int px = (int)((unsigned)__builtin_amdgcn_readlane((int)ry, g * 8 + u) >> 26);
          if (px != cur) { flush(); cur = px; }
          const float w = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)rx, g * 8 + u));
          acc.x = fmaf(w, v[u].x, acc.x); acc.y = fmaf(w, v[u].y, acc.y);
          acc.z = fmaf(w, v[u].z, acc.z); acc.w = fmaf(w, v[u].w, acc.w);
        }
      }
    };
    float4 v0[8], v1[8];
    pg_issue(p.table, rc.y, 0, lane, v0);
    for (int bb = s; bb < e; bb += 64) {
      const bool more = bb + 64 < e;
      uint2 rn = rc;
      if (more) rn = fetch(bb + 64);
      const int ng = min(8, (e - bb + 7) >> 3);
#pragma unroll
      for (int g = 0; g < 8; g += 2) {
        if (g + 1 < ng) pg_issue(p.table, rc.y, g + 1, lane, v1);
        if (g < ng) fma_group(rc.x, rc.y, g, v0);
        if (g + 2 < ng) pg_issue(p.table, rc.y, g + 2, lane, v0);
        else if (g + 2 == 8 && more) pg_issue(p.table, rn.y, 0, lane, v0);     // the next batch's first group
        if (g + 1 < ng) fma_group(rc.x, rc.y, g + 1, v1);
      }
      rc = rn;
    }
    flush();
  }
  if (lane == 0) s_head[wave] = head;
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < 8; ++w) {
      const int px = s_head[w];
      if (px < 0) continue;
      float4 a = *reinterpret_cast<const float4*>(&s_tp[px * PG_PITCH + lane * 4]);
      const float4 b = *reinterpret_cast<const float4*>(s_part + w * kChannels + lane * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      *reinterpret_cast<float4*>(&s_tp[px * PG_PITCH + lane * 4]) = a;
    }
  }
  __syncthreads();
  if (NHWC) {
    // channels-last output (R, H, W, C): a pixel is one contiguous 1-KB row; wave w writes pixels w, w + 8, ...
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (int px = wave; px < npx; px += 8) {
      const int yy = (cy << chs) + (px >> cws), xx = (cx << cws) + (px & ((1 << cws) - 1));
      if (yy < H && xx < W) {
        const float4 v = *reinterpret_cast<const float4*>(&s_tp[px * PG_PITCH + lane * 4]);
        __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w},
                                    reinterpret_cast<f4v*>(p.out[l] + (((size_t)row * H + yy) * W + xx) * kChannels) + lane);
      }
    }
    return;
  }
  // wave w writes channels 32 w .. 32 w + 31; lane = pixel of the chunk (runs of cw pixels).  Non-temporal: kept in the L2
  // (plain stores) the written lines push out the table rows - the write-out alone is twice as fast, the kernel slower.
  const int py = lane >> cws, pxx = lane & ((1 << cws) - 1);
  const int y = (cy << chs) + py, x = (cx << cws) + pxx;
  if (py < (1 << chs) && y < H && x < W) {
    float* gp = p.out[l] + ((size_t)row * kChannels + wave * 32) * ((size_t)H * W) + (size_t)y * W + x;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) __builtin_nontemporal_store(s_tp[lane * PG_PITCH + wave * 32 + i], gp + (size_t)i * H * W);
  }
}

}  // namespace gd4d

extern "C" int gd4d_value_proj_heads_bwd(const float* grad_out, const float* weight, const float* bias, float* grad_agg,
                                         float* beta, int M, int Hh, int C, void* stream) {
  using namespace gd4d;
  if (!grad_out || !weight || !grad_agg || M <= 0) return GD4D_EINVAL;
  if (C != kChannels || (Hh != 4 && Hh != 8 && Hh != 16)) return GD4D_EUNSUPPORTED;
  const dim3 grid(((M + HB_ROWS - 1) / HB_ROWS) * Hh);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (kChannels / Hh) {
    case 64: hipLaunchKernelGGL(value_proj_heads_bwd_kernel<64>, grid, dim3(256), 0, s, grad_out, weight, bias, grad_agg, beta, M, Hh); break;
    case 32: hipLaunchKernelGGL(value_proj_heads_bwd_kernel<32>, grid, dim3(256), 0, s, grad_out, weight, bias, grad_agg, beta, M, Hh); break;
    case 16: hipLaunchKernelGGL(value_proj_heads_bwd_kernel<16>, grid, dim3(256), 0, s, grad_out, weight, bias, grad_agg, beta, M, Hh); break;
    default: return GD4D_EUNSUPPORTED;
  }
  return check_launch();
}

extern "C" size_t gd4d_value_proj_heads_bwd_weight_workspace_bytes(void) {
  return (size_t)gd4d::HW_KS * (gd4d::kChannels + 1) * gd4d::kChannels * sizeof(float);
}

extern "C" int gd4d_value_proj_heads_bwd_weight(const float* grad_out, const float* agg, const float* wsum, float* grad_weight,
                                                float* grad_bias, void* workspace, size_t workspace_bytes, int M, int Hh, int C,
                                                int accumulate, void* stream) {
  using namespace gd4d;
  if (!grad_out || !agg || !grad_weight || !workspace || M <= 0 || (grad_bias && !wsum)) return GD4D_EINVAL;
  if (C != kChannels || (Hh != 4 && Hh != 8 && Hh != 16)) return GD4D_EUNSUPPORTED;
  if (workspace_bytes < gd4d_value_proj_heads_bwd_weight_workspace_bytes()) return GD4D_EWORKSPACE;
  if (!aligned16(workspace)) return GD4D_EALIGN;
  float* part = static_cast<float*>(workspace);
  float* part_b = grad_bias ? part + (size_t)HW_KS * kChannels * kChannels : nullptr;
  const dim3 grid(Hh * (kChannels / 64) * HW_KS);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (kChannels / Hh) {
    case 64: hipLaunchKernelGGL(value_proj_heads_bwd_weight_kernel<64>, grid, dim3(256), 0, s, grad_out, agg, wsum, part, part_b, M, Hh); break;
    case 32: hipLaunchKernelGGL(value_proj_heads_bwd_weight_kernel<32>, grid, dim3(256), 0, s, grad_out, agg, wsum, part, part_b, M, Hh); break;
    case 16: hipLaunchKernelGGL(value_proj_heads_bwd_weight_kernel<16>, grid, dim3(256), 0, s, grad_out, agg, wsum, part, part_b, M, Hh); break;
    default: return GD4D_EUNSUPPORTED;
  }
  if (int rc = check_launch()) return rc;
  hipLaunchKernelGGL(value_proj_heads_bwd_weight_sum_kernel, dim3(kChannels * kChannels / 4 / 256), dim3(256), 0, s, part, part_b, grad_weight,
                     grad_bias, accumulate ? 1 : 0);
  return check_launch();
}

extern "C" int gd4d_value_proj_heads_bwd_weight_group(const void* const* grad_out, const void* const* agg, const void* const* wsum,
                                                      void* const* grad_weight, void* const* grad_bias, const int32_t* rows, int count,
                                                      void* workspace, size_t workspace_bytes, int Hh, int C, int accumulate, void* stream) {
  using namespace gd4d;
  if (!grad_out || !agg || !wsum || !grad_weight || !grad_bias || !rows || !workspace || count <= 0) return GD4D_EINVAL;
  if (count > HW_GROUP || C != kChannels || (Hh != 4 && Hh != 8 && Hh != 16)) return GD4D_EUNSUPPORTED;
  if (workspace_bytes < (size_t)count * gd4d_value_proj_heads_bwd_weight_workspace_bytes()) return GD4D_EWORKSPACE;
  if (!aligned16(workspace)) return GD4D_EALIGN;
  HwGroup q{};
  for (int i = 0; i < HW_GROUP; ++i) {
    const int j = i < count ? i : 0;
    if (i < count && (!grad_out[j] || !agg[j] || !grad_weight[j] || rows[j] <= 0 || (grad_bias[j] && !wsum[j]))) return GD4D_EINVAL;
    q.g[i] = static_cast<const float*>(grad_out[j]); q.agg[i] = static_cast<const float*>(agg[j]);
    q.wsum[i] = static_cast<const float*>(wsum[j]); q.gw[i] = static_cast<float*>(grad_weight[j]);
    q.gb[i] = static_cast<float*>(grad_bias[j]); q.M[i] = rows[j];
  }
  q.part = static_cast<float*>(workspace); q.count = count; q.accumulate = accumulate ? 1 : 0;
  const dim3 grid(Hh * (kChannels / 64) * HW_KS, count);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (kChannels / Hh) {
    case 64: hipLaunchKernelGGL(value_proj_heads_bwd_weight_group_kernel<64>, grid, dim3(256), 0, s, q, Hh); break;
    case 32: hipLaunchKernelGGL(value_proj_heads_bwd_weight_group_kernel<32>, grid, dim3(256), 0, s, q, Hh); break;
    case 16: hipLaunchKernelGGL(value_proj_heads_bwd_weight_group_kernel<16>, grid, dim3(256), 0, s, q, Hh); break;
    default: return GD4D_EUNSUPPORTED;
  }
  if (int rc = check_launch()) return rc;
  hipLaunchKernelGGL(value_proj_heads_bwd_weight_sum_group_kernel, dim3(kChannels * kChannels / 4 / 256, count), dim3(256), 0, s, q);
  return check_launch();
}

extern "C" size_t gd4d_cross_attn_dot_bytes(int B, int N, int Q, int Hh, int P) {
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || P <= 0) return 0;
  return (size_t)gd4d::kSlices * B * Q * Hh * gd4d::plan_cap_t(N, P) * 64 * sizeof(float);
}

static int dot_sliced_impl(const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan,
                          const float* grad_agg, void* dpart, size_t dpart_bytes, int B, int N, int Q, int Hh,
                          int C, int L, int P, int feats_dtype, const int32_t* query_order, void* stream, gd4d::WgradGuest* wg) {
  using namespace gd4d;
  if (!level_ptrs || !plan || !grad_agg || !dpart) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || L <= 0) return GD4D_EINVAL;
  if (C != kChannels || (P != kPoints && P != 8) || L > 4 || N > 64 || B > 16) return GD4D_EUNSUPPORTED;    // (P only sizes the plan)
  if (feats_dtype != GD4D_F32 && feats_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (Hh != 4 && Hh != 8 && Hh != 16) return GD4D_EUNSUPPORTED;
  const int es = feats_dtype == GD4D_BF16 ? 2 : 4;
  if (!aligned16(grad_agg) || !aligned16(plan) || slice_stride_bytes % (4 * es)) return GD4D_EALIGN;
  if (dpart_bytes < gd4d_cross_attn_dot_bytes(B, N, Q, Hh, P)) return GD4D_EWORKSPACE;
  DotParams p{};
  for (int l = 0; l < L; ++l) {
    if (!level_ptrs[l]) return GD4D_EINVAL;
    if (reinterpret_cast<uintptr_t>(level_ptrs[l]) % (4 * es)) return GD4D_EALIGN;
    p.lvl_base[l] = static_cast<const char*>(level_ptrs[l]);
  }
  for (int l = L; l < 4; ++l) p.lvl_base[l] = p.lvl_base[0];
  p.slice_stride = slice_stride_bytes;
  p.hdr = static_cast<const int*>(plan);
  p.pair = reinterpret_cast<const uint2*>(static_cast<const char*>(plan) + plan_hdr_bytes(B, Q));
  p.order = query_order; p.gagg = grad_agg;
  p.dpart = static_cast<float*>(dpart);
  p.BQ = B * Q; p.per_xcd = (B * Q + 7) / 8; p.cap_t = plan_cap_t(N, P);
  p.dslice = (long long)B * Q * Hh * p.cap_t * 64;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(8 * p.per_xcd * kSlices);
  if (wg) {                                              // queued weight gradients ride along
    if (Hh != 8 || L != 4 || feats_dtype != GD4D_F32) return GD4D_EUNSUPPORTED;
    wg->guest_groups = (wg->tiles + 7) / 8;
    wg->total_groups = (int)(grid.x / 8) + wg->guest_groups;
    wg->p = p;
    hipLaunchKernelGGL((cross_attn_dot_sliced_wgrad_kernel<8, 4, float, 6>), dim3(8 * wg->total_groups), dim3(64 * 8),
                       (size_t)8 * 6 * 8 * 80, s, *wg);
    return check_launch();
  }
  auto go = [&](auto kern, int hh) -> int {
    const size_t lds = (size_t)hh * 6 * 8 * 80;
    if (lds > 65536 && !allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(64 * hh), lds, s, p);
    return check_launch();
  };
#define GD4D_DOT_L(HH_, VT_, OCC_)                                                  \
  switch (L) {                                                                      \
    case 1: return go(cross_attn_dot_sliced_kernel<HH_, 1, VT_, OCC_>, HH_);        \
    case 2: return go(cross_attn_dot_sliced_kernel<HH_, 2, VT_, OCC_>, HH_);        \
    case 3: return go(cross_attn_dot_sliced_kernel<HH_, 3, VT_, OCC_>, HH_);        \
    default: return go(cross_attn_dot_sliced_kernel<HH_, 4, VT_, OCC_>, HH_);       \
  }
  const bool bf16 = feats_dtype == GD4D_BF16;
  switch (Hh) {
    case 4: if (bf16) { GD4D_DOT_L(4, uint16_t, 6) } else { GD4D_DOT_L(4, float, 6) }
    case 8: if (bf16) { GD4D_DOT_L(8, uint16_t, 6) } else { GD4D_DOT_L(8, float, 6) }
    default: if (bf16) { GD4D_DOT_L(16, uint16_t, 4) } else { GD4D_DOT_L(16, float, 4) }
  }
#undef GD4D_DOT_L
}

extern "C" int gd4d_cross_attn_dot_sliced(const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan,
                                          const float* grad_agg, void* dpart, size_t dpart_bytes, int B, int N, int Q, int Hh,
                                          int C, int L, int P, int feats_dtype, const int32_t* query_order, void* stream) {
  return dot_sliced_impl(level_ptrs, slice_stride_bytes, plan, grad_agg, dpart, dpart_bytes, B, N, Q, Hh, C, L, P, feats_dtype,
                         query_order, stream, nullptr);
}

extern "C" int gd4d_cross_attn_dot_sliced_wgrad(const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan,
                                                const float* grad_agg, void* dpart, size_t dpart_bytes, int B, int N, int Q, int Hh,
                                                int C, int L, int P, int feats_dtype, const int32_t* query_order,
                                                const void* const* x, const void* const* grad_y, void* const* grad_w,
                                                void* const* grad_b, const int32_t* dims, int count, int accumulate, void* stream) {
  using namespace gd4d;
  WgradGuest wg{};
  if (int rc = fill_lin_bwd_group(wg.g, wg.tiles, x, grad_y, grad_w, grad_b, dims, count, accumulate)) return rc;
  return dot_sliced_impl(level_ptrs, slice_stride_bytes, plan, grad_agg, dpart, dpart_bytes, B, N, Q, Hh, C, L, P, feats_dtype,
                         query_order, stream, &wg);
}

extern "C" int gd4d_cross_attn_plan_bwd(const float* ref, const float* offsets, const float* attn_logits, const float* cam_logits,
                                        const float* lidar2img, const double* pc_range, float img_h, float img_w,
                                        const int32_t* level_hw, const void* plan, const void* dpart, const float* beta,
                                        float* grad_ref, float* grad_offsets, float* grad_attn_logits, float* grad_cam_logits,
                                        void* workspace, size_t workspace_bytes, int32_t* status, int B, int N, int Q, int Hh, int L,
                                        int P, int flags, const int32_t* query_order, void* stream) {
  using namespace gd4d;
  if (!ref || !offsets || !attn_logits || !cam_logits || !lidar2img || !pc_range || !level_hw || !plan || !dpart || !grad_ref ||
      !grad_offsets || !grad_attn_logits || !grad_cam_logits)
    return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  if ((P != kPoints && !(P == 8 && Hh == 8)) || L > 4 || N > 64 || B > 16) return GD4D_EUNSUPPORTED;
  if (Hh != 4 && Hh != 8 && Hh != 16) return GD4D_EUNSUPPORTED;
  const size_t need = B > 1 ? (size_t)B * B * Q * Hh * L * P * sizeof(float) : 0;
  if (B > 1 && (!workspace || workspace_bytes < need)) return GD4D_EWORKSPACE;
  PlanBwdParams pp{};
  CrossAttnParams& p = pp.c;
  p.ref = ref; p.offsets = offsets; p.attn_logits = attn_logits; p.cam_logits = cam_logits; p.lidar2img = lidar2img;
  p.order = query_order;
  p.B = B; p.N = N; p.Q = Q; p.L = L; p.P = P;
  p.raw_cam = (flags & GD4D_CA_RAW_CAM_WEIGHTS) ? 1 : 0;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  for (int l = 0; l < 4; ++l) { pp.lvl_w[l] = 1; pp.lvl_h[l] = 1; }
  for (int l = 0; l < L; ++l) {
    if (level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    pp.lvl_h[l] = level_hw[2 * l]; pp.lvl_w[l] = level_hw[2 * l + 1];
  }
  pp.hdr = static_cast<const int*>(plan);
  pp.dpart = static_cast<const float*>(dpart);
  pp.cap_t = plan_cap_t(N, P);
  pp.dslice = (long long)B * Q * Hh * pp.cap_t * 64;
  pp.beta = beta;
  pp.grad_ref = grad_ref; pp.grad_offsets = grad_offsets; pp.grad_attn_logits = grad_attn_logits;
  pp.grad_cam_logits = grad_cam_logits; pp.ga_part = static_cast<float*>(workspace); pp.status = status;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ncand = N * P;
  auto lds = [&](int LT, int waves) {
    return (size_t)N * Hh * P * sizeof(float2) + (size_t)N * 12 * sizeof(float) + (size_t)((N + 3) & ~3) * sizeof(float) +
           (size_t)((B * Hh * LT * P + 3) & ~3) * sizeof(float) + (size_t)Hh * P * 4 * sizeof(float) +
           (size_t)waves * ncand * (sizeof(float4) + 8 * sizeof(float)) + (size_t)Hh * ncand * sizeof(float) + (size_t)Hh * 4 * sizeof(float);
  };
  auto go = [&](auto kern, int waves, size_t bytes) -> int {
    if (bytes > 160 * 1024) return GD4D_EUNSUPPORTED;
    if (bytes > 65536 && !allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(B * Q), dim3(64 * waves), bytes, s, pp);
    return check_launch();
  };
  int rc;
#define GD4D_PB_GO(HH_, LT_, PT_)                                                                                              \
  rc = B > 1 ? go(cross_attn_plan_bwd_kernel<HH_, LT_, (HH_ < 8 ? HH_ : 8), true, PT_>, (HH_ < 8 ? HH_ : 8), lds(LT_, (HH_ < 8 ? HH_ : 8)))   \
             : go(cross_attn_plan_bwd_kernel<HH_, LT_, (HH_ < 8 ? HH_ : 8), false, PT_>, (HH_ < 8 ? HH_ : 8), lds(LT_, (HH_ < 8 ? HH_ : 8))); \
  if (rc == GD4D_OK && B > 1) {                                                                                                \
    hipLaunchKernelGGL((plan_bwd_logits_kernel<LT_ * PT_>), dim3(B * Q), dim3(64), 0, s, pp.ga_part, attn_logits,              \
                       grad_attn_logits, B, Q, Hh);                                                                            \
    rc = check_launch();                                                                                                       \
  }                                                                                                                            \
  return rc;
#define GD4D_PB_L(HH_, PT_)                \
  switch (L) {                             \
    case 1: GD4D_PB_GO(HH_, 1, PT_)        \
    case 2: GD4D_PB_GO(HH_, 2, PT_)        \
    case 3: GD4D_PB_GO(HH_, 3, PT_)        \
    default: GD4D_PB_GO(HH_, 4, PT_)       \
  }
  if (P == 8) { GD4D_PB_L(8, 8) }
  switch (Hh) {
    case 4: GD4D_PB_L(4, 4)
    case 8: GD4D_PB_L(8, 4)
    default: GD4D_PB_L(16, 4)
  }
#undef GD4D_PB_L
#undef GD4D_PB_GO
}

extern "C" int64_t gd4d_pyramid_grad_chunks(const int32_t* level_hw, int R, int L) {
  gd4d::PgChunks g{};
  if (!level_hw || gd4d::fill_chunks(g, level_hw, nullptr, 0, R, L) != GD4D_OK) return 0;
  return g.total;
}

extern "C" size_t gd4d_pyramid_grad_slots_bytes(int B, int N, int Q, int Hh, int P) {
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || P <= 0) return 0;
  return (size_t)B * Q * Hh * gd4d::plan_cap_t(N, P) * 64 * sizeof(uint2);
}

extern "C" int gd4d_pyramid_grad_count(const void* plan, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                                       int64_t pix_stride_bytes, int32_t* count, void* slots, size_t slots_bytes, int B, int N,
                                       int Q, int Hh, int L, int P, void* stream) {
  using namespace gd4d;
  if (!plan || !level_hw || !cam_stride_bytes || !count || !slots) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || L <= 0) return GD4D_EINVAL;
  if ((P != kPoints && P != 8) || L > 4 || N > 64 || B > 16 || Hh > kPlanHdr) return GD4D_EUNSUPPORTED;
  if (slots_bytes < gd4d_pyramid_grad_slots_bytes(B, N, Q, Hh, P)) return GD4D_EWORKSPACE;
  PgChunks g{};
  if (int rc = fill_chunks(g, level_hw, cam_stride_bytes, pix_stride_bytes, B * N, L)) return rc;
  const int* hdr = static_cast<const int*>(plan);
  const uint2* pair = reinterpret_cast<const uint2*>(static_cast<const char*>(plan) + plan_hdr_bytes(B, Q));
  hipLaunchKernelGGL(pyramid_grad_count_kernel, dim3((B * Q * Hh + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), hdr, pair,
                     plan_cap_t(N, P), Hh, B * Q, g, count, static_cast<uint2*>(slots));
  return check_launch();
}

extern "C" int gd4d_pyramid_grad_fill(const void* plan, const void* slots, const int32_t* start, void* records, uint32_t id_base,
                                      const int32_t* query_order, int B, int N, int Q, int Hh, int P, void* stream) {
  using namespace gd4d;
  if (!plan || !slots || !start || !records) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0) return GD4D_EINVAL;
  if ((P != kPoints && P != 8) || N > 64 || B > 16 || Hh > kPlanHdr) return GD4D_EUNSUPPORTED;
  if ((unsigned long long)id_base + (unsigned long long)B * Q * Hh > (1ull << 26)) return GD4D_EUNSUPPORTED;
  const int* hdr = static_cast<const int*>(plan);
  const uint2* pair = reinterpret_cast<const uint2*>(static_cast<const char*>(plan) + plan_hdr_bytes(B, Q));
  hipLaunchKernelGGL(pyramid_grad_fill_kernel, dim3((B * Q * Hh + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), hdr, pair,
                     static_cast<const uint2*>(slots), plan_cap_t(N, P), Hh, B * Q, start, static_cast<uint2*>(records), query_order,
                     id_base);
  return check_launch();
}

extern "C" size_t gd4d_pyramid_grad_scan_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (size_t)((n + gd4d::SCAN_CHUNK - 1) / gd4d::SCAN_CHUNK) * sizeof(int);
}

extern "C" int gd4d_pyramid_grad_scan(const int32_t* count, int32_t* cursor, void* workspace, size_t workspace_bytes, int64_t n,
                                      void* stream) {
  using namespace gd4d;
  if (!count || !cursor || !workspace || n <= 0) return GD4D_EINVAL;
  if (n >= (1ll << 31)) return GD4D_EUNSUPPORTED;
  if (workspace_bytes < gd4d_pyramid_grad_scan_workspace_bytes(n)) return GD4D_EWORKSPACE;
  const int nb = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  int* bsum = static_cast<int*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pg_scan_sums_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, count, (long long)n, bsum);
  if (int rc = check_launch()) return rc;
  hipLaunchKernelGGL(pg_scan_bsums_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, bsum, nb);
  if (int rc = check_launch()) return rc;
  hipLaunchKernelGGL(pg_scan_apply_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, count, (long long)n, bsum, cursor);
  return check_launch();
}

extern "C" int gd4d_pyramid_grad_chunk_geometry(const int32_t* level_hw, int R, int L, int32_t* out) {
  gd4d::PgChunks g{};
  if (!level_hw || !out) return GD4D_EINVAL;
  if (int rc = gd4d::fill_chunks(g, level_hw, nullptr, 0, R, L)) return rc;
  for (int l = 0; l < L; ++l) {
    out[5 * l] = g.cws[l]; out[5 * l + 1] = g.chs[l]; out[5 * l + 2] = g.CW[l]; out[5 * l + 3] = g.CH[l]; out[5 * l + 4] = g.chunk_base[l];
  }
  return GD4D_OK;
}

extern "C" int gd4d_pyramid_grad_sort(const int32_t* count, const int32_t* start, const void* records, void* sorted,
                                      int32_t* pxoff, int64_t chunks, void* stream) {
  using namespace gd4d;
  if (!count || !start || !records || !sorted || !pxoff || chunks <= 0) return GD4D_EINVAL;
  if (chunks >= (1ll << 25)) return GD4D_EUNSUPPORTED;
  hipLaunchKernelGGL(pyramid_grad_sort_kernel, dim3((unsigned)chunks), dim3(PG_THREADS), 0, static_cast<hipStream_t>(stream), count, start,
                     static_cast<const uint2*>(records), static_cast<uint2*>(sorted), pxoff);
  return check_launch();
}

extern "C" int gd4d_pyramid_grad_reduce(const int32_t* start, const int32_t* pxoff, const void* sorted, const float* table,
                                        void* const* grads, const int32_t* level_hw, const int32_t* chunk_order, int R, int C, int L,
                                        int channels_last, void* stream) {
  using namespace gd4d;
  if (!start || !pxoff || !sorted || !table || !grads || !level_hw || R <= 0 || L <= 0) return GD4D_EINVAL;
  if (C != kChannels || L > GD4D_MAX_LEVELS) return GD4D_EUNSUPPORTED;
  if (!aligned16(table)) return GD4D_EALIGN;
  PgReduceParams p{};
  if (int rc = fill_chunks(p.g, level_hw, nullptr, 0, R, L)) return rc;
  int per = 0;
  for (int l = 0; l < L; ++l) {
    if (!grads[l]) return GD4D_EINVAL;
    p.out[l] = static_cast<float*>(grads[l]);
    p.per_xcd[l] = (p.g.chunk_base[l + 1] - p.g.chunk_base[l] + 7) / 8;
    per += p.per_xcd[l];
  }
  p.start = start; p.pxoff = pxoff; p.rec = static_cast<const uint2*>(sorted); p.table = table;
  p.R = R; p.L = L;
  p.chunk_order = chunk_order;
  p.per = (p.g.total + 7) / 8;
  if (chunk_order) per = p.per;
  const size_t lds = ((size_t)PG_PX * PG_PITCH + PG_PART) * sizeof(float);
  auto go = [&](auto kern) -> int {
    if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(8 * per), dim3(PG_THREADS), lds, static_cast<hipStream_t>(stream), p);
    return check_launch();
  };
  if (channels_last) {
    for (int l = 0; l < L; ++l)
      if (!aligned16(grads[l])) return GD4D_EALIGN;
    return go(pyramid_grad_reduce_kernel<true>);
  }
  return go(pyramid_grad_reduce_kernel<false>);
}

// Aggregate-then-project form of Deform3DCrossAttn's value path (inference, gfx950).
//
// Reference: Deform3DCrossAttn.forward, deform3d_cross_attn.py:264-280 (flatten + transpose + cat + value_proj over the
// whole pyramid, once per decoder layer), :281-284, :301-304 (mmcv MSDA gather), :320-324 (camera-weighted sum).
//
// value_proj is LINEAR and the gather is a weighted sum of value rows, so for head h of query q
//
//   out[q, h*Dh + d] = sum_i w_i * (W_h x_i + b_h)[d]  =  ( W_h * (sum_i w_i x_i) )[d]  +  b_h[d] * sum_i w_i
//
// with i running over the in-bounds bilinear corners of every visible (camera, level, point) sample of that head, x_i the
// RAW 256-channel feature vector of the corner's pixel and w_i = softmax weight * camera weight * bilinear weight
// (out-of-map corners contribute nothing - mmcv's zero padding acts on the projected value, bias included, hence the
// b * sum of IN-BOUNDS weights).  Projecting the aggregate costs Q * Hh * Dh * C MACs per layer (59 MFLOP at 900 queries)
// instead of projecting all 739 800 pixel rows (97 GFLOP, 757 MB written) to read back 17 % of them: the six per-layer
// value tensors (4.5 GB) are never materialised.  The price is a gather of C instead of Dh channels per corner from a
// channels-last copy of the pyramid, which is made once per sample for all layers:
//
//   gd4d_pyramid_channels_last_fwd  NCHW levels -> (R, S, C)      (the reference's own flatten/transpose/cat, :264-276)
//   gd4d_cross_attn_agg_fwd         projection + mask + weights + gather of raw features -> agg (B*Q, Hh, C), wsum (B*Q, Hh)
//   gd4d_value_proj_heads_fwd       out = W_h agg_h + b_h wsum_h  (exact fp32 MFMA) -> (B*Q, Hh*Dh), the input of output_proj
//
// Same visibility mask and image coordinates as gd4d_cross_attn_fwd (shared project_entry, -ffp-contract=off).
#include <stdlib.h>

#include "gd4d_common.h"
#include "gd4d_cross_attn_shared.h"

namespace gd4d {

// ---------------------------------------------------------------------------------------------------------------
// NCHW -> channels-last.  One workgroup = 32 pixels x 256 channels of one (camera row, level): 128-byte pixel runs in,
// one contiguous 32-KB block out, turned through LDS (pitch 260 floats: the dword writes of a half-wave are 2-way
// conflicted at most, which costs nothing on ds_write_b32; the float4 reads are aligned).
GD4D_TRACE_UNIT(late)

struct ClParams {
  const float* in[GD4D_MAX_LEVELS];    // level l: (R, C, HW_l) fp32
  int hw[GD4D_MAX_LEVELS];
  int start[GD4D_MAX_LEVELS];
  int tiles[GD4D_MAX_LEVELS];
  int tile_base[GD4D_MAX_LEVELS + 1];  // prefix over levels of R * tiles[l]
  void* out;                           // (R, S, C) fp32 or bf16
  int R, L, S, total;
};

constexpr int CL_PX = 32, CL_C = 256, CL_PITCH = 260;

// stores 4 consecutive channels of one pixel: 16 bytes fp32 or 8 bytes bf16 (round to nearest even)
template <bool OUT_BF16>
__device__ __forceinline__ void cl_store4(void* base, size_t elem, float4 v) {
  if (OUT_BF16) {
    uint2 pk;
    pk.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
    pk.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(base) + elem) = pk;
  } else {
    *reinterpret_cast<float4*>(static_cast<float*>(base) + elem) = v;
  }
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void pyramid_channels_last_kernel(const ClParams p) {
  __shared__ __attribute__((aligned(16))) float s_t[CL_PX * CL_PITCH];
  const int t = blockIdx.x;
  const float* src = p.in[0];
  int hw = p.hw[0], ostart = p.start[0], tiles = p.tiles[0], tbase = 0;
#pragma unroll
  for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
    if (l < p.L && t >= p.tile_base[l]) { src = p.in[l]; hw = p.hw[l]; ostart = p.start[l]; tiles = p.tiles[l]; tbase = p.tile_base[l]; }
  const int rel = t - tbase;
  const int row = rel / tiles;
  const int pix0 = (rel - row * tiles) * CL_PX;
  const int npx = min(CL_PX, hw - pix0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // reads: a wave instruction covers 2 channels x 32 pixels
  const int px = lane & 31, ch_lo = lane >> 5;
  const float* gp = src + ((size_t)row * CL_C) * hw + pix0 + min(px, npx - 1);
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = wave * 64 + 2 * i + ch_lo;
    v[i] = gp[(size_t)c * hw];
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) s_t[px * CL_PITCH + wave * 64 + 2 * i + ch_lo] = v[i];
  __syncthreads();
  const size_t obase = ((size_t)row * p.S + ostart + pix0) * CL_C;
#pragma unroll
  for (int i = 0; i < CL_PX / 4; ++i) {
    const int q = wave + 4 * i;                                  // pixel of this wave instruction: 1 KB (512 B) contiguous
    if (q < npx)
      cl_store4<OUT_BF16>(p.out, obase + (size_t)q * CL_C + lane * 4, *reinterpret_cast<const float4*>(&s_t[q * CL_PITCH + lane * 4]));
  }
}

// The same copy as ONE persistent 512-thread workgroup per compute unit on `max_cus` of them (its LDS request is padded
// past half of the CU's 160 KB so that no second one fits): the compute units it leaves alone stay completely free -
// LDS included - for the first decoder layer's self-attention and row chains (132 KB of LDS per workgroup), which need
// nothing from the pyramid and run underneath the copy.  Tiles of 64 pixels; the loads of the next tile are in flight
// while the current one is written out.
constexpr int CLP_PX = 64, CLP_THREADS = 512;
constexpr int CLP_LDS = 84 * 1024;                              // > 80 KB: one workgroup per CU

template <bool OUT_BF16>
__global__ __launch_bounds__(CLP_THREADS) void pyramid_channels_last_persistent_kernel(const ClParams p) {
  extern __shared__ __attribute__((aligned(16))) float s_tp[];  // [CLP_PX][CL_PITCH] (+ padding)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // 8 waves; wave w moves channels 32 w .. 32 w + 31
  trace_mark(g_trace_late, 4ull);
  float v[32];
  int npx = 0, ostart = 0, pix0 = 0, row = 0;
  auto locate = [&](int t, const float*& src, int& hw) {
    src = p.in[0];
    hw = p.hw[0];
    int tiles = p.tiles[0], tbase = 0, os = p.start[0];
#pragma unroll
    for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
      if (l < p.L && t >= p.tile_base[l]) { src = p.in[l]; hw = p.hw[l]; os = p.start[l]; tiles = p.tiles[l]; tbase = p.tile_base[l]; }
    const int rel = t - tbase;
    row = rel / tiles;
    pix0 = (rel - row * tiles) * CLP_PX;
    npx = min(CLP_PX, hw - pix0);
    ostart = os;
  };
  auto load_tile = [&](int t) {
    const float* src;
    int hw;
    locate(t, src, hw);
    const float* gp = src + ((size_t)row * CL_C + wave * 32) * hw + pix0 + min(lane, npx - 1);
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = gp[(size_t)i * hw];
  };
  int t = blockIdx.x;
  if (t < p.total) load_tile(t);
  while (t < p.total) {
    const int c_npx = npx, c_ostart = ostart, c_pix0 = pix0, c_row = row;
#pragma unroll
    for (int i = 0; i < 32; ++i) s_tp[lane * CL_PITCH + wave * 32 + i] = v[i];
    __syncthreads();
    const int tn = t + gridDim.x;
    if (tn < p.total) load_tile(tn);                             // in flight during the store phase
    const size_t obase = ((size_t)c_row * p.S + c_ostart + c_pix0) * CL_C;
#pragma unroll
    for (int i = 0; i < CLP_PX / 8; ++i) {
      const int q = wave + 8 * i;
      if (q < c_npx)
        cl_store4<OUT_BF16>(p.out, obase + (size_t)q * CL_C + lane * 4, *reinterpret_cast<const float4*>(&s_tp[q * CL_PITCH + lane * 4]));
    }
    __syncthreads();
    t = tn;
  }
  trace_mark(g_trace_late, 0x84ull);
}

// ---------------------------------------------------------------------------------------------------------------
// Gather of the raw features.  One workgroup (4 waves) per (batch, query); phase A (projection, mask, softmax, camera
// weights) is gd4d_cross_attn_fwd's.  Phase B: every head belongs to ONE wave (head h -> wave h % 4), which walks the
// head's compacted list of visible (camera, point) items.  A corner of an item is one 1-KB wave load (lane c4 takes
// channels [4 c4, 4 c4 + 4) of the corner's pixel), so the lanes of a wave share every address and weight: they are
// computed SIMD-fashion - lane (item % 4, level, corner) for four items at a time - and broadcast with v_readlane into
// scalar registers; the loads use them as scalar offsets, the FMAs as scalar factors.  No cross-wave reduction, a fixed
// summation order (deterministic), and only visible points are touched.
template <int HH, int LT, bool LEVEL_MAJOR, int WAVES, typename VT = float>
__global__ __launch_bounds__(64 * WAVES) void cross_attn_agg_kernel(const CrossAttnParams p) {
  constexpr int ES = sizeof(VT);                                // bytes per stored channel (4: fp32, 2: bf16)
  constexpr int PT = kPoints;
  constexpr int E = HH * PT;
  constexpr int LP = LT * PT;
  constexpr int THREADS = GD4D_WAVE * WAVES;
  constexpr int HPW = (HH + WAVES - 1) / WAVES;               // heads per wave
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* s_uv = reinterpret_cast<float2*>(smem_raw);                        // [N][E]; x < 0: not visible
  float* s_aw = reinterpret_cast<float*>(s_uv + p.N * E);                    // [HH][LP] softmax weights
  float* s_mat = s_aw + HH * LP;                                             // [N][12]
  float* s_cw = s_mat + p.N * 12;                                            // [N] camera weights
  int* s_cnt = reinterpret_cast<int*>(s_cw + ((p.N + 3) & ~3));              // [HH] visible items per head
  uint8_t* s_items = reinterpret_cast<uint8_t*>(s_cnt + HH);                 // [HH][N * PT]: camera * PT + point

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  trace_mark(g_trace_late, 3ull);
  int bq = blockIdx.x;
  if (p.order) {                                   // XCD-contiguous ranges of the locality order (see gd4d_cross_attn_fwd)
    const int per_xcd = (p.B * p.Q + 7) >> 3;
    const int pos = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || pos >= p.B * p.Q) return;
    bq = p.order[pos];
  } else if (bq >= p.B * p.Q) {
    return;
  }
  const int b = bq / p.Q;
  const int q = bq - b * p.Q;

  // ---------------- phase A ----------------
  for (int i = tid; i < p.N * 12; i += THREADS) s_mat[i] = p.lidar2img[((size_t)b * p.N + i / 12) * 16 + i % 12];
  if (tid < p.N) {
    const float cl = p.cam_logits[(size_t)b * p.Q * p.N + (size_t)tid * p.Q + q];   // raw-view scramble (:211-212)
    s_cw[tid] = p.raw_cam ? cl : 1.0f / (1.0f + expf(-cl));
  }
  if ((LP & (LP - 1)) == 0 && LP <= 32) {          // softmax over L * P logits per head, one thread per logit
    if (tid < HH * LP) {
      const int hd = tid / LP, i = tid % LP;
      const float x = p.attn_logits[((size_t)bq * HH + hd) * LP + i];
      float mx = x;
#pragma unroll
      for (int o = 1; o < LP; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float e = expf(x - mx);
      float sum = e;
#pragma unroll
      for (int o = 1; o < LP; o <<= 1) sum += __shfl_xor(sum, o);
      s_aw[hd * LP + i] = e * (1.0f / sum);
    }
  } else if (tid < HH) {
    float w[LP];
    softmax_lp(p.attn_logits + ((size_t)bq * HH + tid) * LP, LP, w);
    for (int i = 0; i < LP; ++i) s_aw[tid * LP + i] = w[i];
  }
  __syncthreads();
  {
    const int total = p.N * E;
    static_assert(E <= GD4D_WAVE && GD4D_WAVE % E == 0 && THREADS % E == 0, "a thread keeps its (head, point)");
    const int hp = tid % E;
    const float* rp = p.ref + (size_t)bq * 3;
    const float* offs = p.offsets + ((size_t)bq * E + hp) * 3;
    const float X = (rp[0] * p.rng_scale[0] + p.rng_lo[0]) + offs[0];     // two roundings, then the offset (:222-229)
    const float Y = (rp[1] * p.rng_scale[1] + p.rng_lo[1]) + offs[1];
    const float Z = (rp[2] * p.rng_scale[2] + p.rng_lo[2]) + offs[2];
    for (int e0 = wave * GD4D_WAVE; e0 < total; e0 += THREADS) {
      const int e = e0 + lane;
      if (e < total) {
        const int n = e / E;
        float u, v;
        const bool vis = project_entry(p, s_mat + n * 12, X, Y, Z, u, v);
        s_uv[e] = vis ? make_float2(u, v) : make_float2(-1.f, -1.f);
        const size_t o = (((size_t)b * p.N + n) * p.Q + q) * E + hp;
        if (p.mask_out) p.mask_out[o] = vis ? 1 : 0;
        if (p.uv_out) { p.uv_out[o * 2] = u; p.uv_out[o * 2 + 1] = v; }
      }
    }
  }
  __syncthreads();

  // ---------------- item lists: wave w compacts the visible (camera, point) pairs of its heads ----------------
  const int ncand = p.N * PT;
#pragma unroll
  for (int hi = 0; hi < HPW; ++hi) {
    const int h = wave + hi * WAVES;
    if (h >= HH) break;
    int base = 0;
    for (int c0 = 0; c0 < ncand; c0 += GD4D_WAVE) {
      const int cand = c0 + lane;
      const bool vis = cand < ncand && s_uv[(cand / PT) * E + h * PT + (cand % PT)].x >= 0.f;
      const unsigned long long bal = __ballot(vis);
      if (vis) s_items[h * ncand + base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)cand;
      base += __popcll(bal);
    }
    if (lane == 0) s_cnt[h] = base;
  }
  __syncthreads();

  // ---------------- phase B ----------------
  if (LEVEL_MAJOR) {
    // Level-major: every wave of every (co-resident) workgroup walks the pyramid coarse -> fine, so the small, heavily
    // re-read maps (level 3: 9 MB for 24 cameras, level 2: 36 MB) are gathered while they fit the XCD's 4-MB L2 instead of
    // being evicted by the level-0 stream (332 MB of nearly unique lines) between two uses.
    // Lane (item % 16, corner): 16 items x 4 corners of ONE level per set-up; a load round is 4 items x 4 corners.
    const int it_of = lane >> 2, c_of = lane & 3;
    const char* vbase = static_cast<const char*>(p.value) + lane * 4 * ES;
    float4 acc[HPW];
    float wsum_lane[HPW];
#pragma unroll
    for (int hi = 0; hi < HPW; ++hi) { acc[hi] = make_float4(0.f, 0.f, 0.f, 0.f); wsum_lane[hi] = 0.f; }
#pragma unroll
    for (int li = 0; li < LT; ++li) {
      const int l = LT - 1 - li;
      const int lw = p.lvl_w[l], lh = p.lvl_h[l], ls = p.lvl_start[l];
#pragma unroll
      for (int hi = 0; hi < HPW; ++hi) {
        const int h = wave + hi * WAVES;
        if (h >= HH) break;
        const int M = __builtin_amdgcn_readfirstlane(s_cnt[h]);
        for (int it0 = 0; it0 < M; it0 += 16) {
          const int item = it0 + it_of;
          const bool valid = item < M;
          const int cand = s_items[h * ncand + min(item, M - 1)];
          const int n = cand / PT, k = cand % PT;
          const float2 uv = s_uv[n * E + h * PT + k];
          const float x = fmaf(uv.x, (float)lw, -0.5f);
          const float y = fmaf(uv.y, (float)lh, -0.5f);
          const float xf = floorf(x), yf = floorf(y);
          const float dx = x - xf, dy = y - yf;
          const int xi = (int)xf + (c_of & 1), yi = (int)yf + (c_of >> 1);
          const bool ok = valid && xi >= 0 && xi < lw && yi >= 0 && yi < lh;
          const float wl = s_aw[h * LP + l * PT + k] * s_cw[n];
          const float wx = (c_of & 1) ? dx : 1.f - dx, wy = (c_of >> 1) ? dy : 1.f - dy;
          const float w = ok ? wl * wx * wy : 0.f;
          const int xc = min(max(xi, 0), lw - 1), yc = min(max(yi, 0), lh - 1);
          const unsigned pixel = valid ? (unsigned)((b * p.N + n) * p.S + ls + yc * lw + xc) : 0u;
          wsum_lane[hi] += w;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (it0 + 4 * s >= M) break;                       // wave-uniform
            float4 val[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              unsigned px = (unsigned)__builtin_amdgcn_readlane((int)pixel, s * 16 + j);
              val[j] = Quad<VT>::load(reinterpret_cast<const VT*>(vbase + (size_t)px * (kChannels * ES)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const float wj = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(w), s * 16 + j));
              acc[hi].x = fmaf(wj, val[j].x, acc[hi].x); acc[hi].y = fmaf(wj, val[j].y, acc[hi].y);
              acc[hi].z = fmaf(wj, val[j].z, acc[hi].z); acc[hi].w = fmaf(wj, val[j].w, acc[hi].w);
            }
          }
        }
      }
    }
#pragma unroll
    for (int hi = 0; hi < HPW; ++hi) {
      const int h = wave + hi * WAVES;
      if (h >= HH) break;
      if (p.agg) *reinterpret_cast<float4*>(p.agg + ((size_t)bq * HH + h) * kChannels + lane * 4) = acc[hi];
      float t = wsum_lane[hi];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (lane == 0 && p.wsum) p.wsum[(size_t)bq * HH + h] = t;
    }
    return;
  }
  // item-major (the first form; GD4D_AGG_VARIANT=0): lane (item % 4, level, corner), all levels of an item per load round
  const int sub = lane >> 4, l_of = (lane >> 2) & 3, c_of = lane & 3;
  int lw = p.lvl_w[0], lh = p.lvl_h[0], ls = p.lvl_start[0];
#pragma unroll
  for (int l = 1; l < LT; ++l)
    if (l_of == l) { lw = p.lvl_w[l]; lh = p.lvl_h[l]; ls = p.lvl_start[l]; }
  const char* vbase = static_cast<const char*>(p.value) + lane * 4 * ES;     // + pixel * (1024 or 512)
#pragma unroll
  for (int hi = 0; hi < HPW; ++hi) {
    const int h = wave + hi * WAVES;
    if (h >= HH) break;
    const int M = __builtin_amdgcn_readfirstlane(s_cnt[h]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float wsum_lane = 0.f;
    for (int it0 = 0; it0 < M; it0 += 4) {
      // this lane's (item, level, corner): pixel index and weight
      const int item = it0 + sub;
      const bool valid = item < M && l_of < LT;
      const int cand = s_items[h * ncand + min(item, M - 1)];
      const int n = cand / PT, k = cand % PT;
      const float2 uv = s_uv[n * E + h * PT + k];
      const float x = fmaf(uv.x, (float)lw, -0.5f);
      const float y = fmaf(uv.y, (float)lh, -0.5f);
      const float xf = floorf(x), yf = floorf(y);
      const float dx = x - xf, dy = y - yf;
      const int xi = (int)xf + (c_of & 1), yi = (int)yf + (c_of >> 1);
      const bool ok = valid && xi >= 0 && xi < lw && yi >= 0 && yi < lh;
      const float wl = s_aw[h * LP + min(l_of, LT - 1) * PT + k] * s_cw[n];
      const float wx = (c_of & 1) ? dx : 1.f - dx, wy = (c_of >> 1) ? dy : 1.f - dy;
      const float w = ok ? wl * wx * wy : 0.f;
      // corners outside the map contribute 0 (zero padding); their loads are clamped onto the map
      const int xc = min(max(xi, 0), lw - 1), yc = min(max(yi, 0), lh - 1);
      const unsigned pixel = valid ? (unsigned)((b * p.N + n) * p.S + ls + yc * lw + xc) : 0u;
      wsum_lane += w;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (it0 + s >= M) break;                               // wave-uniform
        float4 val[LT * 4];
#pragma unroll
        for (int j = 0; j < LT * 4; ++j) {
          unsigned px = (unsigned)__builtin_amdgcn_readlane((int)pixel, s * 16 + j);
          val[j] = Quad<VT>::load(reinterpret_cast<const VT*>(vbase + (size_t)px * (kChannels * ES)));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < LT * 4; ++j) {
          const float wj = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(w), s * 16 + j));
          acc.x = fmaf(wj, val[j].x, acc.x); acc.y = fmaf(wj, val[j].y, acc.y);
          acc.z = fmaf(wj, val[j].z, acc.z); acc.w = fmaf(wj, val[j].w, acc.w);
        }
      }
    }
    if (p.agg) *reinterpret_cast<float4*>(p.agg + ((size_t)bq * HH + h) * kChannels + lane * 4) = acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum_lane += __shfl_xor(wsum_lane, o);
    if (lane == 0 && p.wsum) p.wsum[(size_t)bq * HH + h] = wsum_lane;
    if (p.vp_w) {
      // value_proj of this head's aggregate in the epilogue (exact fp32): out[h*Dh + d] = <W[h*Dh + d, :], agg> + b wsum.
      // A lane holds 4 of the 256 channels, so every output is a 4-term partial per lane and a sum over the wave; the
      // Dh sums are taken together by a butterfly (each exchange halves the values a lane keeps: 16 + 8 + 4 + 2 + 1
      // shuffles for 32 outputs instead of 32 x 6).  The rows of W are 1-KB wave loads from the L2 (every workgroup
      // reads the same 256 KB).  The other workgroups keep the memory system busy meanwhile, and the row chain that
      // follows starts with a 1-KB row per query instead of gathering 8 KB of aggregates per query cold.
      constexpr int DH = kChannels / HH;
      constexpr int NB = DH < 32 ? DH : 32;             // outputs per butterfly
      const float* wrow = p.vp_w + (size_t)(h * DH) * kChannels + lane * 4;
#pragma unroll
      for (int d0 = 0; d0 < DH; d0 += NB) {
        float v[NB];
#pragma unroll
        for (int r0 = 0; r0 < NB; r0 += 8) {
          float4 wv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[j] = *reinterpret_cast<const float4*>(wrow + (size_t)(d0 + r0 + j) * kChannels);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[r0 + j] = fmaf(wv[j].x, acc.x, fmaf(wv[j].y, acc.y, fmaf(wv[j].z, acc.z, wv[j].w * acc.w)));
        }
        int n = NB, mask = 32;
#pragma unroll
        for (; n > 1; n >>= 1, mask >>= 1) {
          const bool upper = (lane & mask) != 0;
#pragma unroll
          for (int j = 0; j < n / 2; ++j) {
            const float send = upper ? v[j] : v[j + n / 2];
            const float keep = upper ? v[j + n / 2] : v[j];
            v[j] = keep + __shfl_xor(send, mask);
          }
        }
        float tot = v[0];
#pragma unroll
        for (; mask > 0; mask >>= 1) tot += __shfl_xor(tot, mask);
        // NB = 32: lane L holds output L >> 1;  NB = 16: output L >> 2
        constexpr int SH = NB == 32 ? 1 : 2;
        const int d = d0 + (lane >> SH);
        if ((lane & ((1 << SH) - 1)) == 0) {
          const float bv = p.vp_b ? p.vp_b[h * DH + d] : 0.f;
          p.out[(size_t)bq * kChannels + h * DH + d] = fmaf(bv, wsum_lane, tot);
        }
      }
    }
  }
  trace_mark(g_trace_late, 0x83ull);                  // (end of workgroup 0's first wave - a sample, not the kernel's end)
}

// ---------------------------------------------------------------------------------------------------------------
// out[r, h*Dh + d] = sum_c W[h*Dh + d][c] * agg[r][h][c] + bias[h*Dh + d] * wsum[r][h]   on v_mfma_f32_16x16x4_f32 (exact
// fp32 products, fp32 accumulate).  One wave per (16 rows, head): A rows = queries, B columns = the head's Dh outputs.
struct HeadProjParams {
  const float* agg; const float* wsum; const float* w; const float* bias; float* out;
  int M, HH;
};

typedef __attribute__((ext_vector_type(4))) float hp4;

template <int NT>   // Dh = 16 * NT
__global__ __launch_bounds__(64) void value_proj_heads_kernel(const HeadProjParams p) {
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 16, h = blockIdx.y;
  const int row = min(r0 + i, p.M - 1);
  const float* ap = p.agg + ((size_t)row * p.HH + h) * kChannels + 8 * g;
  const float* wp = p.w + ((size_t)(h * 16 * NT + i)) * kChannels + 8 * g;
  hp4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = hp4{0.f, 0.f, 0.f, 0.f};
  // the epilogue's operands, requested before the K loop (a wave is alone on its SIMD here: read where they are used - per
  // element, between the stores - each wsum load waited for the store before it)
  float e_bias[NT], e_ws[4];
#pragma unroll
  for (int t = 0; t < NT; ++t) e_bias[t] = p.bias ? p.bias[h * 16 * NT + t * 16 + i] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) e_ws[r] = p.wsum[(size_t)min(r0 + 4 * g + r, p.M - 1) * p.HH + h];
#pragma unroll 4
  for (int kb = 0; kb < kChannels / 32; ++kb) {                // the k order inside a block is (8 g + e): any order sums
    const float4 a0 = *reinterpret_cast<const float4*>(ap + 32 * kb), a1 = *reinterpret_cast<const float4*>(ap + 32 * kb + 4);
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float* wt = wp + (size_t)t * 16 * kChannels + 32 * kb;
      const float4 b0 = *reinterpret_cast<const float4*>(wt), b1 = *reinterpret_cast<const float4*>(wt + 4);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bb[e], acc[t], 0, 0, 0);
    }
  }
  asm volatile("" : "+v"(e_ws[0]), "+v"(e_ws[1]), "+v"(e_ws[2]), "+v"(e_ws[3]));
#pragma unroll
  for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(e_bias[t]));
  // D layout: lane (n = i, g) register r = row 4 g + r
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = h * 16 * NT + t * 16 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = r0 + 4 * g + r;
      if (orow < p.M) p.out[(size_t)orow * kChannels + col] = fmaf(e_bias[t], e_ws[r], acc[t][r]);
    }
  }
}

template <int HH>
static int launch_agg(const CrossAttnParams& p, hipStream_t s, bool bf16) {
  const dim3 grid(p.order ? ((p.B * p.Q + 7) / 8) * 8 : p.B * p.Q);
  auto lds_bytes = [&](int LT) {
    return (size_t)p.N * HH * kPoints * sizeof(float2) + (size_t)HH * LT * kPoints * sizeof(float) + (size_t)p.N * 12 * sizeof(float) +
           (size_t)((p.N + 3) & ~3) * sizeof(float) + (size_t)HH * sizeof(int) + (size_t)HH * p.N * kPoints;
  };
  // (item-major walk, one wave per head: the heads of a query work on the same camera at the same time, and what they share at
  //  the coarse levels is still in the L2; the level-major / two-heads-per-wave forms of rounds 2 - 3 measured slower and are gone)
  auto go = [&](auto kern, int threads, size_t lds) {
    if (lds > 65536) (void)allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, p);
  };
#define GD4D_AGG_GO(LT_)                                                                            \
  if (bf16) go(cross_attn_agg_kernel<HH, LT_, false, HH, uint16_t>, 64 * HH, lds_bytes(LT_));       \
  else go(cross_attn_agg_kernel<HH, LT_, false, HH>, 64 * HH, lds_bytes(LT_));
  switch (p.L) {
    case 1: GD4D_AGG_GO(1) break;
    case 2: GD4D_AGG_GO(2) break;
    case 3: GD4D_AGG_GO(3) break;
    case 4: GD4D_AGG_GO(4) break;
    default: return GD4D_EUNSUPPORTED;
  }
#undef GD4D_AGG_GO
  return check_launch();
}

}  // namespace gd4d

extern "C" void gd4d_trace_set_late(unsigned long long* p) { gd4d::trace_set_late(p); }

extern "C" int gd4d_pyramid_channels_last_fwd(const void* const* feats, const int32_t* level_hw, void* out, int R, int C,
                                              int L, int in_dtype, int out_dtype, int max_cus, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !out || R <= 0 || C <= 0 || L <= 0) return GD4D_EINVAL;
  if (C != CL_C || L > GD4D_MAX_LEVELS || in_dtype != GD4D_F32) return GD4D_EUNSUPPORTED;
  if (out_dtype != GD4D_F32 && out_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  const bool ob = out_dtype == GD4D_BF16;
  if (!aligned16(out)) return GD4D_EALIGN;
  const bool persistent = max_cus > 0;
  const int px = persistent ? CLP_PX : CL_PX;
  ClParams p{};
  int s = 0, base = 0;
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    const int hw = level_hw[2 * l] * level_hw[2 * l + 1];
    p.in[l] = static_cast<const float*>(feats[l]); p.hw[l] = hw; p.start[l] = s; p.tiles[l] = (hw + px - 1) / px;
    p.tile_base[l] = base;
    s += hw;
    base += R * p.tiles[l];
  }
  for (int l = L; l <= GD4D_MAX_LEVELS; ++l) p.tile_base[l] = base;
  for (int l = L; l < GD4D_MAX_LEVELS; ++l) { p.tiles[l] = 1; p.hw[l] = 1; }
  p.out = out; p.R = R; p.L = L; p.S = s; p.total = base;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (persistent) {
    if (ob) {
      if (!allow_dynamic_lds(reinterpret_cast<const void*>(pyramid_channels_last_persistent_kernel<true>), CLP_LDS)) return GD4D_ELAUNCH;
      hipLaunchKernelGGL(pyramid_channels_last_persistent_kernel<true>, dim3(min(max_cus, base)), dim3(CLP_THREADS), CLP_LDS, st, p);
    } else {
      if (!allow_dynamic_lds(reinterpret_cast<const void*>(pyramid_channels_last_persistent_kernel<false>), CLP_LDS)) return GD4D_ELAUNCH;
      hipLaunchKernelGGL(pyramid_channels_last_persistent_kernel<false>, dim3(min(max_cus, base)), dim3(CLP_THREADS), CLP_LDS, st, p);
    }
  } else if (ob) {
    hipLaunchKernelGGL(pyramid_channels_last_kernel<true>, dim3(base), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(pyramid_channels_last_kernel<false>, dim3(base), dim3(256), 0, st, p);
  }
  return check_launch();
}

extern "C" int gd4d_cross_attn_agg_fwd(const void* feats_cl, const int32_t* level_hw, const float* ref, const float* offsets,
                                       const float* attn_logits, const float* cam_logits, const float* lidar2img,
                                       const double* pc_range, float img_h, float img_w, float* agg, float* wsum,
                                       uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh, int C, int L, int P,
                                       int feats_dtype, int flags, const int32_t* query_order, const float* vp_weight,
                                       const float* vp_bias, float* out, void* stream) {
  using namespace gd4d;
  if (!feats_cl || !level_hw || !ref || !offsets || !attn_logits || !cam_logits || !lidar2img || !pc_range) return GD4D_EINVAL;
  if (vp_weight ? !out : (!agg || !wsum)) return GD4D_EINVAL;       // either the projected output or the raw aggregates
  if ((agg != nullptr) != (wsum != nullptr)) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  // B > 1 pairs value rows with the logits of batch (row % B) (deform3d_cross_attn.py:277): gd4d_cross_attn_fwd has that form
  if (C != kChannels || P != kPoints || L > 4 || N > 64 || B != 1) return GD4D_EUNSUPPORTED;
  if (feats_dtype != GD4D_F32 && feats_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (Hh != 4 && Hh != 8 && Hh != 16) return GD4D_EUNSUPPORTED;
  if (!aligned16(feats_cl) || (agg && !aligned16(agg)) || (vp_weight && !aligned16(vp_weight))) return GD4D_EALIGN;
  CrossAttnParams p{};
  p.vp_w = vp_weight; p.vp_b = vp_bias; p.out = out;
  p.value = feats_cl; p.ref = ref; p.offsets = offsets; p.attn_logits = attn_logits; p.cam_logits = cam_logits;
  p.lidar2img = lidar2img; p.agg = agg; p.wsum = wsum; p.mask_out = mask_out; p.uv_out = uv_out; p.order = query_order;
  p.B = B; p.N = N; p.Q = Q; p.L = L; p.P = P;
  p.raw_cam = (flags & GD4D_CA_RAW_CAM_WEIGHTS) ? 1 : 0;
  int start = 0;
  for (int l = 0; l < L; ++l) {
    const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
    if (h <= 0 || w <= 0) return GD4D_EINVAL;
    p.lvl_h[l] = h; p.lvl_w[l] = w; p.lvl_start[l] = start;
    start += h * w;
  }
  p.S = start;
  if ((unsigned long long)B * N * start >= (1ull << 31)) return GD4D_EUNSUPPORTED;   // pixel indices are 32-bit
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool bf16 = feats_dtype == GD4D_BF16;
  switch (Hh) {
    case 4: return launch_agg<4>(p, s, bf16);
    case 8: return launch_agg<8>(p, s, bf16);
    default: return launch_agg<16>(p, s, bf16);
  }
}

extern "C" int gd4d_value_proj_heads_fwd(const float* agg, const float* wsum, const float* weight, const float* bias,
                                         float* out, int M, int Hh, int C, void* stream) {
  using namespace gd4d;
  if (!agg || !wsum || !weight || !out || M <= 0 || Hh <= 0) return GD4D_EINVAL;
  if (C != kChannels || (Hh != 4 && Hh != 8 && Hh != 16)) return GD4D_EUNSUPPORTED;
  if (!aligned16(agg) || !aligned16(weight)) return GD4D_EALIGN;
  HeadProjParams p{agg, wsum, weight, bias, out, M, Hh};
  const dim3 grid((M + 15) / 16, Hh);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (Hh) {
    case 4: hipLaunchKernelGGL(value_proj_heads_kernel<4>, grid, dim3(64), 0, s, p); break;
    case 8: hipLaunchKernelGGL(value_proj_heads_kernel<2>, grid, dim3(64), 0, s, p); break;
    default: hipLaunchKernelGGL(value_proj_heads_kernel<1>, grid, dim3(64), 0, s, p); break;
  }
  return check_launch();
}

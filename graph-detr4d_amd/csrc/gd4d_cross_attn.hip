// gd4d_cross_attn_fwd: fused 3D->2D projection + visibility mask + masked softmax weights +
// multi-camera / multi-level bilinear gather + sigmoid-camera-weighted reduction, for gfx950.
//
// Reference semantics: Deform3DCrossAttn.forward, deform3d_cross_attn.py:220-258 (projection,
// mask), :281-284 (softmax * mask), :301-304 (third-party mmcv MSDA gather), :320-324 (camera
// weights, sum over cameras).  Maths restated in SURVEY.md Appendix A.1.
//
// Work mapping (wave64): one WORKGROUP of waves = one (batch, query) (a first version gave a query one wave: latency-bound on
// ~17 dependent load batches; it is gone).  The 256 output channels of a query
// are exactly 64 lanes x float4, so lane `c4` owns channels [4*c4, 4*c4+4) and belongs to head
// h = 4*c4 / Dh.  With Hh = 8, Dh = 32 the 8 lanes of a head read one 128-byte line per bilinear
// corner (a full L2 line, fully coalesced); no cross-lane reduction is ever needed and the wave
// writes its 1 KiB output row once.  Cameras the query cannot see (about 82 % of (query, camera)
// pairs on a surround rig) are skipped wave-uniformly.
//
// Phase A (projection) is spread over the 64 lanes: entry e = (camera, head, point) -> one lane
// computes u, v and the visibility bit with the reference's exact fp32 operation order
// (no FMA contraction, IEEE division) and parks (u, v) in LDS; phase B re-reads the four points of
// the lane's head per camera (LDS broadcast within the head's lane group).
#include <stdlib.h>

#include "gd4d_common.h"
#include "gd4d_cross_attn_shared.h"

namespace gd4d {

// ---------------------------------------------------------------------------------------------
// One WORKGROUP of WAVES wavefronts per (batch, query).  Same lane -> (head, channel quad)
// mapping as above inside each wave; the query's VISIBLE cameras are compacted (ballot) and dealt
// round-robin to the waves, so a query seen by many cameras no longer serialises its gathers in
// one wave.  Inside a camera
// the code is branch-free - an invisible point of a visible camera (~5 %) gets weight 0 and reads
// the map centre - so all 4 points x L levels x 4 corners loads can be issued back to back.
// Partial sums are combined through LDS in fixed wave order (deterministic).
// B == 1 (every shipped config): the per-head softmax weights and the projected points are read from LDS when needed
// instead of living in registers -> 128 VGPRs, 4 waves per SIMD, no spills (39.0 vs 42.1 us cache-cold at 900 queries
// x 24 cameras).  B > 1 pairs value rows with the logits of batch (row % B) (see gd4d.h): per-camera softmax in
// registers, 3 waves per SIMD.
#ifndef GD4D_GATHER_OCC
#ifdef GD4D_GATHER_HALF
#define GD4D_GATHER_OCC 4
#else
#define GD4D_GATHER_OCC 3
#endif
#endif
#ifndef GD4D_GATHER_WAVES
#define GD4D_GATHER_WAVES 4
#endif
template <typename VT, int HH, int LT, int WAVES, bool BMULTI, int PT>
__global__ __launch_bounds__(GD4D_WAVE * WAVES, (BMULTI || sizeof(VT) == 2) ? GD4D_GATHER_OCC : 4)
void cross_attn_fwd_block(const CrossAttnParams p) {
  constexpr int DH = kChannels / HH;
  constexpr int LANES_PER_HEAD = DH / 4;
  constexpr int E = HH * PT;
  constexpr int LMAX = LT > 0 ? LT : GD4D_MAX_LEVELS;
  constexpr int THREADS = GD4D_WAVE * WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float4* s_red = reinterpret_cast<float4*>(smem_raw);                       // [WAVES-1][64]
  float2* s_uv = reinterpret_cast<float2*>(smem_raw + (WAVES - 1) * GD4D_WAVE * sizeof(float4));  // [N][E]
  int* s_camvis = reinterpret_cast<int*>(s_uv + p.N * E);                    // [N]
  float* s_aw = reinterpret_cast<float*>(s_camvis + ((p.N + 3) & ~3));      // [HH][LMAX*4] softmax weights (B == 1)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup i runs on XCD i % 8 (round-robin dispatch), each XCD with a private L2.  With a locality order of the
  // queries every XCD takes a contiguous range of it, so queries that look at the same camera region share an L2.
  int bq = blockIdx.x;
  if (p.order) {
    const int per_xcd = (p.B * p.Q + 7) >> 3;
    const int pos = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || pos >= p.B * p.Q) return;
    bq = p.order[pos];
  } else if (bq >= p.B * p.Q) {
    return;
  }
  const int b = bq / p.Q;
  const int q = bq - b * p.Q;
  const int L = LT > 0 ? LT : p.L;

  // ---------------- phase A: projection, spread over the whole workgroup ----------------
  // While phase A runs the memory system idles (all workgroups of a launch are resident at once, so the phases are
  // globally aligned): it is kept short - the N camera matrices go through LDS once instead of 12 global loads per entry,
  // a thread's 3-D point (its (head, point) never changes between its entries) is formed once, and the per-head softmax is
  // computed by HH x (L * PT) threads with cross-lane reductions instead of HH threads looping (11 us -> 4 us of 40).
  float* s_mat = s_aw + HH * (LMAX * PT);          // [N][12]: rows 0-2 of lidar2img
  for (int i = tid; i < p.N * 12; i += THREADS) s_mat[i] = p.lidar2img[((size_t)b * p.N + i / 12) * 16 + i % 12];
  if (!BMULTI) {                                   // softmax over L * PT logits per head, one thread per logit
    constexpr int LP = LMAX * PT;                  // <= 32 for the compiled shapes with LT > 0
    const int n_lp = L * PT;
    if (LT > 0 && (LP & (LP - 1)) == 0 && LP <= 32) {
      // LP is a power of two: head = tid / LP, groups of LP lanes reduce with xor shuffles
      if (tid < HH * LP) {
        const int hd = tid / LP, i = tid % LP;
        const float x = p.attn_logits[((size_t)bq * HH + hd) * n_lp + i];
        float mx = x;
#pragma unroll
        for (int o = 1; o < LP; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = expf(x - mx);
        float sum = e;
#pragma unroll
        for (int o = 1; o < LP; o <<= 1) sum += __shfl_xor(sum, o);
        s_aw[hd * LP + i] = e * (1.0f / sum);
      }
    } else if (tid < HH) {                         // other shapes: one thread per head
      float w[LP];
      softmax_lp(p.attn_logits + ((size_t)bq * HH + tid) * n_lp, n_lp, w);
      for (int i = 0; i < n_lp; ++i) s_aw[tid * LP + i] = w[i];
    }
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
      bool vis = false;
      if (e < total) {
        const int n = e / E;
        float u, v;
        vis = project_entry(p, s_mat + n * 12, X, Y, Z, u, v);
        s_uv[e] = vis ? make_float2(u, v) : make_float2(-1.f, -1.f);
        const size_t o = (((size_t)b * p.N + n) * p.Q + q) * E + hp;
        if (p.mask_out) p.mask_out[o] = vis ? 1 : 0;
        if (p.uv_out) { p.uv_out[o * 2] = u; p.uv_out[o * 2 + 1] = v; }
      }
      // per-camera "any point of any head visible": E consecutive entries belong to one camera
      const unsigned long long bal = __ballot(vis);
      constexpr int CPW = GD4D_WAVE / E;          // whole cameras covered by one wave iteration
      if (lane < CPW) {
        const int n = e0 / E + lane;
        const unsigned long long ones = E == 64 ? ~0ull : ((1ull << (E & 63)) - 1ull);
        if (n < p.N) s_camvis[n] = (bal & (ones << (lane * (E & 63)))) ? 1 : 0;
      }
    }
  }
  __syncthreads();

  // ---------------- phase B ----------------
  const int h = lane / LANES_PER_HEAD;
  float aw_reg[BMULTI ? LMAX * PT : 1];
  const float* aw_lds = s_aw + h * (LMAX * PT);

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned long long cams = __ballot(lane < p.N && s_camvis[lane < p.N ? lane : 0] != 0);
#ifdef GD4D_GATHER_DBG_NOB
  cams = 0;                                                  // dev ablation: projection + softmax only
#endif
  // Work items are dealt round-robin to the waves.  B == 1: an item is one POINT of one visible camera (all levels and
  // corners: 16 loads in flight), so a wave's chain of dependent load rounds is ceil(P * cameras / WAVES) long - with whole
  // cameras as items it was P * ceil(cameras / WAVES): 8 rounds instead of 5 for a query seen by five cameras, 4 instead
  // of 2 for two.  The kernel is latency-bound (all workgroups are resident at once: it ends when the longest chain does).
  // B > 1 keeps whole cameras (the per-camera softmax of the row-pairing quirk lives in registers).
  int idx = 0;
  while (cams) {
    const int n = __builtin_ctzll(cams);
    cams &= cams - 1;
    if (BMULTI && (idx++ % WAVES) != wave) continue;         // round-robin deal of visible cameras

    const float2* su = s_uv + n * E + h * PT;
    float2 pu[BMULTI ? PT : 1];
    if (BMULTI) {
#pragma unroll
      for (int k = 0; k < PT; ++k) pu[BMULTI ? k : 0] = su[k];
    }
    const int row = b * p.N + n;
    if (BMULTI) {
      const int bb = row % p.B;
      softmax_lp(p.attn_logits + (((size_t)bb * p.Q + q) * HH + h) * L * PT, L * PT, aw_reg);
    }
    const float cl = p.cam_logits[(size_t)b * p.Q * p.N + (size_t)n * p.Q + q];
    const float cw = p.raw_cam ? cl : 1.0f / (1.0f + expf(-cl));
    const VT* vrow = static_cast<const VT*>(p.value) + (size_t)row * p.S * kChannels;   // wave-uniform
    // pixel stride / lane offset of the two layouts (elements)
    const unsigned pix_stride = p.head_major ? (unsigned)DH : (unsigned)kChannels;
    const unsigned lane_off = p.head_major ? (unsigned)(h * p.S * DH + (lane % LANES_PER_HEAD) * 4) : (unsigned)(lane * 4);

#pragma unroll
    for (int k = 0; k < PT; ++k) {
      if (!BMULTI && (idx++ % WAVES) != wave) continue;      // round-robin deal of (visible camera, point) items
      const float2 puk = BMULTI ? pu[BMULTI ? k : 0] : su[k];
      const bool pv = puk.x >= 0.f;
      const float u = pv ? puk.x : 0.5f, v = pv ? puk.y : 0.5f;
      const float cwk = pv ? cw : 0.f;
      // Stage 1: addresses of all L x 4 corners of this point.  Stage 2: issue every load.  Stage 3
      // (below a scheduling barrier): weights + FMAs.  Left to itself the compiler keeps only 2-4
      // loads in flight per wave to save registers; a gather kernel lives on memory-level parallelism.
      unsigned off[LMAX][4];
      float fx[LMAX], fy[LMAX];
      unsigned okm = 0;                              // 4 bits per level: x0ok, x1ok, y0ok, y1ok
#pragma unroll
      for (int l = 0; l < LMAX; ++l) {
        if (LT == 0 && l >= L) break;
        const int W = p.lvl_w[l], H = p.lvl_h[l];
        const float x = fmaf(u, (float)W, -0.5f);
        const float y = fmaf(v, (float)H, -0.5f);
        const float xf = floorf(x), yf = floorf(y);
        fx[l] = x - xf; fy[l] = y - yf;
        const int x0 = (int)xf, y0 = (int)yf;
        const bool x0ok = x0 >= 0, x1ok = x0 + 1 < W;
        const bool y0ok = y0 >= 0, y1ok = y0 + 1 < H;
        okm |= ((x0ok ? 1u : 0u) | (x1ok ? 2u : 0u) | (y0ok ? 4u : 0u) | (y1ok ? 8u : 0u)) << (4 * l);
        // corners outside the map contribute 0 (zero padding); their loads are clamped onto the map
        const int xa = x0ok ? x0 : 0, xb = x1ok ? x0 + 1 : W - 1;
        const int ya = y0ok ? y0 : 0, yb = y1ok ? y0 + 1 : H - 1;
        const unsigned r0 = (unsigned)(p.lvl_start[l] + ya * W), r1 = (unsigned)(p.lvl_start[l] + yb * W);
        off[l][0] = (r0 + xa) * pix_stride + lane_off;
        off[l][1] = (r0 + xb) * pix_stride + lane_off;
        off[l][2] = (r1 + xa) * pix_stride + lane_off;
        off[l][3] = (r1 + xb) * pix_stride + lane_off;
      }
#ifdef GD4D_GATHER_HALF
      constexpr int NB = 2;
#else
      constexpr int NB = 1;
#endif
      constexpr int LB = (LMAX + NB - 1) / NB;
#pragma unroll
      for (int hb = 0; hb < NB; ++hb) {
        float4 val[LB][4];
#pragma unroll
        for (int li = 0; li < LB; ++li) {
          const int l = hb * LB + li;
          if (l >= LMAX || (LT == 0 && l >= L)) break;
#pragma unroll
          for (int c = 0; c < 4; ++c) val[li][c] = Quad<VT>::load(vrow + off[l][c]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int li = 0; li < LB; ++li) {
          const int l = hb * LB + li;
          if (l >= LMAX || (LT == 0 && l >= L)) break;
          const float wl = (BMULTI ? aw_reg[BMULTI ? l * PT + k : 0] : aw_lds[l * PT + k]) * cwk;
          const float dx = fx[l], dy = fy[l];
          const unsigned ok = okm >> (4 * l);
          const float w00 = ((ok & 5u) == 5u) ? wl * (1.f - dx) * (1.f - dy) : 0.f;
          const float w01 = ((ok & 6u) == 6u) ? wl * dx * (1.f - dy) : 0.f;
          const float w10 = ((ok & 9u) == 9u) ? wl * (1.f - dx) * dy : 0.f;
          const float w11 = ((ok & 10u) == 10u) ? wl * dx * dy : 0.f;
          const float wc[4] = {w00, w01, w10, w11};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc.x = fmaf(wc[c], val[li][c].x, acc.x); acc.y = fmaf(wc[c], val[li][c].y, acc.y);
            acc.z = fmaf(wc[c], val[li][c].z, acc.z); acc.w = fmaf(wc[c], val[li][c].w, acc.w);
          }
        }
      }
    }
  }

  // ---------------- deterministic cross-wave reduction ----------------
  if (WAVES > 1) {
    if (wave > 0) s_red[(wave - 1) * GD4D_WAVE + lane] = acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 1; w < WAVES; ++w) {
        const float4 o = s_red[(w - 1) * GD4D_WAVE + lane];
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
      }
    }
  }
  if (wave == 0) *reinterpret_cast<float4*>(p.out + (size_t)bq * kChannels + lane * 4) = acc;
}

template <typename VT, int HH, int LT, int PT>
static void launch_block(const CrossAttnParams& p, hipStream_t s) {
  constexpr int WAVES = GD4D_GATHER_WAVES;
  const dim3 grid(p.order ? ((p.B * p.Q + 7) / 8) * 8 : p.B * p.Q);
  const size_t lds = (WAVES - 1) * GD4D_WAVE * sizeof(float4) + (size_t)p.N * HH * PT * sizeof(float2) +
                     (size_t)((p.N + 3) & ~3) * sizeof(int) +
                     (size_t)HH * (LT > 0 ? LT : GD4D_MAX_LEVELS) * PT * sizeof(float) + (size_t)p.N * 12 * sizeof(float);
  if (p.B > 1)
    hipLaunchKernelGGL((cross_attn_fwd_block<VT, HH, LT, WAVES, true, PT>), grid, dim3(GD4D_WAVE * WAVES), lds, s, p);
  else
    hipLaunchKernelGGL((cross_attn_fwd_block<VT, HH, LT, WAVES, false, PT>), grid, dim3(GD4D_WAVE * WAVES), lds, s, p);
}

template <typename VT, int HH, int LT>
static void launch_one(const CrossAttnParams& p, hipStream_t s) {
  if (p.P == 1) {                                      // one point per level (Deform3DCrossAttnMP's neighbour pass)
    launch_block<VT, HH, LT, 1>(p, s);
  } else {
    launch_block<VT, HH, LT, kPoints>(p, s);
  }
}

template <typename VT, int HH>
static int launch_levels(const CrossAttnParams& p, hipStream_t s) {
  switch (p.L) {
    case 1: launch_one<VT, HH, 1>(p, s); break;
    case 2: launch_one<VT, HH, 2>(p, s); break;
    case 3: launch_one<VT, HH, 3>(p, s); break;
    case 4: launch_one<VT, HH, 4>(p, s); break;
    default: launch_one<VT, HH, 0>(p, s); break;
  }
  return check_launch();
}

template <typename VT>
static int launch_heads(const CrossAttnParams& p, int Hh, hipStream_t s) {
  switch (Hh) {
    case 4: return launch_levels<VT, 4>(p, s);
    case 8: return launch_levels<VT, 8>(p, s);
    case 16: return launch_levels<VT, 16>(p, s);
    default: return GD4D_EUNSUPPORTED;
  }
}

}  // namespace gd4d

extern "C" int gd4d_cross_attn_fwd(const void* value, const int32_t* level_hw, const float* ref,
                                   const float* offsets, const float* attn_logits,
                                   const float* cam_logits, const float* lidar2img,
                                   const double* pc_range, float img_h, float img_w, float* out,
                                   uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh,
                                   int Dh, int L, int P, int value_dtype, int value_layout, int flags,
                                   const int32_t* query_order, void* stream) {
  using namespace gd4d;
  if (!value || !level_hw || !ref || !offsets || !attn_logits || !cam_logits || !lidar2img ||
      !pc_range || !out)
    return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || Dh <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f))
    return GD4D_EINVAL;
  if (Hh * Dh != kChannels || (P != kPoints && P != 1) || L > GD4D_MAX_LEVELS || N > 64) return GD4D_EUNSUPPORTED;
  if (value_dtype != GD4D_F32 && value_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (value_layout != GD4D_LAYOUT_PIXEL_MAJOR && value_layout != GD4D_LAYOUT_HEAD_MAJOR) return GD4D_EUNSUPPORTED;
  if (!aligned16(value) || !aligned16(out)) return GD4D_EALIGN;

  CrossAttnParams p{};
  p.value = value; p.ref = ref; p.offsets = offsets; p.attn_logits = attn_logits;
  p.cam_logits = cam_logits; p.lidar2img = lidar2img; p.out = out; p.mask_out = mask_out;
  p.uv_out = uv_out; p.order = query_order;
  p.B = B; p.N = N; p.Q = Q; p.L = L; p.P = P;
  p.head_major = value_layout == GD4D_LAYOUT_HEAD_MAJOR;
  p.raw_cam = (flags & GD4D_CA_RAW_CAM_WEIGHTS) ? 1 : 0;
  int start = 0;
  for (int l = 0; l < L; ++l) {
    const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
    if (h <= 0 || w <= 0) return GD4D_EINVAL;
    p.lvl_h[l] = h; p.lvl_w[l] = w; p.lvl_start[l] = start;
    start += h * w;
  }
  p.S = start;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return value_dtype == GD4D_F32 ? launch_heads<float>(p, Hh, s) : launch_heads<uint16_t>(p, Hh, s);
}

// Channel-sliced aggregate-then-project gather (inference, gfx950): gd4d_cross_attn_plan_fwd + gd4d_cross_attn_agg_sliced_fwd.
//
// Reference: Deform3DCrossAttn.forward, deform3d_cross_attn.py:220-258 (projection, mask), :277, :281-284 (softmax x mask,
// the row % B pairing), :301-304 (mmcv MSDA gather), :320-324 (camera-weighted sum) - the same arithmetic as
// gd4d_cross_attn_agg_fwd (gd4d_cross_attn_late.hip): per head h of query q
//     agg[q, h, :] = sum_i w_i x_i,   wsum[q, h] = sum_i w_i       (x_i = RAW 256-channel pixel of an in-bounds corner)
// and value_proj is applied to the 900 x Hh aggregates afterwards (HEADGEMM of the row chain / gd4d_value_proj_heads_fwd).
//
// What is different is the SHAPE OF THE WORK, chosen for the 4-MB private L2 of an XCD.  The one-workgroup-per-query form
// reads a corner as one 1-KB pixel row, so the ~112 queries an XCD owns touch 45 MB of unique lines while 64 of them are
// in flight: a pixel shared by two queries is re-fetched unless both happen to want it within the ~6 us a line survives
// (705 MB of L2 misses per launch for 327 MB of unique bytes, profiles/r02_pmc_cross_attn_agg.json).  Here the 256
// channels are cut into 8 SLICES of 32 (one 128-byte line per corner) and a workgroup is one (query, slice); the
// workgroups of an XCD are issued SLICE-MAJOR, so at any time the XCD works on one slice of all its queries: 1/8 of the
// footprint (5.6 MB) with the same number of bytes in flight.  The pyramid copy is stored slice-planar (8, R, S, 32) so a
// phase reads one contiguous 95-MB plane (HBM channels and TLB reach see a dense region instead of 128 B out of every KB).
//
//   gd4d_cross_attn_plan_fwd        per query, once: projection + mask (bit-exact, shared project_entry) + softmax +
//                                   camera weights -> per head a compacted list of visible (camera, point) items
//                                   {u, v, w_level[4], camera row}: what every slice of that query needs
//   gd4d_cross_attn_agg_sliced_fwd  per (query, slice): one wave per head walks the head's items; a load instruction
//                                   covers 8 corners x 128 B (lane group g = corner slot, lane c = channels 4c..4c+3);
//                                   pixel offsets and weights are computed once per wave, lane = (item, level, corner),
//                                   and handed to the loading lanes with ds_bpermute
//
// Built with -ffp-contract=off like the other cross-attention units (bit-exact mask / uv).
#include <stdlib.h>

#include <type_traits>

#include "gd4d_common.h"
#include "gd4d_cross_attn_shared.h"
#include "gd4d_cross_attn_sliced.h"
#include "gd4d_pyramid_count.h"

namespace gd4d {

GD4D_TRACE_UNIT(sliced)


// ---------------------------------------------------------------------------------------------------------------
// Plan.  One workgroup (4 waves) per position of the locality order (pos -> bq = order[pos]); the plan is indexed by
// POSITION, so the gather finds it without first reading the order.  Everything that does not depend on the slice is
// done here, once per query instead of once per (query, slice):
//     hdr[pos][h]                  M = visible (camera, point) items of head h
//     pair[pos][h][pass t][g][j]   {byte offset inside the level, weight} of the corner that lane group g fetches with
//                                  load j of pass t:  item 4 t + 2 (j >> 2) + (g >> 2), level j & 3, corner g & 3;
//                                  items past M repeat the last item with weight 0 (their lines are already in flight)
//     wsum[bq][h]                  sum of the in-bounds weights
// The offsets are row * cam_stride[level] + pixel * pix_stride of the pyramid the gather will read (PyramidGeom).
struct PlanParams {
  CrossAttnParams c;       // ref, offsets, logits, lidar2img, ranges, mask_out / uv_out, order, wsum, B, N, Q, L
  PyramidGeom g;
  int* hdr;
  uint2* pair;
  int cap_t;               // passes reserved per head
  float4* item;            // GD4D_CA_PLAN_ITEMS: the plan holds ITEMS (two float4 each, below) instead of pairs; else nullptr
  int cap_i;               // items reserved per head (a multiple of 16)
  int both;                // GD4D_CA_PLAN_BOTH: `item` is a second region; the pairs are written as well
};

// PT = sampling points per head (the reference's num_points, deform3d_cross_attn.py:52-69; every shipped config: 4).  An item is
// a visible (camera, point) pair whatever PT is, so the gather does not know it.
template <int HH, int LT, int WAVES, int PT = kPoints>
__device__ __forceinline__ void cross_attn_plan_body(const PlanParams& pp, const int pos, char* smem_raw) {
  const CrossAttnParams& p = pp.c;
  constexpr int E = HH * PT, LP = LT * PT, THREADS = 64 * WAVES;
  float2* s_uv = reinterpret_cast<float2*>(smem_raw);                        // [N][E]; x < 0: not visible
  float* s_mat = reinterpret_cast<float*>(s_uv + p.N * E);                   // [N][12]
  float* s_cw = s_mat + p.N * 12;                                            // [N]
  float* s_aw = s_cw + ((p.N + 3) & ~3);                                     // [B][HH][LP] softmax weights per logit batch
  float4* s_items = reinterpret_cast<float4*>(s_aw + ((p.B * HH * LP + 3) & ~3));   // [WAVES][N * PT]: {u, v, row * PT + point, camera weight}
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bq = p.order ? p.order[pos] : pos;
  const int b = bq / p.Q, q = bq - b * p.Q;

  for (int i = tid; i < p.N * 12; i += THREADS) s_mat[i] = p.lidar2img[((size_t)b * p.N + i / 12) * 16 + i % 12];
  if (tid < p.N) {
    const float cl = p.cam_logits[(size_t)b * p.Q * p.N + (size_t)tid * p.Q + q];   // raw-view scramble (:211-212)
    s_cw[tid] = p.raw_cam ? cl : 1.0f / (1.0f + expf(-cl));
  }
  // softmax over L * P logits per head; value row i = b*N + n is paired with the logits of batch (i % B) (:277)
  if ((LP & (LP - 1)) == 0 && LP <= 32 && (HH * LP) % GD4D_WAVE == 0) {      // one thread per logit, xor-shuffle reductions
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
    const float X = (rp[0] * p.rng_scale[0] + p.rng_lo[0]) + offs[0];     // two roundings, then the offset (:222-229)
    const float Y = (rp[1] * p.rng_scale[1] + p.rng_lo[1]) + offs[1];
    const float Z = (rp[2] * p.rng_scale[2] + p.rng_lo[2]) + offs[2];
    __syncthreads();
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
  // wave w: heads w, w + WAVES, ...: compact the visible (camera, point) pairs, then the corners: lane = (item % 16, level)
  // works out the four corners of its (item, level) - 16 items per step of the wave (with lane = (item % 4, level, corner)
  // the per-(item, level) arithmetic was done four times over: the pass loop was 2.2 M of the launch's 3.9 M vector
  // instructions)
  const int ncand = p.N * PT;
  float4* items = s_items + wave * ncand;
  const int i_of = lane >> 2, l_of = lane & 3;
  int lw = pp.g.lvl_w[0], lh = pp.g.lvl_h[0];
  unsigned cstr = pp.g.cam_stride[0];
#pragma unroll
  for (int l = 1; l < LT; ++l)
    if (l_of == l) { lw = pp.g.lvl_w[l]; lh = pp.g.lvl_h[l]; cstr = pp.g.cam_stride[l]; }
  const float flw = (float)lw, flh = (float)lh;
  const float lvl_on = l_of < LT ? 1.f : 0.f;
  // where the four pairs of this lane go inside their pass: [g][j], g = (item & 1, corner), j = (item >> 1 & 1, level)
  const int slot0 = (((i_of & 1) << 2) << 3) | (((i_of >> 1) & 1) << 2) | l_of;
  for (int h = wave; h < HH; h += WAVES) {
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                 // wave-private list: no workgroup barrier
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (pp.item) {
      // ITEMS form (the inference step): 32 bytes per visible (camera, point) item - {u, v, camera row, M} {weight of
      // level 0..3} - instead of 4 levels x 4 corners x {offset, weight} = 128: the eight slices of a query re-read their
      // plan a phase (~12 us) apart, too far for the L2, and the pairs were 107 of the gather's 499 MB of fabric reads
      // (profiles/r03_pmc_cross_attn_sliced.json).  The corner arithmetic moves into the gather, lane = (item, level), 16
      // items per step (gd4d_cross_attn_agg_items_fwd), which also sums wsum.  M rides in every record: the gather needs
      // one round trip, not header-then-items (record 0 is written for M = 0 too).
      float4* out = pp.item + ((size_t)pos * HH + h) * pp.cap_i * 2;
      for (int it0 = 0; it0 < max(M, 1); it0 += GD4D_WAVE) {
        const int item = it0 + lane;
        if (item < max(M, 1)) {
          const float4 rec = M ? items[item] : make_float4(0.f, 0.f, 0.f, 0.f);
          const int rk = __float_as_int(rec.z);
          const int row = rk / PT, k = rk % PT;
          const int lb = p.B == 1 ? 0 : row % p.B;                          // logits of batch (row % B) (:277)
          const float* aw = s_aw + lb * HH * LP + h * LP + k;
          float wl[4];
#pragma unroll
          for (int l = 0; l < 4; ++l) wl[l] = (l < LT && M) ? aw[l * PT] * rec.w : 0.f;
          out[item * 2] = make_float4(rec.x, rec.y, __int_as_float(row), __int_as_float(M));
          out[item * 2 + 1] = make_float4(wl[0], wl[1], wl[2], wl[3]);
        }
      }
      if (!pp.both) {
        if (lane == 0) pp.hdr[pos * kPlanHdr + h] = M;
        __builtin_amdgcn_wave_barrier();                                   // the list is rewritten for the next head
        continue;
      }                                                                    // both forms: the pairs follow (they write hdr and wsum)
    }
    uint2* out = pp.pair + ((size_t)pos * HH + h) * pp.cap_t * 64;
    const float* aw_h = s_aw + h * LP + min(l_of, LT - 1) * PT;             // + (row % B) * HH * LP + point
    const int m_pad = (M + 3) & ~3;                                         // whole passes: items past M repeat the last one, weight 0
    float wsum_lane = 0.f;
    for (int it0 = 0; it0 < M; it0 += 16) {
      const int item = it0 + i_of;
      const float4 rec = items[min(item, M - 1)];
      const int rk = __float_as_int(rec.z);
      const int row = rk / PT, k = rk % PT;
      const int lb = p.B == 1 ? 0 : row % p.B;                              // logits of batch (row % B) (:277)
      const float wl = aw_h[lb * HH * LP + k] * rec.w;
      const float x = fmaf(rec.x, flw, -0.5f);
      const float y = fmaf(rec.y, flh, -0.5f);
      const float xf = floorf(x), yf = floorf(y);
      const float dx = x - xf, dy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      const float live = item < M ? lvl_on : 0.f;
      const unsigned rbase = (unsigned)row * cstr;
      uint2* dst = out + (size_t)(item >> 2) * 64 + slot0;
#pragma unroll
      for (int c_of = 0; c_of < 4; ++c_of) {
        const int xi = x0 + (c_of & 1), yi = y0 + (c_of >> 1);
        // corners outside the map contribute 0 (zero padding); their loads are clamped onto the map
        const int xc = min(max(xi, 0), lw - 1), yc = min(max(yi, 0), lh - 1);
        const float in = (xc == xi && yc == yi) ? live : 0.f;
        const float wx = (c_of & 1) ? dx : 1.f - dx, wy = (c_of >> 1) ? dy : 1.f - dy;
        const float w = (wl * wx * wy) * in;
        const unsigned off = rbase + (unsigned)(yc * lw + xc) * pp.g.pix_stride;
        wsum_lane += w;
        if (item < m_pad) dst[c_of << 3] = make_uint2(off, __float_as_uint(w));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum_lane += __shfl_xor(wsum_lane, o);
    if (lane == 0) {
      pp.hdr[pos * kPlanHdr + h] = M;
      if (p.wsum) p.wsum[(size_t)bq * HH + h] = wsum_lane;
    }
    __builtin_amdgcn_wave_barrier();                                       // the list is rewritten for the next head
  }
}

template <int HH, int LT, int WAVES, int PT = kPoints>
__global__ __launch_bounds__(64 * WAVES) void cross_attn_plan_kernel(const PlanParams pp) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  trace_mark(g_trace_sliced, 5ull);
  cross_attn_plan_body<HH, LT, WAVES, PT>(pp, blockIdx.x, smem_raw);
  trace_mark(g_trace_sliced, 0x85ull);
}

// ---------------------------------------------------------------------------------------------------------------
// Gather.  Grid = 8 XCDs x slices x per_xcd; workgroup i runs on XCD i % 8 (round-robin dispatch) and, within its XCD,
// index j = i / 8 walks slice-major: slice = j / per_xcd, query position = xcd * per_xcd + j % per_xcd of the locality
// order (an XCD keeps its contiguous azimuth sector, as in gd4d_cross_attn_agg_fwd).
//
// A wave is one head of one (query, slice): ~17 items = ~5 passes of 8 loads.  The first version computed offsets and
// weights in the wave (lane = (item, level, corner), handed to the loading lanes with 16 ds_bpermute per pass): without
// any feature load it ran 65 us of the 133 us the launch took - issue-bound on work that is the same for all 8 slices.
// Now a pass is: 512 B of plan pairs (one coalesced 8-byte load per lane, staged in the wave's own LDS patch), four
// 16-byte LDS reads per lane (the 8 lanes of a corner slot read one address: broadcast), then per load one add, the
// load, two packed FMAs.
struct SlicedParams {
  const char* lvl_base[4];      // level l: address of (camera row 0, pixel 0, slice 0, channel 0)
  long long slice_stride;       // bytes between slices
  const int* hdr;
  const uint2* pair;
  const int32_t* order;
  float* agg;                   // (BQ, HH, 256)
  int BQ, per_xcd, cap_t;
  int slice_lo, slice_n;        // slices [slice_lo, slice_lo + slice_n) in this launch
  int blk;                      // queries per block of the walk: within an XCD, block-major, then slice, then query
};

typedef float f2v __attribute__((ext_vector_type(2)));

// (the walk: which (position, slice) workgroup `block` is; false: none)
__device__ __forceinline__ bool sliced_walk(const SlicedParams& p, int block, int& pos, int& sl) {
  const int xcd = block & 7, jb = block >> 3;
  const int per_blk = p.blk * p.slice_n;                   // workgroups of one block of queries
  const int kb = jb / per_blk, rb = jb - kb * per_blk;
  sl = rb / p.blk;
  const int qi = kb * p.blk + (rb - sl * p.blk);
  pos = xcd * p.per_xcd + qi;
  return qi < p.per_xcd && pos < p.BQ;
}

template <int HH, int LT, typename VT>
__device__ __forceinline__ void cross_attn_agg_sliced_body(const SlicedParams& p, const int pos, const int sl, char* s_raw) {
  constexpr int ES = sizeof(VT);
  constexpr int CH = 6;                                   // passes staged per round (30 KB per workgroup: 4 per CU)
  constexpr int GP = 80;                                  // LDS bytes per corner slot: 64 B of pairs + 16 B pad (conflict-free b128)
  constexpr int PASS = 8 * GP;
  const int lane = threadIdx.x & 63;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = p.slice_lo + sl;
  const uint2* pp = p.pair + ((size_t)pos * HH + h) * p.cap_t * 64 + lane;
  const uint2 first = pp[0];                              // speculative: pass 0 (garbage, unused, when M = 0)
  const int M = __builtin_amdgcn_readfirstlane(p.hdr[pos * kPlanHdr + h]);
  const int bq = p.order ? p.order[pos] : pos;
  const int T = (M + 3) >> 2;
  char* my = s_raw + h * (CH * PASS);
  const int g = lane >> 3, c = lane & 7;
  char* wr = my + g * GP + c * 8;                         // where this lane's pair of a pass goes ([g][j], j = lane & 7)
  const char* rd = my + g * GP;
  const char* base[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) base[l] = p.lvl_base[l] + (size_t)s * p.slice_stride;
  const unsigned lane_off = (unsigned)(c * 4 * ES);

  f2v acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
  for (int t0 = 0; t0 < T; t0 += CH) {
    const int nt = min(CH, T - t0);
    if (t0 > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (k >= nt) break;
      const uint2 v = (k == 0 && t0 == 0) ? first : pp[(size_t)(t0 + k) * 64];
      *reinterpret_cast<uint2*>(wr + k * PASS) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                 // wave-private LDS patch: no workgroup barrier
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = 0; k < nt; ++k) {
      const uint4* row = reinterpret_cast<const uint4*>(rd + k * PASS);
      uint4 pr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = row[i];                          // pairs j = 2 i, 2 i + 1: {off, w, off, w}
      const bool second = (t0 + k) * 4 + 2 < M;                            // wave-uniform: items 2, 3 of the pass exist
      float4 val[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((j & 3) >= LT) continue;
        if (j >= 4 && !second) break;
        const unsigned o = ((j & 1) ? pr[j >> 1].z : pr[j >> 1].x) + lane_off;
        const VT* ap = reinterpret_cast<const VT*>(base[j & 3] + o);
        val[j] = Quad<VT>::load(ap);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((j & 3) >= LT) continue;
        if (j >= 4 && !second) break;
        const float w = __uint_as_float((j & 1) ? pr[j >> 1].w : pr[j >> 1].y);
        const f2v ww = {w, w};
        acc0 = __builtin_elementwise_fma(ww, f2v{val[j].x, val[j].y}, acc0);
        acc1 = __builtin_elementwise_fma(ww, f2v{val[j].z, val[j].w}, acc1);
      }
    }
  }
  float4 acc = make_float4(acc0.x, acc0.y, acc1.x, acc1.y);
  // the 8 corner slots meet in a fixed order (deterministic)
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
    acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
  }
  if (g == 0) *reinterpret_cast<float4*>(p.agg + ((size_t)bq * HH + h) * kChannels + s * kSlice + c * 4) = acc;
}

template <int HH, int LT, typename VT, int OCC>
__global__ __launch_bounds__(64 * HH, OCC) void cross_attn_agg_sliced_kernel(const SlicedParams p) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];   // [HH][CH][8][GP]
  trace_mark(g_trace_sliced, 6ull);
  int pos, sl;
  if (!sliced_walk(p, blockIdx.x, pos, sl)) return;
  cross_attn_agg_sliced_body<HH, LT, VT>(p, pos, sl, s_raw);
  trace_mark(g_trace_sliced, 0x86ull);
}

// ---------------------------------------------------------------------------------------------------------------
// Gather from the ITEMS form of the plan (GD4D_CA_PLAN_ITEMS; the inference step's default).  Same walk, same pass loop
// and the same products in the same order as cross_attn_agg_sliced_body - results are bit-identical to it -
// but the {offset, weight} pairs of a pass are worked out HERE from 32-byte item records, lane = (item % 16, level), 16 items
// (four passes) per step, and parked in the wave's LDS patch in the layout the pass loop reads.  That is ~45 vector
// instructions per four passes on top of the ~160 the passes cost, for a plan a quarter of the size: the pairs were
// 107 MB of the 499 MB a launch pulled over the fabric, because the eight slices of a query read them a phase apart.
// The slice-0 workgroup of a query also sums wsum (same lanes, same order as the plan kernel's pairs form).
struct ItemsParams {
  SlicedParams s;
  PyramidGeom g;
  const float4* item;
  float* wsum;                  // (BQ, HH), written by the workgroups of slice 0
  int cap_i;
  // gd4d_cross_attn_agg_items_coarse_fwd: the coarse levels [2, 4) are gathered from PROJECTED rows (value_proj applied per pixel,
  // bias included: (R, H_l W_l, 256) fp32, head h = bytes [128 h, 128 h + 128) of a pixel's row) by a ninth "slice" of workgroups
  PyramidGeom gc;               // their geometry (entries 2, 3)
  const char* proj_base[4];     // entries 2, 3
  float* pagg;                  // (BQ, 256): sum over the coarse levels' corners of w * projected row, head-major columns
};

// WIDE: a level spans 4 GiB or more (VoVNet-99 level 0 stored channels-last, two samples: 4.6 GB) - the offsets parked in LDS
// are then in units of 16 bytes (every stride is a multiple of 16; the host hands cam_stride / pix_stride pre-divided) and a
// load forms its address with a 64-bit shift-add.
//
// LAP = level slots of an item's lane set (4: lane = (item % 16, level); 2: lane = (item % 32, level) - a pass of 8 loads then
// covers 8 items x 2 levels), LV0 = the first level gathered (levels LV0 .. LV0 + LT - 1 of the records and the geometry).
// COARSE (gd4d_cross_attn_agg_items_coarse_fwd): the workgroup is the query's ninth "slice": wave h gathers head h's 128 bytes of
// the PROJECTED rows of levels LV0 .. (ip.gc / ip.proj_base) and writes pagg instead of agg.
template <int HH, int LT, typename VT, bool WIDE, int LAP = 4, int LV0 = 0, bool COARSE = false>
__device__ __forceinline__ void cross_attn_agg_items_body(const ItemsParams& ip, const int pos, const int sl, char* s_raw) {
  const SlicedParams& p = ip.s;
  const PyramidGeom& geo = COARSE ? ip.gc : ip.g;
  constexpr int ES = sizeof(VT);
  constexpr int LA = LT;                                  // a lane set = the LAP level slots of an item
  static_assert(LA <= LAP && LV0 + LAP <= 4 && (LAP == 2 || LAP == 4), "level slots");
  static_assert(!COARSE || (std::is_same<VT, float>::value && !WIDE && HH * kSlice == kChannels), "projected rows: fp32, 32 channels per head");
  constexpr int IPP = 16 / LAP;                           // items per pass: 4 (LAP = 2: 8)
  constexpr int IPS = 64 / LAP;                           // items per step (four passes): 16 (32)
  constexpr int CH = 4;                                   // passes per step (20 KB of LDS per workgroup of 8 waves)
  constexpr int GP = 80;                                  // LDS bytes per corner slot: 64 B of pairs + 16 B pad (conflict-free b128)
  constexpr int PASS = 8 * GP;
  const int lane = threadIdx.x & 63;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = COARSE ? h : p.slice_lo + sl;
  const int i_of = lane / LAP, l_of = lane % LAP;
  const float4* rec = ip.item + ((size_t)pos * HH + h) * ip.cap_i * 2;
  // speculative: the first records (what lies past M is replaced below before it is used)
  const int i_first = min(i_of, ip.cap_i - 1);
  float4 a = rec[i_first * 2];
  float wl = reinterpret_cast<const float*>(rec + i_first * 2 + 1)[LV0 + l_of];
  const int bq = p.order ? p.order[pos] : pos;
  char* my = s_raw + h * (CH * PASS);
  const int g = lane >> 3, c = lane & 7;
  const char* rd = my + g * GP;
  // where the four pairs of this lane go: pass i_of / IPP, inside it [g][j], g = (item & 1, corner), j = (item >> 1, level)
  const int ip_ = i_of % IPP;
  char* wr = my + (i_of / IPP) * PASS + ((ip_ & 1) << 2) * GP + ((ip_ >> 1) * LAP + l_of) * 8;
  const char* base[LA];
#pragma unroll
  for (int l = 0; l < LA; ++l)
    base[l] = COARSE ? ip.proj_base[LV0 + l] + (size_t)s * (kSlice * sizeof(float)) : p.lvl_base[LV0 + l] + (size_t)s * p.slice_stride;
  const unsigned lane_off = (unsigned)(c * 4 * ES);
  int lw = geo.lvl_w[LV0], lh = geo.lvl_h[LV0];
  unsigned cstr = geo.cam_stride[LV0];
#pragma unroll
  for (int l = 1; l < LA; ++l)
    if (l_of == l) { lw = geo.lvl_w[LV0 + l]; lh = geo.lvl_h[LV0 + l]; cstr = geo.cam_stride[LV0 + l]; }
  const float flw = (float)lw, flh = (float)lh;
  const float lvl_on = l_of < LA ? 1.f : 0.f;

  const int M = __builtin_amdgcn_readfirstlane(__float_as_int(a.w));       // every record carries its head's count
  const int T = (M + IPP - 1) / IPP;
  const int m_even = (M + 1) & ~1;                                         // a load covers a PAIR of items: the odd one out is
  f2v acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};                                // written with weight 0 (its line is in flight already)
  float wsum_lane = 0.f;
  if (M < IPS && M > 0) {                                                   // wave-uniform: items past M repeat the last one
    const int src = min(i_of, M - 1) * LAP + l_of;
    a.x = __shfl(a.x, src); a.y = __shfl(a.y, src); a.z = __shfl(a.z, src); wl = __shfl(wl, src);
  }
  for (int it0 = 0; it0 < M; it0 += IPS) {
    const int item = it0 + i_of;
    {
      const int row = __float_as_int(a.z);
      const float x = fmaf(a.x, flw, -0.5f);
      const float y = fmaf(a.y, flh, -0.5f);
      const float xf = floorf(x), yf = floorf(y);
      const float dx = x - xf, dy = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      const float live = item < M ? lvl_on : 0.f;
      const unsigned rbase = (unsigned)row * cstr;
      if (it0 > 0) __builtin_amdgcn_wave_barrier();                         // the previous step's passes have read the patch
#pragma unroll
      for (int c_of = 0; c_of < 4; ++c_of) {
        const int xi = x0 + (c_of & 1), yi = y0 + (c_of >> 1);
        // corners outside the map contribute 0 (zero padding); their loads are clamped onto the map
        const int xc = min(max(xi, 0), lw - 1), yc = min(max(yi, 0), lh - 1);
        const float in = (xc == xi && yc == yi) ? live : 0.f;
        const float wx = (c_of & 1) ? dx : 1.f - dx, wy = (c_of >> 1) ? dy : 1.f - dy;
        const float w = (wl * wx * wy) * in;
        const unsigned off = rbase + (unsigned)(yc * lw + xc) * geo.pix_stride;
        wsum_lane += w;
        if (item < m_even && l_of < LA) *reinterpret_cast<uint2*>(wr + c_of * GP) = make_uint2(off, __float_as_uint(w));
      }
    }
    if (it0 + IPS < M) {                                                    // the next step's records fly under this step's passes
      const int nx = min(item + IPS, M - 1);
      a = rec[nx * 2];
      wl = reinterpret_cast<const float*>(rec + nx * 2 + 1)[LV0 + l_of];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                 // wave-private LDS patch: no workgroup barrier
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int t0 = it0 / IPP;
    const int nt = min(CH, T - t0);
    for (int k = 0; k < nt; ++k) {
      const uint4* row = reinterpret_cast<const uint4*>(rd + k * PASS);
      uint4 pr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pr[i] = row[i];                          // pairs j = 2 i, 2 i + 1: {off, w, off, w}
      const int left = M - (t0 + k) * IPP;                                 // wave-uniform: items of this pass; load j's pair exists
      float4 val[8];                                                       // while 2 (j / LAP) < left
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((j % LAP) >= LA) continue;
        if (2 * (j / LAP) >= left) break;
        const unsigned raw = (j & 1) ? pr[j >> 1].z : pr[j >> 1].x;
        const unsigned o = raw + lane_off;
        const VT* ap = WIDE ? reinterpret_cast<const VT*>(base[j % LAP] + (((size_t)raw << 4) + lane_off))
                            : reinterpret_cast<const VT*>(base[j % LAP] + o);
        val[j] = Quad<VT>::load(ap);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((j % LAP) >= LA) continue;
        if (2 * (j / LAP) >= left) break;
        const float w = __uint_as_float((j & 1) ? pr[j >> 1].w : pr[j >> 1].y);
        const f2v ww = {w, w};
        acc0 = __builtin_elementwise_fma(ww, f2v{val[j].x, val[j].y}, acc0);
        acc1 = __builtin_elementwise_fma(ww, f2v{val[j].z, val[j].w}, acc1);
      }
    }
  }
  float4 acc = make_float4(acc0.x, acc0.y, acc1.x, acc1.y);
  // the 8 corner slots meet in a fixed order (deterministic)
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
    acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
  }
  if (COARSE) {
    if (g == 0) *reinterpret_cast<float4*>(ip.pagg + (size_t)bq * kChannels + h * kSlice + c * 4) = acc;
    return;
  }
  if (g == 0) *reinterpret_cast<float4*>(p.agg + ((size_t)bq * HH + h) * kChannels + s * kSlice + c * 4) = acc;
  if (s == 0 && ip.wsum) {                                                 // wave-uniform
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum_lane += __shfl_xor(wsum_lane, o);
    if (lane == 0) ip.wsum[(size_t)bq * HH + h] = wsum_lane;
  }
}

template <int HH, int LT, typename VT, int OCC, bool WIDE = false>
__global__ __launch_bounds__(64 * HH, OCC) void cross_attn_agg_items_kernel(const ItemsParams ip) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];   // [HH][CH][8][GP]
  trace_mark(g_trace_sliced, 6ull);
  int pos, sl;
  if (!sliced_walk(ip.s, blockIdx.x, pos, sl)) return;
  cross_attn_agg_items_body<HH, LT, VT, WIDE>(ip, pos, sl, s_raw);
  trace_mark(g_trace_sliced, 0x86ull);
}

// gd4d_cross_attn_agg_items_coarse_fwd (4 levels, 8 heads): the fine levels 0, 1 from the RAW pyramid as above - 8 items x 2 levels
// per pass -, the coarse levels 2, 3 from rows value_proj has already been applied to.  A raw corner is 1 KB through the L1s
// (8 slices x 128 B) whatever its level; a projected corner of a head is its own 128 B.  Levels 2-3 are 6 % of the pixels
// but were 42 us of the 124-us launch (docs/measurements_r05.md section 1): projecting their 43 800 rows per layer costs less
// than gathering them raw.  The projected rows are gathered by a NINTH workgroup per query (walk: slice index 8, the last
// phase of an XCD), wave = head as everywhere: the same number of loads as one raw slice of the two fine levels, so the
// launch stays balanced; its sums go to pagg (BQ, 256), which HEADGEMM adds to W_h agg_h + b_h wsum_h (wsum then
// holds the fine levels' weights only: the projected rows carry their bias).
template <int HH, typename VT, int OCC, bool WIDE = false>
__global__ __launch_bounds__(64 * HH, OCC) void cross_attn_agg_items_coarse_kernel(const ItemsParams ip) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];   // [HH][CH][8][GP]
  trace_mark(g_trace_sliced, 6ull);
  int pos, sl;
  if (!sliced_walk(ip.s, blockIdx.x, pos, sl)) return;
  if (sl == kSlices) cross_attn_agg_items_body<HH, 2, float, false, 2, 2, true>(ip, pos, sl, s_raw);
  else cross_attn_agg_items_body<HH, 2, VT, WIDE, 2, 0, false>(ip, pos, sl, s_raw);
  trace_mark(g_trace_sliced, 0x86ull);
}

// The forward gather of a TRAINING step and the first step of the pyramid-gradient bookkeeping (gd4d_pyramid_count.h: a slot per
// record, one returning atomic per group of lanes that share a chunk) in ONE launch.  Both read the plan only; the count lives on
// L2 atomic round trips (45 us per layer by itself), the gather on the fabric: one after the other they took 135 + 45 us.  Every
// (every + 1)-th group of 8 workgroups counts - 8 (position, head) waves per workgroup, while count groups remain -, the others
// are the gather's, renumbered (a multiple of 8 leaves: a workgroup keeps its XCD and the walk its order).
struct CountArgs {
  const int* hdr;
  const uint2* pair;
  int* count;
  uint2* slots;
  PgChunks g;
  int cap_t, BQ, groups, every;    // groups = count workgroups / 8 (rounded up); one count group after `every` gather groups
};

template <int HH, int LT, typename VT, int OCC>
__global__ __launch_bounds__(64 * HH, OCC) void cross_attn_agg_items_count_kernel(const ItemsParams ip, const CountArgs ca) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  const int grp = blockIdx.x >> 3, period = ca.every + 1;
  const int k = grp / period;
  if (grp - k * period == ca.every && k < ca.groups) {
    pyramid_grad_count_body(ca.hdr, ca.pair, ca.cap_t, HH, ca.BQ, ca.g, ca.count, ca.slots,
                            (k * 8 + (int)(blockIdx.x & 7)) * HH + (int)(threadIdx.x >> 6));
    return;
  }
  int pos, sl;
  if (!sliced_walk(ip.s, (int)blockIdx.x - 8 * min(k, ca.groups), pos, sl)) return;
  cross_attn_agg_items_body<HH, LT, VT, false>(ip, pos, sl, s_raw);
}

// ---------------------------------------------------------------------------------------------------------------
// NCHW levels -> slice-planar copy (8, R, S, 32): the reference's flatten / transpose / cat (:264-276) with the channel
// axis cut into 8 planes.  One workgroup = 32 pixels x 256 channels of one (camera row, level), turned through LDS; a
// wave store instruction writes 8 consecutive pixels of one slice (1 KB contiguous; bf16: 512 B).
struct SpParams {
  const float* in[GD4D_MAX_LEVELS];
  int hw[GD4D_MAX_LEVELS];
  int start[GD4D_MAX_LEVELS];
  int tiles[GD4D_MAX_LEVELS];
  int tile_base[GD4D_MAX_LEVELS + 1];
  void* out;
  int R, L, S, total;
};

constexpr int SP_PX = 64, SP_C = 256, SP_PITCH = 260, SP_THREADS = 512;
constexpr int SP_LDS = 84 * 1024;                                // persistent form: > 80 KB, one workgroup per CU

template <bool OUT_BF16>
__device__ __forceinline__ void sp_store4(void* base, size_t elem, float4 v) {
  if (OUT_BF16) {
    uint2 pk;
    pk.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
    pk.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(base) + elem) = pk;
  } else {
    *reinterpret_cast<float4*>(static_cast<float*>(base) + elem) = v;
  }
}

// One 512-thread workgroup per tile of 64 pixels (PERSISTENT: one workgroup per compute unit, grid-stride over tiles,
// the next tile's loads in flight while the current one is written; its LDS request is padded so that no second one
// fits a CU and the units it leaves alone stay free for the first layer's query side, as the pixel-major copy does).
template <bool OUT_BF16, bool PERSISTENT>
__global__ __launch_bounds__(SP_THREADS) void pyramid_slice_planar_kernel(const SpParams p) {
  extern __shared__ __attribute__((aligned(16))) float s_tp[];  // [SP_PX][SP_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // 8 waves; wave w loads channels 32 w .. 32 w + 31
  trace_mark(g_trace_sliced, 4ull);
  float v[32];
  int npx = 0, ostart = 0, pix0 = 0, row = 0;
  auto load_tile = [&](int t) {
    const float* src = p.in[0];
    int hw = p.hw[0], tiles = p.tiles[0], tbase = 0, os = p.start[0];
#pragma unroll
    for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
      if (l < p.L && t >= p.tile_base[l]) { src = p.in[l]; hw = p.hw[l]; os = p.start[l]; tiles = p.tiles[l]; tbase = p.tile_base[l]; }
    const int rel = t - tbase;
    row = rel / tiles;
    pix0 = (rel - row * tiles) * SP_PX;
    npx = min(SP_PX, hw - pix0);
    ostart = os;
    const float* gp = src + ((size_t)row * SP_C + wave * 32) * hw + pix0 + min(lane, npx - 1);
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = gp[(size_t)i * hw];
  };
  int t = blockIdx.x;
  if (t < p.total) load_tile(t);
  const size_t plane = (size_t)p.R * p.S * kSlice;                 // elements per slice plane
  while (t < p.total) {
    const int c_npx = npx, c_ostart = ostart, c_pix0 = pix0, c_row = row;
#pragma unroll
    for (int i = 0; i < 32; ++i) s_tp[lane * SP_PITCH + wave * 32 + i] = v[i];
    __syncthreads();
    const int tn = t + gridDim.x;
    if (PERSISTENT && tn < p.total) load_tile(tn);               // in flight during the store phase
    // wave w writes slice w: 8 store instructions of 8 pixels x 128 B
    const size_t obase = (size_t)wave * plane + ((size_t)c_row * p.S + c_ostart + c_pix0) * kSlice;
#pragma unroll
    for (int i = 0; i < SP_PX / 8; ++i) {
      const int px = 8 * i + (lane >> 3);
      if (px < c_npx)
        sp_store4<OUT_BF16>(p.out, obase + (size_t)px * kSlice + (lane & 7) * 4,
                            *reinterpret_cast<const float4*>(&s_tp[px * SP_PITCH + wave * 32 + (lane & 7) * 4]));
    }
    if (!PERSISTENT) break;
    __syncthreads();
    t = tn;
  }
  trace_mark(g_trace_sliced, 0x84ull);
}

}  // namespace gd4d

extern "C" void gd4d_trace_set_sliced(unsigned long long* p) { gd4d::trace_set_sliced(p); }

extern "C" size_t gd4d_cross_attn_plan_bytes(int B, int N, int Q, int Hh, int P) {
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || P <= 0) return 0;
  const size_t pairs = (size_t)B * Q * Hh * gd4d::plan_cap_t(N, P) * 64 * sizeof(uint2);      // (the items form needs a quarter)
  return gd4d::plan_hdr_bytes(B, Q) + pairs;
}

namespace gd4d {
static int fill_plan_params(PlanParams& pp, const float* ref, const float* offsets, const float* attn_logits,
                            const float* cam_logits, const float* lidar2img, const double* pc_range, float img_h, float img_w,
                            const int32_t* level_hw, const int64_t* cam_stride_bytes, int64_t pix_stride_bytes, void* plan,
                            size_t plan_bytes, float* wsum, uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh, int L, int P,
                            int flags, const int32_t* query_order);
}

extern "C" int gd4d_cross_attn_plan_fwd(const float* ref, const float* offsets, const float* attn_logits,
                                        const float* cam_logits, const float* lidar2img, const double* pc_range,
                                        float img_h, float img_w, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                                        int64_t pix_stride_bytes, void* plan, size_t plan_bytes, float* wsum,
                                        uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh, int L, int P, int flags,
                                        const int32_t* query_order, void* stream) {
  using namespace gd4d;
  PlanParams pp{};
  if (int rc = fill_plan_params(pp, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, level_hw, cam_stride_bytes,
                                pix_stride_bytes, plan, plan_bytes, wsum, mask_out, uv_out, B, N, Q, Hh, L, P, flags, query_order))
    return rc;
  const CrossAttnParams& p = pp.c;
  (void)p;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(B * Q);
  auto lds = [&](int LT) {
    return (size_t)N * Hh * P * sizeof(float2) + (size_t)N * 12 * sizeof(float) + (size_t)((N + 3) & ~3) * sizeof(float) +
           (size_t)((B * Hh * LT * P + 3) & ~3) * sizeof(float) + (size_t)8 * N * P * sizeof(float4);
  };
  // one wave per head up to 8 heads (the heads' pass loops are the kernel's longest dependent chain)
  auto go = [&](auto kern, int waves, size_t bytes) -> int {
    if (bytes > 65536 && !allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)bytes)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(64 * waves), bytes, s, pp);
    return check_launch();
  };
#define GD4D_PLAN_GO(HH_, LT_) return go(cross_attn_plan_kernel<HH_, LT_, (HH_ < 8 ? HH_ : 8)>, (HH_ < 8 ? HH_ : 8), lds(LT_))
#define GD4D_PLAN_L(HH_)                  \
  switch (L) {                            \
    case 1: GD4D_PLAN_GO(HH_, 1);         \
    case 2: GD4D_PLAN_GO(HH_, 2);         \
    case 3: GD4D_PLAN_GO(HH_, 3);         \
    default: GD4D_PLAN_GO(HH_, 4);        \
  }
  if (lds(L) > 160 * 1024) return GD4D_EUNSUPPORTED;
  if (P != kPoints) {                                 // num_points 1 / 2 / 8: compiled for the reference's default of 8 heads
#define GD4D_PLAN_P(PT_)                                                                  \
    switch (L) {                                                                          \
      case 1: return go(cross_attn_plan_kernel<8, 1, 8, PT_>, 8, lds(1));                 \
      case 2: return go(cross_attn_plan_kernel<8, 2, 8, PT_>, 8, lds(2));                 \
      case 3: return go(cross_attn_plan_kernel<8, 3, 8, PT_>, 8, lds(3));                 \
      default: return go(cross_attn_plan_kernel<8, 4, 8, PT_>, 8, lds(4));                \
    }
    switch (P) {
      case 1: GD4D_PLAN_P(1)
      case 2: GD4D_PLAN_P(2)
      default: GD4D_PLAN_P(8)
    }
#undef GD4D_PLAN_P
  }
  switch (Hh) {
    case 4: GD4D_PLAN_L(4)
    case 8: GD4D_PLAN_L(8)
    default: GD4D_PLAN_L(16)
  }
#undef GD4D_PLAN_L
#undef GD4D_PLAN_GO
}

static int gd4d_fill_plan_params_impl(gd4d::PlanParams& pp, const float* ref, const float* offsets, const float* attn_logits,
                                      const float* cam_logits, const float* lidar2img, const double* pc_range, float img_h,
                                      float img_w, const int32_t* level_hw, const int64_t* cam_stride_bytes, int64_t pix_stride_bytes,
                                      void* plan, size_t plan_bytes, float* wsum, uint8_t* mask_out, float* uv_out, int B, int N,
                                      int Q, int Hh, int L, int P, int flags, const int32_t* query_order) {
  using namespace gd4d;
  if (!ref || !offsets || !attn_logits || !cam_logits || !lidar2img || !pc_range || !plan || !level_hw || !cam_stride_bytes)
    return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  if ((P != 1 && P != 2 && P != 4 && P != 8) || (P != kPoints && Hh != 8) || L > 4 || N > 64 || B > 16) return GD4D_EUNSUPPORTED;
  if (Hh != 4 && Hh != 8 && Hh != 16) return GD4D_EUNSUPPORTED;
  if (!aligned16(plan)) return GD4D_EALIGN;
  if (plan_bytes < gd4d_cross_attn_plan_bytes(B, N, Q, Hh, P)) return GD4D_EWORKSPACE;
  if (pix_stride_bytes <= 0 || pix_stride_bytes >= (1ll << 31)) return GD4D_EINVAL;
  CrossAttnParams& p = pp.c;
  p.ref = ref; p.offsets = offsets; p.attn_logits = attn_logits; p.cam_logits = cam_logits; p.lidar2img = lidar2img;
  p.mask_out = mask_out; p.uv_out = uv_out; p.order = query_order; p.wsum = wsum;
  p.B = B; p.N = N; p.Q = Q; p.L = L; p.P = P;
  p.raw_cam = (flags & GD4D_CA_RAW_CAM_WEIGHTS) ? 1 : 0;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  for (int l = 0; l < 4; ++l) { pp.g.lvl_w[l] = 1; pp.g.lvl_h[l] = 1; pp.g.cam_stride[l] = 0; }
  for (int l = 0; l < L; ++l) {
    const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
    if (h <= 0 || w <= 0 || cam_stride_bytes[l] < 0) return GD4D_EINVAL;
    // 32-bit byte offsets inside a level: the last pixel of the last camera row
    const unsigned long long span = (unsigned long long)(B * N - 1) * (unsigned long long)cam_stride_bytes[l] +
                                    (unsigned long long)(h * w) * (unsigned long long)pix_stride_bytes;
    // (the ITEMS form stores no offsets: its gather checks the spans itself and switches to 16-byte units past 4 GiB)
    if (!(flags & GD4D_CA_PLAN_ITEMS) && (span >= (1ull << 32) || cam_stride_bytes[l] >= (1ll << 32))) return GD4D_EUNSUPPORTED;
    pp.g.lvl_w[l] = w; pp.g.lvl_h[l] = h; pp.g.cam_stride[l] = (unsigned)cam_stride_bytes[l];
  }
  pp.g.pix_stride = (unsigned)pix_stride_bytes;
  pp.hdr = static_cast<int*>(plan);
  pp.pair = reinterpret_cast<uint2*>(static_cast<char*>(plan) + plan_hdr_bytes(B, Q));
  pp.cap_t = plan_cap_t(N, P);
  pp.item = nullptr;
  pp.cap_i = plan_cap_items(N, P);
  pp.both = 0;
  if (flags & GD4D_CA_PLAN_ITEMS) pp.item = reinterpret_cast<float4*>(pp.pair);   // same place, a quarter of the bytes
  if (flags & GD4D_CA_PLAN_BOTH) {
    // A training step: the forward gather walks the ITEMS (a quarter of the plan bytes), the backward kernels the pairs.  The
    // buffer holds two plans back to back - [0, bytes): header + pairs, [bytes, 2 bytes): a header's room + items, so that
    // plan + bytes is what gd4d_cross_attn_agg_items_fwd takes.
    const size_t one = gd4d_cross_attn_plan_bytes(B, N, Q, Hh, P);
    if ((flags & GD4D_CA_PLAN_ITEMS) || plan_bytes < 2 * one) return GD4D_EINVAL;
    pp.item = reinterpret_cast<float4*>(static_cast<char*>(plan) + one + plan_hdr_bytes(B, Q));
    pp.both = 1;
  }
  return GD4D_OK;
}

namespace gd4d {
static int fill_plan_params(PlanParams& pp, const float* ref, const float* offsets, const float* attn_logits,
                            const float* cam_logits, const float* lidar2img, const double* pc_range, float img_h, float img_w,
                            const int32_t* level_hw, const int64_t* cam_stride_bytes, int64_t pix_stride_bytes, void* plan,
                            size_t plan_bytes, float* wsum, uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh, int L, int P,
                            int flags, const int32_t* query_order) {
  return gd4d_fill_plan_params_impl(pp, ref, offsets, attn_logits, cam_logits, lidar2img, pc_range, img_h, img_w, level_hw,
                                    cam_stride_bytes, pix_stride_bytes, plan, plan_bytes, wsum, mask_out, uv_out, B, N, Q, Hh, L, P, flags,
                                    query_order);
}
}  // namespace gd4d

namespace gd4d {
template <int HH, typename VT>
static int launch_sliced(const SlicedParams& p, int L, hipStream_t s) {
  const size_t lds = (size_t)HH * 6 * 8 * 80;                       // [HH][CH][8][GP]
  const dim3 grid(8 * ((p.per_xcd + p.blk - 1) / p.blk) * p.blk * p.slice_n);
  auto go = [&](auto kern) -> int {
    if (lds > 65536 && !allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(64 * HH), lds, s, p);
    return check_launch();
  };
  constexpr int OCC = HH == 16 ? 4 : 6;                             // waves per SIMD
  switch (L) {
    case 1: return go(cross_attn_agg_sliced_kernel<HH, 1, VT, OCC>);
    case 2: return go(cross_attn_agg_sliced_kernel<HH, 2, VT, OCC>);
    case 3: return go(cross_attn_agg_sliced_kernel<HH, 3, VT, OCC>);
    default: return go(cross_attn_agg_sliced_kernel<HH, 4, VT, OCC>);
  }
}
}  // namespace gd4d

namespace gd4d {
static int fill_sliced_params(SlicedParams& p, const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan, float* agg,
                              int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype, const int32_t* query_order,
                              int slice_lo, int slice_n) {
  if (!level_ptrs || !plan || !agg) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || L <= 0) return GD4D_EINVAL;
  if (C != kChannels || (P != 1 && P != 2 && P != 4 && P != 8) || (P != kPoints && Hh != 8) || L > 4 || N > 64 || B > 16)
    return GD4D_EUNSUPPORTED;
  if (feats_dtype != GD4D_F32 && feats_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (Hh != 4 && Hh != 8 && Hh != 16) return GD4D_EUNSUPPORTED;
  if (slice_lo < 0 || slice_n <= 0 || slice_lo + slice_n > kSlices) return GD4D_EINVAL;
  if (!aligned16(agg) || !aligned16(plan)) return GD4D_EALIGN;
  const int es = feats_dtype == GD4D_BF16 ? 2 : 4;
  if (slice_stride_bytes % (4 * es)) return GD4D_EALIGN;
  for (int l = 0; l < L; ++l) {
    if (!level_ptrs[l]) return GD4D_EINVAL;
    if (reinterpret_cast<uintptr_t>(level_ptrs[l]) % (4 * es)) return GD4D_EALIGN;
    p.lvl_base[l] = static_cast<const char*>(level_ptrs[l]);
  }
  for (int l = L; l < 4; ++l) p.lvl_base[l] = p.lvl_base[0];
  p.slice_stride = slice_stride_bytes;
  p.hdr = static_cast<const int*>(plan);
  p.pair = reinterpret_cast<const uint2*>(static_cast<const char*>(plan) + plan_hdr_bytes(B, Q));
  p.order = query_order; p.agg = agg;
  p.BQ = B * Q; p.per_xcd = (B * Q + 7) / 8; p.cap_t = plan_cap_t(N, P);
  p.slice_lo = slice_lo; p.slice_n = slice_n;
  p.blk = p.per_xcd;                                // one block = the XCD's whole sector: plain slice-major
  return GD4D_OK;
}
}  // namespace gd4d

extern "C" int gd4d_cross_attn_agg_sliced_fwd(const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan,
                                              float* agg, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                              const int32_t* query_order, int slice_lo, int slice_n, void* stream) {
  using namespace gd4d;
  SlicedParams p{};
  if (int rc = fill_sliced_params(p, level_ptrs, slice_stride_bytes, plan, agg, B, N, Q, Hh, C, L, P, feats_dtype, query_order, slice_lo,
                                  slice_n))
    return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool bf16 = feats_dtype == GD4D_BF16;
  switch (Hh) {
    case 4: return bf16 ? launch_sliced<4, uint16_t>(p, L, s) : launch_sliced<4, float>(p, L, s);
    case 8: return bf16 ? launch_sliced<8, uint16_t>(p, L, s) : launch_sliced<8, float>(p, L, s);
    default: return bf16 ? launch_sliced<16, uint16_t>(p, L, s) : launch_sliced<16, float>(p, L, s);
  }
}

namespace gd4d {
template <int HH, typename VT>
static int launch_items(const ItemsParams& ip, int L, bool wide, hipStream_t s, CountArgs* ca = nullptr) {
  const SlicedParams& p = ip.s;
  const size_t lds = (size_t)HH * 4 * 8 * 80;                        // [HH][CH][8][GP]
  const dim3 grid(8 * ((p.per_xcd + p.blk - 1) / p.blk) * p.blk * p.slice_n);
  if (ca) {                                            // gather + record count in one launch (training forward)
    if (HH != 8 || L != 4 || sizeof(VT) != 4 || wide) return GD4D_EUNSUPPORTED;
    const int wgs = (ca->BQ * HH + HH - 1) / HH;       // a count workgroup = HH (position, head) waves
    ca->groups = (wgs + 7) / 8;
    const int ggroups = (int)(grid.x / 8);
    ca->every = ggroups / ca->groups;
    if (ca->every < 1) return GD4D_EUNSUPPORTED;
    hipLaunchKernelGGL((cross_attn_agg_items_count_kernel<8, 4, float, 6>), dim3(grid.x + 8 * ca->groups), dim3(64 * 8), lds, s, ip, *ca);
    return check_launch();
  }
  auto go = [&](auto kern) -> int {
    if (lds > 65536 && !allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(64 * HH), lds, s, ip);
    return check_launch();
  };
  constexpr int OCC = HH == 16 ? 4 : 6;                             // waves per SIMD
  if (wide) {                                                       // (a 64-bit address per load: 4 waves per SIMD, no spills)
    switch (L) {
      case 1: return go(cross_attn_agg_items_kernel<HH, 1, VT, 4, true>);
      case 2: return go(cross_attn_agg_items_kernel<HH, 2, VT, 4, true>);
      case 3: return go(cross_attn_agg_items_kernel<HH, 3, VT, 4, true>);
      default: return go(cross_attn_agg_items_kernel<HH, 4, VT, 4, true>);
    }
  }
  switch (L) {
    case 1: return go(cross_attn_agg_items_kernel<HH, 1, VT, OCC>);
    case 2: return go(cross_attn_agg_items_kernel<HH, 2, VT, OCC>);
    case 3: return go(cross_attn_agg_items_kernel<HH, 3, VT, OCC>);
    default: return go(cross_attn_agg_items_kernel<HH, 4, VT, OCC>);
  }
}

// geometry of an items-form gather: 32-bit offsets inside a level - bytes while every level spans < 4 GiB, units of 16 bytes
// otherwise (< 64 GiB; `wide`)
static int fill_items_geom(PyramidGeom& g, bool& wide, const int32_t* level_hw, const int64_t* cam_stride_bytes, int64_t pix_stride_bytes,
                    int rows, int L) {
  if (!level_hw || !cam_stride_bytes) return GD4D_EINVAL;
  if (pix_stride_bytes <= 0 || pix_stride_bytes >= (1ll << 31)) return GD4D_EINVAL;
  for (int l = 0; l < 4; ++l) { g.lvl_w[l] = 1; g.lvl_h[l] = 1; g.cam_stride[l] = 0; }
  wide = false;
  unsigned long long span_max = 0;
  for (int l = 0; l < L; ++l) {
    const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
    if (h <= 0 || w <= 0 || cam_stride_bytes[l] < 0) return GD4D_EINVAL;
    const unsigned long long span = (unsigned long long)(rows - 1) * (unsigned long long)cam_stride_bytes[l] +
                                    (unsigned long long)(h * w) * (unsigned long long)pix_stride_bytes;
    span_max = span > span_max ? span : span_max;
    if (span >= (1ull << 32) || cam_stride_bytes[l] >= (1ll << 32)) wide = true;
    if ((cam_stride_bytes[l] & 15) || (pix_stride_bytes & 15)) {
      if (span >= (1ull << 32)) return GD4D_EUNSUPPORTED;
    }
  }
  if (wide && ((pix_stride_bytes & 15) || span_max >= (1ull << 36))) return GD4D_EUNSUPPORTED;
  for (int l = 0; l < L; ++l) {
    if (wide && ((cam_stride_bytes[l] & 15) || (cam_stride_bytes[l] >> 4) >= (1ll << 32))) return GD4D_EUNSUPPORTED;
    g.lvl_w[l] = level_hw[2 * l + 1]; g.lvl_h[l] = level_hw[2 * l];
    g.cam_stride[l] = (unsigned)(wide ? cam_stride_bytes[l] >> 4 : cam_stride_bytes[l]);
  }
  g.pix_stride = (unsigned)(wide ? pix_stride_bytes >> 4 : pix_stride_bytes);
  return GD4D_OK;
}
}  // namespace gd4d

static int items_fwd_impl(const void* const* level_ptrs, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                         int64_t pix_stride_bytes, int64_t slice_stride_bytes, const void* plan, float* agg,
                         float* wsum, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                         const int32_t* query_order, int slice_lo, int slice_n, void* stream, gd4d::CountArgs* ca) {
  using namespace gd4d;
  ItemsParams ip{};
  if (int rc = fill_sliced_params(ip.s, level_ptrs, slice_stride_bytes, plan, agg, B, N, Q, Hh, C, L, P, feats_dtype, query_order,
                                  slice_lo, slice_n))
    return rc;
  bool wide = false;
  if (int rc = fill_items_geom(ip.g, wide, level_hw, cam_stride_bytes, pix_stride_bytes, B * N, L)) return rc;
  ip.item = reinterpret_cast<const float4*>(ip.s.pair);
  ip.wsum = wsum;
  ip.cap_i = plan_cap_items(N, P);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool bf16 = feats_dtype == GD4D_BF16;
  if (ca) return (Hh == 8 && !bf16) ? launch_items<8, float>(ip, L, wide, s, ca) : GD4D_EUNSUPPORTED;
  switch (Hh) {
    case 4: return bf16 ? launch_items<4, uint16_t>(ip, L, wide, s) : launch_items<4, float>(ip, L, wide, s);
    case 8: return bf16 ? launch_items<8, uint16_t>(ip, L, wide, s) : launch_items<8, float>(ip, L, wide, s);
    default: return bf16 ? launch_items<16, uint16_t>(ip, L, wide, s) : launch_items<16, float>(ip, L, wide, s);
  }
}

extern "C" int gd4d_cross_attn_agg_items_fwd(const void* const* level_ptrs, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                                             int64_t pix_stride_bytes, int64_t slice_stride_bytes, const void* plan, float* agg,
                                             float* wsum, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                             const int32_t* query_order, int slice_lo, int slice_n, void* stream) {
  return items_fwd_impl(level_ptrs, level_hw, cam_stride_bytes, pix_stride_bytes, slice_stride_bytes, plan, agg, wsum, B, N, Q, Hh, C,
                        L, P, feats_dtype, query_order, slice_lo, slice_n, stream, nullptr);
}

extern "C" int gd4d_cross_attn_agg_items_coarse_fwd(const void* const* level_ptrs, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                                                    int64_t pix_stride_bytes, int64_t slice_stride_bytes, const void* const* proj_ptrs,
                                                    const int64_t* proj_cam_stride_bytes, const void* plan, float* agg, float* wsum,
                                                    float* pagg, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                                    const int32_t* query_order, void* stream) {
  using namespace gd4d;
  if (!proj_ptrs || !proj_cam_stride_bytes || !pagg || !wsum) return GD4D_EINVAL;
  if (L != 4 || Hh != 8) return GD4D_EUNSUPPORTED;                   // (two fine + two coarse levels; a head = one 32-channel slice)
  ItemsParams ip{};
  if (int rc = fill_sliced_params(ip.s, level_ptrs, slice_stride_bytes, plan, agg, B, N, Q, Hh, C, L, P, feats_dtype, query_order, 0, kSlices))
    return rc;
  bool wide = false;
  if (int rc = fill_items_geom(ip.g, wide, level_hw, cam_stride_bytes, pix_stride_bytes, B * N, 2)) return rc;   // (spans of the fine levels)
  if (!aligned16(pagg)) return GD4D_EALIGN;
  ip.gc = ip.g;
  for (int l = 2; l < 4; ++l) {
    const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
    if (h <= 0 || w <= 0 || !proj_ptrs[l - 2] || proj_cam_stride_bytes[l - 2] < (int64_t)h * w * kChannels * 4) return GD4D_EINVAL;
    if (!aligned16(proj_ptrs[l - 2]) || (proj_cam_stride_bytes[l - 2] & 15)) return GD4D_EALIGN;
    const unsigned long long span = (unsigned long long)(B * N) * (unsigned long long)proj_cam_stride_bytes[l - 2];
    if (span >= (1ull << 32)) return GD4D_EUNSUPPORTED;
    ip.gc.lvl_w[l] = w; ip.gc.lvl_h[l] = h; ip.gc.cam_stride[l] = (unsigned)proj_cam_stride_bytes[l - 2];
    ip.proj_base[l] = static_cast<const char*>(proj_ptrs[l - 2]);
  }
  ip.gc.pix_stride = kChannels * 4;
  ip.proj_base[0] = ip.proj_base[1] = ip.proj_base[2];
  ip.item = reinterpret_cast<const float4*>(ip.s.pair);
  ip.wsum = wsum; ip.pagg = pagg;
  ip.cap_i = plan_cap_items(N, P);
  ip.s.slice_n = kSlices + 1;                                        // the walk's ninth phase: the projected rows
  const SlicedParams& p = ip.s;
  const size_t lds = (size_t)8 * 4 * 8 * 80;                         // [HH][CH][8][GP]
  const dim3 grid(8 * ((p.per_xcd + p.blk - 1) / p.blk) * p.blk * p.slice_n);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto go = [&](auto kern) -> int {
    hipLaunchKernelGGL(kern, grid, dim3(64 * 8), lds, s, ip);
    return check_launch();
  };
  if (feats_dtype == GD4D_BF16) return wide ? go(cross_attn_agg_items_coarse_kernel<8, uint16_t, 4, true>) : go(cross_attn_agg_items_coarse_kernel<8, uint16_t, 6>);
  return wide ? go(cross_attn_agg_items_coarse_kernel<8, float, 4, true>) : go(cross_attn_agg_items_coarse_kernel<8, float, 6>);
}

extern "C" int gd4d_cross_attn_agg_items_count_fwd(const void* const* level_ptrs, const int32_t* level_hw,
                                                   const int64_t* cam_stride_bytes, int64_t pix_stride_bytes,
                                                   int64_t slice_stride_bytes, const void* plan_items, float* agg, float* wsum, int B,
                                                   int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                                   const int32_t* query_order, const void* plan_pairs, int32_t* count, void* slots,
                                                   size_t slots_bytes, void* stream) {
  using namespace gd4d;
  if (!plan_pairs || !count || !slots || !level_hw || !cam_stride_bytes) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || L <= 0) return GD4D_EINVAL;
  if ((P != kPoints && P != 8) || L > 4 || N > 64 || B > 16 || Hh > kPlanHdr) return GD4D_EUNSUPPORTED;
  if (slots_bytes < (size_t)B * Q * Hh * plan_cap_t(N, P) * 64 * sizeof(uint2)) return GD4D_EWORKSPACE;
  CountArgs ca{};
  if (int rc = fill_chunks(ca.g, level_hw, cam_stride_bytes, pix_stride_bytes, B * N, L)) return rc;
  ca.hdr = static_cast<const int*>(plan_pairs);
  ca.pair = reinterpret_cast<const uint2*>(static_cast<const char*>(plan_pairs) + plan_hdr_bytes(B, Q));
  ca.count = count;
  ca.slots = static_cast<uint2*>(slots);
  ca.cap_t = plan_cap_t(N, P);
  ca.BQ = B * Q;
  return items_fwd_impl(level_ptrs, level_hw, cam_stride_bytes, pix_stride_bytes, slice_stride_bytes, plan_items, agg, wsum, B, N, Q, Hh,
                        C, L, P, feats_dtype, query_order, 0, 8, stream, &ca);
}

extern "C" int gd4d_pyramid_slice_planar_fwd(const void* const* feats, const int32_t* level_hw, void* out, int R, int C,
                                             int L, int in_dtype, int out_dtype, int max_cus, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !out || R <= 0 || C <= 0 || L <= 0) return GD4D_EINVAL;
  if (C != SP_C || L > GD4D_MAX_LEVELS || in_dtype != GD4D_F32) return GD4D_EUNSUPPORTED;
  if (out_dtype != GD4D_F32 && out_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (!aligned16(out)) return GD4D_EALIGN;
  SpParams p{};
  int s = 0, base = 0;
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    const int hw = level_hw[2 * l] * level_hw[2 * l + 1];
    p.in[l] = static_cast<const float*>(feats[l]); p.hw[l] = hw; p.start[l] = s; p.tiles[l] = (hw + SP_PX - 1) / SP_PX;
    p.tile_base[l] = base;
    s += hw;
    base += R * p.tiles[l];
  }
  for (int l = L; l <= GD4D_MAX_LEVELS; ++l) p.tile_base[l] = base;
  for (int l = L; l < GD4D_MAX_LEVELS; ++l) { p.tiles[l] = 1; p.hw[l] = 1; }
  p.out = out; p.R = R; p.L = L; p.S = s; p.total = base;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool ob = out_dtype == GD4D_BF16;
  const size_t tile_lds = (size_t)SP_PX * SP_PITCH * sizeof(float);
  auto go = [&](auto kern, int blocks, size_t lds) -> int {
    if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(SP_THREADS), lds, st, p);
    return check_launch();
  };
  if (max_cus > 0) {
    const int blocks = max_cus < base ? max_cus : base;
    return ob ? go(pyramid_slice_planar_kernel<true, true>, blocks, SP_LDS) : go(pyramid_slice_planar_kernel<false, true>, blocks, SP_LDS);
  }
  return ob ? go(pyramid_slice_planar_kernel<true, false>, base, tile_lds) : go(pyramid_slice_planar_kernel<false, false>, base, tile_lds);
}

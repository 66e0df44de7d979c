// The pyramid-gradient bookkeeping's first step as a device function: gd4d_cross_attn_sliced_bwd.hip launches it as a kernel of its
// own (gd4d_pyramid_grad_count), gd4d_cross_attn_sliced.hip beside the forward gather of a training step, in one launch
// (gd4d_cross_attn_agg_items_count_fwd).  Reference: see gd4d_cross_attn_sliced_bwd.hip.
#pragma once
#include "gd4d_common.h"
#include "gd4d_cross_attn_sliced.h"

namespace gd4d {

struct PgChunks {
  unsigned cam_stride[4];
  unsigned pix_stride;
  int lvl_w[4], lvl_h[4];
  int cws[4], chs[4];            // log2 of the chunk width / height
  int CW[4], CH[4];              // chunks across / down one camera row of level l
  int chunk_base[5];             // first chunk of level l; within a level (row, cy, cx)
  int total;
};

// One wave per (position, head) = ph, one pass of 64 pairs at a time: every pair with a non-zero weight gets a slot in its
// chunk's bucket.  Lanes with the same chunk are matched first (no memory traffic), their leader asks for the whole
// group's slots with one returning atomic; {chunk << 6 | pixel-in-chunk, slot} is parked in the plan's layout for the
// fill (which runs after the scan over the counts of ALL layers).
__device__ __forceinline__ void pyramid_grad_count_body(const int* __restrict__ hdr, const uint2* __restrict__ pair, int cap_t, int HH,
                                                        int BQ, const PgChunks& g, int* __restrict__ count, uint2* __restrict__ slots,
                                                        const int ph) {
  const int lane = threadIdx.x & 63;
  if (ph >= BQ * HH) return;
  const int pos = ph / HH, h = ph - pos * HH;
  const int M = hdr[pos * kPlanHdr + h];
  const int T = (M + 3) >> 2;
  const int l = lane & 3;
  const unsigned cs = g.cam_stride[l];
  const int W = g.lvl_w[l], cws = g.cws[l], chs = g.chs[l], CWl = g.CW[l], CHl = g.CH[l], cbase = g.chunk_base[l];
  const unsigned long long lt = (1ull << lane) - 1ull;
  const size_t prow = (size_t)ph * cap_t * 64 + lane;
  // four passes per round: their plan rows requested together, then the four rounds of matching, then the four returning
  // atomics in flight together (a pass at a time, every pass waited for its plan row and then for its own atomic's round
  // trip; two per round with the second row requested after the first match: 52 us per launch)
  constexpr int NP = 4;
  struct Match { int key, pxin, leader, rank, n; bool valid; };
  auto match = [&](int t, const uint2 pr) -> Match {
    Match m{0, 0, lane, 0, 0, false};
    if (t >= T) return m;
    m.valid = __uint_as_float(pr.y) != 0.f;
    const unsigned row = pr.x / cs;
    const unsigned pix = (pr.x - row * cs) / g.pix_stride;
    const int y = (int)pix / W, x = (int)pix - y * W;
    m.key = cbase + ((int)row * CHl + (y >> chs)) * CWl + (x >> cws);
    m.pxin = ((y & ((1 << chs) - 1)) << cws) | (x & ((1 << cws) - 1));
    unsigned long long rem = __ballot(m.valid);
    while (rem) {
      const int ld = __ffsll((long long)rem) - 1;
      const int k = __builtin_amdgcn_readlane(m.key, ld);
      const bool mine = m.valid && m.key == k;
      const unsigned long long mm = __ballot(mine);
      if (mine) { m.leader = ld; m.rank = __popcll(mm & lt); }
      if (lane == ld) m.n = __popcll(mm);
      rem &= ~mm;
    }
    return m;
  };
  auto finish = [&](int t, const Match& m, int base) {
    if (t >= T) return;
    base = __shfl(base, m.leader);
    slots[prow + (size_t)t * 64] = m.valid ? make_uint2(((unsigned)m.key << 6) | (unsigned)m.pxin, (unsigned)(base + m.rank))
                                           : make_uint2(0xffffffffu, 0xffffffffu);
  };
  for (int t = 0; t < T; t += NP) {
    uint2 pr[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) pr[i] = pair[prow + (size_t)min(t + i, T - 1) * 64];
    Match m[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) m[i] = match(t + i, pr[i]);
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    static_assert(NP == 4, "four returning atomics in flight");
    // (the four addresses exist before the first atomic is issued: computed between them, the compiler re-used the registers
    // of an atomic in flight and waited for it first - four round trips in turn)
    typedef __attribute__((address_space(1))) int gint;           // (global, not generic: global_atomic_add, not flat_)
    gint* a0 = (gint*)(count + m[0].key); gint* a1 = (gint*)(count + m[1].key);
    gint* a2 = (gint*)(count + m[2].key); gint* a3 = (gint*)(count + m[3].key);
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    if (m[0].valid && m[0].leader == lane) b0 = __hip_atomic_fetch_add(a0, m[0].n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m[1].valid && m[1].leader == lane) b1 = __hip_atomic_fetch_add(a1, m[1].n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m[2].valid && m[2].leader == lane) b2 = __hip_atomic_fetch_add(a2, m[2].n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m[3].valid && m[3].leader == lane) b3 = __hip_atomic_fetch_add(a3, m[3].n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int b[NP] = {b0, b1, b2, b3};
#pragma unroll
    for (int i = 0; i < NP; ++i) finish(t + i, m[i], b[i]);
  }
}


// chunk shapes: ~64 pixels on the two finest levels, a quarter per level beyond (records per pixel grow 4x per level)
static inline int fill_chunks(PgChunks& g, const int32_t* level_hw, const int64_t* cam_stride_bytes, int64_t pix_stride_bytes, int R, int L) {
  if (L <= 0 || L > 4 || R <= 0) return GD4D_EINVAL;
  long long base = 0;
  for (int l = 0; l < 4; ++l) { g.cam_stride[l] = 1; g.lvl_w[l] = 1; g.lvl_h[l] = 1; g.cws[l] = 0; g.chs[l] = 0; g.CW[l] = 1; g.CH[l] = 1; }
  const long long hw0 = (long long)level_hw[0] * level_hw[1];
  for (int l = 0; l < L; ++l) {
    const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
    if (h <= 0 || w <= 0) return GD4D_EINVAL;
    if (cam_stride_bytes) {
      if (cam_stride_bytes[l] <= 0 || cam_stride_bytes[l] >= (1ll << 32)) return GD4D_EINVAL;
      g.cam_stride[l] = (unsigned)cam_stride_bytes[l];
    }
    long long want = 256ll * h * w / (hw0 > 0 ? hw0 : 1);            // 64 x (pixels of this level / pixels of level 1)
    int px = 64;
    while (px > 8 && px > want) px >>= 1;                           // (8 waves of the reduce kernel: at least a pixel each)
    int chs = 1, cws = 0;
    while ((2 << cws) * 2 <= px) ++cws;                             // cw = px / 2, ch = 2
    g.lvl_w[l] = w; g.lvl_h[l] = h; g.cws[l] = cws; g.chs[l] = chs;
    g.CW[l] = (w + (1 << cws) - 1) >> cws; g.CH[l] = (h + (1 << chs) - 1) >> chs;
    g.chunk_base[l] = (int)base;
    base += (long long)R * g.CW[l] * g.CH[l];
    if (base >= (1ll << 25)) return GD4D_EUNSUPPORTED;
  }
  for (int l = L; l <= 4; ++l) g.chunk_base[l] = (int)base;
  if (cam_stride_bytes && (pix_stride_bytes <= 0 || pix_stride_bytes >= (1ll << 31))) return GD4D_EINVAL;
  g.pix_stride = cam_stride_bytes ? (unsigned)pix_stride_bytes : 1u;
  g.total = (int)base;
  return GD4D_OK;
}

}  // namespace gd4d

// What the channel-sliced kernels share: the layout of the plan (gd4d_cross_attn_sliced.hip writes it; the gather, the
// training backward in gd4d_cross_attn_sliced_bwd.hip and the pyramid-gradient kernels read it).
#pragma once
#include "gd4d_common.h"
#include "gd4d_cross_attn_shared.h"

namespace gd4d {

constexpr int kSlice = 32;                       // channels per slice: 128 bytes fp32 = one L2 line
constexpr int kSlices = kChannels / kSlice;      // 8
constexpr int kPlanHdr = 16;                     // ints per query in the plan header: item count per head

// How the pyramid the gather reads is addressed: offset of (camera row, level l, pixel) inside level l's base pointer
// = row * cam_stride[l] + pixel * pix_stride  (the plan stores these byte offsets).
struct PyramidGeom {
  unsigned cam_stride[4];       // bytes between camera rows of level l
  int lvl_w[4], lvl_h[4];
  unsigned pix_stride;          // bytes between pixels
};

// plan = [header: kPlanHdr ints per position, padded to 256 B][pairs: (position, head, pass < cap_t, 64) uint2]
// GD4D_CA_PLAN_ITEMS: [the same header][items: (position, head, item < cap_i) x {u, v, camera row, M}{level weight 0..3}];
// cap_i * 32 <= cap_t * 512, so gd4d_cross_attn_plan_bytes covers both forms
static inline int plan_cap_t(int N, int P) { return (N * P + 3) / 4; }      // passes of 4 items a head can need
static inline int plan_cap_items(int N, int P) { return (N * P + 15) & ~15; }  // GD4D_CA_PLAN_ITEMS: 32-byte records a head can need
static inline size_t plan_hdr_bytes(int B, int Q) { return (((size_t)B * Q * kPlanHdr * sizeof(int)) + 255) & ~(size_t)255; }

}  // namespace gd4d

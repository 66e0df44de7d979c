// Shared pieces of the fused cross-attention kernels (gd4d_cross_attn.hip: gather from projected values;
// gd4d_cross_attn_late.hip: aggregate the raw features per head, project afterwards).  Both translation units are built
// with -ffp-contract=off: project_entry() must reproduce the reference's un-fused fp32 arithmetic bit for bit.
#pragma once
#include "gd4d_common.h"

namespace gd4d {

struct CrossAttnParams {
  const void* value;
  const float* ref;
  const float* offsets;
  const float* attn_logits;
  const float* cam_logits;
  const float* lidar2img;
  float* out;
  uint8_t* mask_out;
  float* uv_out;
  const int32_t* order;   // optional permutation of [0, B*Q): locality order of the queries (gd4d_query_order_fwd)
  int B, N, Q, L, S, P;
  int raw_cam;         // 1: camera weights are the raw logits (Deform3DCrossAttnMP's neighbour pass), 0: sigmoid
  int head_major;      // value layout: 0 = (B*N, S, Hh, Dh) pixel-major, 1 = (B*N, Hh, S, Dh) head-major planes
  int lvl_h[GD4D_MAX_LEVELS];
  int lvl_w[GD4D_MAX_LEVELS];
  int lvl_start[GD4D_MAX_LEVELS];
  float rng_scale[3];  // float(double(hi) - double(lo))
  float rng_lo[3];     // float(lo)
  float img_h, img_w;
  float* agg;          // gd4d_cross_attn_agg_fwd only: (B*Q, Hh, C) per-head aggregates of the raw features
  float* wsum;         //                               (B*Q, Hh) sum of the in-bounds sampling weights per head
  const float* vp_w;   // gd4d_cross_attn_agg_fwd, optional: value_proj weight (C, C) / bias (C) applied to the aggregates in
  const float* vp_b;   //   the kernel's epilogue: out (B*Q, C) = W_h agg_h + b_h wsum_h  (agg / wsum may then be NULL)
  unsigned dbg_wrap;   // dev (GD4D_AGG_DBG_WRAP): pixel indices are ANDed with this mask (0 = off): an all-L2-hit run
};

constexpr int kPoints = 4;   // sampling points per head (reference configs: num_points=4)
constexpr int kChannels = 256;

// ---- phase A: bit-exact projection of one (camera, head, point) entry ------------------------
// Order of operations is the reference's torch-CPU arithmetic:
//   p = ref*scale + lo (two roundings); X = p + off; c = ((m0*X + m1*Y) + m2*Z) + m3;
//   u = (cx / max(cz, eps)) / W; v = (cy / max(cz, eps)) / H; all IEEE, no contraction.
__device__ __forceinline__ bool project_entry(const CrossAttnParams& p, const float* __restrict__ m,
                                              float X, float Y, float Z, float& u, float& v) {
  // NOTE: this translation unit is built with -ffp-contract=off; HIP's __fmul_rn/__fadd_rn are
  // plain operators that the compiler would otherwise contract into v_fma_f32 (seen in the ISA).
  const float eps = 1e-5f;
  const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
  const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
  const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
  bool vis = cz > eps;
  const float zc = fmaxf(cz, eps);
  u = (cx / zc) / p.img_w;       // IEEE-correct division (v_div_scale/fmas/fixup)
  v = (cy / zc) / p.img_h;
  vis = vis && (u > 0.f) && (u < 1.f) && (v > 0.f) && (v < 1.f);
  return vis;
}

template <typename VT> struct Quad;  // 4 consecutive channels of one pixel/head
template <> struct Quad<float> {
  static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
};
template <> struct Quad<uint16_t> {
  static __device__ __forceinline__ float4 load(const uint16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
  }
};

__device__ __forceinline__ void softmax_lp(const float* __restrict__ logits, int n, float* w) {
  float mx = logits[0];
  for (int i = 1; i < n; ++i) mx = fmaxf(mx, logits[i]);
  float sum = 0.f;
  for (int i = 0; i < n; ++i) { w[i] = expf(logits[i] - mx); sum += w[i]; }
  float inv = 1.0f / sum;
  for (int i = 0; i < n; ++i) w[i] *= inv;
}

}  // namespace gd4d

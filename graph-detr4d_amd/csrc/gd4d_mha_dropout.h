// Dropout of the self-attention probabilities (training): which (batch, head, query, key) elements are kept.
//
// nn.MultiheadAttention applies F.dropout to the softmax output (torch.nn.functional.multi_head_attention_forward;
// the reference's decoder sets attn_drop = 0.1, projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:74-78).
// The forward kernel (gd4d_self_attn.hip) and the two backward kernels (gd4d_train.hip) each regenerate the keep
// decision from the element's index and a 64-bit seed in device memory, so no (heads, Q, Q) mask is ever stored and a
// replayed hipGraph draws a new mask whenever the seed words are advanced on the device.  The decision is a function of
// (seed, element) only: tests rebuild the mask on the host (tests/test_dense_gpu.py) and compare against torch with
// that mask.
#pragma once
#include <cstdint>

namespace gd4d {

// id of element (b, h, q, key = 0): ((b H + h) Lq + q) Lk; the entry points refuse B H Lq Lk >= 2^32
__host__ __device__ __forceinline__ uint32_t mha_drop_row(int b, int h, int q, int H, int Lq, int Lk) {
  return (uint32_t)(((uint32_t)(b * H + h) * (uint32_t)Lq + (uint32_t)q) * (uint32_t)Lk);
}

// kept  <=>  mix(seed, id) >= thresh, thresh = round(p 2^32): two rounds of multiply / xor-shift (the murmur3 finaliser
// with the second seed word folded in between)
__host__ __device__ __forceinline__ bool mha_drop_keep(uint32_t seed_lo, uint32_t seed_hi, uint32_t id, uint32_t thresh) {
  uint32_t x = id ^ seed_lo;
  x *= 0x9E3779B1u; x ^= x >> 16;
  x += seed_hi;
  x *= 0x85EBCA6Bu; x ^= x >> 13;
  x *= 0xC2B2AE35u; x ^= x >> 16;
  return x >= thresh;
}

inline uint32_t mha_drop_thresh(float p) {
  const double t = (double)p * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(t + 0.5));
}

}  // namespace gd4d

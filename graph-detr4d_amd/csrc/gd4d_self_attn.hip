// gd4d_mha_core_fwd: softmax(q k^T / sqrt(d) [+ mask]) v for the decoder self-attention, gfx950.
//
// Reference: third-party mmcv MultiheadAttention -> nn.MultiheadAttention (config call site
// projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:74-78); H-DETR passes a
// boolean self-attention mask (h_detr3d_transformer.py:149-157).  The packed in-projection and the
// out-projection are gd4d_linear_fwd launches; this kernel is the part in between and never
// materialises the (heads, Q, Q) score tensor the reference writes and re-reads.
//
// Everything is fp32 on v_mfma_f32_16x16x4_f32 (exact fp32 products).  One workgroup = 8 waves =
// one (batch, head, 16-query tile); the waves split the keys 8 ways (flash-decoding style) and merge
// their (max, sum, O) triples through LDS.
//
// Layout trick: the scores are computed TRANSPOSED, S^T = K Q^T (MFMA rows = keys, cols = queries),
// with MFMA row rho of a 16-key tile holding key (rho>>2) + 4*(rho&3).  The C/D layout then leaves
// lane (query i = lane&15, g = lane>>4), register r with key g + 4r - which is exactly the
// B-operand layout (k = g + 4s at step s) of the second product O^T = V^T P^T.  The probabilities
// never move between lanes; the softmax reductions are 4 in-lane values + two xor-shuffles.
#include <stdlib.h>

#include "gd4d_common.h"
#include "gd4d_mha_dropout.h"
#include "gd4d_mha_body.h"

namespace gd4d {


GD4D_TRACE_UNIT(mha)

// DROP: each probability is kept with chance 1 - p and scaled by 1 / (1 - p) before it multiplies v, as F.dropout on the
// softmax output inside nn.MultiheadAttention; the normaliser (and the saved log-sum-exp) are those of the full softmax.
// MASK: 0 none, 1 bool, 2 additive float (compile-time: the unmasked kernel of the decoder carries neither the registers nor
// the branches).  A tile's four mask entries are requested with its K rows and V columns - read where they are used, per
// score, each was a branch with its own load and wait (28.7 us per layer with H-DETR's mask against 20.8 without).
template <bool DROP, int MASK>
__global__ __launch_bounds__(64 * MHA_WAVES) void mha_core_kernel(const MhaParams p) {
  trace_mark(g_trace_mha, 2ull);
  __shared__ float s_m[MHA_WAVES][16];
  __shared__ float s_l[MHA_WAVES][16];
  __shared__ float s_o[MHA_WAVES][MHA_D][17];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = lane & 15;            // query column of this lane
  const int g = lane >> 4;             // lane group
  const int q0 = blockIdx.x * 16;
  const int h = blockIdx.y, b = blockIdx.z;
  const float NEG_INF = -__builtin_inff();
  // The softmax runs in base 2: q is scaled by scale * log2(e), so a probability is ONE v_exp_f32 (exp2) instead of expf's
  // range reduction + exp2 + ldexp (~15 instructions, five times per tile of keys: the kernel issued 195 vector instructions
  // per tile and was bound by them - 5.0 M per launch = 8 us of the 13.4, profiles/r04_pmc_step_inflight1.txt).  The saved
  // log-sum-exp is converted back to natural units; an additive mask is scaled like the scores.
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const float qscale = p.scale * LOG2E;

  // Q^T as B operand: lane (col = query qi, g) holds q[qi][8g + s] at k-step s (any k<->dim map works
  // as long as A uses the same one).  Rows beyond Lq are clamped (their outputs are never written).
  float qf[8];
  {
    const int qrow = min(q0 + qi, p.Lq - 1);
    const float* src = p.q + ((size_t)qrow * p.B + b) * p.ldq + h * MHA_D + 8 * g;
    const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
    qf[0] = a.x * qscale; qf[1] = a.y * qscale; qf[2] = a.z * qscale; qf[3] = a.w * qscale;
    qf[4] = c.x * qscale; qf[5] = c.y * qscale; qf[6] = c.z * qscale; qf[7] = c.w * qscale;
  }
  const int krho = (qi >> 2) + 4 * (qi & 3);     // key (within a tile) held by MFMA row rho = lane&15

  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};    // O^T rows d = 4g + r, and 16 + 4g + r
  float m = NEG_INF, l = 0.f;                                     // running max (per query), partial sum (per lane)

  const int ntiles = (p.Lk + 15) / 16;
  const size_t hoff = (size_t)h * MHA_D;
  uint32_t seed_lo = 0, seed_hi = 0, drop_row = 0;
  if (DROP) {
    seed_lo = p.seed[0]; seed_hi = p.seed[1];
    drop_row = mha_drop_row(b, h, min(q0 + qi, p.Lq - 1), p.H, p.Lq, p.Lk);
  }
  // A wave walks its tiles of keys in chunks of MHA_PF tiles whose K rows and V columns are requested together.  Measured
  // at 900 x 900 (ms per decoder step): MHA_PF = 1 (70 registers, three workgroups per compute unit) 1.759, 2 (96) 1.760,
  // 4 (128) 1.762, 8 (220 registers, one workgroup per unit) 1.827; one tile of look-ahead across iterations (the first
  // version) 1.778.  q / k / v arrive cold from another XCD (~2 us per round trip), but other workgroups on the unit hide
  // that better than a deeper request queue in this one does.
  const size_t mask_row = (size_t)min(q0 + qi, p.Lq - 1) * p.Lk;
  auto fetch = [&](int kt, float4& a, float4& c, float* vdst, float* mdst) {
    const int kbase = kt * 16;
    if (MASK) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t mi = mask_row + min(kbase + g + 4 * r, p.Lk - 1);
        if (MASK == 1) mdst[r] = static_cast<const uint8_t*>(p.mask)[mi] ? 1.f : 0.f;
        else mdst[r] = static_cast<const float*>(p.mask)[mi];
      }
    }
    const int krow = min(kbase + krho, p.Lk - 1);
    const float* src = p.k + ((size_t)krow * p.B + b) * p.ldk + hoff + 8 * g;
    a = *reinterpret_cast<const float4*>(src);
    c = *reinterpret_cast<const float4*>(src + 4);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int vrow = min(kbase + g + 4 * st, p.Lk - 1);      // out-of-range keys get probability 0
      const float* vs = p.v + ((size_t)vrow * p.B + b) * p.ldv + hoff + qi;
      vdst[2 * st] = vs[0];
      vdst[2 * st + 1] = vs[16];
    }
  };
  for (int base = wave; base < ntiles; base += MHA_WAVES * MHA_PF) {
    float4 ka[MHA_PF], kc[MHA_PF];
    float vv[MHA_PF][8];
    float mk[MHA_PF][4];
#pragma unroll
    for (int i = 0; i < MHA_PF; ++i) fetch(min(base + i * MHA_WAVES, ntiles - 1), ka[i], kc[i], vv[i], mk[i]);
#pragma unroll
    for (int i = 0; i < MHA_PF; ++i) {
      const int kt = base + i * MHA_WAVES;
      if (kt >= ntiles) break;                                   // wave-uniform
      const int kbase = kt * 16;
      // ---- S^T = K Q^T ----
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].x, qf[0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].y, qf[1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].z, qf[2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i].w, qf[3], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[i].x, qf[4], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[i].y, qf[5], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[i].z, qf[6], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[i].w, qf[7], s, 0, 0, 0);
      // lane (qi, g) reg r  <->  key kbase + g + 4r
      float sc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kbase + g + 4 * r;
        float val = s[r];
        if (key >= p.Lk) {
          val = NEG_INF;
        } else if (MASK == 1) {
          if (mk[i][r] != 0.f) val = NEG_INF;
        } else if (MASK == 2) {
          val += mk[i][r] * LOG2E;
        }
        sc[r] = val;
      }
      float tmax = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m, tmax);
      // all keys so far masked: keep everything at zero weight without creating NaN here
      const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
      const float corr = __builtin_amdgcn_exp2f(m - m_use);           // m = -inf -> 0
      float pr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(sc[r] - m_use);
      l = l * corr + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
      m = m_new;
      if (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pr[r] = mha_drop_keep(seed_lo, seed_hi, drop_row + (uint32_t)(kbase + g + 4 * r), p.drop_thresh) ? pr[r] * p.inv_keep : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { o0[r] *= corr; o1[r] *= corr; }
      // ---- O^T += V^T P^T : A = V^T[d = lane&15 (+16)][key = kbase + g + 4s], B = P^T = pr[s] ----
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[i][2 * st], pr[st], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[i][2 * st + 1], pr[st], o1, 0, 0, 0);
      }
    }
  }

  // ---- merge the key-slices ----
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (g == 0) { s_m[wave][qi] = m; s_l[wave][qi] = l; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s_o[wave][4 * g + r][qi] = o0[r];
    s_o[wave][16 + 4 * g + r][qi] = o1[r];
  }
  __syncthreads();
  for (int e = tid; e < 16 * MHA_D; e += 64 * MHA_WAVES) {
    const int i = e / MHA_D, d = e % MHA_D;                    // consecutive threads -> consecutive d
    if (q0 + i >= p.Lq) continue;
    float mm = s_m[0][i];
#pragma unroll
    for (int w = 1; w < MHA_WAVES; ++w) mm = fmaxf(mm, s_m[w][i]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < MHA_WAVES; ++w) {
      const float f = __builtin_amdgcn_exp2f(s_m[w][i] - mm);  // fully masked row: -inf - -inf = NaN, as ATen
      num += s_o[w][d][i] * f;
      den += s_l[w][i] * f;
    }
    p.out[((size_t)(q0 + i) * p.B + b) * p.ldo + h * MHA_D + d] = num / den;
    if (p.lse && d == 0) p.lse[((size_t)(q0 + i) * p.B + b) * p.H + h] = mm * LN2 + logf(den);   // natural units; saved for gd4d_mha_core_bwd
  }
  trace_mark(g_trace_mha, 0x82ull);
}


// The split-bf16 form (inference; training in eval mode: it also writes the log-sum-exp): gd4d_mha_body.h
template <int MASK>
__global__ __launch_bounds__(64 * MHA_WAVES) void mha_core_bf16x3_kernel(const MhaParams p) {
  trace_mark(g_trace_mha, 2ull);
  __shared__ MhaShared sh;
#ifndef MHA_AHEAD_N
#define MHA_AHEAD_N 1          // (4 = all of a wave's rows up front: the fused launch's form; measured for this kernel too: see docs)
#endif
  mha_core_bf16x3_body<MASK, MHA_AHEAD_N>(p, blockIdx.x, blockIdx.y, blockIdx.z, sh);
  trace_mark(g_trace_mha, 0x82ull);
}

// ... with K and V as pre-split bf16 planes (written by the in-projection: GD4D_CHAIN_SPLIT_KV); batch 1
template <int MASK, bool DROP>
__global__ __launch_bounds__(64 * MHA_WAVES) void mha_core_presplit_kernel(const MhaParams p) {
  trace_mark(g_trace_mha, 2ull);
  __shared__ MhaShared sh;
  mha_core_bf16x3_body<MASK, 1, true, DROP>(p, blockIdx.x, blockIdx.y, 0, sh);
  trace_mark(g_trace_mha, 0x82ull);
}

}  // namespace gd4d

extern "C" void gd4d_trace_set_mha(unsigned long long* p) { gd4d::trace_set_mha(p); }

extern "C" int gd4d_mha_core_fwd(const float* q, const float* k, const float* v, const void* mask,
                                 float* out, int Lq, int Lk, int B, int H, int D, int ldq, int ldk,
                                 int ldv, int ldo, int mask_kind, float scale, float* lse, float drop_p, const void* seed,
                                 void* stream) {
  using namespace gd4d;
  if (!q || !k || !v || !out || Lq <= 0 || Lk <= 0 || B <= 0 || H <= 0) return GD4D_EINVAL;
  if (!(drop_p >= 0.f && drop_p < 1.f) || (drop_p > 0.f && !seed)) return GD4D_EINVAL;
  if (drop_p > 0.f && (double)B * H * Lq * Lk >= 4294967296.0) return GD4D_EUNSUPPORTED;   // element ids are 32 bits
  if (D != MHA_D || mask_kind < 0 || mask_kind > 2 || (mask_kind && !mask)) return GD4D_EUNSUPPORTED;
  if (ldq < H * D || ldk < H * D || ldv < H * D || ldo < H * D) return GD4D_EINVAL;
  if (!aligned16(q) || !aligned16(k) || (ldq % 4) || (ldk % 4)) return GD4D_EALIGN;
  MhaParams p{q, k, v, mask, out, lse, Lq, Lk, B, H, ldq, ldk, ldv, ldo, mask_kind, scale,
              static_cast<const uint32_t*>(seed), mha_drop_thresh(drop_p), 1.f / (1.f - drop_p)};
  const dim3 grid((Lq + 15) / 16, H, B), block(64 * MHA_WAVES);
  hipStream_t st = static_cast<hipStream_t>(stream);
  static const bool fp32_only = [] { const char* e = getenv("GD4D_MHA_FP32"); return e && e[0] == '1'; }();
  if (drop_p == 0.f && !fp32_only) {                  // no dropout: split-bf16 x3 products (GD4D_MHA_FP32=1: the fp32 kernel)
    if (mask_kind == 0) hipLaunchKernelGGL((mha_core_bf16x3_kernel<0>), grid, block, 0, st, p);
    else if (mask_kind == 1) hipLaunchKernelGGL((mha_core_bf16x3_kernel<1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mha_core_bf16x3_kernel<2>), grid, block, 0, st, p);
  } else if (drop_p > 0.f) {
    if (mask_kind == 0) hipLaunchKernelGGL((mha_core_kernel<true, 0>), grid, block, 0, st, p);
    else if (mask_kind == 1) hipLaunchKernelGGL((mha_core_kernel<true, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mha_core_kernel<true, 2>), grid, block, 0, st, p);
  } else {
    if (mask_kind == 0) hipLaunchKernelGGL((mha_core_kernel<false, 0>), grid, block, 0, st, p);
    else if (mask_kind == 1) hipLaunchKernelGGL((mha_core_kernel<false, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mha_core_kernel<false, 2>), grid, block, 0, st, p);
  }
  return check_launch();
}

extern "C" int gd4d_mha_core_presplit_fwd(const float* q, const void* k_planes, const void* v_planes, float* out, int L, int H, int D,
                                          int ldq, int ldo, long long k_plane_stride, long long v_plane_stride, const void* mask,
                                          int mask_kind, float scale, float* lse, float drop_p, const void* seed, void* stream) {
  using namespace gd4d;
  if (!q || !k_planes || !v_planes || !out || L <= 0 || H <= 0) return GD4D_EINVAL;
  if (!(drop_p >= 0.f && drop_p < 1.f) || (drop_p > 0.f && !seed)) return GD4D_EINVAL;
  if (drop_p > 0.f && (double)H * L * L >= 4294967296.0) return GD4D_EUNSUPPORTED;   // element ids are 32 bits
  if (D != MHA_D || mask_kind < 0 || mask_kind > 2 || (mask_kind && !mask)) return GD4D_EUNSUPPORTED;
  const long long tiles = (L + 15) / 16, steps = (L + 31) / 32;
  if (ldq < H * D || ldo < H * D || k_plane_stride < H * tiles * 512 || v_plane_stride < H * steps * 1024 ||
      (k_plane_stride & 7) || (v_plane_stride & 7))
    return GD4D_EINVAL;
  if (!aligned16(q) || (ldq % 4) || !aligned16(k_planes) || !aligned16(v_planes)) return GD4D_EALIGN;
  MhaParams p{q, nullptr, nullptr, mask, out, lse, L, L, 1, H, ldq, 0, 0, ldo, mask_kind, scale,
              static_cast<const uint32_t*>(seed), mha_drop_thresh(drop_p), 1.f / (1.f - drop_p),
              static_cast<const unsigned short*>(k_planes), static_cast<const unsigned short*>(v_planes), k_plane_stride, v_plane_stride};
  const dim3 grid((L + 15) / 16, H, 1), block(64 * MHA_WAVES);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (drop_p > 0.f) {
    if (mask_kind == 0) hipLaunchKernelGGL((mha_core_presplit_kernel<0, true>), grid, block, 0, st, p);
    else if (mask_kind == 1) hipLaunchKernelGGL((mha_core_presplit_kernel<1, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mha_core_presplit_kernel<2, true>), grid, block, 0, st, p);
  } else {
    if (mask_kind == 0) hipLaunchKernelGGL((mha_core_presplit_kernel<0, false>), grid, block, 0, st, p);
    else if (mask_kind == 1) hipLaunchKernelGGL((mha_core_presplit_kernel<1, false>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mha_core_presplit_kernel<2, false>), grid, block, 0, st, p);
  }
  return check_launch();
}

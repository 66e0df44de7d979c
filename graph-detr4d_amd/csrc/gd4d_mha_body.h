// The split-bf16 self-attention core as a device function: gd4d_self_attn.hip launches it as a kernel of its own,
// gd4d_rowchain.hip as the attention workgroups of a fused [attention | chain A] launch (gd4d_row_chain_mha_fwd).
// Reference: see gd4d_self_attn.hip.
#pragma once
#include "gd4d_common.h"
#include "gd4d_mha_dropout.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct MhaParams {
  const float* q; const float* k; const float* v;   // row (l*B + b), row strides ldq/ldk/ldv, head h at +32*h
  const void* mask;                                  // (Lq, Lk) uint8 (nonzero = masked) or float additive
  float* out;                                        // (Lq*B, heads*32)
  float* lse;                                        // optional (Lq, B, heads): log sum exp of the scaled, masked scores
  int Lq, Lk, B, H, ldq, ldk, ldv, ldo, mask_kind;   // 0 none, 1 bool, 2 float
  float scale;
  const uint32_t* seed;                              // dropout of the probabilities (training): two words, see mha_dropout.h
  uint32_t drop_thresh; float inv_keep;
  // pre-split operands (gd4d_mha_core_presplit_fwd; written by the in-projection's epilogue, GD4D_CHAIN_SPLIT_KV; B = 1), bf16, in
  // the MFMA operand layout itself - a wave's load of a fragment is 1 KB contiguous:
  const unsigned short* ksp;   // K: [plane hi, lo][head][tile of 16 keys][lane = 16 (c / 8) + key % 16][8 channels 8 (c / 8) ..]
  const unsigned short* vsp;   // V: [plane][head][step of 32 keys][half = d / 16][lane = 16 g + d % 16][j] <-> key 32 s + 16 (j >> 2) + 4 g + (j & 3)
  long long ks_plane, vs_plane;   // elements per plane
};

constexpr int MHA_D = 32;
#ifndef MHA_WAVES_N
#define MHA_WAVES_N 8
#endif
constexpr int MHA_WAVES = MHA_WAVES_N;
#ifndef MHA_PF_N
#define MHA_PF_N 1
#endif
constexpr int MHA_PF = MHA_PF_N;       // tiles of keys a wave requests ahead (one chunk)

// ---------------------------------------------------------------------------------------------------------------
// Inference form (no dropout, no saved log-sum-exp): both products on v_mfma_f32_16x16x32_bf16 with split operands
// (x = hi + lo, a b ~= a_hi b_hi + a_lo b_hi + a_hi b_lo, fp32 accumulation: ~2^-16 relative per product - the arithmetic of
// the row chains' GEMMs that feed and drain this kernel).  After the base-2 softmax the fp32 kernel above was bound by its
// matrix pipe: 416 k v_mfma_f32_16x16x4_f32 of 32 cycles = 5.4 us of the 11.7; the same contractions are 12 bf16 MFMAs of
// 16 cycles per 32 keys here (1.0 us).  Training keeps the fp32 kernel (its backward recomputes the probabilities in fp32).
//
// A wave takes 32 keys per step: two score tiles S^T = K Q^T (MFMA rows = keys kbase + 16 t + rho, columns = queries).  The
// C/D layout leaves lane (query qi, g) the scores of keys kbase + 16 t + 4 g + r; numbering the k index of the second product
// O^T = V^T P^T as 8 g + j <-> (t = j >> 2, r = j & 3) makes those eight probabilities exactly the lane's B operand: they
// never move between lanes (the trick of the fp32 kernel, for the 32-deep instruction).
typedef __attribute__((ext_vector_type(8))) __bf16 mha_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned mha_u4;

__device__ __forceinline__ unsigned mha_cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}
__device__ __forceinline__ void mha_split8(const float* v, mha_u4& h, mha_u4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = mha_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    ll[i] = mha_cvt_pk_bf16(v[2 * i] - __uint_as_float(hh[i] << 16), v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u));
  }
  h = mha_u4{hh[0], hh[1], hh[2], hh[3]};
  l = mha_u4{ll[0], ll[1], ll[2], ll[3]};
}
__device__ __forceinline__ f32x4 mha_mfma3(mha_u4 ah, mha_u4 al, mha_u4 bh, mha_u4 bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mha_bf16x8, al), __builtin_bit_cast(mha_bf16x8, bh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mha_bf16x8, ah), __builtin_bit_cast(mha_bf16x8, bl), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mha_bf16x8, ah), __builtin_bit_cast(mha_bf16x8, bh), acc, 0, 0, 0);
  return acc;
}

// LDS of one workgroup (the kernel's own static arrays, or a piece of the fused launch's dynamic allocation)
struct MhaShared {
  float m[MHA_WAVES][16];
  float l[MHA_WAVES][16];
  float o[MHA_WAVES][MHA_D][17];
};

// One workgroup (8 waves) = one (16-query tile qblock, head h, batch b); the stores of `out` are the last thing it issues.
// DROP (training, modules in train mode): the probabilities are dropped as in the fp32 kernel of gd4d_self_attn.hip - same element
// ids, same hash, the normaliser and the saved log-sum-exp those of the full softmax.
template <int MASK, int AHEAD = 1, bool PRE = false, bool DROP = false>
__device__ __forceinline__ void mha_core_bf16x3_body(const MhaParams& p, const int qblock, const int h, const int b, MhaShared& sh) {
  float (&s_m)[MHA_WAVES][16] = sh.m;
  float (&s_l)[MHA_WAVES][16] = sh.l;
  float (&s_o)[MHA_WAVES][MHA_D][17] = sh.o;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = lane & 15;            // query column of this lane (score / output tiles); key row rho of a K tile; channel of a V^T tile
  const int g = lane >> 4;             // lane group
  const int q0 = qblock * 16;
  const float NEG_INF = -__builtin_inff();
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const float qscale = p.scale * LOG2E;                       // base-2 softmax, as above

  mha_u4 qh, ql;                                              // Q^T as B operand: lane (query qi, g) holds q[qi][8 g .. 8 g + 7]
  {
    const int qrow = min(q0 + qi, p.Lq - 1);
    const float* src = p.q + ((size_t)qrow * p.B + b) * p.ldq + h * MHA_D + 8 * g;
    const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
    const float qf[8] = {a.x * qscale, a.y * qscale, a.z * qscale, a.w * qscale, c.x * qscale, c.y * qscale, c.z * qscale, c.w * qscale};
    mha_split8(qf, qh, ql);
  }
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};    // O^T rows d = 4g + r, and 16 + 4g + r
  float m = NEG_INF, l = 0.f;
  uint32_t seed_lo = 0, seed_hi = 0, drop_row = 0;
  if (DROP) {
    seed_lo = p.seed[0]; seed_hi = p.seed[1];
    drop_row = mha_drop_row(b, h, min(q0 + qi, p.Lq - 1), p.H, p.Lq, p.Lk);
  }
  const int nsteps = (p.Lk + 31) / 32;
  const size_t hoff = (size_t)h * MHA_D;
  const size_t mask_row = (size_t)min(q0 + qi, p.Lq - 1) * p.Lk;
  // One step of look-ahead (MHA_LOOKAHEAD, default on): q / k / v arrive cold from other XCDs (~2 us per round trip) and a wave
  // has only ~4 steps, so "request, wait, compute" per step was a chain of four exposed round trips; the next step's rows are
  // requested before this step's are consumed.
  // PRE: K and V arrive as the MFMA operands themselves (bf16 hi / lo planes): 8 loads of 16 / 8 bytes per step instead of 4
  // float4 + 16 dwords, and none of the 112 conversion instructions per step (two thirds of the kernel's splits)
  struct StepData { float4 ka[2], kc[2]; float v0[8], v1[8], mk[8]; mha_u4 pkh[2], pkl[2], pvh[2], pvl[2]; };
  auto fetch = [&](int kt, StepData& d) {
    const int kbase = min(kt, nsteps - 1) * 32;
    if (PRE) {
      const int tiles = (p.Lk + 15) / 16;
#pragma unroll
      for (int t = 0; t < 2; ++t) {                          // (a tile past the end: the last one again - its scores are -inf)
        const unsigned short* src = p.ksp + (((size_t)h * tiles + min(kbase / 16 + t, tiles - 1)) * 64 + lane) * 8;
        d.pkh[t] = *reinterpret_cast<const mha_u4*>(src);
        d.pkl[t] = *reinterpret_cast<const mha_u4*>(src + p.ks_plane);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const unsigned short* src = p.vsp + ((((size_t)h * nsteps + kbase / 32) * 2 + half) * 64 + lane) * 8;
        d.pvh[half] = *reinterpret_cast<const mha_u4*>(src);
        d.pvl[half] = *reinterpret_cast<const mha_u4*>(src + p.vs_plane);
      }
      if (MASK) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const size_t mi = mask_row + min(kbase + 16 * (j >> 2) + 4 * g + (j & 3), p.Lk - 1);
          if (MASK == 1) d.mk[j] = static_cast<const uint8_t*>(p.mask)[mi] ? 1.f : 0.f;
          else d.mk[j] = static_cast<const float*>(p.mask)[mi];
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int krow = min(kbase + 16 * t + qi, p.Lk - 1);
      const float* src = p.k + ((size_t)krow * p.B + b) * p.ldk + hoff + 8 * g;
      d.ka[t] = *reinterpret_cast<const float4*>(src);
      d.kc[t] = *reinterpret_cast<const float4*>(src + 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = kbase + 16 * (j >> 2) + 4 * g + (j & 3);
      const int vrow = min(key, p.Lk - 1);                     // out-of-range keys get probability 0
      const float* vs = p.v + ((size_t)vrow * p.B + b) * p.ldv + hoff + qi;
      d.v0[j] = vs[0];
      d.v1[j] = vs[16];
      if (MASK) {
        const size_t mi = mask_row + vrow;
        if (MASK == 1) d.mk[j] = static_cast<const uint8_t*>(p.mask)[mi] ? 1.f : 0.f;
        else d.mk[j] = static_cast<const float*>(p.mask)[mi];
      }
    }
  };
#ifndef MHA_LOOKAHEAD
#define MHA_LOOKAHEAD 1
#endif
  auto step = [&](const int kt, const StepData& d) {
    const int kbase = kt * 32;
    const float4 (&ka)[2] = d.ka;
    const float4 (&kc)[2] = d.kc;
    const float (&v0)[8] = d.v0;
    const float (&v1)[8] = d.v1;
    const float (&mk)[8] = d.mk;
    // ---- S^T = K Q^T, two tiles of 16 keys ----
    float sc[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float kf[8] = {ka[t].x, ka[t].y, ka[t].z, ka[t].w, kc[t].x, kc[t].y, kc[t].z, kc[t].w};
      mha_u4 kh, kl;
      if (PRE) { kh = d.pkh[t]; kl = d.pkl[t]; }
      else mha_split8(kf, kh, kl);
      const f32x4 st = mha_mfma3(kh, kl, qh, ql, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r) {                            // lane (qi, g) reg r <-> key kbase + 16 t + 4 g + r
        const int key = kbase + 16 * t + 4 * g + r;
        float val = st[r];
        if (key >= p.Lk) {
          val = NEG_INF;
        } else if (MASK == 1) {
          if (mk[4 * t + r] != 0.f) val = NEG_INF;
        } else if (MASK == 2) {
          val += mk[4 * t + r] * LOG2E;
        }
        sc[4 * t + r] = val;
      }
    }
    float tmax = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m, tmax);
    const float m_use = (m_new == NEG_INF) ? 0.f : m_new;      // all keys so far masked: zero weights, no NaN here
    const float corr = __builtin_amdgcn_exp2f(m - m_use);      // m = -inf -> 0
    float pr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pr[j] = __builtin_amdgcn_exp2f(sc[j] - m_use);
    l = l * corr + (((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7])));
    m = m_new;
    if (DROP) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        pr[j] = mha_drop_keep(seed_lo, seed_hi, drop_row + (uint32_t)(kbase + 16 * (j >> 2) + 4 * g + (j & 3)), p.drop_thresh)
                    ? pr[j] * p.inv_keep : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { o0[r] *= corr; o1[r] *= corr; }
    // ---- O^T += V^T P^T: A = V^T[d = lane & 15 (+ 16)][k = 8 g + j], B = P^T = this lane's own eight probabilities ----
    mha_u4 ph, pl, vh, vl;
    mha_split8(pr, ph, pl);
    if (PRE) { vh = d.pvh[0]; vl = d.pvl[0]; }
    else mha_split8(v0, vh, vl);
    o0 = mha_mfma3(vh, vl, ph, pl, o0);
    if (PRE) { vh = d.pvh[1]; vl = d.pvl[1]; }
    else mha_split8(v1, vh, vl);
    o1 = mha_mfma3(vh, vl, ph, pl, o1);
  };
  if (AHEAD > 1) {
    // the fused launch's form (gd4d_row_chain_mha_fwd): ONE workgroup per compute unit there, nobody else's work hides a round
    // trip - with one step of look-ahead a wave waited ~1.5 us per step for rows (9.1 us per workgroup for 4.5 us of issue).
    // All rows of AHEAD steps are requested before the first is consumed.
    StepData d[AHEAD];
    for (int base = wave; base < nsteps; base += MHA_WAVES * AHEAD) {
#pragma unroll
      for (int i = 0; i < AHEAD; ++i) fetch(base + MHA_WAVES * i, d[i]);        // (clamped: past the end = the last step again)
#pragma unroll
      for (int i = 0; i < AHEAD; ++i)
        if (base + MHA_WAVES * i < nsteps) step(base + MHA_WAVES * i, d[i]);
    }
  } else {
    StepData cur, nxt;
    if (wave < nsteps) fetch(wave, cur);
    for (int kt = wave; kt < nsteps; kt += MHA_WAVES) {
      if (MHA_LOOKAHEAD) {
        if (kt + MHA_WAVES < nsteps) fetch(kt + MHA_WAVES, nxt);
      } else if (kt != wave) {
        fetch(kt, cur);
      }
      step(kt, cur);
      if (MHA_LOOKAHEAD) cur = nxt;
    }
  }
  // ---- merge the key-slices (as the fp32 kernel) ----
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
    const int i = e / MHA_D, d = e % MHA_D;
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
    if (p.lse && d == 0) p.lse[((size_t)(q0 + i) * p.B + b) * p.H + h] = mm * LN2 + logf(den);   // a training step's forward: for gd4d_mha_core_bwd
  }
}

}  // namespace gd4d

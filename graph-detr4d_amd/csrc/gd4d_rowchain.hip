// gd4d_row_chain_fwd: a CHAIN of row-local operations (Linear, LayerNorm, adds, ReLU, reference-point refinement) over
// blocks of 16 complete rows, one launch.
//
// Between two points where a decoder layer needs all queries at once (the attention core, the fused sample-aggregate
// kernel) everything the reference does is row-local: nn.MultiheadAttention's out_proj + residual, norms[i], the
// Linears of Deform3DCrossAttn on query + query_pos (deform3d_cross_attn.py:211, :227, :281), output_proj + residuals
// (:326-336), mmcv's FFN, the next layer's in_proj, the head's reg branch and the reference-point refinement
// (detr3d_transformer.py:199-214), position_encoder (:104-111).  The reference launches ~40 ATen kernels per layer for
// them; round 1 of this build 14-18 (each 5-25 us, latency-bound, 3-6 % MFMA busy).  Here a workgroup owns 16 rows,
// keeps their activations in LDS (four 16 x 516 fp32 buffers) and walks a small PROGRAM of operations over them; the
// weights stream from L2 straight into MFMA operand registers.  A decoder layer becomes: attention core, chain A
// (out_proj .. cross-attention Linears), fused sample-aggregate, chain B (output_proj .. FFN .. next in_proj .. reg branch).
//
// Arithmetic of the GEMMs: split-bf16 x3 on v_mfma_f32_16x16x32_bf16 (x = hi + lo, a w ~= a_hi w_hi + a_lo w_hi +
// a_hi w_lo, fp32 accumulate: <= ~2^-16 relative per product, the numerics class of gd4d_value_proj_fwd).  With 16 rows
// per workgroup only ceil(M / 16) = 57 workgroups exist, so a chain is bound by what ONE compute unit can do: the
// fp32-input MFMA (64 FLOP/clk/SIMD) made chain B 121 us (1.3 GFLOP on 57 CUs), the bf16 pipe is 16x faster per
// product.  The weights are pre-split into MFMA B-fragment order (gd4d_chain_weight_image, cached by the caller while
// the weights do not change) so they stream from L2 as whole 1-KB pieces; the activations are split when read from LDS.
// LayerNorm two-pass like ATen (mean, centred sum of squares, biased variance, eps inside the square root).
#include <stdlib.h>

#include "gd4d_common.h"
#include "gd4d_pyramid_fill.h"
#include "gd4d_mha_dropout.h"
#include "gd4d_value_proj_body.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(4))) float rc4;

GD4D_TRACE_UNIT(rowchain)

constexpr int RC_M = 16;            // rows per workgroup
// Waves per workgroup x 16-column tiles per wave and pass: 8 x 2 (two waves per SIMD, 163 registers) against the first
// version's 4 x 4 (one wave per SIMD, 276 registers): the same bytes in flight per compute unit, but twice the waves to
// issue the fragment loads and to hide each other's MFMA chains - 1.79 against 1.87 ms per step, results bit-identical
// (a column's sum does not depend on which wave owns it).  -DRC_WAVES_N=4 -DRC_TILES_N=4 rebuilds the first form;
// 16 x 1 (128 registers, 33 spilled) measured 1.83 ms.
#ifndef RC_WAVES_N
#define RC_WAVES_N 8
#endif
#ifndef RC_TILES_N
#define RC_TILES_N 2
#endif
constexpr int RC_WAVES = RC_WAVES_N;
// LDS row pitch and the K-permutation of the GEMMs' A operand (round 4).  Lane (row i16, k-group g) of v_mfma_f32_16x16x32
// takes 8 of a k-step's 32 activations as two ds_read_b128.  With the natural choice (k = 8 g .. 8 g + 7) and a 516-float
// pitch the 16 lanes one LDS cycle serves ({0-3, 12-15, 20-27}, ...: rows 0-3 / 12-15 of one k-group with rows 4-11 of the
// next) land on 14 of the 16 four-bank slots: a 2-way conflict in every lane group, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 39 % (profiles/r03_pmc_step_inflight1.txt).  A sum over k may be taken in any order as long as both operands agree, so k-group
// g takes k = 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 (the weight image and HEADGEMM's global A rows follow): neighbouring
// k-groups are now ONE slot apart, and with a pitch of 520 floats (8 mod 64: a row = two slots) the rows of a lane group fall
// on the even slots and the other k-group's rows on the odd ones - conflict-free.  -DRC_KPERM=0 -DRC_LD_N=516 rebuilds round 3.
#ifndef RC_KPERM
#define RC_KPERM 1
#endif
#ifndef RC_LD_N
#define RC_LD_N 520
#endif
constexpr int RC_W = 512;           // widest row an operation may use
constexpr int RC_LD = RC_LD_N;      // floats per LDS row
static_assert(RC_LD >= RC_W + 4 && RC_LD % 4 == 0, "row pitch");
constexpr int RC_KG = RC_KPERM ? 4 : 8;     // floats between the first pieces of neighbouring k-groups
constexpr int RC_K2 = RC_KPERM ? 16 : 4;    // floats from a k-group's first piece (4 floats) to its second
constexpr int RC_BUFS = 4;
#ifndef RC_DEPTH_N
#define RC_DEPTH_N 2
#endif
constexpr int RC_DEPTH = RC_DEPTH_N; // weight-fragment ring depth in k-steps of 32 (8 waves: 1, 2 equal, 4 +0.8 %, 8 +2 % per step)
constexpr int RC_TILES = RC_TILES_N; // 16-column MFMA tiles per wave and pass
constexpr int RC_COLS = 16 * RC_TILES;   // columns per wave and pass
constexpr int RC_ROWS_PER_WAVE = RC_M / RC_WAVES;
static_assert((RC_TILES == 1 || RC_TILES % 2 == 0) && RC_M % RC_WAVES == 0 && RC_WAVES * RC_COLS == 256, "a pass of all waves covers 256 columns");

// One operation; the program is an array of these in device memory (gd4d.h: gd4d_chain_op).
typedef gd4d_chain_op ChainOp;

__device__ __forceinline__ float rc_act_in(float v, int flags) { return (flags & GD4D_CHAIN_INV_SIGMOID) ? inv_sigmoid(v) : v; }

#ifndef RC_ROTATE
#define RC_ROTATE 1   // per-workgroup starting k-step of the GEMMs (0: every workgroup walks K from 0 - dev A/B)
#endif
#ifndef RC_PREFETCH
#define RC_PREFETCH 1   // bit 0: touch the program's weight images; bit 1: also its small operands; bit 2: images after the leading LOADs
#endif
#ifndef RC_TRACE_OPS
#define RC_TRACE_OPS 0   // dev: stamp every operation's end in the device timeline (tools/trace_step.py)
#endif
#ifndef RC_DBG
#define RC_DBG 0      // dev ablation (compile with -DRC_DBG=n): 1 = no MFMAs, 2 = no weight loads
#endif
typedef __attribute__((ext_vector_type(8))) __bf16 rc_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned rc_u4;

__device__ __forceinline__ unsigned rc_cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}
// 8 consecutive floats -> bf16 hi halves and bf16 lo halves (x ~= hi + lo)
__device__ __forceinline__ void rc_split8(const float* v, rc_u4& h, rc_u4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = rc_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    ll[i] = rc_cvt_pk_bf16(v[2 * i] - __uint_as_float(hh[i] << 16), v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u));
  }
  h = rc_u4{hh[0], hh[1], hh[2], hh[3]};
  l = rc_u4{ll[0], ll[1], ll[2], ll[3]};
}
// 8 consecutive floats -> three bf16 pieces (x = hi + mid + lo to ~2^-25: GD4D_CHAIN_EXACT)
__device__ __forceinline__ void rc_split8x3(const float* v, rc_u4& h, rc_u4& m, rc_u4& l) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = rc_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    const float r0 = v[2 * i] - __uint_as_float(hh[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u);
    mm[i] = rc_cvt_pk_bf16(r0, r1);
    ll[i] = rc_cvt_pk_bf16(r0 - __uint_as_float(mm[i] << 16), r1 - __uint_as_float(mm[i] & 0xffff0000u));
  }
  h = rc_u4{hh[0], hh[1], hh[2], hh[3]};
  m = rc_u4{mm[0], mm[1], mm[2], mm[3]};
  l = rc_u4{ll[0], ll[1], ll[2], ll[3]};
}
__device__ __forceinline__ rc_bf16x8 rc_frag(rc_u4 v) { return __builtin_bit_cast(rc_bf16x8, v); }

// Weight image (gd4d_chain_weight_image): [tile t of 16 output columns][k-step s of 32][hi, lo][lane][8 bf16]; lane l of
// a fragment holds W[n = 16 t + (l & 15)][k = 32 s + {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3}], g = l >> 4 (RC_KPERM above) - the
// B operand of v_mfma_f32_16x16x32_bf16.
// PLANES = 3 (gd4d_chain_weight_image_exact): [hi, mid, lo] - the operand of a GD4D_CHAIN_EXACT GEMM.
template <int PLANES>
__global__ __launch_bounds__(256) void chain_weight_image_kernel(const float* __restrict__ w, char* __restrict__ img, int N, int K) {
  const int ksteps = K / 32;
  const int frag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // fragment = t * ksteps + s
  const int tiles = (N + 15) / 16;
  if (frag >= tiles * ksteps) return;
  const int t = frag / ksteps, s = frag - t * ksteps;
  const int n = 16 * t + (lane & 15);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = n < N ? w[(size_t)n * K + 32 * s + RC_KG * (lane >> 4) + (j < 4 ? j : RC_K2 + j - 4)] : 0.f;
  char* dst = img + (size_t)frag * (PLANES * 1024) + lane * 16;
  if (PLANES == 3) {
    rc_u4 h, m, l;
    rc_split8x3(v, h, m, l);
    *reinterpret_cast<rc_u4*>(dst) = h;
    *reinterpret_cast<rc_u4*>(dst + 1024) = m;
    *reinterpret_cast<rc_u4*>(dst + 2048) = l;
  } else {
    rc_u4 h, l;
    rc_split8(v, h, l);
    *reinterpret_cast<rc_u4*>(dst) = h;
    *reinterpret_cast<rc_u4*>(dst + 1024) = l;
  }
}

// Every weight image a TRAINING step needs - forward (W) and backward (W^T) operands of its chains - in ONE launch: the weights
// change with every optimizer step, and inside a replayed hipGraph nothing can notice that from the host.  A job describes a
// logical matrix S = up to three row blocks stacked (the Linears of one input; unused blocks have 0 rows) of `cols` columns;
// its image is that of S (N = rows, K = cols) or, transposed, of S^T (N = cols, K = rows, zero-padded to a multiple of 64: the
// backward GEMM's A operand carries zeros there).  planes = 2 / 3: bf16 hi / lo (/ mid); planes = 0: no image - the job
// concatenates 1-D segments (cols = 1: the stacked bias) into fp32 `image`.  frag0 = the job's first fragment in the launch
// (a fragment = 64 lanes x 8 values; jobs sorted by frag0).
__global__ __launch_bounds__(256) void chain_weight_image_group_kernel(const gd4d_image_job* __restrict__ jobs, int count, int total) {
  // which job: a binary search over the jobs' first fragments - from LDS (seven dependent global loads per wave otherwise:
  // the launch was bound by them, 57 us for 16 k fragments)
  __shared__ int s_frag0[GD4D_IMAGE_JOBS_MAX];
  for (int i = threadIdx.x; i < count; i += 256) s_frag0[i] = jobs[i].frag0;
  __syncthreads();
  const int frag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (frag >= total) return;
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (s_frag0[mid] <= frag) lo = mid; else hi = mid - 1;
  }
  const gd4d_image_job J = jobs[lo];
  const int f = frag - J.frag0;
  const int R = J.rows[0] + J.rows[1] + J.rows[2];
  auto elem = [&](int r, int c) -> float {
    if (r >= R || c >= J.cols) return 0.f;
    int j = 0;
    if (r >= J.rows[0]) { r -= J.rows[0]; j = 1; if (r >= J.rows[1]) { r -= J.rows[1]; j = 2; } }
    return J.seg[j][(size_t)r * J.cols + c];
  };
  if (J.planes == 0) {
    const int i = f * 64 + lane;
    if (i < R) static_cast<float*>(J.image)[i] = elem(i, 0);
    return;
  }
  const int N = J.transposed ? J.cols : R, K = J.transposed ? R : J.cols;
  const int ksteps = ((K + 63) & ~63) / 32;
  const int t = f / ksteps, s = f - t * ksteps;
  const int n = 16 * t + (lane & 15);
  float v[8];
  if (!J.transposed && (J.cols & 3) == 0 && RC_KG == 4) {
    // a lane's row is fixed: its block once, then its two pieces of the k-step as 16-byte loads (K = cols is a multiple of 64 here)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (n < N) {
      int r = n, j = 0;
      if (r >= J.rows[0]) { r -= J.rows[0]; j = 1; if (r >= J.rows[1]) { r -= J.rows[1]; j = 2; } }
      const float* rowp = J.seg[j] + (size_t)r * J.cols + 32 * s + RC_KG * (lane >> 4);
      if (32 * s + RC_KG * (lane >> 4) + 3 < J.cols) a = *reinterpret_cast<const float4*>(rowp);
      if (32 * s + RC_KG * (lane >> 4) + RC_K2 + 3 < J.cols) b = *reinterpret_cast<const float4*>(rowp + RC_K2);
    }
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * s + RC_KG * (lane >> 4) + (j < 4 ? j : RC_K2 + j - 4);
      v[j] = n < N ? (J.transposed ? elem(k, n) : elem(n, k)) : 0.f;
    }
  }
  char* dst = static_cast<char*>(J.image) + (size_t)f * (J.planes * 1024) + lane * 16;
  if (J.planes == 3) {
    rc_u4 h, m, l;
    rc_split8x3(v, h, m, l);
    *reinterpret_cast<rc_u4*>(dst) = h;
    *reinterpret_cast<rc_u4*>(dst + 1024) = m;
    *reinterpret_cast<rc_u4*>(dst + 2048) = l;
  } else {
    rc_u4 h, l;
    rc_split8(v, h, l);
    *reinterpret_cast<rc_u4*>(dst) = h;
    *reinterpret_cast<rc_u4*>(dst + 1024) = l;
  }
}

// GEMM over the workgroup's 16 rows: out[:, n] = act(sum_k in[:, k] * W[n, k] + bias[n]) (+ residuals), n < N.
// Wave w owns columns [256 pass + RC_COLS w, + RC_COLS) of every pass; A fragments are split from the LDS buffer, the W fragments
// come pre-split from the image (global / L2) through a register ring RC_DEPTH k-steps deep.
// EXACT (GD4D_CHAIN_EXACT): both operands cut into THREE bf16 pieces and the six products of combined order <= 2 summed
// (x y ~= sum_{i + j <= 2} x_i y_j, ~2^-24 relative: fp32-class), for the GEMMs whose outputs become reference points -
// a point's error is multiplied by the 102-m range and the focal length before it selects pixels.
template <bool EXACT, bool TRAIN>
__device__ __forceinline__ void rc_gemm(const ChainOp& op, float (*bufs)[RC_M][RC_LD], int m0, int M, int lane, int wave) {
  constexpr int FRAG = EXACT ? 3072 : 2048;
  const int i16 = lane & 15, g = lane >> 4;
  const int K = op.K, N = op.N;
  const int steps = K / 32, tiles = (N + 15) / 16;
  const char* img = reinterpret_cast<const char*>(op.p0);
  // GD4D_CHAIN_SRC2: the passes from column ld0 on (a multiple of 256 = one pass of all waves) take their A operand from buffer
  // `res` instead of `src` - the packed in-projection (q, k from x + pos, v from x) as ONE operation.
  // GD4D_CHAIN_SPLIT_OUT: the N columns go to three global tensors (gout | p2 | p3, each as wide as its row stride) - the
  // three Linears of query + query_pos (camera logits, offsets, attention logits) as ONE operation over their stacked weights.
  const bool src2 = (op.flags & GD4D_CHAIN_SRC2) != 0, split_out = (op.flags & GD4D_CHAIN_SPLIT_OUT) != 0;
  // GD4D_CHAIN_SPLIT_KV: pass 1 (K) and pass 2 (V) of the packed in-projection also leave as bf16 hi / lo planes (p2: K row-major,
  // p3: V^T) - the operands gd4d_mha_core_presplit_fwd feeds to its MFMAs without converting anything
  const bool split_kv = (op.flags & GD4D_CHAIN_SPLIT_KV) != 0;
  const bool kv_fp32_too = (op.flags & GD4D_CHAIN_SPLIT_KV_KEEP) != 0;    // (a training step: the attention BACKWARD reads fp32 rows)
  // GD4D_CHAIN_MASK_P2: p2 is not an addend but the OUTPUT a ReLU produced in the forward pass - the result (the gradient at that
  // ReLU's output) passes where it was > 0, times `eps` when that is non-zero (the 1 / (1 - p) of a dropout that followed the ReLU
  // and left its zeros in p2 as well)
  const bool mask_p2 = TRAIN && (op.flags & GD4D_CHAIN_MASK_P2) != 0;      // (TRAIN: the training operations live in their own instantiation of the kernel - the inference step's chains ran 10 % slower with them compiled in)
  const float mask_scale = op.eps != 0.f ? op.eps : 1.f;
  // GD4D_CHAIN_DROPOUT (a training step with the modules in train mode): the output - after bias / activation, before the
  // residuals - is dropped like nn.Dropout does: kept elements times eps = 1 / (1 - p).  Which elements: csrc/gd4d_mha_dropout.h's
  // hash of (the 64-bit seed at p3, m N + n) against the threshold in `reserved` - a backward chain regenerates it (DROPMASK).
  const bool drop = TRAIN && (op.flags & GD4D_CHAIN_DROPOUT) != 0;
  const uint32_t drop_lo = drop ? reinterpret_cast<const uint32_t*>(op.p3)[0] : 0u;
  const uint32_t drop_hi = drop ? reinterpret_cast<const uint32_t*>(op.p3)[1] : 0u;
  for (int n_base = RC_COLS * wave; n_base < N; n_base += RC_COLS * RC_WAVES) {
    const float* a_row = &bufs[(src2 && n_base >= op.ld0) ? op.res : op.src][i16][RC_KG * g];
    const char* wf[RC_TILES];                                  // tiles past the end re-read the last one (never stored)
#pragma unroll
    for (int c = 0; c < RC_TILES; ++c) wf[c] = img + (size_t)min(n_base / 16 + c, tiles - 1) * steps * FRAG + lane * 16;
    rc4 acc[RC_TILES];
#pragma unroll
    for (int c = 0; c < RC_TILES; ++c) acc[c] = rc4{0.f, 0.f, 0.f, 0.f};
    // bias: unconditional loads from clamped addresses, issued before the K loop (a load under a predicate becomes its
    // own basic block closed by s_waitcnt vmcnt(0): a dozen serialised L2 round trips in the epilogue otherwise)
    float e_bias[RC_TILES];
    {
      const float* bias_p = op.p1 ? op.p1 : reinterpret_cast<const float*>(op.p0);
      const float bias_on = op.p1 ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < RC_TILES; ++c) e_bias[c] = bias_p[min(n_base + 16 * c + i16, N - 1)] * bias_on;
    }
    // global addends p2 (+ p3) of the epilogue - the residual rows a LOAD operation would otherwise park in a buffer first:
    // requested here, consumed after the K loop (their fabric round trip hides under the weight stream)
    float e_add[RC_TILES][4];
    if (op.p2 && !split_out && !split_kv) {
#pragma unroll
      for (int c = 0; c < RC_TILES; ++c) {
        const int n = min(n_base + 16 * c + i16, N - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t m = (size_t)min(m0 + 4 * g + r, M - 1);
          const float a2 = op.p2[m * op.ld2 + n];
          e_add[c][r] = (op.p3 && !drop) ? a2 + op.p3[m * op.ld1 + n] : a2;
        }
      }
    }
    rc_u4 bh[RC_DEPTH][RC_TILES], bl[RC_DEPTH][RC_TILES], bm[EXACT ? RC_DEPTH : 1][RC_TILES];
    auto issue = [&](int slot, int j) {
      if (RC_DBG & 2) return;
#pragma unroll
      for (int c = 0; c < RC_TILES; ++c) {
        bh[slot][c] = *reinterpret_cast<const rc_u4*>(wf[c] + (size_t)j * FRAG);
        if (EXACT) bm[slot][c] = *reinterpret_cast<const rc_u4*>(wf[c] + (size_t)j * FRAG + 1024);
        bl[slot][c] = *reinterpret_cast<const rc_u4*>(wf[c] + (size_t)j * FRAG + (EXACT ? 2048 : 1024));
      }
    };
    if (RC_DBG & 2) {
#pragma unroll
      for (int d = 0; d < RC_DEPTH; ++d)
#pragma unroll
        for (int c = 0; c < RC_TILES; ++c) { bh[d][c] = rc_u4{1u, 2u, 3u, (unsigned)lane}; bl[d][c] = bh[d][c]; if (EXACT) bm[d][c] = bh[d][c]; }
    }
    // The workgroups of a launch run in lock step and stream the SAME weight image: un-rotated, the ~7 workgroups that
    // share an XCD ask one L2 channel for one fragment at the same instant and take turns (17 B/clk per CU measured).
    // Each workgroup therefore walks K from its own starting k-step (a sum may be taken in any order; the order is a
    // function of the workgroup index only, so results are run-to-run identical).
    const int rot = RC_ROTATE ? (int)(((unsigned)(m0 / RC_M) >> 3) % (unsigned)steps) : 0;   // by ROW BLOCK: the same in a one- and a two-program launch
    auto kstep = [&](int j) { const int r = j + rot; return r >= steps ? r - steps : r; };
#pragma unroll
    for (int d = 0; d < RC_DEPTH; ++d) issue(d, kstep(min(d, steps - 1)));
    auto consume = [&](int d, int j0_) {
      const int j = kstep(j0_);
      const float4 t0 = *reinterpret_cast<const float4*>(a_row + 32 * j);
      const float4 t1 = *reinterpret_cast<const float4*>(a_row + 32 * j + RC_K2);
      const float a[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      if (EXACT) {
        rc_u4 ah, am, al;
        rc_split8x3(a, ah, am, al);
#pragma unroll
        for (int c = 0; c < RC_TILES; ++c) {                   // smallest terms first
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(al), rc_frag(bh[d][c]), acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(ah), rc_frag(bl[d][c]), acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(am), rc_frag(bm[d][c]), acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(am), rc_frag(bh[d][c]), acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(ah), rc_frag(bm[d][c]), acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(ah), rc_frag(bh[d][c]), acc[c], 0, 0, 0);
        }
        return;
      }
      rc_u4 ah, al;
      rc_split8(a, ah, al);
      if (RC_DBG & 1) { asm volatile("" ::"v"(ah), "v"(al), "v"(bh[d][0]), "v"(bl[d][3])); return; }
#pragma unroll
      for (int c = 0; c < RC_TILES; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(ah), rc_frag(bh[d][c]), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(al), rc_frag(bh[d][c]), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(ah), rc_frag(bl[d][c]), acc[c], 0, 0, 0);
      }
    };
    // host guarantees steps % RC_DEPTH == 0 (K % (32 RC_DEPTH) == 0)
    for (int j0 = 0; j0 + RC_DEPTH < steps; j0 += RC_DEPTH) {
#pragma unroll
      for (int d = 0; d < RC_DEPTH; ++d) {
        consume(d, j0 + d);
        issue(d, kstep(j0 + d + RC_DEPTH));
      }
    }
#pragma unroll
    for (int d = 0; d < RC_DEPTH; ++d) consume(d, steps - RC_DEPTH + d);
    // C/D of 16x16x32: col = lane & 15 (+ 16 c), row = 4 * (lane >> 4) + r
#pragma unroll
    for (int c = 0; c < RC_TILES; ++c) {
      const int n = n_base + 16 * c + i16;
      if (n >= N) continue;
      float kv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r, m = m0 + row;
        float v = acc[c][r] + e_bias[c];
        if (op.flags & GD4D_CHAIN_RELU) v = fmaxf(v, 0.f);
        if (op.flags & GD4D_CHAIN_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        if (drop) v = mha_drop_keep(drop_lo, drop_hi, (uint32_t)m * (uint32_t)N + (uint32_t)n, (uint32_t)op.reserved) ? v * op.eps : 0.f;
        if (mask_p2) v = e_add[c][r] > 0.f ? v * mask_scale : 0.f;
        if (op.res >= 0 && !src2) v += bufs[op.res][row][n];
        if (op.p2 && !split_out && !mask_p2 && !split_kv) v += e_add[c][r];
        if (split_kv) kv[r] = v;
        if (op.dst >= 0) bufs[op.dst][row][op.dst_col + n] = v;
        if (split_out) {
          if (m < M) {
            const int c1 = op.ldg, c2 = op.ldg + op.ld2;
            float* o = n < c1 ? op.gout + (size_t)m * op.ldg + n
                     : n < c2 ? const_cast<float*>(op.p2) + (size_t)m * op.ld2 + (n - c1)
                              : const_cast<float*>(op.p3) + (size_t)m * op.ld1 + (n - c2);
            *o = v;
          }
        } else if (op.gout && m < M && !(split_kv && !kv_fp32_too && n_base >= RC_COLS * RC_WAVES)) {   // (K and V leave as planes only)
          op.gout[(size_t)m * op.ldg + n] = v;
        }
      }
      if (split_kv && n_base >= RC_COLS * RC_WAVES) {          // (pass-uniform: pass 1 = K, pass 2 = V)
        const unsigned h01 = rc_cvt_pk_bf16(kv[0], kv[1]), h23 = rc_cvt_pk_bf16(kv[2], kv[3]);
        const unsigned l01 = rc_cvt_pk_bf16(kv[0] - __uint_as_float(h01 << 16), kv[1] - __uint_as_float(h01 & 0xffff0000u));
        const unsigned l23 = rc_cvt_pk_bf16(kv[2] - __uint_as_float(h23 << 16), kv[3] - __uint_as_float(h23 & 0xffff0000u));
        constexpr int C3 = RC_COLS * RC_WAVES;                 // = N / 3
        const int blk = m0 / RC_M, blocks = (M + RC_M - 1) / RC_M;
        if (n_base < 2 * C3) {
          // K: [head][tile = this row block][lane = 16 (c / 8) + key % 16][c % 8] - the A operand of S^T = K Q^T (rows past M too)
          const int nn = n - C3, hd = nn >> 5, cc = nn & 31;
          unsigned short* kh = reinterpret_cast<unsigned short*>(const_cast<float*>(op.p2)) +
                               (((size_t)hd * blocks + blk) * 64 + (cc >> 3) * 16 + 4 * g) * 8 + (cc & 7);
          unsigned short* kl = kh + op.ld2;
          const unsigned hh[4] = {h01 & 0xffffu, h01 >> 16, h23 & 0xffffu, h23 >> 16};
          const unsigned ll[4] = {l01 & 0xffffu, l01 >> 16, l23 & 0xffffu, l23 >> 16};
#pragma unroll
          for (int r = 0; r < 4; ++r) { kh[r * 8] = (unsigned short)hh[r]; kl[r * 8] = (unsigned short)ll[r]; }
        } else {
          // V: [head][step of 32 keys][half = d / 16][lane = 16 g + d % 16][j = 4 (block & 1) + r]: the A operand of O^T = V^T P^T;
          // the four keys of this lane are one 8-byte store per plane.  An odd last block also zeroes the step's other half
          // (nobody else writes it; its probabilities are 0 and 0 x anything finite is 0).
          const int nn = n - 2 * C3, hd = nn >> 5, dd = nn & 31, steps = (blocks + 1) >> 1;
          unsigned short* vh = reinterpret_cast<unsigned short*>(const_cast<float*>(op.p3)) +
                               ((((size_t)hd * steps + (blk >> 1)) * 2 + (dd >> 4)) * 64 + 16 * g + (dd & 15)) * 8 + 4 * (blk & 1);
          unsigned short* vl = vh + op.ld1;
          *reinterpret_cast<uint2*>(vh) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(vl) = make_uint2(l01, l23);
          if (blk == blocks - 1 && !(blk & 1)) {
            *reinterpret_cast<uint2*>(vh + 4) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(vl + 4) = make_uint2(0u, 0u);
          }
        }
      }
    }
  }
}

// HEADGEMM: value_proj applied to per-head aggregates that live in GLOBAL memory (gd4d_cross_attn_agg_fwd's output):
//   v[m, n] = sum_k agg[m][h][k] * W[n][k] + bias[n] * wsum[m][h],   h = n / (N / heads)
// i.e. gd4d_value_proj_heads_fwd as the first operation of chain B (its result is the input of output_proj).  Same
// column ownership, weight image and arithmetic as rc_gemm; the A fragments of a k-step come from the rows of the two
// heads a wave's 64 columns belong to (Dh = 32: tiles 0, 1 -> one head, tiles 2, 3 -> the next; Dh = 64: one head),
// through a 2-deep register ring beside the weight fragments.  Requires Dh % 32 == 0, K % 64 == 0.
__device__ __forceinline__ void rc_headgemm(const ChainOp& op, float (*bufs)[RC_M][RC_LD], int m0, int M, int lane, int wave) {
  constexpr int HD = 2;                                        // ring depth (k-steps)
  const int i16 = lane & 15, g = lane >> 4;
  const int K = op.K, N = op.N, heads = op.ld0, Dh = N / heads;
  const int steps = K / 32, tiles = (N + 15) / 16;
  const char* img = reinterpret_cast<const char*>(op.p0);
  const int m_ld = min(m0 + i16, M - 1);                       // rows past M repeat the last row (never stored)
  for (int n_base = RC_COLS * wave; n_base < N; n_base += RC_COLS * RC_WAVES) {
    const char* wf[RC_TILES];
#pragma unroll
    for (int c = 0; c < RC_TILES; ++c) wf[c] = img + (size_t)min(n_base / 16 + c, tiles - 1) * steps * 2048 + lane * 16;
    constexpr int TPG = RC_TILES >= 2 ? 2 : 1;                 // tiles per group: up to 32 columns, inside one head
    constexpr int NG = RC_TILES / TPG;
    int hg[NG];
    const float* ag[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      hg[gi] = min((n_base + 16 * TPG * gi) / Dh, heads - 1);
      ag[gi] = op.p2 + ((size_t)m_ld * heads + hg[gi]) * K + RC_KG * g;
    }
    rc4 acc[RC_TILES];
#pragma unroll
    for (int c = 0; c < RC_TILES; ++c) acc[c] = rc4{0.f, 0.f, 0.f, 0.f};
    float e_bias[RC_TILES];
    {
      const float* bias_p = op.p1 ? op.p1 : reinterpret_cast<const float*>(op.p0);
      const float bias_on = op.p1 ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < RC_TILES; ++c) e_bias[c] = bias_p[min(n_base + 16 * c + i16, N - 1)] * bias_on;
    }
    // GD4D_CHAIN_ADD_GOUT: the (M, N) addend at gout, requested before the k loop (the lane's own output elements)
    float addend[RC_TILES][4];
    if (op.flags & GD4D_CHAIN_ADD_GOUT) {
#pragma unroll
      for (int c = 0; c < RC_TILES; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          addend[c][r] = op.gout[(size_t)min(m0 + 4 * g + r, M - 1) * op.ldg + min(n_base + 16 * c + i16, N - 1)];
    }
    // (Round 6, measured and not kept: ALL k-steps' aggregate rows requested at once - 64 registers - instead of this two-deep
    //  ring: chain B' 55.7 against 55.7 us.  HEADGEMM's 18 us in block 0's timeline are the launch (~5 us until a first operation
    //  completes in any chain) and the program's weight touches, which its first loads queue behind: not its own ring.)
    rc_u4 bh[HD][RC_TILES], bl[HD][RC_TILES];
    float4 av[HD][NG][2];
    auto issue = [&](int slot, int j) {
#pragma unroll
      for (int c = 0; c < RC_TILES; ++c) {
        bh[slot][c] = *reinterpret_cast<const rc_u4*>(wf[c] + (size_t)j * 2048);
        bl[slot][c] = *reinterpret_cast<const rc_u4*>(wf[c] + (size_t)j * 2048 + 1024);
      }
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        av[slot][gi][0] = *reinterpret_cast<const float4*>(ag[gi] + 32 * j);
        av[slot][gi][1] = *reinterpret_cast<const float4*>(ag[gi] + 32 * j + RC_K2);
      }
    };
    auto consume = [&](int d) {
#pragma unroll
      for (int hh = 0; hh < NG; ++hh) {
        const float a[8] = {av[d][hh][0].x, av[d][hh][0].y, av[d][hh][0].z, av[d][hh][0].w,
                            av[d][hh][1].x, av[d][hh][1].y, av[d][hh][1].z, av[d][hh][1].w};
        rc_u4 ah, al;
        rc_split8(a, ah, al);
#pragma unroll
        for (int c = TPG * hh; c < TPG * hh + TPG; ++c) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(ah), rc_frag(bh[d][c]), acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(al), rc_frag(bh[d][c]), acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rc_frag(ah), rc_frag(bl[d][c]), acc[c], 0, 0, 0);
        }
      }
    };
#pragma unroll
    for (int d = 0; d < HD; ++d) issue(d, min(d, steps - 1));
    for (int j0 = 0; j0 + HD < steps; j0 += HD) {              // steps is even (K % 64 == 0)
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        consume(d);
        issue(d, j0 + d + HD);
      }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) consume(d);
#pragma unroll
    for (int c = 0; c < RC_TILES; ++c) {
      const int n = n_base + 16 * c + i16;
      if (n >= N) continue;
      const int hc = hg[c / TPG];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r, m = m0 + row;
        float v = fmaf(e_bias[c], op.p3[(size_t)min(m, M - 1) * heads + hc], acc[c][r]);
        if (op.res >= 0) v += bufs[op.res][row][n];
        if (op.flags & GD4D_CHAIN_ADD_GOUT) v += addend[c][r];
        if (op.dst >= 0) bufs[op.dst][row][op.dst_col + n] = v;
        if (op.gout && !(op.flags & GD4D_CHAIN_ADD_GOUT) && m < M) op.gout[(size_t)m * op.ldg + n] = v;
      }
    }
  }
}

// LayerNorm over N columns (N % 64 == 0, N <= 512) of the 16 rows: wave w normalises rows 4 w .. 4 w + 3, 16 lanes per
// row, a lane owns columns 64 ch + 4 l16 .. + 4 of every 64-column chunk.  gamma / beta are requested first.
__device__ __forceinline__ void rc_layernorm(const ChainOp& op, float (*bufs)[RC_M][RC_LD], int m0, int M, int lane, int wave) {
  constexpr int MAXCH = RC_W / 64;
  if (wave >= RC_M / 4) return;                                // 4 rows per wave: with more than 4 waves the rest wait at the barrier
  const int row = 4 * wave + (lane >> 4), l16 = lane & 15, N = op.N, nch = N / 64;
  float4 gm[MAXCH], bt[MAXCH], x[MAXCH], ad[MAXCH];
  // second output (res >= 0 with p2): buf[res] = result + p2[m, :] - the ADD operation that would follow (x + query_pos for
  // the next projection), its global rows requested here with gamma / beta
  const bool second = op.res >= 0 && op.p2;
  const size_t m_ld = (size_t)min(m0 + row, M - 1);
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) {
    const int n = min(64 * ch, N - 64) + 4 * l16;              // chunks past N repeat the last one (unused)
    gm[ch] = *reinterpret_cast<const float4*>(op.p0 + n);
    bt[ch] = *reinterpret_cast<const float4*>(op.p1 + n);
    if (second) ad[ch] = *reinterpret_cast<const float4*>(op.p2 + m_ld * op.ld2 + n);
    x[ch] = *reinterpret_cast<const float4*>(&bufs[op.src][row][n]);
  }
  float s = 0.f;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (ch < nch) s += (x[ch].x + x[ch].y) + (x[ch].z + x[ch].w);
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)N;
  float q = 0.f;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (ch < nch) {
      const float a = x[ch].x - mean, b = x[ch].y - mean, c = x[ch].z - mean, d = x[ch].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)N + op.eps);
  const int m = m0 + row;
  const bool relu = op.flags & GD4D_CHAIN_RELU;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) {
    if (ch >= nch) break;
    const int n = 64 * ch + 4 * l16;
    float4 v;
    v.x = (x[ch].x - mean) * rstd * gm[ch].x + bt[ch].x; v.y = (x[ch].y - mean) * rstd * gm[ch].y + bt[ch].y;
    v.z = (x[ch].z - mean) * rstd * gm[ch].z + bt[ch].z; v.w = (x[ch].w - mean) * rstd * gm[ch].w + bt[ch].w;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (op.dst >= 0) *reinterpret_cast<float4*>(&bufs[op.dst][row][n]) = v;
    if (op.gout && m < M) *reinterpret_cast<float4*>(op.gout + (size_t)m * op.ldg + n) = v;
    if (second)
      *reinterpret_cast<float4*>(&bufs[op.res][row][n]) = make_float4(ad[ch].x + v.x, ad[ch].y + v.y, ad[ch].z + v.z, ad[ch].w + v.w);
  }
}

// LN_BWD: the backward of LAYERNORM ([ReLU] LN(x) gamma + beta) over the 16 rows: buf[src] = gradient of the output, buf[res] = the
// forward's INPUT x (mean / rstd are recomputed, two-pass, as the forward did); dx -> buf[dst] (dst == src allowed: a lane
// reads its columns before it writes them) and / or gout; the row block's partial dgamma / dbeta -> p2[(block * 2 + {0, 1}) * N + c]
// (the layout of gd4d_layernorm_bwd's workspace: gd4d_layernorm_bwd_reduce_group adds the blocks in order).  Rows past M count
// as zero gradients.  Waves 0 .. 3 own 4 rows each, 16 lanes per row; the partial sums meet in `scratch` ([4][2][RC_W] floats).
__device__ __forceinline__ void rc_layernorm_bwd(const ChainOp& op, float (*bufs)[RC_M][RC_LD], float* scratch, int m0, int M, int wg,
                                                 int tid) {
  constexpr int MAXCH = RC_W / 64;
  const int lane = tid & 63, wave = tid >> 6;
  const int N = op.N, nch = N / 64;
  const bool relu = op.flags & GD4D_CHAIN_RELU;
  if (wave < RC_M / 4) {
    const int row = 4 * wave + (lane >> 4), l16 = lane & 15;
    const bool live = m0 + row < M;
    float4 gm[MAXCH], bt[MAXCH], x[MAXCH], d[MAXCH];
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      const int n = min(64 * ch, N - 64) + 4 * l16;
      gm[ch] = *reinterpret_cast<const float4*>(op.p0 + n);
      bt[ch] = relu ? *reinterpret_cast<const float4*>(op.p1 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      // the forward's input: from buf[res], or (p3 given) straight from its rows in global memory - a LOAD operation and its
      // barrier less per LayerNorm of a backward chain
      x[ch] = op.p3 ? *reinterpret_cast<const float4*>(op.p3 + (size_t)min(m0 + row, M - 1) * op.ld1 + n)
                    : *reinterpret_cast<const float4*>(&bufs[op.res][row][n]);
      d[ch] = *reinterpret_cast<const float4*>(&bufs[op.src][row][n]);
      if (!live) d[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      if (ch < nch) s += (x[ch].x + x[ch].y) + (x[ch].z + x[ch].w);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)N;
    float q = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      if (ch < nch) {
        const float a = x[ch].x - mean, b = x[ch].y - mean, c = x[ch].z - mean, e = x[ch].w - mean;
        q += (a * a + b * b) + (c * c + e * e);
      }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)N + op.eps);
    float s1 = 0.f, s2 = 0.f;
    float4 xh[MAXCH], g[MAXCH];
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      xh[ch] = make_float4((x[ch].x - mean) * rstd, (x[ch].y - mean) * rstd, (x[ch].z - mean) * rstd, (x[ch].w - mean) * rstd);
      if (relu) {                                          // the forward clamped at 0: those outputs pass no gradient
        if (xh[ch].x * gm[ch].x + bt[ch].x <= 0.f) d[ch].x = 0.f;
        if (xh[ch].y * gm[ch].y + bt[ch].y <= 0.f) d[ch].y = 0.f;
        if (xh[ch].z * gm[ch].z + bt[ch].z <= 0.f) d[ch].z = 0.f;
        if (xh[ch].w * gm[ch].w + bt[ch].w <= 0.f) d[ch].w = 0.f;
      }
      g[ch] = make_float4(d[ch].x * gm[ch].x, d[ch].y * gm[ch].y, d[ch].z * gm[ch].z, d[ch].w * gm[ch].w);
      if (ch < nch) {
        s1 += (g[ch].x + g[ch].y) + (g[ch].z + g[ch].w);
        s2 += (g[ch].x * xh[ch].x + g[ch].y * xh[ch].y) + (g[ch].z * xh[ch].z + g[ch].w * xh[ch].w);
      }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float m1 = s1 / (float)N, m2 = s2 / (float)N;
    const int m = m0 + row;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      if (ch >= nch) break;
      const int n = 64 * ch + 4 * l16;
      const float4 v = make_float4(rstd * (g[ch].x - m1 - xh[ch].x * m2), rstd * (g[ch].y - m1 - xh[ch].y * m2),
                                   rstd * (g[ch].z - m1 - xh[ch].z * m2), rstd * (g[ch].w - m1 - xh[ch].w * m2));
      if (op.dst >= 0) *reinterpret_cast<float4*>(&bufs[op.dst][row][n]) = v;
      if (op.gout && m < M) *reinterpret_cast<float4*>(op.gout + (size_t)m * op.ldg + n) = v;
      if (op.p2) {                                         // the wave's four rows, in row order
        float4 a = make_float4(d[ch].x * xh[ch].x, d[ch].y * xh[ch].y, d[ch].z * xh[ch].z, d[ch].w * xh[ch].w), b = d[ch];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
          b.x += __shfl_xor(b.x, o); b.y += __shfl_xor(b.y, o); b.z += __shfl_xor(b.z, o); b.w += __shfl_xor(b.w, o);
        }
        if (lane < 16) {
          *reinterpret_cast<float4*>(scratch + (wave * 2 + 0) * RC_W + n) = a;
          *reinterpret_cast<float4*>(scratch + (wave * 2 + 1) * RC_W + n) = b;
        }
      }
    }
  }
  if (!op.p2) return;
  __syncthreads();
  float* part = const_cast<float*>(op.p2);
  for (int e = tid; e < 2 * N; e += 64 * RC_WAVES) {
    const int k = e / N, c = e - k * N;
    const float t = (scratch[(0 * 2 + k) * RC_W + c] + scratch[(1 * 2 + k) * RC_W + c]) +
                    (scratch[(2 * 2 + k) * RC_W + c] + scratch[(3 * 2 + k) * RC_W + c]);
    part[((size_t)wg * 2 + k) * N + c] = t;
  }
}

// Rows of global tensors into / onto an LDS buffer, float4 per lane, every load issued before the first use:
//   LOAD: dst[:, dst_col + n] = f(p0[m, n]) (+ p1[m, n]);   ADD: dst[:, n] = src[:, n] (+ res[:, n]) (+ p2[m, n])
// wave w handles RC_ROWS_PER_WAVE consecutive rows; N % 4 == 0, N <= 512 (two 256-column chunks per row).
template <bool IS_ADD, bool TRAIN>
__device__ __forceinline__ void rc_rows(const ChainOp& op, float (*bufs)[RC_M][RC_LD], int m0, int M, int lane, int wave) {
  const int N = op.N;
  const float* ga = IS_ADD ? op.p2 : op.p0;                  // first global operand (may be null for ADD)
  const float* gb = IS_ADD ? nullptr : op.p1;                 // second global operand (LOAD only)
  const int lda = IS_ADD ? op.ld2 : op.ld0, ldb = op.ld1;
  float4 va[RC_ROWS_PER_WAVE][2], vb[RC_ROWS_PER_WAVE][2];
#pragma unroll
  for (int r = 0; r < RC_ROWS_PER_WAVE; ++r) {
    const int m = min(m0 + RC_ROWS_PER_WAVE * wave + r, M - 1);              // rows past M repeat the last row (never stored)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = min(256 * cb + 4 * lane, N - 4);
      va[r][cb] = ga ? *reinterpret_cast<const float4*>(ga + (size_t)m * lda + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      vb[r][cb] = gb ? *reinterpret_cast<const float4*>(gb + (size_t)m * ldb + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int r = 0; r < RC_ROWS_PER_WAVE; ++r) {
    const int row = RC_ROWS_PER_WAVE * wave + r;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = 256 * cb + 4 * lane;
      if (n >= N) continue;
      float4 v = va[r][cb];
      if (!IS_ADD && (op.flags & GD4D_CHAIN_INV_SIGMOID)) { v.x = inv_sigmoid(v.x); v.y = inv_sigmoid(v.y); v.z = inv_sigmoid(v.z); v.w = inv_sigmoid(v.w); }
      v.x += vb[r][cb].x; v.y += vb[r][cb].y; v.z += vb[r][cb].z; v.w += vb[r][cb].w;
      if (IS_ADD) {
        const float4 s0 = *reinterpret_cast<const float4*>(&bufs[op.src][row][n]);
        v.x += s0.x; v.y += s0.y; v.z += s0.z; v.w += s0.w;
        if (op.res >= 0) {
          const float4 s1 = *reinterpret_cast<const float4*>(&bufs[op.res][row][n]);
          v.x += s1.x; v.y += s1.y; v.z += s1.z; v.w += s1.w;
        }
      }
      *reinterpret_cast<float4*>(&bufs[op.dst][row][(IS_ADD ? 0 : op.dst_col) + n]) = v;
      // gout: the rows also leave for global memory (a training step keeps x + query_pos, the input of the projections that
      // follow, for their weight gradients; a backward chain's running sums)
      if (TRAIN && op.gout && m0 + row < M) *reinterpret_cast<float4*>(op.gout + (size_t)(m0 + row) * op.ldg + n) = v;
    }
  }
}

// The program travels BY VALUE as the kernel argument (<= 2.9 KB of the 4 KB kernarg segment): no upload, no lifetime
// to manage, and a hipGraph capture keeps its own copy.  It is read through the kernarg pointer with a run-time index
// (wave-uniform scalar loads); indexing the by-value struct directly would make the compiler copy it to scratch.
struct ChainProgram {
  int nops, M, nops2, split;      // split > 0: workgroups [0, blocks) run ops[0, nops), workgroups [split, split + blocks) run
  ChainOp ops[GD4D_CHAIN_MAX_OPS];   // ops[nops, nops + nops2) over the SAME rows (gd4d_row_chain2_fwd); split = blocks up to a multiple of 8
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) ChainProgram* rc_prog_ptr_t;   // the kernel argument segment: scalar loads
#else
typedef const ChainProgram* rc_prog_ptr_t;
#endif

// bx: the workgroup's index in the launch (blockIdx.x).
template <bool TRAIN>
__device__ __forceinline__ void row_chain_body(const rc_prog_ptr_t pp, const int bx) {
  const int M = pp->M, split = pp->split;
  const int blocks = (M + RC_M - 1) / RC_M;
  // Two programs: the second one's workgroups start at `split` = blocks rounded up to a multiple of 8, so that row block i of
  // both programs sits on the SAME XCD (workgroup j is dispatched to XCD j % 8) behind the same L2 - what a SIGNAL / WAIT pair
  // between the programs relies on; the workgroups in [blocks, split) have nothing to do.
  if (split > 0 && bx >= blocks && bx < split) return;
  const bool second = split > 0 && bx >= split;                    // workgroup-uniform: which of the two programs
  const int op_base = second ? pp->nops : 0;
  const int nops = second ? pp->nops2 : pp->nops;
  const int wg = second ? bx - split : bx;                         // row block
  const int wg_lo = second ? split : 0, wg_hi = wg_lo + blocks;
  extern __shared__ __attribute__((aligned(16))) char rc_smem[];
  float (*bufs)[RC_M][RC_LD] = reinterpret_cast<float (*)[RC_M][RC_LD]>(rc_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = wg * RC_M;
  // (row block 0 of EITHER program stamps the timeline: kind 1 = the first program, 7 = the second)
  trace_mark_if(g_trace_rowchain, (second ? 7ull : 1ull) | ((unsigned long long)nops << 8), wg == 0);
#if RC_PREFETCH
  // The weight images of a program (2.6 MB for chain B) are cold in this XCD's L2 when the launch starts - another
  // kernel streamed through it since their last use - and the GEMMs below fetch them with 32 KB per wave in flight: at
  // the ~2 us of a fabric round trip that is 17-24 B/clk per compute unit, which is what bounded every chain.  So the
  // workgroups that share an XCD (blockIdx = xcd mod 8) first TOUCH the images of all the program's GEMMs, each its
  // slice, one dword per 64 bytes, all requests in flight at once: the cold misses overlap instead of queueing behind
  // a 4-deep ring.  The touches are LDS-DMA loads (global_load_lds_dword: no destination register - an in-flight load
  // into a register the compiler believes to be free corrupts whatever it puts there next) into a 256-byte dump area per
  // wave behind the row buffers that nobody reads.
  typedef __attribute__((address_space(3))) void rc_lds_void_t;
  typedef const __attribute__((address_space(1))) void rc_glb_void_t;
  rc_lds_void_t* rc_dump = (rc_lds_void_t*)(rc_smem + sizeof(float) * RC_BUFS * RC_M * RC_LD + 256 * wave);
  auto touch1 = [&](const char* q) { __builtin_amdgcn_global_load_lds((rc_glb_void_t*)q, rc_dump, 4, 0, 0); };
  // the workgroups of THIS program that sit on this XCD: indices first, first + 8, ... below wg_hi
  const unsigned xcd = (unsigned)bx & 7u;
  const unsigned first = (unsigned)wg_lo + ((xcd - (unsigned)wg_lo) & 7u);
  const unsigned mine = ((unsigned)bx - first) >> 3, share = ((unsigned)wg_hi - 1u - first) / 8u + 1u;
  auto touch = [&](const void* base, size_t bytes) {
    if (!base || bytes < 4) return;
    const char* b = reinterpret_cast<const char*>(base);
    const unsigned pieces = (unsigned)((bytes + 63) / 64) + 1u;     // + 1: a range that is not 64-byte aligned
    for (unsigned c = tid; c < pieces; c += 64 * RC_WAVES) touch1(b + min((size_t)c * 64, bytes - 4));
  };
  // (bit 1) The SMALL operands of the operations from `from` on - this workgroup's rows of the global tensors the LOAD /
  // ADD / REFINE operations read, bias vectors, LayerNorm parameters, SMALL_LINEAR weights: a few KB, all cold (activations
  // were written by a kernel on another XCD, parameters were evicted by the gather).  Untouched, each operation starts
  // with a fabric round trip of its own (~2.5 us, one after the other along the chain: 31 of chain B's 50 us were left
  // with neither MFMAs nor weight loads).
  // MEASURED, ms per step over 200 steps, twice: RC_PREFETCH=1 (default) 1.864 / 1.862; =5 (images after the leading
  // LOADs) 1.863 / 1.869; =7 (+ small operands) 1.880 / 1.874; =3 1.886 / 1.886.  Walking the program for the small
  // touches (scalar loads of every operation's fields, a loop per operand) costs every launch more than the warm
  // parameters save chain B (-4 us in block 0's timeline): bits 1 and 2 stay off.
  auto touch_small = [&](int from) {
    const int rows = min(RC_M, M - m0);
    for (int oi = from; oi < op_base + nops; ++oi) {
      const ChainOp& o = pp->ops[oi];
      const size_t nb = (size_t)o.N * 4;
      switch (o.kind) {
        case GD4D_CHAIN_LOAD:
          for (int r = 0; r < rows; ++r) {
            touch(o.p0 + (size_t)(m0 + r) * o.ld0, nb);
            if (o.p1) touch(o.p1 + (size_t)(m0 + r) * o.ld1, nb);
          }
          break;
        case GD4D_CHAIN_ADD:
          if (o.p2) for (int r = 0; r < rows; ++r) touch(o.p2 + (size_t)(m0 + r) * o.ld2, nb);
          break;
        case GD4D_CHAIN_GEMM: touch(o.p1, nb); break;
        case GD4D_CHAIN_HEADGEMM: touch(o.p1, nb); touch(o.p3 + (size_t)m0 * o.ld0, (size_t)rows * o.ld0 * 4); break;
        case GD4D_CHAIN_LAYERNORM: touch(o.p0, nb); touch(o.p1, nb); break;
        case GD4D_CHAIN_SMALL_LINEAR: touch(o.p0, nb * o.K); touch(o.p1, nb); break;
        case GD4D_CHAIN_REFINE: touch(o.p0 + (size_t)m0 * 3, (size_t)rows * 12); break;
        default: break;
      }
    }
  };
  // (bit 0) the weight images of the program's GEMMs, this workgroup's share of each
  auto touch_images = [&]() {
    for (int oi = op_base; oi < op_base + nops; ++oi) {
      if (pp->ops[oi].kind == GD4D_CHAIN_HEADGEMM) {           // this workgroup's own aggregate rows (written by another XCD: cold)
        const int rows = min(RC_M, M - m0);
        const char* a = reinterpret_cast<const char*>(pp->ops[oi].p2 + (size_t)m0 * pp->ops[oi].ld0 * pp->ops[oi].K);
        const unsigned pieces = (unsigned)rows * (unsigned)pp->ops[oi].ld0 * (unsigned)pp->ops[oi].K / 16u;
        for (unsigned c = tid; c < pieces; c += 64 * RC_WAVES) touch1(a + (size_t)c * 64);
      }
      if (pp->ops[oi].kind != GD4D_CHAIN_GEMM && pp->ops[oi].kind != GD4D_CHAIN_HEADGEMM) continue;
      const char* img = reinterpret_cast<const char*>(pp->ops[oi].p0);
      const unsigned chunks = (unsigned)((pp->ops[oi].N + 15) / 16) * (unsigned)(pp->ops[oi].K / 32) *
                              ((pp->ops[oi].flags & GD4D_CHAIN_EXACT) ? 48u : 32u);   // 64-byte pieces (2 or 3 planes of 1 KB per fragment)
      const unsigned lo = (unsigned)((unsigned long long)chunks * mine / share), hi = (unsigned)((unsigned long long)chunks * (mine + 1) / share);
      for (unsigned c = lo + tid; c < hi; c += 64 * RC_WAVES) touch1(img + (size_t)c * 64);
    }
  };
  // (bit 2) The image touches go out AFTER the program's leading LOAD operations: vmcnt retires in order, so a LOAD
  // issued behind ~370 KB of touches per workgroup sees its rows only when all of those have arrived.
  int lead = 0;
  if (RC_PREFETCH & 4)
    while (lead < nops && pp->ops[op_base + lead].kind == GD4D_CHAIN_LOAD) ++lead;
  if (RC_PREFETCH & 2) touch_small(op_base + lead);
  if ((RC_PREFETCH & 1) && lead == 0) touch_images();
#endif
  // A WAIT that gives up (below) poisons what the program LOADs afterwards: a hand-off that failed must not look like a result.
  __shared__ int rc_poison;
  if (tid == 0) rc_poison = 0;
  bool poisoned = false;
  for (int oi = op_base; oi < op_base + nops; ++oi) {
#if RC_PREFETCH
    if ((RC_PREFETCH & 1) && lead > 0 && oi == op_base + lead) touch_images();
#endif
    const ChainOp op = pp->ops[oi];                        // uniform: scalar loads
    switch (op.kind) {
      case GD4D_CHAIN_LOAD: {                              // dst[:, :N] = f(p0[m, :N]) (+ p1[m, :N])
        if ((op.N & 3) == 0 && (op.dst_col & 3) == 0) { rc_rows<false, TRAIN>(op, bufs, m0, M, lane, wave); break; }
        for (int e = tid; e < RC_M * op.N; e += 64 * RC_WAVES) {      // a handful of columns (reference points)
          const int row = e / op.N, n = e - row * op.N;
          const int m = min(m0 + row, M - 1);
          float v = rc_act_in(op.p0[(size_t)m * op.ld0 + n], op.flags);
          if (op.p1) v += op.p1[(size_t)m * op.ld1 + n];
          bufs[op.dst][row][op.dst_col + n] = v;
          if (TRAIN && op.gout && m0 + row < M) op.gout[(size_t)(m0 + row) * op.ldg + n] = v;
        }
        break;
      }
      case GD4D_CHAIN_GEMM:
        if (op.flags & GD4D_CHAIN_EXACT) rc_gemm<true, TRAIN>(op, bufs, m0, M, lane, wave);
        else rc_gemm<false, TRAIN>(op, bufs, m0, M, lane, wave);
        break;
      case GD4D_CHAIN_HEADGEMM: rc_headgemm(op, bufs, m0, M, lane, wave); break;
      case GD4D_CHAIN_LAYERNORM: rc_layernorm(op, bufs, m0, M, lane, wave); break;
      case GD4D_CHAIN_ADD: rc_rows<true, TRAIN>(op, bufs, m0, M, lane, wave); break;   // dst = src + (res buffer) + (p2 global)
      case GD4D_CHAIN_DROPMASK: {                         // buf[dst] = nn.Dropout's mask of a forward GEMM applied to buf[src]
        if (!TRAIN) break;
        const uint32_t lo = reinterpret_cast<const uint32_t*>(op.p0)[0], hi = reinterpret_cast<const uint32_t*>(op.p0)[1];
        const int nv = op.N / 4;
        for (int e = tid; e < RC_M * nv; e += 64 * RC_WAVES) {
          const int row = e / nv, n = 4 * (e - row * nv);
          const uint32_t id = (uint32_t)(m0 + row) * (uint32_t)op.N + (uint32_t)n;
          float4 v = *reinterpret_cast<const float4*>(&bufs[op.src][row][n]);
          v.x = mha_drop_keep(lo, hi, id, (uint32_t)op.reserved) ? v.x * op.eps : 0.f;
          v.y = mha_drop_keep(lo, hi, id + 1, (uint32_t)op.reserved) ? v.y * op.eps : 0.f;
          v.z = mha_drop_keep(lo, hi, id + 2, (uint32_t)op.reserved) ? v.z * op.eps : 0.f;
          v.w = mha_drop_keep(lo, hi, id + 3, (uint32_t)op.reserved) ? v.w * op.eps : 0.f;
          *reinterpret_cast<float4*>(&bufs[op.dst][row][n]) = v;
          if (op.gout && m0 + row < M) *reinterpret_cast<float4*>(op.gout + (size_t)(m0 + row) * op.ldg + n) = v;
        }
        break;
      }
      case GD4D_CHAIN_LN_BWD:
        if (TRAIN)
          rc_layernorm_bwd(op, bufs, reinterpret_cast<float*>(rc_smem + sizeof(float) * RC_BUFS * RC_M * RC_LD + 256 * RC_WAVES), m0, M, wg, tid);
        break;
      case GD4D_CHAIN_SMALL_LINEAR: {                      // K <= 8 inputs (position_encoder's first Linear): plain FMAs
        // The first version evaluated inverse_sigmoid (a division and a logarithm) for every (row, output, input) and
        // fetched the weights inside the loop: 5.1 us for 16 x 256 x 3 MACs, on the critical path of the dual launch.
        // Now: the K inputs of a row are transformed once (into columns 8 .. 8 + K of the source buffer, which nothing
        // reads), and when every thread keeps one output column its weights are fetched once.
        const int K = op.K, N = op.N;
        if (op.flags & GD4D_CHAIN_INV_SIGMOID) {
          if (tid < RC_M * K) {
            const int row = tid / K, k = tid - row * K;
            bufs[op.src][row][8 + k] = inv_sigmoid(bufs[op.src][row][k]);
          }
          __syncthreads();
        }
        const int col0 = (op.flags & GD4D_CHAIN_INV_SIGMOID) ? 8 : 0;
        if ((64 * RC_WAVES) % N == 0) {                    // thread -> one column n, rows tid / N + i * (threads / N)
          const int n = tid % N;
          float w[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) w[k] = k < K ? op.p0[(size_t)n * K + k] : 0.f;
          const float b = op.p1 ? op.p1[n] : 0.f;
          for (int row = tid / N; row < RC_M; row += (64 * RC_WAVES) / N) {
            float v = b;
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (k < K) v = fmaf(bufs[op.src][row][col0 + k], w[k], v);
            if (op.flags & GD4D_CHAIN_RELU) v = fmaxf(v, 0.f);
            bufs[op.dst][row][n] = v;
            if (TRAIN && op.gout && m0 + row < M) op.gout[(size_t)(m0 + row) * op.ldg + n] = v;
          }
        } else {
          for (int e = tid; e < RC_M * N; e += 64 * RC_WAVES) {
            const int row = e / N, n = e - row * N;
            float v = op.p1 ? op.p1[n] : 0.f;
            for (int k = 0; k < K; ++k) v = fmaf(bufs[op.src][row][col0 + k], op.p0[(size_t)n * K + k], v);
            if (op.flags & GD4D_CHAIN_RELU) v = fmaxf(v, 0.f);
            bufs[op.dst][row][n] = v;
            if (TRAIN && op.gout && m0 + row < M) op.gout[(size_t)(m0 + row) * op.ldg + n] = v;
          }
        }
        break;
      }
      case GD4D_CHAIN_REFINE: {                            // detr3d_transformer.py:201-214 on src = reg-branch output
        if (tid < RC_M && m0 + tid < M) {
          const int m = m0 + tid;
          const float* t = bufs[op.src][tid];
          const float* r = op.p0 + (size_t)m * 3;
          float* o = op.gout + (size_t)m * 3;
          const float x = t[0] + inv_sigmoid(r[0]), y = t[1] + inv_sigmoid(r[1]), z = t[4] + inv_sigmoid(r[2]);
          o[0] = 1.0f / (1.0f + expf(-x));
          o[1] = 1.0f / (1.0f + expf(-y));
          o[2] = 1.0f / (1.0f + expf(-z));
          if (op.dst >= 0) { bufs[op.dst][tid][0] = o[0]; bufs[op.dst][tid][1] = o[1]; bufs[op.dst][tid][2] = o[2]; }
        } else if (tid < RC_M && op.dst >= 0) {              // rows past M: finite filler (never stored)
          bufs[op.dst][tid][0] = bufs[op.dst][tid][1] = bufs[op.dst][tid][2] = 0.5f;
        }
        break;
      }
      case GD4D_CHAIN_SIGNAL: {
        // This row block's global outputs so far (gout of the operations above) are handed to row block wg of the OTHER
        // program of the launch.  Both sit on one XCD (see `split`), so the L2 is their point of coherence: every thread
        // waits until its stores have been acknowledged by the L2 (the L1 is write-through), the workgroup meets, one
        // thread raises the flag with a relaxed agent-scope atomic (performed in the L2; no cache maintenance - an
        // agent-scope RELEASE would write the whole L2 back on this multi-XCD part).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(op.gout) + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      case GD4D_CHAIN_WAIT: {
        // ... and the consumer: one thread polls the flag in the L2 (bounded: ~0.2 s - then the error word at gout is raised,
        // the rows this program LOADs from here on are replaced by NaN, and the program goes on: a loud wrong result instead of
        // a hung GPU; the host reads the error word, ops.check_handoff), the barrier below releases the others.  The rows the next LOAD reads were never in this CU's L1 (invalidated at the launch's
        // start; row blocks are 16 rows of >= 1 KB: no line is shared with another block), so plain loads see them.
        if (tid == 0) {
          const unsigned* f = reinterpret_cast<const unsigned*>(op.p0) + wg;
          unsigned spins = 0;
          while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            if (++spins > (1u << 20)) {
              if (op.gout) atomicAdd(reinterpret_cast<unsigned*>(op.gout), 1u);
              rc_poison = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(8);
          }
          // The consumer takes the flag down again: the flags of a request return to zero by themselves, so that one buffer per
          // request slot serves every request (no fill in the replayed graph).  The next SIGNAL for this row block belongs to a
          // later launch of the same stream.  (A WAIT that gave up leaves the flag alone: the host zeroes the buffers when it
          // reports the error word, ops.check_handoff.)
          if (!rc_poison) __hip_atomic_store(const_cast<unsigned*>(f), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
      default: break;
    }
    __syncthreads();
    if (op.kind == GD4D_CHAIN_WAIT) poisoned = rc_poison != 0;             // (workgroup-uniform)
    if (poisoned && op.kind == GD4D_CHAIN_LOAD && op.dst >= 0) {
      for (int e = tid; e < RC_M * op.N; e += 64 * RC_WAVES) {
        const int row = e / op.N, n = e - row * op.N;
        bufs[op.dst][row][op.dst_col + n] = __builtin_nanf("");
        if (TRAIN && op.gout && m0 + row < M) op.gout[(size_t)(m0 + row) * op.ldg + n] = __builtin_nanf("");
      }
      __syncthreads();
    }
#if RC_TRACE_OPS
    trace_mark_if(g_trace_rowchain, 0x40ull | ((unsigned long long)op.kind << 8) | ((unsigned long long)op.N << 16) | ((unsigned long long)op.K << 32) | (second ? 1ull << 63 : 0ull), wg == 0);
#endif
  }
  trace_mark_if(g_trace_rowchain, (second ? 0x87ull : 0x81ull) | ((unsigned long long)nops << 8), wg == 0);
}

template <bool TRAIN>
__global__ __launch_bounds__(64 * RC_WAVES) void row_chain_kernel(const ChainProgram by_value) {
#if defined(__HIP_DEVICE_COMPILE__)
  const rc_prog_ptr_t pp = (rc_prog_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // explicit arguments start at 0
#else
  const ChainProgram* pp = &by_value;
#endif
  (void)by_value;
  row_chain_body<TRAIN>(pp, (int)blockIdx.x);
}

// gd4d_row_chain_guest_fwd: the row chain(s) of a launch plus GUEST workgroups that run one decoder layer's value_proj over a few
// pyramid levels (value_proj_astat_body, gd4d_value_proj_body.h).  A chain occupies ceil(M / 16) (two programs: twice that)
// of the 256 compute units for 15-50 us and is bound by the latency of ONE unit; the others idle.  The guests are dispatched
// after the chain's workgroups (higher indices), ask for the same LDS (one workgroup per compute unit) and share nothing with
// the chain: no hand-off, no second stream, no cross-stream edge in a replayed graph - the launch ends when both have.
template <bool IN_CHLAST>
__global__ __launch_bounds__(64 * RC_WAVES) void row_chain_guest_kernel(const ChainProgram by_value, const VpaParams guest, const int gbase,
                                                                         const int gcount) {
#if defined(__HIP_DEVICE_COMPILE__)
  const rc_prog_ptr_t pp = (rc_prog_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // explicit arguments start at 0
#else
  const ChainProgram* pp = &by_value;
#endif
  (void)by_value;
  if ((int)blockIdx.x >= gbase) {                                       // workgroup-uniform
    extern __shared__ __attribute__((aligned(16))) char rc_smem[];
    trace_mark_if(g_trace_rowchain, 9ull, (int)blockIdx.x == gbase);
    value_proj_astat_body<RC_WAVES, false, false, false, 0, IN_CHLAST>(guest, (int)blockIdx.x - gbase, gcount, rc_smem);
    trace_mark_if(g_trace_rowchain, 0x89ull, (int)blockIdx.x == gbase);
    return;
  }
  row_chain_body<false>(pp, (int)blockIdx.x);
}

// A TRAINING chain with the pyramid gradient's record fills as guests (gd4d_row_chain_fill_fwd): the fills of a step need the scan
// over all layers' counts and nothing from the backward pass; they stream plans and scatter 8-byte records - memory work a backward
// chain (57 of the 256 compute units for 50-70 us, bound by the latency of one) leaves room for.  Until round 6 they rode in the
// attention backward's dk / dv launch, which they made 20-38 us longer (53-71 against 33 us).  Guest workgroups sit behind the
// chain's, stay (gcount of them) and walk the (position, head) rows of up to two jobs: wave w of guest g takes rows
// g RC_WAVES + w, + gcount RC_WAVES, ...
#ifndef RC_FILL_FP
#define RC_FILL_FP 8
#endif
__global__ __launch_bounds__(64 * RC_WAVES) void row_chain_fill_kernel(const ChainProgram by_value, const FillGuest fg, const int gbase,
                                                                        const int gcount) {
#if defined(__HIP_DEVICE_COMPILE__)
  const rc_prog_ptr_t pp = (rc_prog_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // explicit arguments start at 0
#else
  const ChainProgram* pp = &by_value;
#endif
  (void)by_value;
  if ((int)blockIdx.x >= gbase) {                                       // workgroup-uniform
    const int rows0 = fg.BQ[0] * fg.HH, rows1 = fg.hdr[1] ? fg.BQ[1] * fg.HH : 0;
    const int wave = (int)(threadIdx.x >> 6), step = gcount * RC_WAVES;
    for (int ph = ((int)blockIdx.x - gbase) * RC_WAVES + wave; ph < rows0 + rows1; ph += step) {
      const int job = ph >= rows0 ? 1 : 0;
      pyramid_grad_fill_body<RC_FILL_FP>(fg.hdr[job], fg.pair[job], fg.slots[job], fg.cap_t, fg.HH, fg.BQ[job], fg.start, fg.rec,
                                         fg.order[job], fg.id_base[job], ph - (job ? rows0 : 0));
    }
    return;
  }
  row_chain_body<true>(pp, (int)blockIdx.x);
}

// XCC id of every workgroup of a launch (gd4d_xcd_placement_probe): the hand-offs above rely on workgroups j and j + 8 k sharing
// an XCD; the host checks that once per device before it builds programs with SIGNAL / WAIT.
__global__ void xcd_placement_probe_kernel(int32_t* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = (int32_t)(v & 0xfu);
}

}  // namespace gd4d

extern "C" void gd4d_trace_set_rowchain(unsigned long long* p) { gd4d::trace_set_rowchain(p); }

extern "C" size_t gd4d_chain_op_bytes(void) { return sizeof(gd4d_chain_op); }

extern "C" size_t gd4d_chain_weight_image_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
  return (size_t)((N + 15) / 16) * (K / 32) * 2048;
}

extern "C" int gd4d_chain_weight_image(const float* weight, int N, int K, void* image, void* stream) {
  using namespace gd4d;
  if (!weight || !image || N <= 0 || K <= 0) return GD4D_EINVAL;
  if (K % 64 != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(image)) return GD4D_EALIGN;
  const int frags = ((N + 15) / 16) * (K / 32);
  hipLaunchKernelGGL(chain_weight_image_kernel<2>, dim3((frags + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), weight,
                     static_cast<char*>(image), N, K);
  return check_launch();
}

extern "C" size_t gd4d_chain_weight_image_exact_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 32 != 0) return 0;
  return (size_t)((N + 15) / 16) * (K / 32) * 3072;
}

extern "C" int gd4d_chain_weight_image_exact(const float* weight, int N, int K, void* image, void* stream) {
  using namespace gd4d;
  if (!weight || !image || N <= 0 || K <= 0) return GD4D_EINVAL;
  if (K % 64 != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(image)) return GD4D_EALIGN;
  const int frags = ((N + 15) / 16) * (K / 32);
  hipLaunchKernelGGL(chain_weight_image_kernel<3>, dim3((frags + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), weight,
                     static_cast<char*>(image), N, K);
  return check_launch();
}

extern "C" size_t gd4d_image_job_bytes(void) { return sizeof(gd4d_image_job); }

extern "C" int gd4d_chain_weight_image_group(const gd4d_image_job* jobs_device, int count, int total_frags, void* stream) {
  using namespace gd4d;
  if (!jobs_device || count <= 0 || total_frags <= 0) return GD4D_EINVAL;
  if (count > GD4D_IMAGE_JOBS_MAX) return GD4D_EUNSUPPORTED;
  hipLaunchKernelGGL(chain_weight_image_group_kernel, dim3((total_frags + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                     jobs_device, count, total_frags);
  return check_launch();
}

// which: 0 = the only program of a launch, 1 = the first of two, 2 = the second of two
static int rc_validate(const gd4d_chain_op* program, int nops, int which) {
  const bool two_programs = which != 0;
  using namespace gd4d;
  for (int i = 0; i < nops; ++i) {
    const gd4d_chain_op& op = program[i];
    const bool buf_ok = op.src < RC_BUFS && op.dst < RC_BUFS && op.res < RC_BUFS;
    if (!buf_ok) return GD4D_EINVAL;
    switch (op.kind) {
      case GD4D_CHAIN_LOAD:
        if (!op.p0 || op.dst < 0 || op.N <= 0 || op.dst_col < 0 || op.dst_col + op.N > RC_W) return GD4D_EINVAL;
        if ((op.N & 3) == 0 && (op.dst_col & 3) == 0 &&         // the float4 path
            (!aligned16(op.p0) || (op.ld0 & 3) || (op.p1 && (!aligned16(op.p1) || (op.ld1 & 3))))) return GD4D_EALIGN;
        if (op.gout && (op.N & 3) == 0 && (op.dst_col & 3) == 0 && (!aligned16(op.gout) || (op.ldg & 3))) return GD4D_EALIGN;   // (the float4 path)
        break;
      case GD4D_CHAIN_GEMM:
        if (!op.p0 || op.src < 0 || op.K <= 0 || op.N <= 0 || (op.dst < 0 && !op.gout)) return GD4D_EINVAL;
        if (op.p3 && !op.p2 && !(op.flags & GD4D_CHAIN_DROPOUT)) return GD4D_EINVAL;      // a second addend without a first
        if (op.K % (32 * RC_DEPTH) != 0 || op.K > RC_W) return GD4D_EUNSUPPORTED;
        if (op.dst >= 0 && (op.dst_col < 0 || op.dst_col + op.N > RC_W)) return GD4D_EINVAL;
        if (op.dst >= 0 && op.dst == op.src) return GD4D_EINVAL;          // waves would overwrite rows others still read
        if (!aligned16(op.p0)) return GD4D_EALIGN;
        if (op.flags & GD4D_CHAIN_SRC2)
          if (op.res < 0 || op.res == op.dst || op.ld0 <= 0 || op.ld0 % (RC_COLS * RC_WAVES) != 0 || op.ld0 >= op.N) return GD4D_EINVAL;
        if ((op.flags & GD4D_CHAIN_DROPOUT) && (!op.p3 || ((uintptr_t)op.p3 & 7) || (op.flags & (GD4D_CHAIN_SPLIT_OUT | GD4D_CHAIN_MASK_P2))))
          return GD4D_EINVAL;
        if ((op.flags & GD4D_CHAIN_MASK_P2) && (!op.p2 || op.p3 || (op.flags & (GD4D_CHAIN_SPLIT_OUT | GD4D_CHAIN_SRC2)))) return GD4D_EINVAL;
        if (op.flags & GD4D_CHAIN_SPLIT_OUT)
          if (!op.gout || !op.p2 || !op.p3 || op.ldg <= 0 || op.ld2 <= 0 || op.ld1 <= 0 || op.ldg + op.ld2 + op.ld1 != op.N)
            return GD4D_EINVAL;
        if (op.flags & GD4D_CHAIN_SPLIT_KV) {
          if (op.flags & (GD4D_CHAIN_SPLIT_OUT | GD4D_CHAIN_MASK_P2 | GD4D_CHAIN_DROPOUT | GD4D_CHAIN_RELU | GD4D_CHAIN_SIGMOID)) return GD4D_EINVAL;
          if (op.N != 3 * RC_COLS * RC_WAVES) return GD4D_EUNSUPPORTED;
          if (!op.p2 || !op.p3 || op.ld2 <= 0 || op.ld1 <= 0 || (op.ld1 & 7) || (op.ld2 & 7)) return GD4D_EINVAL;
          if (!aligned16(op.p2) || !aligned16(op.p3)) return GD4D_EALIGN;
        }
        break;
      case GD4D_CHAIN_HEADGEMM:
        if (!op.p0 || !op.p2 || !op.p3 || op.K <= 0 || op.N <= 0 || op.ld0 <= 0 || (op.dst < 0 && !op.gout)) return GD4D_EINVAL;
        if (op.N % op.ld0 != 0 || (op.N / op.ld0) % 32 != 0 || op.K % 64 != 0 || op.N % 64 != 0) return GD4D_EUNSUPPORTED;
        if ((op.flags & GD4D_CHAIN_ADD_GOUT) && (!op.gout || op.dst < 0 || op.ldg < op.N)) return GD4D_EINVAL;
        if (op.dst >= 0 && (op.dst_col < 0 || op.dst_col + op.N > RC_W)) return GD4D_EINVAL;
        if (!aligned16(op.p0) || !aligned16(op.p2)) return GD4D_EALIGN;
        break;
      case GD4D_CHAIN_LAYERNORM:
        if (!op.p0 || !op.p1 || op.src < 0 || op.N <= 0 || op.N > RC_W || (op.dst < 0 && !op.gout)) return GD4D_EINVAL;
        if (op.N % 64 != 0) return GD4D_EUNSUPPORTED;
        if (!aligned16(op.p0) || !aligned16(op.p1) || (op.gout && (!aligned16(op.gout) || (op.ldg & 3)))) return GD4D_EALIGN;
        if (op.p2 && (op.res < 0 || op.res == op.src || op.res == op.dst)) return GD4D_EINVAL;    // second output: its own buffer
        if (op.p2 && (!aligned16(op.p2) || (op.ld2 & 3))) return GD4D_EALIGN;
        break;
      case GD4D_CHAIN_DROPMASK:
        if (!op.p0 || ((uintptr_t)op.p0 & 7) || op.src < 0 || op.dst < 0 || op.N <= 0 || op.N > RC_W || (op.N & 3)) return GD4D_EINVAL;
        if (op.gout && (!aligned16(op.gout) || (op.ldg & 3))) return GD4D_EALIGN;
        break;
      case GD4D_CHAIN_LN_BWD:
        if (!op.p0 || op.src < 0 || op.N <= 0 || op.N > RC_W) return GD4D_EINVAL;
        if (!op.p3 && (op.res < 0 || op.res == op.src || op.res == op.dst)) return GD4D_EINVAL;
        if (op.p3 && (!aligned16(op.p3) || (op.ld1 & 3) || op.ld1 < op.N)) return GD4D_EALIGN;
        if (op.dst < 0 && !op.gout && !op.p2) return GD4D_EINVAL;
        if ((op.flags & GD4D_CHAIN_RELU) && !op.p1) return GD4D_EINVAL;
        if (op.N % 64 != 0) return GD4D_EUNSUPPORTED;
        if (!aligned16(op.p0) || (op.p1 && !aligned16(op.p1)) || (op.gout && (!aligned16(op.gout) || (op.ldg & 3)))) return GD4D_EALIGN;
        break;
      case GD4D_CHAIN_ADD:
        if (op.src < 0 || op.dst < 0 || op.N <= 0 || op.N > RC_W || (op.N & 3)) return GD4D_EINVAL;
        if (op.p2 && (!aligned16(op.p2) || (op.ld2 & 3))) return GD4D_EALIGN;
        if (op.gout && (!aligned16(op.gout) || (op.ldg & 3))) return GD4D_EALIGN;
        break;
      case GD4D_CHAIN_SMALL_LINEAR:
        if (!op.p0 || op.src < 0 || op.dst < 0 || op.dst == op.src || op.K <= 0 || op.K > 8 || op.N <= 0 || op.N > RC_W)
          return GD4D_EINVAL;
        break;
      case GD4D_CHAIN_REFINE:
        if (op.src < 0 || !op.p0 || !op.gout || (op.dst >= 0 && op.dst == op.src)) return GD4D_EINVAL;
        break;
      // Workgroups of a launch are dispatched in index order, the first program's before the second's: a WAIT in the SECOND
      // program is answered by a workgroup that was dispatched before the waiting one - resident or finished, and never
      // blocked itself - whatever else keeps the device busy.  The other direction could leave every compute unit to
      // spinning consumers whose producers are still in the queue (five requests in flight would do): refused.
      case GD4D_CHAIN_SIGNAL:
        if (!op.gout || !two_programs || which != 1) return GD4D_EINVAL;
        break;
      case GD4D_CHAIN_WAIT:       // (what was handed over must enter through a LOAD: that is where a time-out is poisoned)
        if (!op.p0 || !op.gout || !two_programs || which != 2 || i + 1 >= nops || program[i + 1].kind != GD4D_CHAIN_LOAD) return GD4D_EINVAL;
        break;
      default: return GD4D_EINVAL;
    }
  }
  return GD4D_OK;
}

static int rc_launch(const gd4d_chain_op* a, int na, const gd4d_chain_op* b, int nb, int M, void* stream,
                     const gd4d_chain_guest* guest = nullptr, const gd4d::FillGuest* fills = nullptr, int fill_workgroups = 0) {
  using namespace gd4d;
  if (!a || na <= 0 || M <= 0 || nb < 0 || (nb > 0 && !b)) return GD4D_EINVAL;
  if (na + nb > GD4D_CHAIN_MAX_OPS) return GD4D_EUNSUPPORTED;
  if (int rc = rc_validate(a, na, nb > 0 ? 1 : 0)) return rc;
  if (nb > 0)
    if (int rc = rc_validate(b, nb, 2)) return rc;
  for (int w = 0; w < 2; ++w)                              // the K / V planes must hold this launch's row blocks
    for (int i = 0; i < (w ? nb : na); ++i) {
      const gd4d_chain_op& op = (w ? b : a)[i];
      if (op.kind == GD4D_CHAIN_GEMM && (op.flags & GD4D_CHAIN_SPLIT_KV)) {
        const long long blocks = (M + RC_M - 1) / RC_M, heads = RC_COLS * RC_WAVES / 32;
        if (op.ld2 < heads * blocks * 512 || op.ld1 < heads * ((blocks + 1) / 2) * 1024) return GD4D_EINVAL;
      }
    }
  // the operations only a training step uses (and the stores of LOAD / ADD / SMALL_LINEAR) are compiled into a second
  // instantiation: with them in, the inference step's chains ran 10 % slower (221 against 163 registers, longer epilogues)
  bool train = false;
  for (int w = 0; w < 2; ++w) {
    const gd4d_chain_op* pr = w ? b : a;
    for (int i = 0; i < (w ? nb : na); ++i) {
      const gd4d_chain_op& op = pr[i];
      train = train || op.kind == GD4D_CHAIN_LN_BWD || op.kind == GD4D_CHAIN_DROPMASK ||
              (op.kind == GD4D_CHAIN_GEMM && (op.flags & (GD4D_CHAIN_MASK_P2 | GD4D_CHAIN_DROPOUT))) ||
              ((op.kind == GD4D_CHAIN_LOAD || op.kind == GD4D_CHAIN_ADD || op.kind == GD4D_CHAIN_SMALL_LINEAR) && op.gout);
    }
  }
  // the guests' job (gd4d_value_proj_fwd's geometry: one layer, pixel-major fp32 rows) - checked before anything touches the device
  VpaParams g{};
  if (guest) {
    if (train) return GD4D_EUNSUPPORTED;
    if (int rc = va_guest_params(guest, g)) return rc;
  }
  const size_t lds = sizeof(float) * RC_BUFS * RC_M * RC_LD + 256 * RC_WAVES + (train ? sizeof(float) * 8 * RC_W : 0);   // row buffers + the prefetch dump area (+ LN_BWD's partial sums)
  const void* kern = train ? reinterpret_cast<const void*>(row_chain_kernel<true>) : reinterpret_cast<const void*>(row_chain_kernel<false>);
  if (!allow_dynamic_lds(kern, (int)lds)) return GD4D_ELAUNCH;
  const int blocks = (M + RC_M - 1) / RC_M;
  ChainProgram prog{};
  const int split = nb > 0 ? (blocks + 7) & ~7 : 0;      // second program: same XCD per row block (see the kernel)
  prog.nops = na; prog.M = M; prog.nops2 = nb; prog.split = split;
  for (int i = 0; i < na; ++i) prog.ops[i] = a[i];
  for (int i = 0; i < nb; ++i) prog.ops[na + i] = b[i];
  if (guest) {
    const int base = g.total;
    if (va_lds_bytes(1, false, RC_WAVES) > lds) return GD4D_EUNSUPPORTED;
    const int chain_wgs = nb > 0 ? split + blocks : blocks;
    const int gbase = chain_wgs;                           // (every index below it is a chain workgroup with rows to work on)
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int need = (base + RC_WAVES - 1) / RC_WAVES;
    // One workgroup per compute unit (the launch's LDS request).  Default: one workgroup per 8 tiles up to the number of compute
    // units - every wave makes ONE pass; the guests that find no free unit start when the first program's workgroups end (20 of
    // chain B's 55 us).  Measured at 174 workgroups of work, 142 units free: 142 guests x two passes 645 samples/s, 174 x one 659-661.
    int gcount = guest->workgroups > 0 ? guest->workgroups : cus;
    if (gcount < 8) gcount = 8;
    if (gcount > need) gcount = need;
    auto go = [&](auto kern) -> int {
      if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds)) return GD4D_ELAUNCH;
      hipLaunchKernelGGL(kern, dim3(gbase + gcount), dim3(64 * RC_WAVES), lds, static_cast<hipStream_t>(stream), prog, g, gbase, gcount);
      return check_launch();
    };
    return g.in_chlast ? go(row_chain_guest_kernel<true>) : go(row_chain_guest_kernel<false>);
  }
  if (fills) {                                             // (the training instantiation, whatever the program holds)
    if (guest) return GD4D_EINVAL;
    const void* fk = reinterpret_cast<const void*>(row_chain_fill_kernel);
    const size_t flds = sizeof(float) * RC_BUFS * RC_M * RC_LD + 256 * RC_WAVES + sizeof(float) * 8 * RC_W;
    if (!allow_dynamic_lds(fk, (int)flds)) return GD4D_ELAUNCH;
    const int gbase = nb > 0 ? split + blocks : blocks;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int rows = fills->BQ[0] * fills->HH + (fills->hdr[1] ? fills->BQ[1] * fills->HH : 0);
    const int need = (rows + RC_WAVES - 1) / RC_WAVES;
    int gcount = fill_workgroups > 0 ? fill_workgroups : 2 * cus;     // (the chain's LDS request: the guests share compute units at most two by two)
    if (gcount > need) gcount = need;
    if (gcount < 1) gcount = 1;
    hipLaunchKernelGGL(row_chain_fill_kernel, dim3(gbase + gcount), dim3(64 * RC_WAVES), flds, static_cast<hipStream_t>(stream), prog, *fills,
                       gbase, gcount);
    return check_launch();
  }
  if (train)
    hipLaunchKernelGGL(row_chain_kernel<true>, dim3(nb > 0 ? split + blocks : blocks), dim3(64 * RC_WAVES), lds, static_cast<hipStream_t>(stream), prog);
  else
    hipLaunchKernelGGL(row_chain_kernel<false>, dim3(nb > 0 ? split + blocks : blocks), dim3(64 * RC_WAVES), lds, static_cast<hipStream_t>(stream), prog);
  return check_launch();
}

extern "C" int gd4d_xcd_placement_probe(int32_t* out, int blocks, void* stream) {
  if (!out || blocks <= 0) return GD4D_EINVAL;
  hipLaunchKernelGGL(gd4d::xcd_placement_probe_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out);
  return gd4d::check_launch();
}

extern "C" int gd4d_row_chain_fwd(const gd4d_chain_op* program, int nops, int M, void* stream) {
  return rc_launch(program, nops, nullptr, 0, M, stream);
}

extern "C" int gd4d_row_chain2_fwd(const gd4d_chain_op* program_a, int nops_a, const gd4d_chain_op* program_b, int nops_b, int M,
                                   void* stream) {
  if (!program_b || nops_b <= 0) return GD4D_EINVAL;
  return rc_launch(program_a, nops_a, program_b, nops_b, M, stream);
}

extern "C" size_t gd4d_chain_guest_bytes(void) { return sizeof(gd4d_chain_guest); }

extern "C" int gd4d_row_chain_guest_fwd(const gd4d_chain_op* program_a, int nops_a, const gd4d_chain_op* program_b, int nops_b, int M,
                                        const gd4d_chain_guest* guest, void* stream) {
  if (!guest || nops_b < 0 || (nops_b > 0 && !program_b)) return GD4D_EINVAL;
  return rc_launch(program_a, nops_a, nops_b > 0 ? program_b : nullptr, nops_b, M, stream, guest);
}

// A training chain (one or two programs) that carries record fills of the pyramid gradient as guest workgroups: see row_chain_fill_kernel.
// jobs / start / records / fill_*: as gd4d_mha_core_bwd_fill.  workgroups: guest workgroups (0: two per compute unit).
extern "C" int gd4d_row_chain_fill_fwd(const gd4d_chain_op* program_a, int nops_a, const gd4d_chain_op* program_b, int nops_b, int M,
                                       const gd4d_fill_job* jobs, int njobs, const int32_t* start, void* records, int fill_B, int fill_N,
                                       int fill_Hh, int fill_P, int workgroups, void* stream) {
  using namespace gd4d;
  if (!jobs || njobs < 1 || njobs > 2 || !start || !records || fill_B <= 0 || fill_N <= 0 || fill_Hh <= 0) return GD4D_EINVAL;
  if (nops_b < 0 || (nops_b > 0 && !program_b) || workgroups < 0) return GD4D_EINVAL;
  if ((fill_P != kPoints && fill_P != 8) || fill_N > 64 || fill_B > 16 || fill_Hh > kPlanHdr) return GD4D_EUNSUPPORTED;
  FillGuest fg{};
  for (int j = 0; j < njobs; ++j) {
    const gd4d_fill_job& jb = jobs[j];
    if (!jb.plan || !jb.slots || jb.Q <= 0) return GD4D_EINVAL;
    if ((unsigned long long)jb.id_base + (unsigned long long)fill_B * jb.Q * fill_Hh > (1ull << 26)) return GD4D_EUNSUPPORTED;
    fg.hdr[j] = static_cast<const int*>(jb.plan);
    fg.pair[j] = reinterpret_cast<const uint2*>(static_cast<const char*>(jb.plan) + plan_hdr_bytes(fill_B, jb.Q));
    fg.slots[j] = static_cast<const uint2*>(jb.slots);
    fg.order[j] = jb.query_order;
    fg.id_base[j] = jb.id_base;
    fg.BQ[j] = fill_B * jb.Q;
  }
  fg.HH = fill_Hh;
  fg.cap_t = plan_cap_t(fill_N, fill_P);
  fg.start = start;
  fg.rec = static_cast<uint2*>(records);
  return rc_launch(program_a, nops_a, nops_b > 0 ? program_b : nullptr, nops_b, M, stream, nullptr, &fg, workgroups);
}

// The step right after the decoder: the head's per-layer box epilogue (gd4d_box_head_fwd) and
// NMSFreeCoder.decode_single (gd4d_nms_free_decode_fwd), one workgroup per batch element.
//
// Reference: projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py:47-96 and denormalize_bbox
// (projects/mmdet3d_plugin/core/bbox/util.py:58-87): sigmoid over (num_query x num_classes) logits, top-`max_num`
// of the flattened scores, label = index % num_classes, box = bbox_preds[index // num_classes], de-normalise
// (exp of the log sizes, atan2 of the sine/cosine pair), keep boxes whose centre lies in post_center_range
// (and above score_threshold).  The reference runs this as ~15 torch ops (topk, gathers, cat, masks).
//
// One workgroup (1024 threads) per batch element:
//   1. scores = sigmoid(logit) as IEEE bits (positive floats order like unsigned ints), kept in LDS;
//   2. radix select of the K-th largest key: four 8-bit histogram passes (per-wave private LDS histograms, the
//      digit walk as a shuffle suffix-scan);
//   3. gather the keys above the threshold and the lowest-index ties at the threshold: exactly K survivors;
//   4. rank sort (each survivor counts the survivors that precede it: key descending, ties by ascending index,
//      deterministic), decode + range test, each thread writing its survivor to its rank's row.
// The boolean compaction of the kept boxes (a host-visible size) stays with the caller, as in the reference.
#include "gd4d_common.h"

#ifndef GD4D_DEC_STOP
#define GD4D_DEC_STOP 0      // dev ablation builds: return after phase N
#endif

namespace gd4d {

constexpr int DEC_THREADS = 1024;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int DEC_KMAX = 1024;          // max_num supported (one sorted entry per thread)
constexpr size_t DEC_MAX_DYN_LDS = 128 * 1024;   // score keys live in LDS: Q*C <= 32768

struct DecodeParams {
  const float* cls;        // (B, Q, C) logits
  const float* bbox;       // (B, Q, code)
  float* boxes;            // (B, K, 9 or 7)
  float* scores;           // (B, K)
  int32_t* labels;         // (B, K)
  uint8_t* keep;           // (B, K)
  int Q, C, code, K;
  float range_lo[3], range_hi[3];
  float score_thr;         // < 0: none
};

__device__ __forceinline__ unsigned score_key(float logit) {
  const float s = 1.0f / (1.0f + expf(-logit));      // in [0, 1]: non-negative, bit pattern is monotone
  return __float_as_uint(s);
}

__device__ __forceinline__ unsigned long long pack_cand(unsigned key, int idx) {
  return ((unsigned long long)key << 32) | (unsigned)~(unsigned)idx;
}

__global__ __launch_bounds__(DEC_THREADS) void nms_free_decode_kernel(const DecodeParams p) {
  extern __shared__ unsigned s_keys[];              // n score keys (computed once)
  __shared__ unsigned s_whist[DEC_WAVES][256];      // one private digit histogram per wave
  __shared__ unsigned s_hist[256];
  __shared__ unsigned long long s_cand[DEC_KMAX];   // survivors as key << 32 | ~index: larger sorts first
  __shared__ unsigned s_wave_total[DEC_WAVES];
  __shared__ unsigned s_prefix, s_remaining, s_eq_total, s_count;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int n = p.Q * p.C;
  const float* cls = p.cls + (size_t)b * n;
  const int K = min(p.K, n);
  const unsigned long long lanes_below = (1ull << lane) - 1ull;

  // one workgroup on one CU cannot hide HBM latency by occupancy: issue every load up front (8 x float4 per thread
  // covers Q*C <= 32768), then convert
  if ((n & 3) == 0) {
    const float4* cls4 = reinterpret_cast<const float4*>(cls);   // b*n*4 bytes: 16 B aligned when n % 4 == 0
    const int n4 = n >> 2;
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * DEC_THREADS;
      v[u] = i < n4 ? cls4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + u * DEC_THREADS;
      if (i < n4) {
        uint4 k;
        k.x = score_key(v[u].x); k.y = score_key(v[u].y); k.z = score_key(v[u].z); k.w = score_key(v[u].w);
        reinterpret_cast<uint4*>(s_keys)[i] = k;
      }
    }
  } else {
    for (int i = tid; i < n; i += DEC_THREADS) s_keys[i] = score_key(cls[i]);
  }
  if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)K; s_count = 0u; }
#if GD4D_DEC_STOP == 1
  return;
#endif

  // ---- radix select: the key of the K-th largest element, 8 bits per pass ----
  for (int pass = 3; pass >= 0; --pass) {
#pragma unroll
    for (int u = 0; u < DEC_WAVES * 256 / DEC_THREADS; ++u) (&s_whist[0][0])[tid + u * DEC_THREADS] = 0u;
    __syncthreads();                                // also: keys / previous pass's prefix are visible
    const unsigned prefix = s_prefix;
    const unsigned hi_mask = pass == 3 ? 0u : (0xffffffffu << (8 * (pass + 1)));
    unsigned* hist = s_whist[wave];
#pragma unroll 4
    for (int i = tid; i < n; i += DEC_THREADS) {
      const unsigned k = s_keys[i];
      if ((k & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(k >> (8 * pass)) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256) {
      unsigned t = 0u;
#pragma unroll
      for (int w = 0; w < DEC_WAVES; ++w) t += s_whist[w][tid];
      s_hist[tid] = t;
    }
    __syncthreads();
    if (wave == 0) {                                // lane l owns bins 4l..4l+3; suffix sums by shuffles
      const unsigned h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2],
                     h3 = s_hist[4 * lane + 3];
      const unsigned tot = h0 + h1 + h2 + h3;
      unsigned x = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned y = __shfl_down(x, off);
        if (lane + off < 64) x += y;
      }
      const unsigned rem = s_remaining;
      const unsigned a3 = x - tot, a2 = a3 + h3, a1 = a2 + h2, a0 = a1 + h1;   // keys in bins above bin b
      int sub = -1;
      unsigned above = 0u, here = 0u;
      if (a3 < rem && rem <= a3 + h3) { sub = 3; above = a3; here = h3; }
      else if (a2 < rem && rem <= a2 + h2) { sub = 2; above = a2; here = h2; }
      else if (a1 < rem && rem <= a1 + h1) { sub = 1; above = a1; here = h1; }
      else if (a0 < rem && rem <= a0 + h0) { sub = 0; above = a0; here = h0; }
      if (sub >= 0) {                               // exactly one lane: the digit holding the K-th key
        s_prefix = prefix | ((unsigned)(4 * lane + sub) << (8 * pass));
        s_remaining = rem - above;
        s_eq_total = here;                          // after the last pass: how many keys equal the threshold
      }
    }
  }
  __syncthreads();
#if GD4D_DEC_STOP == 2
  return;
#endif
  const unsigned thr = s_prefix;                    // K-th largest key
  const unsigned need_eq = s_remaining;             // ties at the threshold to take (lowest indices first)
  const bool all_ties = s_eq_total == need_eq;      // the usual case: every key == thr is a survivor

  // ---- survivors: every key > thr (and every key == thr when all of them fit), in any order ----
  for (int i0 = 0; i0 < n; i0 += DEC_THREADS) {
    const int i = i0 + tid;
    const unsigned k = i < n ? s_keys[i] : 0u;
    const bool take = i < n && (k > thr || (all_ties && k == thr));
    const unsigned long long m = __ballot(take);
    unsigned base = 0u;
    if (lane == 0 && m) base = atomicAdd(&s_count, (unsigned)__popcll(m));
    base = __shfl(base, 0);
    if (take) s_cand[base + (unsigned)__popcll(m & lanes_below)] = pack_cand(k, i);
  }
  if (!all_ties) {
    // more keys == thr than places left: the lowest flat indices win.  Each thread owns a contiguous chunk,
    // counts its ties, a two-level scan ranks them.
    const int chunk = (n + DEC_THREADS - 1) / DEC_THREADS;
    const int c0 = min(tid * chunk, n), c1 = min(c0 + chunk, n);
    unsigned mine = 0u;
    for (int i = c0; i < c1; ++i) mine += s_keys[i] == thr ? 1u : 0u;
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned y = __shfl_up(incl, off);
      if (lane >= off) incl += y;
    }
    if (lane == 63) s_wave_total[wave] = incl;
    __syncthreads();                                // also publishes s_count (== K - need_eq)
    const unsigned ngt = s_count;
    unsigned rank = incl - mine;
    for (int w = 0; w < wave; ++w) rank += s_wave_total[w];
    if (mine > 0u && rank < need_eq) {
      for (int i = c0; i < c1 && rank < need_eq; ++i)
        if (s_keys[i] == thr) { s_cand[ngt + rank] = pack_cand(thr, i); ++rank; }
    }
  }
  __syncthreads();
#if GD4D_DEC_STOP == 3
  return;
#endif

  // ---- rank sort: candidate t's output row = how many candidates sort before it (key desc, index asc).
  // All lanes read the same s_cand[j] (an LDS broadcast), no barriers; K^2 / 1024 compares per thread ----
  if (tid >= K) return;
  const unsigned long long me = s_cand[tid];
  int r = 0;
#pragma unroll 16
  for (int j = 0; j < K; ++j) r += s_cand[j] > me ? 1 : 0;
#if GD4D_DEC_STOP == 4
  if (r == 0x12345) p.scores[0] = 0.f;
  return;
#endif

  // ---- decode my candidate into row r ----
  const int nbox = p.code > 8 ? 9 : 7;
  const int idx = (int)~(unsigned)me;
  const int qi = idx / p.C;
  const size_t o = (size_t)b * p.K + r;
  float* ob = p.boxes + o * nbox;
  const float* bb = p.bbox + ((size_t)b * p.Q + qi) * p.code;
  const float cx = bb[0], cy = bb[1], cz = bb[4];
  ob[0] = cx; ob[1] = cy; ob[2] = cz;
  ob[3] = expf(bb[2]); ob[4] = expf(bb[3]); ob[5] = expf(bb[5]);
  ob[6] = atan2f(bb[6], bb[7]);
  if (nbox == 9) { ob[7] = bb[8]; ob[8] = bb[9]; }
  const float sc = __uint_as_float((unsigned)(me >> 32));
  p.scores[o] = sc;
  p.labels[o] = idx - qi * p.C;
  bool ok = cx >= p.range_lo[0] && cy >= p.range_lo[1] && cz >= p.range_lo[2] &&
            cx <= p.range_hi[0] && cy <= p.range_hi[1] && cz <= p.range_hi[2];
  if (p.score_thr >= 0.f) ok = ok && sc > p.score_thr;
  p.keep[o] = ok ? 1 : 0;
}

// Head box epilogue (dense_heads/detr3d_head_pe.py:571-600), one thread per query.  Un-fused fp32 arithmetic in the
// reference's order (the TU is built with -ffp-contract=off).
__global__ __launch_bounds__(256) void box_head_kernel(const float* __restrict__ tmp, const float* __restrict__ ref,
                                                       float* __restrict__ out, int M, int code, float sx, float sy,
                                                       float sz, float lx, float ly, float lz, float scale,
                                                       int scaled) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float* t = tmp + (size_t)i * code;
  const float* r = ref + (size_t)i * 3;
  float* o = out + (size_t)i * code;
  float x = 1.0f / (1.0f + expf(-(t[0] + inv_sigmoid(r[0]))));
  float y = 1.0f / (1.0f + expf(-(t[1] + inv_sigmoid(r[1]))));
  float z = 1.0f / (1.0f + expf(-(t[4] + inv_sigmoid(r[2]))));
  x = x * sx + lx; y = y * sy + ly; z = z * sz + lz;
  if (scaled) { x *= scale; y *= scale; z *= scale; }
  for (int c = 0; c < code; ++c) {
    const float v = c == 0 ? x : c == 1 ? y : c == 4 ? z : t[c];
    o[c] = v;
  }
}

}  // namespace gd4d

extern "C" int gd4d_nms_free_decode_fwd(const float* cls_scores, const float* bbox_preds,
                                        const float* post_center_range, float score_threshold, float* boxes,
                                        float* scores, int32_t* labels, uint8_t* keep, int B, int Q, int C,
                                        int code_size, int K, void* stream) {
  using namespace gd4d;
  if (!cls_scores || !bbox_preds || !post_center_range || !boxes || !scores || !labels || !keep) return GD4D_EINVAL;
  if (B <= 0 || Q <= 0 || C <= 0 || K <= 0) return GD4D_EINVAL;
  if (K > Q * C) return GD4D_EINVAL;            // torch.topk raises in the reference
  if (K > DEC_KMAX || (code_size != 8 && code_size != 10)) return GD4D_EUNSUPPORTED;
  DecodeParams p{};
  p.cls = cls_scores; p.bbox = bbox_preds; p.boxes = boxes; p.scores = scores; p.labels = labels; p.keep = keep;
  p.Q = Q; p.C = C; p.code = code_size; p.K = K;
  for (int k = 0; k < 3; ++k) { p.range_lo[k] = post_center_range[k]; p.range_hi[k] = post_center_range[k + 3]; }
  p.score_thr = score_threshold;
  const size_t lds = (size_t)Q * C * sizeof(unsigned);
  if (lds > DEC_MAX_DYN_LDS) return GD4D_EUNSUPPORTED;
  if (!allow_dynamic_lds(reinterpret_cast<const void*>(nms_free_decode_kernel), (int)DEC_MAX_DYN_LDS))   // > 64 KB of LDS
    return GD4D_ELAUNCH;
  hipLaunchKernelGGL(nms_free_decode_kernel, dim3(B), dim3(DEC_THREADS), lds, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_box_head_fwd(const float* tmp, const float* ref, const double* pc_range, float scale, float* out,
                                 int M, int code, void* stream) {
  using namespace gd4d;
  if (!tmp || !ref || !pc_range || !out || M <= 0) return GD4D_EINVAL;
  if (code < 5) return GD4D_EUNSUPPORTED;
  const float sx = (float)(pc_range[3] - pc_range[0]), sy = (float)(pc_range[4] - pc_range[1]),
              sz = (float)(pc_range[5] - pc_range[2]);
  hipLaunchKernelGGL(box_head_kernel, dim3((M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), tmp, ref,
                     out, M, code, sx, sy, sz, (float)pc_range[0], (float)pc_range[1], (float)pc_range[2], scale,
                     scale != 1.0f ? 1 : 0);
  return check_launch();
}

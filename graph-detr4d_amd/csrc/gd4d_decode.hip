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
//   1. scores = sigmoid(logit) as IEEE bits (positive floats order like unsigned ints);
//   2. radix select of the K-th largest key: four 8-bit histogram passes in LDS;
//   3. gather the < K keys above the threshold and the lowest-index ties at the threshold into LDS (exactly K
//      candidates, or n if n < K), bitonic sort them descending (ties: ascending index, deterministic);
//   4. decode + range test for the sorted candidates.
// The boolean compaction of the kept boxes (a host-visible size) stays with the caller, as in the reference.
#include "gd4d_common.h"

namespace gd4d {

constexpr int DEC_THREADS = 1024;
constexpr int DEC_KMAX = 1024;          // max_num supported (sorted in LDS)

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

__global__ __launch_bounds__(DEC_THREADS) void nms_free_decode_kernel(const DecodeParams p) {
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_key[DEC_KMAX];
  __shared__ int s_idx[DEC_KMAX];
  __shared__ unsigned s_prefix, s_remaining, s_count_gt;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int n = p.Q * p.C;
  const float* cls = p.cls + (size_t)b * n;
  const int K = min(p.K, n);

  // ---- radix select: find the key of the K-th largest element ----
  if (tid == 0) { s_prefix = 0u; s_remaining = (unsigned)K; }
  __syncthreads();
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) s_hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned hi_mask = pass == 3 ? 0u : (0xffffffffu << (8 * (pass + 1)));
    for (int i = tid; i < n; i += DEC_THREADS) {
      const unsigned k = score_key(cls[i]);
      if ((k & hi_mask) == (prefix & hi_mask)) atomicAdd(&s_hist[(k >> (8 * pass)) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned rem = s_remaining, acc = 0u;
      int d = 255;
      for (; d > 0; --d) {                         // walk digits from the top until the K-th falls inside
        if (acc + s_hist[d] >= rem) break;
        acc += s_hist[d];
      }
      s_prefix = prefix | ((unsigned)d << (8 * pass));
      s_remaining = rem - acc;                      // how many we still need among keys with this digit
    }
    __syncthreads();
  }
  const unsigned thr = s_prefix;                    // K-th largest key
  const unsigned need_eq = s_remaining;             // ties at the threshold to take (lowest indices first)

  // ---- collect candidates: all keys > thr, then `need_eq` of the keys == thr in index order ----
  if (tid == 0) s_count_gt = 0u;
  for (int i = tid; i < DEC_KMAX; i += DEC_THREADS) { s_key[i] = 0u; s_idx[i] = 0x7fffffff; }
  __syncthreads();
  for (int i = tid; i < n; i += DEC_THREADS) {
    const unsigned k = score_key(cls[i]);
    if (k > thr) {
      const unsigned slot = atomicAdd(&s_count_gt, 1u);
      s_key[slot] = k; s_idx[slot] = i;
    }
  }
  __syncthreads();
  const unsigned ngt = s_count_gt;                  // == K - need_eq
  // ties: deterministic lowest-index-first needs an ordered scan; ties are rare, a single thread walks them
  if (tid == 0 && need_eq > 0u) {
    unsigned taken = 0u;
    for (int i = 0; i < n && taken < need_eq; ++i)
      if (score_key(cls[i]) == thr) { s_key[ngt + taken] = thr; s_idx[ngt + taken] = i; ++taken; }
  }
  __syncthreads();

  // ---- bitonic sort of DEC_KMAX (key desc, index asc); padding (key 0, idx INT_MAX) sinks to the end ----
  for (int size = 2; size <= DEC_KMAX; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int i = tid;                            // DEC_THREADS == DEC_KMAX: one element per thread
      const int j = i ^ stride;
      if (j > i) {
        const unsigned ki = s_key[i], kj = s_key[j];
        const int ii = s_idx[i], ij = s_idx[j];
        const bool i_before_j = ki > kj || (ki == kj && ii < ij);     // desired order: i first
        const bool descending_block = (i & size) == 0;
        if (descending_block ? !i_before_j : i_before_j) {
          s_key[i] = kj; s_key[j] = ki; s_idx[i] = ij; s_idx[j] = ii;
        }
      }
      __syncthreads();
    }
  }

  // ---- decode ----
  const int nbox = p.code > 8 ? 9 : 7;
  for (int r = tid; r < p.K; r += DEC_THREADS) {
    float* ob = p.boxes + ((size_t)b * p.K + r) * nbox;
    const size_t o = (size_t)b * p.K + r;
    if (r >= K) {                                   // fewer than K scores exist: pad
      for (int c = 0; c < nbox; ++c) ob[c] = 0.f;
      p.scores[o] = 0.f; p.labels[o] = 0; p.keep[o] = 0;
      continue;
    }
    const int idx = s_idx[r];
    const int qi = idx / p.C;
    const float* bb = p.bbox + ((size_t)b * p.Q + qi) * p.code;
    const float cx = bb[0], cy = bb[1], cz = bb[4];
    ob[0] = cx; ob[1] = cy; ob[2] = cz;
    ob[3] = expf(bb[2]); ob[4] = expf(bb[3]); ob[5] = expf(bb[5]);
    ob[6] = atan2f(bb[6], bb[7]);
    if (nbox == 9) { ob[7] = bb[8]; ob[8] = bb[9]; }
    const float sc = __uint_as_float(s_key[r]);
    p.scores[o] = sc;
    p.labels[o] = idx - qi * p.C;
    bool ok = cx >= p.range_lo[0] && cy >= p.range_lo[1] && cz >= p.range_lo[2] &&
              cx <= p.range_hi[0] && cy <= p.range_hi[1] && cz <= p.range_hi[2];
    if (p.score_thr >= 0.f) ok = ok && sc > p.score_thr;
    p.keep[o] = ok ? 1 : 0;
  }
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
  if (K > DEC_KMAX || (code_size != 8 && code_size != 10)) return GD4D_EUNSUPPORTED;
  DecodeParams p{};
  p.cls = cls_scores; p.bbox = bbox_preds; p.boxes = boxes; p.scores = scores; p.labels = labels; p.keep = keep;
  p.Q = Q; p.C = C; p.code = code_size; p.K = K;
  for (int k = 0; k < 3; ++k) { p.range_lo[k] = post_center_range[k]; p.range_hi[k] = post_center_range[k + 3]; }
  p.score_thr = score_threshold;
  hipLaunchKernelGGL(nms_free_decode_kernel, dim3(B), dim3(DEC_THREADS), 0, static_cast<hipStream_t>(stream), p);
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

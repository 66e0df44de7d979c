// The optimizer step of the reference's training recipe over ONE flat parameter / gradient buffer: gradient-norm clipping
// (mmcv's GradientCumulativeOptimizerHook -> torch.nn.utils.clip_grad_norm_, max_norm 35, L2) followed by AdamW (lr 2e-4,
// weight decay 0.01) - projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:205-213.
//
// Two launches whatever the number of parameters (torch's clip + foreach AdamW: ~12 multi-tensor launches over 230 tensors):
//   gd4d_adamw_flat  (1) per-block sums of squares of the gradients, in a fixed order; one thread advances the step counter
//                    (2) every block adds the partial sums in the same order (the same norm everywhere, run-to-run identical),
//                        forms the clip coefficient min(1, max_norm / (norm + 1e-6)) and updates its slice:
//                            g = coef * grad;  p *= 1 - lr * wd;  m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g
//                            p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step counter lives on the device (a replayed hipGraph cannot change a kernel argument).
#include <algorithm>

#include "gd4d_common.h"

namespace gd4d {

constexpr int OPT_THREADS = 256;
constexpr int OPT_BLOCKS = 512;          // partial sums (<= 1024)

__global__ __launch_bounds__(OPT_THREADS) void adamw_sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial,
                                                                  float* __restrict__ state) {
  __shared__ float red[OPT_THREADS / 64];
  const long long per = ((n + 3) / 4 + gridDim.x - 1) / gridDim.x * 4;        // a block's contiguous range (multiple of 4)
  const long long lo = (long long)blockIdx.x * per, hi = min(n, lo + per);
  float s = 0.f;
  for (long long i = lo + 4ll * threadIdx.x; i < hi; i += 4ll * OPT_THREADS) {
    if (i + 3 < hi) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long long j = i; j < hi; ++j) s += g[j] * g[j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < OPT_THREADS / 64; ++w) t += red[w];
    partial[blockIdx.x] = t;
    if (blockIdx.x == 0) state[0] += 1.0f;                                    // the step this update is
  }
}

__global__ __launch_bounds__(OPT_THREADS) void adamw_apply_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                                  float* __restrict__ v, long long n, const float* __restrict__ partial,
                                                                  int nparts, float* __restrict__ state, float lr, float b1, float b2,
                                                                  float eps, float wd, float max_norm) {
  __shared__ float s_coef, s_c1, s_c2;
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < nparts; ++i) t += (double)partial[i];                 // the same order in every block
    const float norm = (float)sqrt(t);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(1.f, max_norm / (norm + 1e-6f));         // clip_grad_norm_'s clamp
    const float step = state[0];
    s_coef = coef;
    s_c1 = (float)((double)lr / (1.0 - pow((double)b1, (double)step)));     // (bias corrections in double, as torch's host code)
    s_c2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)step)));
    if (blockIdx.x == 0) state[1] = norm;                                     // (for the host: the norm before clipping)
  }
  __syncthreads();
  const float coef = s_coef, step_size = s_c1, inv_bc2 = s_c2, decay = 1.f - lr * wd;
  for (long long i = 4ll * ((long long)blockIdx.x * OPT_THREADS + threadIdx.x); i < n; i += 4ll * OPT_THREADS * gridDim.x) {
    const int cnt = (int)min(4ll, n - i);
    float pv[4], gv[4], mv[4], vv[4];
    if (cnt == 4) {
      *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i);
      *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + i);
      *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m + i);
      *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i);
    } else {
      for (int j = 0; j < cnt; ++j) { pv[j] = p[i + j]; gv[j] = g[i + j]; mv[j] = m[i + j]; vv[j] = v[i + j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= cnt) break;
      const float gj = gv[j] * coef;
      const float pd = pv[j] * decay;
      const float mj = mv[j] + (1.f - b1) * (gj - mv[j]);
      const float vj = b2 * vv[j] + (1.f - b2) * gj * gj;
      pv[j] = pd - step_size * (mj / (sqrtf(vj) * inv_bc2 + eps));
      mv[j] = mj; vv[j] = vj;
    }
    if (cnt == 4) {
      *reinterpret_cast<float4*>(p + i) = *reinterpret_cast<const float4*>(pv);
      *reinterpret_cast<float4*>(m + i) = *reinterpret_cast<const float4*>(mv);
      *reinterpret_cast<float4*>(v + i) = *reinterpret_cast<const float4*>(vv);
    } else {
      for (int j = 0; j < cnt; ++j) { p[i + j] = pv[j]; m[i + j] = mv[j]; v[i + j] = vv[j]; }
    }
  }
}

}  // namespace gd4d

extern "C" size_t gd4d_adamw_flat_workspace_bytes(void) { return (size_t)gd4d::OPT_BLOCKS * sizeof(float); }

extern "C" int gd4d_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* state, void* workspace,
                               size_t workspace_bytes, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float max_norm, void* stream) {
  using namespace gd4d;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !state || !workspace || n <= 0) return GD4D_EINVAL;
  if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps > 0.f) || !(weight_decay >= 0.f)) return GD4D_EINVAL;
  if (!aligned16(params) || !aligned16(grads) || !aligned16(exp_avg) || !aligned16(exp_avg_sq)) return GD4D_EALIGN;
  if (workspace_bytes < gd4d_adamw_flat_workspace_bytes()) return GD4D_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(OPT_BLOCKS), dim3(OPT_THREADS), 0, s, grads, (long long)n, partial, state);
  if (int rc = check_launch()) return rc;
  const long long quads = (n + 3) / 4;
  const int blocks = (int)std::min<long long>((quads + OPT_THREADS - 1) / OPT_THREADS, 2048);
  hipLaunchKernelGGL(adamw_apply_kernel, dim3(blocks), dim3(OPT_THREADS), 0, s, params, grads, exp_avg, exp_avg_sq, (long long)n, partial,
                     OPT_BLOCKS, state, lr, beta1, beta2, eps, weight_decay, max_norm);
  return check_launch();
}

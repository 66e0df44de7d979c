// DGCNNAttn (projects/mmdet3d_plugin/models/utils/dgcnn_attn.py:10-96): kNN-graph EdgeConv
// self-attention.  Two kernels; the 1x1 convolutions become two (N, C) x (C, C) Linears on gd4d_linear_group_fwd
// because conv(cat(x_j, x_i)) = W[:, :C] x_j + W[:, C:] x_i.
//
//   gd4d_knn_farthest_fwd   :84-86  cdist + topk of the LARGEST distances (the reference's choice) - one wave per
//                           query: squared distances straight from the rows (no N x N matrix in memory), then K rounds
//                           of wave arg-max.  The reference materialises (B, N, N) distances.
//   gd4d_edge_conv_max_fwd  :72-74  gather the K neighbour rows of W_a x, add W_b x_i, BatchNorm (eval: per-channel
//                           scale / shift), ReLU, max over K.  The reference materialises (B, 2C, N, K) edge features
//                           (59 MB at N = 900, K = 16) and the conv output of the same size.
#include "gd4d_common.h"

namespace gd4d {

constexpr int KNN_MAX_PER_LANE = 32;      // N <= 2048

__global__ __launch_bounds__(256) void knn_farthest_kernel(const float* __restrict__ x, int32_t* __restrict__ idx,
                                                           int B, int N, int C, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // (b, i)
  if (row >= B * N) return;
  const int b = row / N;
  const float* xb = x + (size_t)b * N * C;
  const float* xi = x + (size_t)row * C;
  float d[KNN_MAX_PER_LANE];
#pragma unroll
  for (int t = 0; t < KNN_MAX_PER_LANE; ++t) {
    const int j = lane + 64 * t;
    d[t] = -1.f;                                                // below any squared distance
    if (j < N) {
      const float4* pj = reinterpret_cast<const float4*>(xb + (size_t)j * C);
      const float4* pi = reinterpret_cast<const float4*>(xi);
      float s = 0.f;
      for (int c4 = 0; c4 < C / 4; ++c4) {
        const float4 a = pi[c4], q = pj[c4];
        const float e0 = a.x - q.x, e1 = a.y - q.y, e2 = a.z - q.z, e3 = a.w - q.w;
        s = fmaf(e0, e0, s); s = fmaf(e1, e1, s); s = fmaf(e2, e2, s); s = fmaf(e3, e3, s);
      }
      d[t] = s;
    }
  }
  int32_t* out = idx + (size_t)row * K;
  for (int k = 0; k < K; ++k) {
    float best = -2.f;                                          // my largest remaining value and its column
    int bj = 0x7fffffff, bt = 0;
#pragma unroll
    for (int t = 0; t < KNN_MAX_PER_LANE; ++t)
      if (d[t] > best) { best = d[t]; bj = lane + 64 * t; bt = t; }
    float wb = best;
    int wj = bj;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                          // wave arg-max, ties to the lower column
      const float ob = __shfl_xor(wb, o);
      const int oj = __shfl_xor(wj, o);
      if (ob > wb || (ob == wb && oj < wj)) { wb = ob; wj = oj; }
    }
    if (wj == bj) {                                             // the winner retires its element
#pragma unroll
      for (int t = 0; t < KNN_MAX_PER_LANE; ++t)
        if (t == bt) d[t] = -1.f;
    }
    if (lane == 0) out[k] = wj;
  }
}

__global__ __launch_bounds__(256) void edge_conv_max_kernel(const float* __restrict__ a, const float* __restrict__ bs,
                                                            const int32_t* __restrict__ idx,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            int B, int N, int C, int K, int lda) {
  // one wave per (b, n); lane owns channels 4*lane + 256*t
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * N) return;
  const int b = row / N;
  const int32_t* nb = idx + (size_t)row * K;
  for (int c = 4 * lane; c < C; c += 256) {
    const float4 self = *reinterpret_cast<const float4*>(bs + (size_t)row * lda + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);                 // max of ReLU outputs: >= 0
    for (int k = 0; k < K; ++k) {
      const int j = nb[k];
      const float4 v = *reinterpret_cast<const float4*>(a + ((size_t)b * N + j) * lda + c);
      m.x = fmaxf(m.x, (v.x + self.x) * sc.x + sh.x);
      m.y = fmaxf(m.y, (v.y + self.y) * sc.y + sh.y);
      m.z = fmaxf(m.z, (v.z + self.z) * sc.z + sh.z);
      m.w = fmaxf(m.w, (v.w + self.w) * sc.w + sh.w);
    }
    *reinterpret_cast<float4*>(out + (size_t)row * C + c) = m;
  }
}

}  // namespace gd4d

extern "C" int gd4d_knn_farthest_fwd(const float* x, int32_t* idx, int B, int N, int C, int K, void* stream) {
  using namespace gd4d;
  if (!x || !idx || B <= 0 || N <= 0 || C <= 0 || K <= 0) return GD4D_EINVAL;
  if (K > N) return GD4D_EINVAL;                                 // torch.topk raises in the reference
  if (N > 64 * KNN_MAX_PER_LANE || C % 4 != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(x)) return GD4D_EALIGN;
  hipLaunchKernelGGL(knn_farthest_kernel, dim3((B * N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, idx,
                     B, N, C, K);
  return check_launch();
}

extern "C" int gd4d_edge_conv_max_fwd(const float* a, const float* b_self, const int32_t* idx, const float* scale,
                                      const float* shift, float* out, int B, int N, int C, int K, int lda,
                                      void* stream) {
  using namespace gd4d;
  if (!a || !b_self || !idx || !scale || !shift || !out || B <= 0 || N <= 0 || C <= 0 || K <= 0 || lda < C)
    return GD4D_EINVAL;
  if (C % 4 != 0 || lda % 4 != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(a) || !aligned16(b_self) || !aligned16(scale) || !aligned16(shift) || !aligned16(out)) return GD4D_EALIGN;
  hipLaunchKernelGGL(edge_conv_max_kernel, dim3((B * N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     b_self, idx, scale, shift, out, B, N, C, K, lda);
  return check_launch();
}

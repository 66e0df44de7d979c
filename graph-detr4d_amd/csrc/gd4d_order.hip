// gd4d_query_order_fwd / gd4d_refine_reference_order_fwd: a locality order of the queries for the fused
// sample-aggregate kernel (gd4d_cross_attn.hip maps workgroup i to XCD i % 8 and hands every XCD a contiguous range of
// this order, so queries that read the same camera region share one L2).
//
// Key of a query = (sample, azimuth of its de-normalised reference point about the lidar origin): the cameras of a rig
// sit near that origin, so queries of similar azimuth project into the same image columns of the same cameras in
// every frame.  (A key built from "first camera that sees the point" was measured worse: the ~15 % of points no camera
// sees directly pile up in one XCD.)  Counting sort in a single workgroup: histogram, scan, scatter - the order inside
// a bin is arbitrary, the gather's result does not depend on it.
//
// The fused variant also performs the reference-point refinement between decoder layers
// (Detr3DTransformerDecoder.forward, detr3d_transformer.py:201-214), so the per-layer order costs no extra launch.
#include "gd4d_common.h"

namespace gd4d {

constexpr int ORD_THREADS = 1024;
constexpr int ORD_MAX_BINS = 4096;

struct OrderParams {
  const float* ref;      // (B*Q, 3) in [0,1]: the points to order (for the fused kernel: written by it first)
  const float* tmp;      // fused refinement only: (B*Q, ldt) regression deltas
  const float* ref_in;   // fused refinement only: (B*Q, 3) current reference points
  float* ref_out;        // fused refinement only: (B*Q, 3)
  int32_t* order;        // (B*Q)
  int B, Q, ldt, abins;
  float rng_scale[2], rng_lo[2];
};

template <bool REFINE>
__global__ __launch_bounds__(ORD_THREADS) void query_order_kernel(const OrderParams p) {
  __shared__ unsigned s_count[ORD_MAX_BINS];
  __shared__ unsigned s_wave[ORD_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = p.B * p.Q;
  for (int i = tid; i < ORD_MAX_BINS; i += ORD_THREADS) s_count[i] = 0u;
  __syncthreads();
  auto bin_of = [&](int bq, float rx, float ry) {
    const float X = rx * p.rng_scale[0] + p.rng_lo[0];
    const float Y = ry * p.rng_scale[1] + p.rng_lo[1];
    const float turn = atan2f(Y, X) * 0.15915494309189535f + 0.5f;             // [0, 1]
    const int a = min(p.abins - 1, max(0, (int)(turn * (float)p.abins)));      // NaN -> 0
    return (bq / p.Q) * p.abins + a;
  };
  constexpr int MAXPER = 4;                          // bins kept in registers for the first 4096 queries
  int mybin[MAXPER];
#pragma unroll
  for (int k = 0; k < MAXPER; ++k) {
    const int bq = tid + k * ORD_THREADS;
    mybin[k] = -1;
    if (bq < total) {
      float rx, ry;
      if (REFINE) {
        const float* t = p.tmp + (size_t)bq * p.ldt;
        const float* r = p.ref_in + (size_t)bq * 3;
        rx = 1.0f / (1.0f + expf(-(t[0] + inv_sigmoid(r[0]))));
        ry = 1.0f / (1.0f + expf(-(t[1] + inv_sigmoid(r[1]))));
        const float rz = 1.0f / (1.0f + expf(-(t[4] + inv_sigmoid(r[2]))));
        float* o = p.ref_out + (size_t)bq * 3;
        o[0] = rx; o[1] = ry; o[2] = rz;
      } else {
        rx = p.ref[(size_t)bq * 3]; ry = p.ref[(size_t)bq * 3 + 1];
      }
      mybin[k] = bin_of(bq, rx, ry);
      atomicAdd(&s_count[mybin[k]], 1u);
    }
  }
  // exclusive scan of the bin counts: 4 consecutive bins per thread, wave scan, wave totals
  __syncthreads();
  unsigned c[4], sum = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) { c[k] = s_count[4 * tid + k]; sum += c[k]; }
  unsigned incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned y = __shfl_up(incl, off);
    if (lane >= off) incl += y;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned base = incl - sum;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_count[4 * tid + k] = base; base += c[k]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < MAXPER; ++k)
    if (mybin[k] >= 0) p.order[atomicAdd(&s_count[mybin[k]], 1u)] = tid + k * ORD_THREADS;
}

static int fill(OrderParams& p, const double* pc_range, int B, int Q) {
  if (B <= 0 || Q <= 0) return GD4D_EINVAL;
  if (B > ORD_MAX_BINS / 8 || (long long)B * Q > 4LL * ORD_THREADS) return GD4D_EUNSUPPORTED;
  p.B = B; p.Q = Q; p.abins = ORD_MAX_BINS / B;
  for (int k = 0; k < 2; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  return GD4D_OK;
}

}  // namespace gd4d

extern "C" int gd4d_query_order_fwd(const float* ref, const double* pc_range, int32_t* order, int B, int Q,
                                    void* stream) {
  using namespace gd4d;
  if (!ref || !pc_range || !order) return GD4D_EINVAL;
  OrderParams p{};
  if (int rc = fill(p, pc_range, B, Q)) return rc;
  p.ref = ref; p.order = order;
  hipLaunchKernelGGL(query_order_kernel<false>, dim3(1), dim3(ORD_THREADS), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_refine_reference_order_fwd(const float* tmp, const float* ref, float* out, const double* pc_range,
                                               int32_t* order, int B, int Q, int ldt, void* stream) {
  using namespace gd4d;
  if (!tmp || !ref || !out || !pc_range || !order) return GD4D_EINVAL;
  if (ldt < 5) return GD4D_EUNSUPPORTED;
  OrderParams p{};
  if (int rc = fill(p, pc_range, B, Q)) return rc;
  p.tmp = tmp; p.ref_in = ref; p.ref_out = out; p.order = order; p.ldt = ldt;
  hipLaunchKernelGGL(query_order_kernel<true>, dim3(1), dim3(ORD_THREADS), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

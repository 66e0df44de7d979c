// gd4d_linear_bwd_weight: weight and bias gradient of the decoder's small dense layers (training),
//   dw[n][k] = sum_m dy[m][n] * x[m][k],   db[n] = sum_m dy[m][n]            (m: the ~900 query rows)
// A library fp32 GEMM picks one 256 x 256 macro tile for these shapes (N, K <= 512, M ~ 900) and runs the whole
// contraction on ONE compute unit: ~205 us per layer where the arithmetic is 0.1 GFLOP (measured with rocprofv3 on the
// training step, 30 such GEMMs per step).  Here every 16 (n) x 32 (k) block of dw is one workgroup; its sixteen waves
// split the rows, run v_mfma_f32_16x16x4_f32 on operands loaded straight from global memory (the reduction index m
// is the ROW of both operands, which is exactly how the fp32 MFMA wants A^T and B: one float per lane, 16 consecutive
// columns per row = one 64 B segment) and are summed through LDS in a fixed order: deterministic, no atomics.
#include "gd4d_common.h"
#include "gd4d_linear_bwd_body.h"

namespace gd4d {

// (the tile, the group descriptor and its host-side filler: gd4d_linear_bwd_body.h)

__global__ __launch_bounds__(64 * LB_WAVES) void linear_bwd_weight_kernel(const LinBwdParams p) {
  __shared__ LinBwdShared<LB_WAVES> sh;
  linear_bwd_weight_tile<LB_WAVES>(p, blockIdx.x * LB_TK, blockIdx.y * LB_TN, blockIdx.x == 0, sh);
}

__global__ __launch_bounds__(64 * LB_WAVES) void linear_bwd_weight_group_kernel(const LinBwdGroup g) {
  __shared__ LinBwdShared<LB_WAVES> sh;
#if defined(__HIP_DEVICE_COMPILE__)
  const lin_group_ptr_t gp = (lin_group_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // explicit arguments start at 0
#else
  const lin_group_ptr_t gp = &g;
#endif
  (void)g;
  linear_bwd_weight_group_tile<LB_WAVES>(gp, (int)blockIdx.x, sh);
}

}  // namespace gd4d

extern "C" int gd4d_linear_bwd_weight_group(const void* const* x, const void* const* grad_y, void* const* grad_w, void* const* grad_b,
                                            const int32_t* dims, int count, int accumulate, void* stream) {
  using namespace gd4d;
  LinBwdGroup g{};
  int tiles = 0;
  if (int rc = fill_lin_bwd_group(g, tiles, x, grad_y, grad_w, grad_b, dims, count, accumulate)) return rc;
  hipLaunchKernelGGL(linear_bwd_weight_group_kernel, dim3(tiles), dim3(64 * LB_WAVES), 0, static_cast<hipStream_t>(stream), g);
  return check_launch();
}

extern "C" int gd4d_linear_bwd_weight(const float* x, const float* grad_y, float* grad_w, float* grad_b, int M, int K,
                                      int N, int ldx, int ldy, int accumulate, void* stream) {
  using namespace gd4d;
  if (!x || !grad_y || !grad_w || M <= 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) return GD4D_EINVAL;
  LinBwdParams p{x, grad_y, grad_w, grad_b, M, N, K, ldx, ldy, accumulate ? 1 : 0};
  const dim3 grid((K + LB_TK - 1) / LB_TK, (N + LB_TN - 1) / LB_TN);
  hipLaunchKernelGGL(linear_bwd_weight_kernel, grid, dim3(64 * LB_WAVES), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

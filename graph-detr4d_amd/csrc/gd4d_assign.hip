// gd4d_linear_sum_assignment_batch: the host side of the Hungarian assignment.
//
// HungarianAssigner3D hands its cost matrix to scipy.optimize.linear_sum_assignment on the CPU
// (core/bbox/assigners/hungarian_assigner_3d.py:125-131), once per decoder layer and sample.  This is the same
// algorithm - the shortest-augmenting-path method for the rectangular problem with dual variables (Crouse 2016, "On
// implementing 2D rectangular assignment algorithms"; what scipy ships) - restated from the published description, for
// a whole batch of independent problems spread over host threads: the six decoder layers of a step are solved
// concurrently on the cost matrices gd4d_match_cost_fwd produced in one launch.  Pure host code (no GPU work).
#include <cmath>
#include <limits>
#include <thread>
#include <vector>

#include "gd4d_common.h"

namespace gd4d {

// nr <= nc, c row-major (nr, nc) in double.  col4row[i] = column assigned to row i.
static bool lsa_solve(int nr, int nc, const double* c, std::vector<int>& col4row) {
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> u(nr, 0.0), v(nc, 0.0), shortest(nc);
  std::vector<int> path(nc, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  col4row.assign(nr, -1);
  for (int cur = 0; cur < nr; ++cur) {
    // shortest augmenting path from row `cur` to any unassigned column
    double min_val = 0.0;
    int i = cur, sink = -1, num_remaining = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(shortest.begin(), shortest.end(), inf);
    while (sink == -1) {
      int index = -1;
      double lowest = inf;
      SR[i] = 1;
      const double* ci = c + (size_t)i * nc;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = remaining[it];
        const double r = min_val + ci[j] - u[i] - v[j];
        if (r < shortest[j]) {
          path[j] = i;
          shortest[j] = r;
        }
        // among equally short paths prefer one that ends in an unassigned column (it terminates the search)
        if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) {
          lowest = shortest[j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == inf) return false;             // infeasible (cannot happen after nan_to_num)
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    // dual update
    u[cur] += min_val;
    for (int r = 0; r < nr; ++r)
      if (SR[r] && r != cur) u[r] += min_val - shortest[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - shortest[j];
    // augment along the path
    int j = sink;
    while (true) {
      const int r = path[j];
      row4col[j] = r;
      std::swap(col4row[r], j);
      if (r == cur) break;
    }
  }
  return true;
}

// one problem: rows x cols row-major; out[r] = assigned column or -1
static bool lsa_problem(const float* c, int rows, int cols, int32_t* out) {
  std::vector<int> a;
  for (int r = 0; r < rows; ++r) out[r] = -1;
  if (rows == 0 || cols == 0) return true;
  std::vector<double> m((size_t)rows * cols);
  if (cols < rows) {                                // more predictions than boxes: solve the transposed problem
    for (int r = 0; r < rows; ++r)
      for (int g = 0; g < cols; ++g) m[(size_t)g * rows + r] = c[(size_t)r * cols + g];
    if (!lsa_solve(cols, rows, m.data(), a)) return false;
    for (int g = 0; g < cols; ++g) out[a[g]] = g;
  } else {
    for (size_t k = 0; k < m.size(); ++k) m[k] = c[k];
    if (!lsa_solve(rows, cols, m.data(), a)) return false;
    for (int r = 0; r < rows; ++r) out[r] = a[r];
  }
  return true;
}

}  // namespace gd4d

extern "C" int gd4d_linear_sum_assignment_batch(const float* cost, const int64_t* cost_offset, const int32_t* rows,
                                                const int32_t* cols, int num_problems, int32_t* col_of_row,
                                                const int64_t* out_offset, int num_threads) {
  using namespace gd4d;
  if (!cost || !cost_offset || !rows || !cols || !col_of_row || !out_offset || num_problems < 0) return GD4D_EINVAL;
  for (int p = 0; p < num_problems; ++p)
    if (rows[p] < 0 || cols[p] < 0) return GD4D_EINVAL;
  if (num_threads < 1) num_threads = 1;
  if (num_threads > num_problems) num_threads = num_problems > 0 ? num_problems : 1;
  std::vector<int> ok(num_problems > 0 ? num_problems : 1, 1);
  auto work = [&](int t) {
    for (int p = t; p < num_problems; p += num_threads)
      ok[p] = lsa_problem(cost + cost_offset[p], rows[p], cols[p], col_of_row + out_offset[p]) ? 1 : 0;
  };
  if (num_threads == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < num_threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  for (int p = 0; p < num_problems; ++p)
    if (!ok[p]) return GD4D_EUNSUPPORTED;
  return GD4D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// gd4d_hungarian_assign_fwd: the same solver ON THE DEVICE (round 5) - the assignment of a training step no
// longer leaves the GPU (no device -> host copy of the cost matrix, no host solve, no copy back: the `--criterion` step is one
// graph).  One workgroup per (decoder layer, sample) problem: all threads turn the problem's (Q, G) float block of
// gd4d_match_cost_fwd into a double matrix with the shorter side as rows (what lsa_problem does), then run lsa_solve above,
// operation for operation in double - the scan over the remaining columns spread over the workgroup's 256 threads (a first version
// gave it to one wave: 266 us for six 900 x 40 problems, bound by the scan's dependent round trips), its arg-min as a
// reduction that reproduces the sequential scan's choice exactly:
//     sequential:  a candidate replaces the best so far if it is shorter, or equally short and its column is unassigned
//     => the winner is, among the positions of minimal length, the LAST one whose column is unassigned if there is one,
//        else the FIRST one                                                  (positions = indices into `remaining`)
// so ties break as scipy's / the host solver's do and the matching is bit-identical to theirs.  Per-column state (v, shortest,
// path, row4col, remaining, SC) lives in LDS (32 B per column), the matrix in global memory (L2-resident).
namespace gd4d {

constexpr int HA_THREADS = 256;

struct HaParams {
  const float* cost;
  const int32_t* gt_start;      // (B + 1) on the device
  int32_t* assigned;            // (NL, B, Q)
  int32_t* status;              // (NL * B): 0 = solved, 1 = a NaN cost (label outside [0, classes): gd4d_match_cost_fwd's marker), 2 = infeasible
  double* work;                 // per problem Q * max_gt doubles
  int NL, B, Q, sum_gt, max_gt;
};

__device__ __forceinline__ double ha_shfl_xor(double x, int o) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __shfl_xor(lo, o); hi = __shfl_xor(hi, o);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void ha_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(HA_THREADS) void hungarian_assign_kernel(const HaParams p) {
  extern __shared__ __attribute__((aligned(16))) char ha_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int prob = blockIdx.x, l = prob / p.B, b = prob - l * p.B;
  const int g0 = p.gt_start[b], G = p.gt_start[b + 1] - g0, Q = p.Q;
  int32_t* out = p.assigned + (size_t)prob * Q;
  for (int q = tid; q < Q; q += HA_THREADS) out[q] = -1;
  if (G <= 0 || Q <= 0) { if (tid == 0) p.status[prob] = 0; return; }
  if (G > p.max_gt) { if (tid == 0) p.status[prob] = 2; return; }
  const float* c = p.cost + (size_t)Q * ((size_t)l * p.sum_gt + g0);        // (Q, G) row-major
  const bool transposed = G < Q;                                             // rows = the shorter side
  const int nr = transposed ? G : Q, nc = transposed ? Q : G;
  double* m = p.work + (size_t)prob * Q * p.max_gt;                          // (nr, nc)
  __shared__ int s_nan;
  if (tid == 0) s_nan = 0;
  __syncthreads();
  bool nan = false;
  for (int idx = tid; idx < nr * nc; idx += HA_THREADS) {
    const int r = idx / nc, j = idx - r * nc;
    const float x = transposed ? c[(size_t)j * G + r] : c[idx];
    nan = nan || (x != x);
    m[idx] = (double)x;
  }
  if (nan) s_nan = 1;
  // LDS: per column v, shortest (double), path, row4col, remaining, SC (int); per row u (double), col4row, SR (int)
  double* v = reinterpret_cast<double*>(ha_smem);
  double* shortest = v + nc;
  double* u = shortest + nc;
  int* path = reinterpret_cast<int*>(u + nr);
  int* row4col = path + nc;
  int* remaining = row4col + nc;
  int* SC = remaining + nc;
  int* col4row = SC + nc;
  int* SR = col4row + nr;
  for (int j = tid; j < nc; j += HA_THREADS) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
  for (int r = tid; r < nr; r += HA_THREADS) { u[r] = 0.0; col4row[r] = -1; }
  __threadfence_block();
  __syncthreads();
  if (s_nan) { if (tid == 0) p.status[prob] = 1; return; }
  // ---- the solver: every thread of the workgroup walks the same control flow; the scan over the remaining columns is spread over
  // all of them (at 900 columns: <= 4 per thread, their cost loads in flight together), its arg-min meets through LDS ----
  __shared__ double s_low[HA_THREADS / 64];
  __shared__ int s_pun[HA_THREADS / 64], s_pfirst[HA_THREADS / 64];
  const int wave = tid >> 6;
  const double inf = __builtin_inf();
  bool ok = true;
  for (int cur = 0; cur < nr && ok; ++cur) {
    for (int j = tid; j < nc; j += HA_THREADS) { remaining[j] = nc - j - 1; SC[j] = 0; shortest[j] = inf; }
    for (int r = tid; r < nr; r += HA_THREADS) SR[r] = 0;
    __syncthreads();
    double min_val = 0.0;
    int i = cur, sink = -1, num_remaining = nc;
    while (sink == -1) {
      if (tid == 0) SR[i] = 1;
      const double ui = u[i];
      const double* ci = m + (size_t)i * nc;
      double lowest = inf;
      int pos_un = -1, pos_first = 0x7fffffff;
      // (the row's costs come from global memory - L2 hits of ~1 us each if taken one by one: four positions' loads go out
      //  together, then the four are worked through in position order)
      for (int it0 = tid; it0 < num_remaining; it0 += HA_THREADS * 4) {
        int js[4];
        double cs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) js[k] = remaining[min(it0 + HA_THREADS * k, num_remaining - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) cs[k] = ci[js[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int it = it0 + HA_THREADS * k;
          if (it >= num_remaining) break;
          const int j = js[k];
          const double r = ((min_val + cs[k]) - ui) - v[j];
          double sj = shortest[j];
          if (r < sj) { path[j] = i; shortest[j] = r; sj = r; }
          const bool un = row4col[j] == -1;
          if (sj < lowest) { lowest = sj; pos_first = it; pos_un = un ? it : -1; }
          else if (sj == lowest) { if (un) pos_un = it; if (it < pos_first) pos_first = it; }
        }
      }
      double wmin = lowest;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const double t = ha_shfl_xor(wmin, o); wmin = t < wmin ? t : wmin; }
      if (lane == 0) s_low[wave] = wmin;
      __syncthreads();
      double gmin = s_low[0];
#pragma unroll
      for (int w = 1; w < HA_THREADS / 64; ++w) gmin = s_low[w] < gmin ? s_low[w] : gmin;
      if (!(lowest == gmin) || tid >= num_remaining) { pos_un = -1; pos_first = 0x7fffffff; }   // (threads without a position hold inf)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        pos_un = max(pos_un, __shfl_xor(pos_un, o));
        pos_first = min(pos_first, __shfl_xor(pos_first, o));
      }
      if (lane == 0) { s_pun[wave] = pos_un; s_pfirst[wave] = pos_first; }
      __syncthreads();
      pos_un = s_pun[0]; pos_first = s_pfirst[0];
#pragma unroll
      for (int w = 1; w < HA_THREADS / 64; ++w) { pos_un = max(pos_un, s_pun[w]); pos_first = min(pos_first, s_pfirst[w]); }
      min_val = gmin;
      if (min_val == inf) { ok = false; break; }                            // infeasible (cannot happen after nan_to_num)
      const int index = pos_un >= 0 ? pos_un : pos_first;
      const int j = remaining[index];
      const int rj = row4col[j];
      const int last = remaining[num_remaining - 1];
      __syncthreads();                                                       // every thread has read before thread 0 rewrites
      if (rj == -1) sink = j; else i = rj;
      --num_remaining;
      if (tid == 0) { SC[j] = 1; remaining[index] = last; }
      __syncthreads();
    }
    if (!ok) break;
    // dual update
    for (int r = tid; r < nr; r += HA_THREADS)
      if (SR[r] && r != cur) u[r] += min_val - shortest[col4row[r]];
    for (int j = tid; j < nc; j += HA_THREADS)
      if (SC[j]) v[j] -= min_val - shortest[j];
    __syncthreads();
    if (tid == 0) {
      u[cur] += min_val;
      int j = sink;                                                          // augment along the path
      while (true) {
        const int r = path[j];
        row4col[j] = r;
        const int t = col4row[r]; col4row[r] = j; j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  if (!ok) { if (tid == 0) p.status[prob] = 2; return; }
  for (int r = tid; r < nr; r += HA_THREADS) {
    if (transposed) out[col4row[r]] = r + g0;                                // row = box r, its column = the prediction
    else out[r] = col4row[r] + g0;
  }
  if (tid == 0) p.status[prob] = 0;
}

}  // namespace gd4d

extern "C" size_t gd4d_hungarian_assign_workspace_bytes(int NL, int B, int Q, int max_gt) {
  if (NL <= 0 || B <= 0 || Q <= 0 || max_gt <= 0) return 0;
  return (size_t)NL * B * Q * max_gt * sizeof(double);
}

extern "C" int gd4d_hungarian_assign_fwd(const float* cost, const int32_t* gt_start, int32_t* assigned, int32_t* status, void* workspace,
                                         size_t workspace_bytes, int NL, int B, int Q, int sum_gt, int max_gt, void* stream) {
  using namespace gd4d;
  if (!cost || !gt_start || !assigned || !status || NL <= 0 || B <= 0 || Q <= 0 || sum_gt < 0 || max_gt < 0) return GD4D_EINVAL;
  if (max_gt > 0 && (!workspace || workspace_bytes < gd4d_hungarian_assign_workspace_bytes(NL, B, Q, max_gt))) return GD4D_EWORKSPACE;
  const int nc = Q > max_gt ? Q : max_gt, nr = Q > max_gt ? max_gt : Q;
  const size_t lds = (size_t)nc * (8 + 8 + 4 + 4 + 4 + 4) + (size_t)nr * (8 + 4 + 4) + 64;
  if (lds > 160 * 1024) return GD4D_EUNSUPPORTED;
  if (lds > 65536 && !allow_dynamic_lds(reinterpret_cast<const void*>(hungarian_assign_kernel), (int)lds)) return GD4D_ELAUNCH;
  HaParams p{cost, gt_start, assigned, status, static_cast<double*>(workspace), NL, B, Q, sum_gt, max_gt};
  hipLaunchKernelGGL(hungarian_assign_kernel, dim3(NL * B), dim3(HA_THREADS), lds, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

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

// Training-side step right after the path: the matching cost of HungarianAssigner3D and the
// per-layer focal / L1 losses of Detr3DHeadPE.loss_single, for ALL decoder layers in one launch each.
//
// The reference runs, per decoder layer and sample: ~15 small torch kernels for the cost matrix, a device -> host copy
// (a synchronisation), scipy on the CPU, two host -> device copies, ~40 small kernels for targets and losses and two
// scalar all-reduces each followed by .item() (two more synchronisations): 6 layers -> 18+ syncs per step
// (hungarian_assigner_3d.py:117-144, detr3d_head_pe.py:782-845).  With these two kernels a step has ONE device -> host
// copy (all layers' cost matrices), the assignment on the host, ONE host -> device copy, one loss launch that also
// produces the gradients, and no .item(): the normalisers are read from device memory.
#include "gd4d_common.h"

namespace gd4d {

constexpr int MC_MAX_GT = 1024;      // ground-truth boxes per sample held in LDS
constexpr int MC_QB = 64;            // queries per workgroup

// core/bbox/util.py:38-58: (cx, cy, cz, w, l, h, rot[, vx, vy]) -> (cx, cy, log w, log l, cz, log h, sin, cos[, vx, vy])
__device__ __forceinline__ void normalize_box(const float* b, int gt_dim, float* o) {
  o[0] = b[0]; o[1] = b[1]; o[2] = logf(b[3]); o[3] = logf(b[4]); o[4] = b[2]; o[5] = logf(b[5]);
  o[6] = sinf(b[6]); o[7] = cosf(b[6]);
  o[8] = gt_dim > 7 ? b[7] : 0.f;
  o[9] = gt_dim > 8 ? b[8] : 0.f;
}

struct MatchCostParams {
  const float* cls;          // (NL, B, Q, C) logits
  const float* box;          // (NL, B, Q, code) head outputs (normalised box code)
  const float* gt_boxes;     // (sumG, gt_dim)
  const int32_t* gt_labels;  // (sumG)
  const int32_t* gt_start;   // device, (B + 1) prefix offsets into the ground-truth arrays
  float* cost;               // block (l, b) at Q * (l * sumG + start_b), shape (Q, G_b) row-major
  int NL, B, Q, C, code, gt_dim, sumG;
  float cls_weight, reg_weight, alpha;
};

// mmdet FocalLossCost (alpha, gamma = 2, eps = 1e-12) + BBox3DL1Cost over the first 8 code entries, then nan_to_num
__global__ __launch_bounds__(256) void match_cost_kernel(const MatchCostParams p) {
  __shared__ float s_gt[MC_MAX_GT * 8];
  __shared__ int s_lab[MC_MAX_GT];
  const int b = blockIdx.y, l = blockIdx.z;
  const int g0 = p.gt_start[b], G = p.gt_start[b + 1] - g0;
  if (G <= 0) return;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float o[10];
    normalize_box(p.gt_boxes + (size_t)(g0 + g) * p.gt_dim, p.gt_dim, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) s_gt[8 * g + k] = o[k];
    s_lab[g] = p.gt_labels[g0 + g];
  }
  __syncthreads();
  const int q0 = blockIdx.x * MC_QB;
  const int nq = min(MC_QB, p.Q - q0);
  const size_t row0 = ((size_t)l * p.B + b) * p.Q;
  float* out = p.cost + (size_t)p.Q * ((size_t)l * p.sumG + g0);
  for (int e = threadIdx.x; e < nq * G; e += blockDim.x) {
    const int ql = e / G, g = e - ql * G;
    const int q = q0 + ql;
    const int lab = s_lab[g];
    if (lab < 0 || lab >= p.C) {                          // the reference's cls_pred[:, gt_labels] raises here; the host
      out[(size_t)q * G + g] = __builtin_nanf("");        // sees the NaN (valid costs are never NaN) and raises too
      continue;
    }
    const float x = p.cls[(row0 + q) * p.C + lab];
    const float pr = 1.0f / (1.0f + expf(-x));
    const float neg = -logf(1.0f - pr + 1e-12f) * (1.0f - p.alpha) * (pr * pr);
    const float pos = -logf(pr + 1e-12f) * p.alpha * ((1.0f - pr) * (1.0f - pr));
    const float* bx = p.box + (row0 + q) * p.code;
    float l1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) l1 += fabsf(bx[k] - s_gt[8 * g + k]);
    float c = (pos - neg) * p.cls_weight + l1 * p.reg_weight;
    if (c != c || c == INFINITY) c = 100.0f;              // torch.nan_to_num(nan=100, posinf=100, neginf=-100)
    else if (c == -INFINITY) c = -100.0f;
    out[(size_t)q * G + g] = c;
  }
}

struct HeadLossParams {
  const float* cls;            // (NL, B, Q, C)
  const float* box;            // (NL, B, Q, code)
  const int32_t* assigned;     // (NL, B, Q): index into the ground-truth arrays, or -1 = background
  const float* gt_boxes;
  const int32_t* gt_labels;
  const float* code_weights;   // (10)
  const float* avg_factors;    // device, 2 floats: cls_avg_factor, num_total_pos (both clamped to >= 1 here)
  float* loss;                 // (NL, 2): loss_cls, loss_bbox
  float* grad_cls;             // like cls
  float* grad_box;             // like box
  int NL, B, Q, C, code, gt_dim, sumG;
  float alpha, cls_weight, box_weight;
};

// One workgroup per decoder layer, one thread per (sample, query) row: sigmoid focal loss (gamma = 2) over the classes
// with label `num_classes` = background, L1 on the positives against the normalised ground truth (rows whose target is
// not finite are dropped, :837-842), both already divided by their normalisers; gradients written in the same pass.
__global__ __launch_bounds__(1024) void head_loss_kernel(const HeadLossParams p) {
  __shared__ float s_red[2][16];
  const int l = blockIdx.x;
  const float cls_avg = fmaxf(p.avg_factors[0], 1.0f), pos_avg = fmaxf(p.avg_factors[1], 1.0f);
  const float kc = p.cls_weight / cls_avg, kb = p.box_weight / pos_avg;
  float sum_cls = 0.f, sum_box = 0.f;
  for (int r = threadIdx.x; r < p.B * p.Q; r += blockDim.x) {
    const size_t row = (size_t)l * p.B * p.Q + r;
    int a = p.assigned[row];
    if (a >= p.sumG) a = -1;                               // never index past the ground truth (host validates; memory safety)
    int label = a >= 0 ? p.gt_labels[a] : p.C;
    if (label < 0 || label > p.C) label = p.C;
    const float* x = p.cls + row * p.C;
    float* gx = p.grad_cls + row * p.C;
    for (int c = 0; c < p.C; ++c) {
      const float v = x[c];
      const bool t = c == label;
      const float pr = 1.0f / (1.0f + expf(-v));
      const float pt = t ? 1.0f - pr : pr;
      const float aw = t ? p.alpha : 1.0f - p.alpha;
      const float fw = aw * (pt * pt);
      // binary_cross_entropy_with_logits: max(v, 0) - v t + log(1 + exp(-|v|))
      const float bce = fmaxf(v, 0.f) - (t ? v : 0.f) + log1pf(expf(-fabsf(v)));
      sum_cls += bce * fw;
      const float dpt = (t ? -1.0f : 1.0f) * pr * (1.0f - pr);
      gx[c] = kc * (fw * (pr - (t ? 1.0f : 0.f)) + bce * aw * 2.0f * pt * dpt);
    }
    float* gb = p.grad_box + row * p.code;
    for (int k = 0; k < p.code; ++k) gb[k] = 0.f;
    if (a >= 0) {
      float tgt[10];
      normalize_box(p.gt_boxes + (size_t)a * p.gt_dim, p.gt_dim, tgt);
      const int nk = min(p.gt_dim > 7 ? 10 : 8, p.code);
      bool ok = true;
      for (int k = 0; k < nk; ++k) ok = ok && (fabsf(tgt[k]) <= 3.402823466e38f);     // finite (NaN compares false)
      if (ok) {
        const float* bx = p.box + row * p.code;
        for (int k = 0; k < nk; ++k) {
          const float d = bx[k] - tgt[k], w = p.code_weights[k];
          sum_box += fabsf(d) * w;
          gb[k] = kb * w * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f));
        }
      }
    }
  }
  // fixed-order reduction: lanes, then the 16 waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum_cls += __shfl_xor(sum_cls, o); sum_box += __shfl_xor(sum_box, o); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[0][wave] = sum_cls; s_red[1][wave] = sum_box; }
  __syncthreads();
  if (threadIdx.x < 2) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += s_red[threadIdx.x][w];
    s *= threadIdx.x == 0 ? kc : kb;
    if (s != s) s = 0.f;                                   // torch.nan_to_num on the two loss terms (:844-845):
    else if (s == INFINITY) s = 3.402823466e38f;           // nan -> 0, +/-inf -> +/-FLT_MAX
    else if (s == -INFINITY) s = -3.402823466e38f;
    p.loss[2 * l + threadIdx.x] = s;
  }
}

}  // namespace gd4d

extern "C" int gd4d_match_cost_fwd(const float* cls, const float* box, const float* gt_boxes, const int32_t* gt_labels,
                                   const int32_t* gt_start, float* cost, int NL, int B, int Q, int C, int code,
                                   int gt_dim, int sum_gt, int max_gt, float cls_weight, float reg_weight, float alpha,
                                   void* stream) {
  using namespace gd4d;
  if (!cls || !box || !gt_boxes || !gt_labels || !gt_start || !cost) return GD4D_EINVAL;
  if (NL <= 0 || B <= 0 || Q <= 0 || C <= 0 || sum_gt <= 0 || max_gt <= 0) return GD4D_EINVAL;
  if (code < 8 || gt_dim < 7 || gt_dim > 9 || max_gt > MC_MAX_GT || B > 65535 || NL > 65535) return GD4D_EUNSUPPORTED;
  MatchCostParams p{cls, box, gt_boxes, gt_labels, gt_start, cost, NL, B, Q, C, code, gt_dim, sum_gt,
                    cls_weight, reg_weight, alpha};
  const dim3 grid((Q + MC_QB - 1) / MC_QB, B, NL);
  hipLaunchKernelGGL(match_cost_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_head_loss_fwd_bwd(const float* cls, const float* box, const int32_t* assigned, const float* gt_boxes,
                                      const int32_t* gt_labels, const float* code_weights, const float* avg_factors,
                                      float* loss, float* grad_cls, float* grad_box, int NL, int B, int Q, int C,
                                      int code, int gt_dim, int sum_gt, float alpha, float loss_cls_weight,
                                      float loss_bbox_weight, void* stream) {
  using namespace gd4d;
  if (!cls || !box || !assigned || !gt_boxes || !gt_labels || !code_weights || !avg_factors || !loss || !grad_cls ||
      !grad_box)
    return GD4D_EINVAL;
  if (NL <= 0 || B <= 0 || Q <= 0 || C <= 0 || sum_gt <= 0) return GD4D_EINVAL;
  if (code < 8 || code > 16 || gt_dim < 7 || gt_dim > 9) return GD4D_EUNSUPPORTED;
  HeadLossParams p{cls, box, assigned, gt_boxes, gt_labels, code_weights, avg_factors, loss, grad_cls, grad_box,
                   NL, B, Q, C, code, gt_dim, sum_gt, alpha, loss_cls_weight, loss_bbox_weight};
  hipLaunchKernelGGL(head_loss_kernel, dim3(NL), dim3(1024), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

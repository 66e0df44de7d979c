// gd4d_detr3d_fwd: DETR3D-baseline cross-attention core for gfx950 - feature_sampling
// (detr3d_transformer.py:397-438) fused with the sigmoid weighting and the sum over cameras and
// levels of Detr3DCrossAtten.forward (detr3d_transformer.py:373-383).  SURVEY.md Appendix A.2.
//
// The feature maps stay in the layout the reference's caller provides: per level (B*N, C, H, W)
// NCHW.  One workgroup = one (batch, query); thread c = channel c, so every global load of a wave
// walks 64 channel planes at the same (y, x): strided by H*W floats, but only the few pixels a
// query projects to are ever touched (about 1 camera x 4 levels x 4 corners per query), which is
// far less traffic than re-laying-out the whole pyramid first.  The projection is spread over the
// first N threads (bit-exact arithmetic, translation unit built with -ffp-contract=off).
#include "gd4d_common.h"

namespace gd4d {

struct Detr3dParams {
  const float* feats[GD4D_MAX_LEVELS];
  const float* ref;
  const float* attn_logits;   // (B, Q, N, P=1, L)
  const float* lidar2img;
  float* out;                 // (B, Q, C) or null
  uint8_t* mask_out;          // (B, N, Q) or null
  float* sampled_out;         // (B, C, Q, N, 1, L) or null
  int B, N, Q, C, L;
  int lvl_h[GD4D_MAX_LEVELS];
  int lvl_w[GD4D_MAX_LEVELS];
  float rng_scale[3];
  float rng_lo[3];
  float img_h, img_w;
};

// ATen grid_sampler, bilinear / zeros / align_corners=False, on grid coordinate g in [-1, 1]:
//   x = ((g + 1) * size - 1) / 2 ; corners floor(x), floor(x)+1 ; out-of-map corners contribute 0.
__device__ __forceinline__ float unnormalize(float g, int size) {
  return ((g + 1.f) * (float)size - 1.f) / 2.f;
}

__global__ __launch_bounds__(256) void detr3d_fwd_kernel(const Detr3dParams p) {
  extern __shared__ float s_mem[];
  float2* s_uv = reinterpret_cast<float2*>(s_mem);            // [N] grid coords in [-1,1]
  float* s_w = s_mem + 2 * p.N;                               // [N*L] sigmoid(logit) * vis
  int* s_vis = reinterpret_cast<int*>(s_w + p.N * p.L);       // [N]

  const int bq = blockIdx.x;
  const int b = bq / p.Q, q = bq - b * p.Q;
  const int t = threadIdx.x;

  if (t < p.N) {
    const int n = t;
    const float* rp = p.ref + (size_t)bq * 3;
    const float X = rp[0] * p.rng_scale[0] + p.rng_lo[0];
    const float Y = rp[1] * p.rng_scale[1] + p.rng_lo[1];
    const float Z = rp[2] * p.rng_scale[2] + p.rng_lo[2];
    const float* m = p.lidar2img + ((size_t)b * p.N + n) * 16;
    const float eps = 1e-5f;
    const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
    const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
    const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
    bool vis = cz > eps;
    const float zc = fmaxf(cz, eps);
    float u = (cx / zc) / p.img_w;
    float v = (cy / zc) / p.img_h;
    u = (u - 0.5f) * 2.f;                                     // detr3d_transformer.py:421
    v = (v - 0.5f) * 2.f;
    vis = vis && (u > -1.f) && (u < 1.f) && (v > -1.f) && (v < 1.f);
    s_uv[n] = make_float2(u, v);
    s_vis[n] = vis ? 1 : 0;
    if (p.mask_out) p.mask_out[((size_t)b * p.N + n) * p.Q + q] = vis ? 1 : 0;
  }
  __syncthreads();
  for (int e = t; e < p.N * p.L; e += blockDim.x) {
    const int n = e / p.L;
    const float lg = p.attn_logits[(size_t)bq * p.N * p.L + e];
    s_w[e] = s_vis[n] ? 1.0f / (1.0f + expf(-lg)) : 0.f;
  }
  __syncthreads();

  const bool want_all = p.sampled_out != nullptr;     // feature_sampling() returns every camera
  for (int c = t; c < p.C; c += blockDim.x) {
    float acc = 0.f;
    for (int n = 0; n < p.N; ++n) {
      if (!want_all && !s_vis[n]) continue;                   // workgroup-uniform
      const float2 g = s_uv[n];
      for (int l = 0; l < p.L; ++l) {
        const int H = p.lvl_h[l], W = p.lvl_w[l];
        const float x = unnormalize(g.x, W), y = unnormalize(g.y, H);
        const float xf = floorf(x), yf = floorf(y);
        const float dx = x - xf, dy = y - yf;
        // float compares first: far-away points can exceed the int range
        const bool x0ok = xf >= 0.f && xf <= (float)(W - 1), x1ok = xf + 1.f >= 0.f && xf + 1.f <= (float)(W - 1);
        const bool y0ok = yf >= 0.f && yf <= (float)(H - 1), y1ok = yf + 1.f >= 0.f && yf + 1.f <= (float)(H - 1);
        float s = 0.f;
        if ((x0ok || x1ok) && (y0ok || y1ok)) {
          const int x0 = (int)xf, y0 = (int)yf;
          const float* plane = p.feats[l] + ((size_t)(b * p.N + n) * p.C + c) * H * W;
          if (x0ok && y0ok) s += (1.f - dx) * (1.f - dy) * plane[y0 * W + x0];
          if (x1ok && y0ok) s += dx * (1.f - dy) * plane[y0 * W + x0 + 1];
          if (x0ok && y1ok) s += (1.f - dx) * dy * plane[(y0 + 1) * W + x0];
          if (x1ok && y1ok) s += dx * dy * plane[(y0 + 1) * W + x0 + 1];
        }
        if (want_all)
          p.sampled_out[((((size_t)b * p.C + c) * p.Q + q) * p.N + n) * p.L + l] = s;
        acc = fmaf(s_w[n * p.L + l], s, acc);
      }
    }
    if (p.out) p.out[(size_t)bq * p.C + c] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// gd4d_detr3d_v2_fwd: Detr3DCrossAttenV2 (detr3d_transformer.py:441-710) - the 2-D-offset deformable variant.
// Same work mapping as above (workgroup = query, thread = channel, NCHW maps untouched); head h = c / Dh reads its
// own sampling locations: grid coordinate of the projected reference point + offset / (W_l, H_l) (:696-700), weights =
// softmax over level x point per (camera, head) times the camera's visibility (:602-611, :617).
// Quirk reproduced: the reference stacks samples as (..., point, level) and multiplies them with weights laid out
// (..., level, point) (:611 against :705-707), so the sample at (point i, level j) meets weight (level i, point j) -
// well-formed only for num_levels == num_points, which the host enforces.
struct Detr3dV2Params {
  const float* feats[GD4D_MAX_LEVELS];
  const float* ref;
  const float* attn_logits;   // (B, Q, N, Hh, L*P)
  const float* offsets;       // (B, Q, N, Hh, L, P, 2) pixels of the level
  const float* lidar2img;
  float* out;                 // (B, Q, C)
  uint8_t* mask_out;          // (B, N, Q) or null
  int B, N, Q, C, L, Hh;
  int lvl_h[GD4D_MAX_LEVELS];
  int lvl_w[GD4D_MAX_LEVELS];
  float rng_scale[3];
  float rng_lo[3];
  float img_h, img_w;
};

__global__ __launch_bounds__(256) void detr3d_v2_fwd_kernel(const Detr3dV2Params p) {
  extern __shared__ float s_mem[];
  const int LP = p.L * p.L;                                   // num_points == num_levels
  float2* s_uv = reinterpret_cast<float2*>(s_mem);            // [N] grid coords in [-1,1]
  float* s_w = s_mem + 2 * p.N;                               // [N][Hh][L*P] softmax * vis
  int* s_vis = reinterpret_cast<int*>(s_w + p.N * p.Hh * LP); // [N]
  const int bq = blockIdx.x;
  const int b = bq / p.Q, q = bq - b * p.Q;
  const int t = threadIdx.x;
  if (t < p.N) {
    const int n = t;
    const float* rp = p.ref + (size_t)bq * 3;
    const float X = rp[0] * p.rng_scale[0] + p.rng_lo[0];
    const float Y = rp[1] * p.rng_scale[1] + p.rng_lo[1];
    const float Z = rp[2] * p.rng_scale[2] + p.rng_lo[2];
    const float* m = p.lidar2img + ((size_t)b * p.N + n) * 16;
    const float eps = 1e-5f;
    const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
    const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
    const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
    bool vis = cz > eps;
    const float zc = fmaxf(cz, eps);
    float u = (cx / zc) / p.img_w;
    float v = (cy / zc) / p.img_h;
    u = (u - 0.5f) * 2.f;                                     // :676
    v = (v - 0.5f) * 2.f;
    vis = vis && (u > -1.f) && (u < 1.f) && (v > -1.f) && (v < 1.f);
    s_uv[n] = make_float2(u, v);
    s_vis[n] = vis ? 1 : 0;
    if (p.mask_out) p.mask_out[((size_t)b * p.N + n) * p.Q + q] = vis ? 1 : 0;
  }
  __syncthreads();
  for (int e = t; e < p.N * p.Hh; e += blockDim.x) {          // one thread per (camera, head): softmax over L*P
    const int n = e / p.Hh;
    const float* lg = p.attn_logits + ((size_t)bq * p.N * p.Hh + e) * LP;
    float* w = s_w + (size_t)e * LP;
    if (!s_vis[n]) { for (int i = 0; i < LP; ++i) w[i] = 0.f; continue; }
    float mx = lg[0];
    for (int i = 1; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
    for (int i = 0; i < LP; ++i) { w[i] = expf(lg[i] - mx); sum += w[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < LP; ++i) w[i] *= inv;
  }
  __syncthreads();
  const int Dh = p.C / p.Hh;
  for (int c = t; c < p.C; c += blockDim.x) {
    const int h = c / Dh;
    float acc = 0.f;
    for (int n = 0; n < p.N; ++n) {
      if (!s_vis[n]) continue;                                // workgroup-uniform
      const float2 g = s_uv[n];
      const float* w = s_w + ((size_t)n * p.Hh + h) * LP;
      const float* off = p.offsets + (((size_t)bq * p.N + n) * p.Hh + h) * LP * 2;
      for (int l = 0; l < p.L; ++l) {
        const int H = p.lvl_h[l], W = p.lvl_w[l];
        const float* plane = p.feats[l] + ((size_t)(b * p.N + n) * p.C + c) * H * W;
        for (int pt = 0; pt < p.L; ++pt) {
          const float gx = g.x + off[(l * p.L + pt) * 2] / (float)W;          // :699-700
          const float gy = g.y + off[(l * p.L + pt) * 2 + 1] / (float)H;
          const float x = unnormalize(gx, W), y = unnormalize(gy, H);
          const float xf = floorf(x), yf = floorf(y);
          const float dx = x - xf, dy = y - yf;
          const bool x0ok = xf >= 0.f && xf <= (float)(W - 1), x1ok = xf + 1.f >= 0.f && xf + 1.f <= (float)(W - 1);
          const bool y0ok = yf >= 0.f && yf <= (float)(H - 1), y1ok = yf + 1.f >= 0.f && yf + 1.f <= (float)(H - 1);
          float s = 0.f;
          if ((x0ok || x1ok) && (y0ok || y1ok)) {
            const int x0 = (int)xf, y0 = (int)yf;
            if (x0ok && y0ok) s += (1.f - dx) * (1.f - dy) * plane[y0 * W + x0];
            if (x1ok && y0ok) s += dx * (1.f - dy) * plane[y0 * W + x0 + 1];
            if (x0ok && y1ok) s += (1.f - dx) * dy * plane[(y0 + 1) * W + x0];
            if (x1ok && y1ok) s += dx * dy * plane[(y0 + 1) * W + x0 + 1];
          }
          acc = fmaf(w[pt * p.L + l], s, acc);                // weight of (level = pt, point = l): the reference's pairing
        }
      }
    }
    p.out[(size_t)bq * p.C + c] = acc;
  }
}

}  // namespace gd4d

namespace gd4d {

// ---------------------------------------------------------------------------------------------
// gd4d_detr3d_bwd: backward of detr3d_fwd_kernel's `out` (the reference: autograd through feature_sampling's
// F.grid_sample per level, the sigmoid weights, the mask product and the sums, detr3d_transformer.py:373-383, :397-438).
// Same work mapping (workgroup = (batch, query), thread = channel).  With g = dL/d out (C), w = sigmoid(logit) * vis and
// s[c, n, l] the bilinear sample:
//   dL/d feats[l][b*N+n, c, corner] += g[c] w[n,l] b_corner                (atomic fp32 add; ~1 camera x L levels x 4 corners)
//   dL/d logit[n,l] = vis w (1 - w) sum_c g[c] s[c,n,l]
//   dL/d (x, y)     = w sum_c g[c] ds/d(x, y)  (pixel units) -> grid coordinate -> u = cx / (cz W_img), v = cy / (cz H_img)
//                     -> lidar2img -> metres -> reference point.  The mask is piecewise constant (no gradient).
// The sums over channels meet in LDS in a fixed order.
struct Detr3dBwdParams {
  const float* feats[GD4D_MAX_LEVELS];
  float* gfeats[GD4D_MAX_LEVELS];   // zero-initialised by the caller, or NULL
  const float* ref;
  const float* attn_logits;
  const float* lidar2img;
  const float* grad_out;            // (B, Q, C)
  float* grad_logits;               // (B, Q, N, 1, L)
  float* grad_ref;                  // (B, Q, 3)
  int B, N, Q, C, L;
  int lvl_h[GD4D_MAX_LEVELS];
  int lvl_w[GD4D_MAX_LEVELS];
  float rng_scale[3];
  float rng_lo[3];
  float img_h, img_w;
};

__global__ __launch_bounds__(256) void detr3d_bwd_kernel(const Detr3dBwdParams p) {
  extern __shared__ float s_mem[];
  float2* s_uv = reinterpret_cast<float2*>(s_mem);            // [N]
  float* s_cam = s_mem + 2 * p.N;                             // [N][4]: cx, cy, cz, vis
  float* s_red = s_cam + 4 * p.N;                             // [4 waves][3]
  float* s_pt = s_red + 12;                                   // [3] dL/d (metre point), accumulated by thread 0
  const int bq = blockIdx.x;
  const int b = bq / p.Q;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* rp = p.ref + (size_t)bq * 3;
  const float X = rp[0] * p.rng_scale[0] + p.rng_lo[0];
  const float Y = rp[1] * p.rng_scale[1] + p.rng_lo[1];
  const float Z = rp[2] * p.rng_scale[2] + p.rng_lo[2];
  if (t < p.N) {
    const float* m = p.lidar2img + ((size_t)b * p.N + t) * 16;
    const float eps = 1e-5f;
    const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
    const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
    const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
    bool vis = cz > eps;
    const float zc = fmaxf(cz, eps);
    float u = (cx / zc) / p.img_w;
    float v = (cy / zc) / p.img_h;
    u = (u - 0.5f) * 2.f;
    v = (v - 0.5f) * 2.f;
    vis = vis && (u > -1.f) && (u < 1.f) && (v > -1.f) && (v < 1.f);
    s_uv[t] = make_float2(u, v);
    s_cam[4 * t] = cx; s_cam[4 * t + 1] = cy; s_cam[4 * t + 2] = cz; s_cam[4 * t + 3] = vis ? 1.f : 0.f;
  }
  if (t < 3) s_pt[t] = 0.f;
  __syncthreads();
  // (C <= 256 * 4: a thread owns up to 4 channels, all of them inside every reduction)
  for (int n = 0; n < p.N; ++n) {
    const bool vis = s_cam[4 * n + 3] != 0.f;                 // workgroup-uniform
    if (!vis) {
      if (t < p.L) p.grad_logits[((size_t)bq * p.N + n) * p.L + t] = 0.f;
      continue;
    }
    const float2 g2 = s_uv[n];
    float gu = 0.f, gv = 0.f;                                 // dL/d(u, v) in [0, 1] image units, summed over the levels (thread 0)
    for (int l = 0; l < p.L; ++l) {
      const int H = p.lvl_h[l], W = p.lvl_w[l];
      const float x = unnormalize(g2.x, W), y = unnormalize(g2.y, H);
      const float xf = floorf(x), yf = floorf(y);
      const float dx = x - xf, dy = y - yf;
      const bool x0ok = xf >= 0.f && xf <= (float)(W - 1), x1ok = xf + 1.f >= 0.f && xf + 1.f <= (float)(W - 1);
      const bool y0ok = yf >= 0.f && yf <= (float)(H - 1), y1ok = yf + 1.f >= 0.f && yf + 1.f <= (float)(H - 1);
      const float lg = p.attn_logits[((size_t)bq * p.N + n) * p.L + l];
      const float w = 1.0f / (1.0f + expf(-lg));
      float part[3] = {0.f, 0.f, 0.f};                        // sum_c g s, sum_c g ds/dx, sum_c g ds/dy
      if ((x0ok || x1ok) && (y0ok || y1ok)) {
        const int x0 = (int)xf, y0 = (int)yf;
        for (int c = t; c < p.C; c += 256) {
          const size_t plane = ((size_t)(b * p.N + n) * p.C + c) * H * W;
          const float* fp = p.feats[l] + plane;
          const float v00 = (x0ok && y0ok) ? fp[y0 * W + x0] : 0.f;
          const float v01 = (x1ok && y0ok) ? fp[y0 * W + x0 + 1] : 0.f;
          const float v10 = (x0ok && y1ok) ? fp[(y0 + 1) * W + x0] : 0.f;
          const float v11 = (x1ok && y1ok) ? fp[(y0 + 1) * W + x0 + 1] : 0.f;
          const float g = p.grad_out[(size_t)bq * p.C + c];
          const float sv = ((1.f - dx) * (1.f - dy) * v00 + dx * (1.f - dy) * v01) + ((1.f - dx) * dy * v10 + dx * dy * v11);
          part[0] += g * sv;
          part[1] += g * ((1.f - dy) * (v01 - v00) + dy * (v11 - v10));
          part[2] += g * ((1.f - dx) * (v10 - v00) + dx * (v11 - v01));
          if (p.gfeats[l]) {
            float* gp = p.gfeats[l] + plane;
            const float gw = g * w;
            if (x0ok && y0ok) atomicAdd(gp + y0 * W + x0, gw * (1.f - dx) * (1.f - dy));
            if (x1ok && y0ok) atomicAdd(gp + y0 * W + x0 + 1, gw * dx * (1.f - dy));
            if (x0ok && y1ok) atomicAdd(gp + (y0 + 1) * W + x0, gw * (1.f - dx) * dy);
            if (x1ok && y1ok) atomicAdd(gp + (y0 + 1) * W + x0 + 1, gw * dx * dy);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part[k] += __shfl_xor(part[k], o);
      }
      __syncthreads();                                        // (s_red of the previous level has been read)
      if (lane == 0) { s_red[wave * 3] = part[0]; s_red[wave * 3 + 1] = part[1]; s_red[wave * 3 + 2] = part[2]; }
      __syncthreads();
      if (t == 0) {
        float tot[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) tot[k] = ((s_red[k] + s_red[3 + k]) + s_red[6 + k]) + s_red[9 + k];
        p.grad_logits[((size_t)bq * p.N + n) * p.L + l] = w * (1.f - w) * tot[0];
        // x = ((gx + 1) W - 1) / 2, gx = (u - 0.5) 2  ->  dx/du = W ; likewise dy/dv = H
        gu += w * tot[1] * (float)W;
        gv += w * tot[2] * (float)H;
      }
    }
    if (t == 0) {
      const float* m = p.lidar2img + ((size_t)b * p.N + n) * 16;
      const float cx = s_cam[4 * n], cy = s_cam[4 * n + 1], cz = s_cam[4 * n + 2];
      const float iz = 1.0f / cz;                             // visible: cz > eps
      const float gcx = gu * iz / p.img_w;
      const float gcy = gv * iz / p.img_h;
      const float gcz = -(gu * cx * iz * iz / p.img_w + gv * cy * iz * iz / p.img_h);
      s_pt[0] += gcx * m[0] + gcy * m[4] + gcz * m[8];
      s_pt[1] += gcx * m[1] + gcy * m[5] + gcz * m[9];
      s_pt[2] += gcx * m[2] + gcy * m[6] + gcz * m[10];
    }
  }
  __syncthreads();
  if (t < 3 && p.grad_ref) p.grad_ref[(size_t)bq * 3 + t] = s_pt[t] * p.rng_scale[t];
}

}  // namespace gd4d

extern "C" int gd4d_detr3d_bwd(const void* const* feats, const int32_t* level_hw, const float* ref, const float* attn_logits,
                               const float* lidar2img, const double* pc_range, float img_h, float img_w, const float* grad_out,
                               void* const* grad_feats, float* grad_logits, float* grad_ref, int B, int N, int Q, int C, int L,
                               int P, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !ref || !attn_logits || !lidar2img || !pc_range || !grad_out || !grad_logits) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || C <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  if (L > GD4D_MAX_LEVELS || P != 1 || N > 256) return GD4D_EUNSUPPORTED;
  Detr3dBwdParams p{};
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    p.feats[l] = static_cast<const float*>(feats[l]);
    p.gfeats[l] = grad_feats ? static_cast<float*>(grad_feats[l]) : nullptr;
    p.lvl_h[l] = level_hw[2 * l];
    p.lvl_w[l] = level_hw[2 * l + 1];
  }
  p.ref = ref; p.attn_logits = attn_logits; p.lidar2img = lidar2img; p.grad_out = grad_out;
  p.grad_logits = grad_logits; p.grad_ref = grad_ref;
  p.B = B; p.N = N; p.Q = Q; p.C = C; p.L = L;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  const size_t lds = sizeof(float) * (size_t)(2 * N + 4 * N + 12 + 3);
  hipLaunchKernelGGL(detr3d_bwd_kernel, dim3(B * Q), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_detr3d_fwd(const void* const* feats, const int32_t* level_hw, const float* ref,
                               const float* attn_logits, const float* lidar2img,
                               const double* pc_range, float img_h, float img_w, float* out,
                               uint8_t* mask_out, float* sampled_out, int B, int N, int Q, int C,
                               int L, int P, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !ref || !attn_logits || !lidar2img || !pc_range) return GD4D_EINVAL;
  if (!out && !sampled_out && !mask_out) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || C <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  if (L > GD4D_MAX_LEVELS || P != 1 || N > 256) return GD4D_EUNSUPPORTED;
  Detr3dParams p{};
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    p.feats[l] = static_cast<const float*>(feats[l]);
    p.lvl_h[l] = level_hw[2 * l];
    p.lvl_w[l] = level_hw[2 * l + 1];
  }
  p.ref = ref; p.attn_logits = attn_logits; p.lidar2img = lidar2img;
  p.out = out; p.mask_out = mask_out; p.sampled_out = sampled_out;
  p.B = B; p.N = N; p.Q = Q; p.C = C; p.L = L;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  const size_t lds = sizeof(float) * (size_t)(2 * N + N * L) + sizeof(int) * (size_t)N;
  hipLaunchKernelGGL(detr3d_fwd_kernel, dim3(B * Q), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_detr3d_v2_fwd(const void* const* feats, const int32_t* level_hw, const float* ref,
                                  const float* attn_logits, const float* offsets, const float* lidar2img,
                                  const double* pc_range, float img_h, float img_w, float* out, uint8_t* mask_out,
                                  int B, int N, int Q, int C, int Hh, int L, int P, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !ref || !attn_logits || !offsets || !lidar2img || !pc_range || !out) return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || C <= 0 || Hh <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  if (L > GD4D_MAX_LEVELS || P != L || N > 256 || C % Hh != 0) return GD4D_EUNSUPPORTED;
  Detr3dV2Params p{};
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    p.feats[l] = static_cast<const float*>(feats[l]);
    p.lvl_h[l] = level_hw[2 * l];
    p.lvl_w[l] = level_hw[2 * l + 1];
  }
  p.ref = ref; p.attn_logits = attn_logits; p.offsets = offsets; p.lidar2img = lidar2img;
  p.out = out; p.mask_out = mask_out;
  p.B = B; p.N = N; p.Q = Q; p.C = C; p.L = L; p.Hh = Hh;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  const size_t lds = sizeof(float) * (size_t)(2 * N + N * Hh * L * L) + sizeof(int) * (size_t)N;
  if (lds > 64 * 1024) return GD4D_EUNSUPPORTED;
  hipLaunchKernelGGL(detr3d_v2_fwd_kernel, dim3(B * Q), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

namespace gd4d {

// ---------------------------------------------------------------------------------------------
// gd4d_detr3d_v2_bwd: backward of detr3d_v2_fwd_kernel (the reference: autograd through Detr3DCrossAttenV2's per-level
// F.grid_sample of every head's channel slice, the softmax over level x point, the (point, level) x (level, point)
// pairing and the sums, detr3d_transformer.py:597-710).  Same work mapping as the forward (workgroup = (batch, query),
// thread = channel, head h = c / Dh).  With g = dL/d out (C), W[n,h,i] = softmax_i(logits[n,h,:]) * vis[n] and
// s[c,n,l,pt] the bilinear sample of channel c at grid g_n + offsets[n,h,l,pt] / (W_l, H_l):
//   dL/d feats[l][b*N+n, c, corner] += g[c] W[n,h,pt*L+l] b_corner               (atomic fp32 add)
//   dL/d W[n,h,pt*L+l]   = sum_{c in h} g[c] s[c,n,l,pt]      ->  softmax backward -> dL/d logits[n,h,:]
//   dL/d offsets[n,h,l,pt] = W[n,h,pt*L+l] sum_{c in h} g[c] ds/d(x, y) / 2       (x = ((gx + 1) W - 1) / 2, gx = g.x + off / W)
//   dL/d g_n = sum_{h,l,pt} W sum_c g ds/d(x, y) (W_l, H_l) / 2  -> u, v -> lidar2img -> metres -> reference point
// The sums over a head's channels meet in LDS in channel order (deterministic).  C <= 256.
struct Detr3dV2BwdParams {
  const float* feats[GD4D_MAX_LEVELS];
  float* gfeats[GD4D_MAX_LEVELS];   // zero-initialised by the caller, or NULL
  const float* ref;
  const float* attn_logits;         // (B, Q, N, Hh, L*P)
  const float* offsets;             // (B, Q, N, Hh, L, P, 2)
  const float* lidar2img;
  const float* grad_out;            // (B, Q, C)
  float* grad_logits;               // (B, Q, N, Hh, L*P)
  float* grad_offsets;              // (B, Q, N, Hh, L, P, 2)
  float* grad_ref;                  // (B, Q, 3) or NULL
  int B, N, Q, C, L, Hh;
  int lvl_h[GD4D_MAX_LEVELS];
  int lvl_w[GD4D_MAX_LEVELS];
  float rng_scale[3];
  float rng_lo[3];
  float img_h, img_w;
};

__global__ __launch_bounds__(256) void detr3d_v2_bwd_kernel(const Detr3dV2BwdParams p) {
  extern __shared__ float s_mem[];
  const int LP = p.L * p.L;                                   // num_points == num_levels
  float2* s_uv = reinterpret_cast<float2*>(s_mem);            // [N]
  float* s_cam = s_mem + 2 * p.N;                             // [N][4]: cx, cy, cz, vis
  float* s_w = s_cam + 4 * p.N;                               // [N][Hh][LP] softmax (zeros for an invisible camera)
  float* s_part = s_w + p.N * p.Hh * LP;                      // [3][256]
  float* s_dw = s_part + 3 * 256;                             // [Hh][LP] dL/dW of the current camera
  float* s_g = s_dw + p.Hh * LP;                              // [Hh][2] dL/d(grid x, y) of the current camera, per head
  float* s_pt = s_g + 2 * p.Hh;                               // [3]
  const int bq = blockIdx.x;
  const int b = bq / p.Q;
  const int t = threadIdx.x;
  const float* rp = p.ref + (size_t)bq * 3;
  const float X = rp[0] * p.rng_scale[0] + p.rng_lo[0];
  const float Y = rp[1] * p.rng_scale[1] + p.rng_lo[1];
  const float Z = rp[2] * p.rng_scale[2] + p.rng_lo[2];
  if (t < p.N) {
    const float* m = p.lidar2img + ((size_t)b * p.N + t) * 16;
    const float eps = 1e-5f;
    const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
    const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
    const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
    bool vis = cz > eps;
    const float zc = fmaxf(cz, eps);
    float u = (cx / zc) / p.img_w;
    float v = (cy / zc) / p.img_h;
    u = (u - 0.5f) * 2.f;
    v = (v - 0.5f) * 2.f;
    vis = vis && (u > -1.f) && (u < 1.f) && (v > -1.f) && (v < 1.f);
    s_uv[t] = make_float2(u, v);
    s_cam[4 * t] = cx; s_cam[4 * t + 1] = cy; s_cam[4 * t + 2] = cz; s_cam[4 * t + 3] = vis ? 1.f : 0.f;
  }
  if (t < 3) s_pt[t] = 0.f;
  __syncthreads();
  for (int e = t; e < p.N * p.Hh; e += blockDim.x) {          // softmax over L*P per (camera, head), as the forward
    const int n = e / p.Hh;
    const float* lg = p.attn_logits + ((size_t)bq * p.N * p.Hh + e) * LP;
    float* w = s_w + (size_t)e * LP;
    if (s_cam[4 * n + 3] == 0.f) { for (int i = 0; i < LP; ++i) w[i] = 0.f; continue; }
    float mx = lg[0];
    for (int i = 1; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
    for (int i = 0; i < LP; ++i) { w[i] = expf(lg[i] - mx); sum += w[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < LP; ++i) w[i] *= inv;
  }
  __syncthreads();
  const int Dh = p.C / p.Hh;
  const int c = t;                                            // C <= 256: one channel per thread
  const bool c_ok = c < p.C;
  const int h = c_ok ? c / Dh : 0;
  const float g = c_ok ? p.grad_out[(size_t)bq * p.C + c] : 0.f;
  for (int n = 0; n < p.N; ++n) {
    const size_t row = ((size_t)bq * p.N + n) * p.Hh;
    if (s_cam[4 * n + 3] == 0.f) {                            // workgroup-uniform: no gradient through an invisible camera
      for (int e = t; e < p.Hh * LP; e += blockDim.x) {
        p.grad_logits[row * LP + e] = 0.f;
        p.grad_offsets[(row * LP + e) * 2] = 0.f;
        p.grad_offsets[(row * LP + e) * 2 + 1] = 0.f;
      }
      continue;
    }
    const float2 g2 = s_uv[n];
    if (t < 2 * p.Hh) s_g[t] = 0.f;
    for (int l = 0; l < p.L; ++l) {
      const int H = p.lvl_h[l], W = p.lvl_w[l];
      for (int pt = 0; pt < p.L; ++pt) {
        float part0 = 0.f, part1 = 0.f, part2 = 0.f;
        if (c_ok) {
          const float* off = p.offsets + ((row + h) * LP + (l * p.L + pt)) * 2;
          const float gx = g2.x + off[0] / (float)W, gy = g2.y + off[1] / (float)H;
          const float x = unnormalize(gx, W), y = unnormalize(gy, H);
          const float xf = floorf(x), yf = floorf(y);
          const float dx = x - xf, dy = y - yf;
          const bool x0ok = xf >= 0.f && xf <= (float)(W - 1), x1ok = xf + 1.f >= 0.f && xf + 1.f <= (float)(W - 1);
          const bool y0ok = yf >= 0.f && yf <= (float)(H - 1), y1ok = yf + 1.f >= 0.f && yf + 1.f <= (float)(H - 1);
          if ((x0ok || x1ok) && (y0ok || y1ok)) {
            const int x0 = (int)xf, y0 = (int)yf;
            const size_t plane = ((size_t)(b * p.N + n) * p.C + c) * H * W;
            const float* fp = p.feats[l] + plane;
            const float v00 = (x0ok && y0ok) ? fp[y0 * W + x0] : 0.f;
            const float v01 = (x1ok && y0ok) ? fp[y0 * W + x0 + 1] : 0.f;
            const float v10 = (x0ok && y1ok) ? fp[(y0 + 1) * W + x0] : 0.f;
            const float v11 = (x1ok && y1ok) ? fp[(y0 + 1) * W + x0 + 1] : 0.f;
            const float sv = ((1.f - dx) * (1.f - dy) * v00 + dx * (1.f - dy) * v01) + ((1.f - dx) * dy * v10 + dx * dy * v11);
            part0 = g * sv;
            part1 = g * ((1.f - dy) * (v01 - v00) + dy * (v11 - v10));
            part2 = g * ((1.f - dx) * (v10 - v00) + dx * (v11 - v01));
            if (p.gfeats[l]) {
              float* gp = p.gfeats[l] + plane;
              const float gw = g * s_w[((size_t)n * p.Hh + h) * LP + pt * p.L + l];     // weight (level = pt, point = l)
              if (x0ok && y0ok) atomicAdd(gp + y0 * W + x0, gw * (1.f - dx) * (1.f - dy));
              if (x1ok && y0ok) atomicAdd(gp + y0 * W + x0 + 1, gw * dx * (1.f - dy));
              if (x0ok && y1ok) atomicAdd(gp + (y0 + 1) * W + x0, gw * (1.f - dx) * dy);
              if (x1ok && y1ok) atomicAdd(gp + (y0 + 1) * W + x0 + 1, gw * dx * dy);
            }
          }
        }
        s_part[t] = part0; s_part[256 + t] = part1; s_part[512 + t] = part2;
        __syncthreads();
        if (t < p.Hh) {                                       // head t: its Dh channels in order
          float tot0 = 0.f, tot1 = 0.f, tot2 = 0.f;
          for (int k = 0; k < Dh; ++k) { tot0 += s_part[t * Dh + k]; tot1 += s_part[256 + t * Dh + k]; tot2 += s_part[512 + t * Dh + k]; }
          const float w = s_w[((size_t)n * p.Hh + t) * LP + pt * p.L + l];
          s_dw[t * LP + pt * p.L + l] = tot0;
          float* go = p.grad_offsets + ((row + t) * LP + (l * p.L + pt)) * 2;
          go[0] = 0.5f * w * tot1;                            // dx / d off_x = (W / 2) (1 / W)
          go[1] = 0.5f * w * tot2;
          s_g[2 * t] += w * tot1 * (0.5f * (float)W);         // dx / d gx = W / 2
          s_g[2 * t + 1] += w * tot2 * (0.5f * (float)H);
        }
        __syncthreads();
      }
    }
    if (t < p.Hh) {                                           // softmax backward of (camera n, head t)
      const float* w = s_w + ((size_t)n * p.Hh + t) * LP;
      float dot = 0.f;
      for (int i = 0; i < LP; ++i) dot += w[i] * s_dw[t * LP + i];
      for (int i = 0; i < LP; ++i) p.grad_logits[(row + t) * LP + i] = w[i] * (s_dw[t * LP + i] - dot);
    }
    if (t == 0) {
      float ggx = 0.f, ggy = 0.f;
      for (int hh = 0; hh < p.Hh; ++hh) { ggx += s_g[2 * hh]; ggy += s_g[2 * hh + 1]; }
      const float gu = 2.f * ggx, gv = 2.f * ggy;             // g = (u - 0.5) 2
      const float* m = p.lidar2img + ((size_t)b * p.N + n) * 16;
      const float cx = s_cam[4 * n], cy = s_cam[4 * n + 1], cz = s_cam[4 * n + 2];
      const float iz = 1.0f / cz;                             // visible: cz > eps
      const float gcx = gu * iz / p.img_w;
      const float gcy = gv * iz / p.img_h;
      const float gcz = -(gu * cx * iz * iz / p.img_w + gv * cy * iz * iz / p.img_h);
      s_pt[0] += gcx * m[0] + gcy * m[4] + gcz * m[8];
      s_pt[1] += gcx * m[1] + gcy * m[5] + gcz * m[9];
      s_pt[2] += gcx * m[2] + gcy * m[6] + gcz * m[10];
    }
    __syncthreads();
  }
  if (t < 3 && p.grad_ref) p.grad_ref[(size_t)bq * 3 + t] = s_pt[t] * p.rng_scale[t];
}

}  // namespace gd4d

extern "C" int gd4d_detr3d_v2_bwd(const void* const* feats, const int32_t* level_hw, const float* ref, const float* attn_logits,
                                  const float* offsets, const float* lidar2img, const double* pc_range, float img_h, float img_w,
                                  const float* grad_out, void* const* grad_feats, float* grad_logits, float* grad_offsets,
                                  float* grad_ref, int B, int N, int Q, int C, int L, int Hh, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !ref || !attn_logits || !offsets || !lidar2img || !pc_range || !grad_out || !grad_logits || !grad_offsets)
    return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || C <= 0 || L <= 0 || Hh <= 0 || C % Hh || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  if (L > GD4D_MAX_LEVELS || N > 256 || C > 256 || Hh > 64) return GD4D_EUNSUPPORTED;
  Detr3dV2BwdParams p{};
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    p.feats[l] = static_cast<const float*>(feats[l]);
    p.gfeats[l] = grad_feats ? static_cast<float*>(grad_feats[l]) : nullptr;
    p.lvl_h[l] = level_hw[2 * l];
    p.lvl_w[l] = level_hw[2 * l + 1];
  }
  p.ref = ref; p.attn_logits = attn_logits; p.offsets = offsets; p.lidar2img = lidar2img; p.grad_out = grad_out;
  p.grad_logits = grad_logits; p.grad_offsets = grad_offsets; p.grad_ref = grad_ref;
  p.B = B; p.N = N; p.Q = Q; p.C = C; p.L = L; p.Hh = Hh;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  const size_t lds = sizeof(float) * ((size_t)6 * N + (size_t)N * Hh * L * L + 3 * 256 + (size_t)Hh * L * L + 2 * Hh + 3);
  if (lds > 65536 && !allow_dynamic_lds(reinterpret_cast<const void*>(detr3d_v2_bwd_kernel), (int)lds)) return GD4D_ELAUNCH;
  hipLaunchKernelGGL(detr3d_v2_bwd_kernel, dim3(B * Q), dim3(256), lds, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

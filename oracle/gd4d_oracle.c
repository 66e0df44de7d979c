/*
 * ORACLE - test infrastructure, not product code.
 *
 * Plain-C (scalar, single-thread) restatement of the fused sample-aggregate step of the
 * reference's Deform3DCrossAttn.forward and of DETR3D's feature_sampling + weighting.  It exists
 * so that the bit-exact parts of the contract (projection, visibility mask) are pinned by
 * arithmetic that is identical on every host: torch's batched matmul may pick different kernels
 * on a different CPU, this file (built with -ffp-contract=off -O2, no -ffast-math) cannot.
 *
 * Reference lines followed (projects/mmdet3d_plugin/models/utils/):
 *   deform3d_cross_attn.py:222-224  de-normalise: ref*(hi-lo)+lo, (hi-lo) formed in double
 *   deform3d_cross_attn.py:227-230  add metre offsets (same 3-D point for every level)
 *   deform3d_cross_attn.py:232-243  [x y z 1] @ lidar2img^T, eps=1e-5 depth clamp, /W, /H
 *   deform3d_cross_attn.py:249-252  mask = z>eps & 0<u<1 & 0<v<1
 *   deform3d_cross_attn.py:277,281-284  softmax over L*P, times mask; logits row (b*N+n) % B
 *   deform3d_cross_attn.py:302-304  third-party mmcv MSDA: bilinear at (u*W-0.5, v*H-0.5), zero pad
 *   deform3d_cross_attn.py:211-212,320-324  scrambled camera logits, sigmoid, sum over cameras
 *   detr3d_transformer.py:397-438   feature_sampling (NCHW grid_sample at one point per query)
 *   detr3d_transformer.py:373-383   sigmoid(logits) * mask, triple sum
 *
 * Pinned by tests/test_c_oracle.py against tests/golden/ (vectors captured from the reference
 * itself by tools/gen_golden.py): uv and mask bit-for-bit, aggregated output to 1e-5.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_EPS 1e-5f

/* torch-CPU arithmetic of the projection: separate IEEE mul/add in
 * k-order, true divisions.  Returns the visibility bit; writes u, v even when invisible. */
static int project_point(const float* m, float X, float Y, float Z, float img_h, float img_w,
                         float* u, float* v) {
  const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
  const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
  const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
  int vis = cz > ORACLE_EPS;
  const float zc = cz > ORACLE_EPS ? cz : ORACLE_EPS;
  *u = (cx / zc) / img_w;
  *v = (cy / zc) / img_h;
  return vis && *u > 0.f && *u < 1.f && *v > 0.f && *v < 1.f;
}

/* value (B*N, S, Hh, Dh) fp32.  Same argument meaning as gd4d_cross_attn_fwd (include/gd4d.h). */
int gd4d_oracle_cross_attn_fwd(const float* value, const int32_t* level_hw, const float* ref,
                               const float* offsets, const float* attn_logits,
                               const float* cam_logits, const float* lidar2img,
                               const double* pc_range, float img_h, float img_w, float* out,
                               uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh, int Dh,
                               int L, int P) {
  int lvl_start[16];
  int S = 0;
  if (L > 16) return -2;
  for (int l = 0; l < L; ++l) { lvl_start[l] = S; S += level_hw[2 * l] * level_hw[2 * l + 1]; }
  const int C = Hh * Dh, LP = L * P;
  float scale[3], lo[3];
  for (int k = 0; k < 3; ++k) { scale[k] = (float)(pc_range[k + 3] - pc_range[k]); lo[k] = (float)pc_range[k]; }
  float* w = (float*)malloc(sizeof(float) * LP);
  memset(out, 0, sizeof(float) * (size_t)B * Q * C);
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q) {
      const float* r = ref + ((size_t)b * Q + q) * 3;
      const float px = r[0] * scale[0] + lo[0], py = r[1] * scale[1] + lo[1], pz = r[2] * scale[2] + lo[2];
      float* o = out + ((size_t)b * Q + q) * C;
      for (int n = 0; n < N; ++n) {
        const int row = b * N + n;
        const int bb = row % B;                         /* query.repeat(N,1,1) row pairing, :277 */
        const float cl = cam_logits[(size_t)b * Q * N + (size_t)n * Q + q];   /* raw .view(), :211-212 */
        const float cw = 1.0f / (1.0f + expf(-cl));
        const float* m = lidar2img + (size_t)row * 16;
        for (int h = 0; h < Hh; ++h) {
          const float* lg = attn_logits + (((size_t)bb * Q + q) * Hh + h) * LP;
          float mx = lg[0], sum = 0.f;
          for (int i = 1; i < LP; ++i) mx = lg[i] > mx ? lg[i] : mx;
          for (int i = 0; i < LP; ++i) { w[i] = expf(lg[i] - mx); sum += w[i]; }
          for (int i = 0; i < LP; ++i) w[i] /= sum;
          for (int p = 0; p < P; ++p) {
            const float* of = offsets + ((((size_t)b * Q + q) * Hh + h) * P + p) * 3;
            float u, v;
            const int vis = project_point(m, px + of[0], py + of[1], pz + of[2], img_h, img_w, &u, &v);
            const size_t e = ((((size_t)b * N + n) * Q + q) * Hh + h) * P + p;
            if (mask_out) mask_out[e] = (uint8_t)vis;
            if (uv_out) { uv_out[2 * e] = u; uv_out[2 * e + 1] = v; }
            if (!vis) continue;
            for (int l = 0; l < L; ++l) {
              const int H = level_hw[2 * l], W = level_hw[2 * l + 1];
              const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
              const float xf = floorf(x), yf = floorf(y);
              const float dx = x - xf, dy = y - yf;
              const int x0 = (int)xf, y0 = (int)yf;
              const float wl = w[l * P + p] * cw;
              for (int cy2 = 0; cy2 < 2; ++cy2)
                for (int cx2 = 0; cx2 < 2; ++cx2) {
                  const int xx = x0 + cx2, yy = y0 + cy2;
                  if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;     /* zero padding */
                  const float bw = (cx2 ? dx : 1.f - dx) * (cy2 ? dy : 1.f - dy) * wl;
                  const float* src = value + (((size_t)row * S + lvl_start[l] + (size_t)yy * W + xx) * Hh + h) * Dh;
                  for (int d = 0; d < Dh; ++d) o[h * Dh + d] += bw * src[d];
                }
            }
          }
        }
      }
    }
  free(w);
  return 0;
}

/* DETR3D baseline (a9/a10).  feats: L pointers to (B*N, C, H_l, W_l) NCHW fp32; ref (B,Q,3);
 * attn_logits (B,Q,N,P=1,L); out (B,Q,C) = sum_{n,l} sigmoid(logit)*vis*bilinear; mask (B,N,Q);
 * uv_out (B,N,Q,2) holds the [-1,1]-mapped grid coordinates of detr3d_transformer.py:421;
 * sampled_out optional (B,C,Q,N,1,L). */
int gd4d_oracle_detr3d_fwd(const float* const* feats, const int32_t* level_hw, const float* ref,
                           const float* attn_logits, const float* lidar2img, const double* pc_range,
                           float img_h, float img_w, float* out, uint8_t* mask_out, float* uv_out,
                           float* sampled_out, int B, int N, int Q, int C, int L) {
  float scale[3], lo[3];
  for (int k = 0; k < 3; ++k) { scale[k] = (float)(pc_range[k + 3] - pc_range[k]); lo[k] = (float)pc_range[k]; }
  memset(out, 0, sizeof(float) * (size_t)B * Q * C);
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Q; ++q) {
      const float* r = ref + ((size_t)b * Q + q) * 3;
      const float X = r[0] * scale[0] + lo[0], Y = r[1] * scale[1] + lo[1], Z = r[2] * scale[2] + lo[2];
      for (int n = 0; n < N; ++n) {
        const float* m = lidar2img + ((size_t)b * N + n) * 16;
        const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
        const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
        const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
        int vis = cz > ORACLE_EPS;
        const float zc = cz > ORACLE_EPS ? cz : ORACLE_EPS;
        float u = (cx / zc) / img_w, v = (cy / zc) / img_h;
        u = (u - 0.5f) * 2.f;                                  /* :421 */
        v = (v - 0.5f) * 2.f;
        vis = vis && u > -1.f && u < 1.f && v > -1.f && v < 1.f;
        const size_t e = ((size_t)b * N + n) * Q + q;
        if (mask_out) mask_out[e] = (uint8_t)vis;
        if (uv_out) { uv_out[2 * e] = u; uv_out[2 * e + 1] = v; }
        for (int l = 0; l < L; ++l) {
          const int H = level_hw[2 * l], W = level_hw[2 * l + 1];
          /* ATen grid_sampler unnormalize, align_corners=False: ((g+1)*size-1)/2 */
          const float x = ((u + 1.f) * (float)W - 1.f) / 2.f, y = ((v + 1.f) * (float)H - 1.f) / 2.f;
          const float xf = floorf(x), yf = floorf(y);
          const float dx = x - xf, dy = y - yf;
          const float lg = attn_logits[(((size_t)b * Q + q) * N + n) * L + l];
          const float wl = vis ? 1.0f / (1.0f + expf(-lg)) : 0.f;
          const float* f = feats[l] + (size_t)(b * N + n) * C * H * W;
          for (int c = 0; c < C; ++c) {
            float s = 0.f;
            for (int cy2 = 0; cy2 < 2; ++cy2)
              for (int cx2 = 0; cx2 < 2; ++cx2) {
                const float xxf = xf + (float)cx2, yyf = yf + (float)cy2;
                if (!(xxf >= 0.f && xxf <= (float)(W - 1) && yyf >= 0.f && yyf <= (float)(H - 1))) continue;
                const float bw = (cx2 ? dx : 1.f - dx) * (cy2 ? dy : 1.f - dy);
                s += bw * f[((size_t)c * H + (int)yyf) * W + (int)xxf];
              }
            if (sampled_out) sampled_out[((((size_t)b * C + c) * Q + q) * N + n) * L + l] = s;
            out[((size_t)b * Q + q) * C + c] += wl * s;
          }
        }
      }
    }
  return 0;
}

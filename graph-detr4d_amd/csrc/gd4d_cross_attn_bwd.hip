// gd4d_cross_attn_bwd: backward of the fused project + sample + aggregate kernel, gfx950.
//
// In the reference the backward of this block is implicit autograd over deform3d_cross_attn.py:220-324:
// the third-party mmcv `ms_deformable_col2im` CUDA kernel (grad of value via atomicAdd, grad of
// sampling locations and attention weights) followed by ~30 elementwise backward launches through the
// softmax, the mask product, the sigmoid camera weights, the divisions and the lidar2img matmul.
// Here it is one kernel with the forward's work mapping (workgroup per query, waves split the visible
// cameras, 8-lane group per head, float4 per lane).
//
// With g = dL/d out (B,Q,256), for camera n, head h, point p, level l:
//   T  = sum_corner b_corner * <g_h, v_corner>              (<.,.> over the head's Dh channels)
//   dL/d value[n, pix_corner, h, :] += cw_n * a_hlp * b_corner * g_h               (atomic fp32 add)
//   dL/d a_hlp    += cw_n * T                     -> softmax backward over the L*P logits of head h
//   dL/d cam_n     = cw_n (1 - cw_n) * sum_{h,l,p} a_hlp * T
//   dL/d x, dL/d y = cw_n * a_hlp * sum_corner (d b_corner / dx, dy) <g_h, v_corner>      (pixel units)
//   -> dL/d u = sum_l W_l dL/dx_l ; dL/d v = sum_l H_l dL/dy_l ; then through u = cx / (cz Wimg),
//      v = cy / (cz Himg), c = M [X Y Z 1]^T to the metre-space point, i.e. to offsets[h,p] and ref.
// The visibility mask is piecewise constant (no gradient), exactly as in the reference.
// B > 1: the forward weights value row i = b*N + n with the logits of batch (i % B) (deform3d_cross_attn.py:277, see
// gd4d.h), so the gradient of attn_logits[bb] collects contributions from the workgroups of EVERY sample b.  The
// geometry owner (b, q) accumulates dL/d a per class bb = (b*N + n) % B in LDS and writes B partial rows into a
// workspace; a second small kernel sums them over b in a fixed order and applies the softmax backward.  B == 1 keeps
// everything in one kernel.  Supported: fp32 value, pixel-major layout.
#include "gd4d_common.h"

namespace gd4d {

struct CrossAttnBwdParams {
  const float* value;
  const float* ref;
  const float* offsets;
  const float* attn_logits;
  const float* cam_logits;
  const float* lidar2img;
  const float* grad_out;       // (B, Q, 256)
  float* grad_value;           // same shape as value, zero-initialised by the caller (atomic adds)
  float* grad_ref;             // (B, Q, 3)
  float* grad_offsets;         // (B, Q, Hh, P, 3)
  float* grad_attn_logits;     // (B, Q, Hh, L, P)
  float* grad_cam_logits;      // (B, Q, N), un-scrambled layout like cam_logits
  const int32_t* order;       // optional locality order of the queries (gd4d_query_order_fwd), as in the forward
  float* ga_part;             // B > 1: (B_geometry, B_class, Q, Hh, L*P) partial dL/d a (before the softmax backward)
  int B, N, Q, L, S;
  int raw_cam;                // GD4D_CA_RAW_CAM_WEIGHTS: the camera weights are the raw logits (no sigmoid), as in the forward
  int lvl_h[GD4D_MAX_LEVELS];
  int lvl_w[GD4D_MAX_LEVELS];
  int lvl_start[GD4D_MAX_LEVELS];
  float rng_scale[3];
  float rng_lo[3];
  float img_h, img_w;
};

constexpr int kBP = 4;        // points per head
constexpr int kBC = 256;      // channels

// sum over the `width` lanes of an aligned lane group (width = lanes per head)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int HH, int LT, int WAVES, bool BMULTI>
__global__ __launch_bounds__(GD4D_WAVE * WAVES) void cross_attn_bwd_block(const CrossAttnBwdParams p) {
  constexpr int DH = kBC / HH;
  constexpr int LPH = DH / 4;                     // lanes per head
  constexpr int E = HH * kBP;
  constexpr int L = LT;
  constexpr int LP = L * kBP;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // [WAVES][HH][LP + 12] partial sums (attention-weight grads, point grads) | [N] camera visibility
  float* s_part = reinterpret_cast<float*>(smem_raw);
  constexpr int PSTRIDE = LP + 3 * kBP;
  int* s_camvis = reinterpret_cast<int*>(s_part + WAVES * HH * PSTRIDE);
  float* s_pt = reinterpret_cast<float*>(s_camvis + 64);                   // [E][3] metre-space points
  float* s_gab = s_pt + E * 3;                                             // BMULTI: [WAVES][B][HH][LP] dL/d a per logit class

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // with a locality order every XCD (workgroup i -> XCD i % 8) takes a contiguous range of it: the atomic adds into
  // grad_value then meet in that XCD's L2 like the forward's reads do
  int bq = blockIdx.x;
  if (p.order) {
    const int per_xcd = (p.B * p.Q + 7) >> 3;
    const int pos = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || pos >= p.B * p.Q) return;
    bq = p.order[pos];
  }
  const int b = bq / p.Q;
  const int q = bq - b * p.Q;
  const int h = lane / LPH;
  const int sub = lane % LPH;                     // position inside the head's lane group
  if (BMULTI)
    for (int i = tid; i < WAVES * p.B * HH * LP; i += GD4D_WAVE * WAVES) s_gab[i] = 0.f;

  // metre-space sample points of this query (same arithmetic as the forward)
  {
    const float* rp = p.ref + (size_t)bq * 3;
    const float* offs = p.offsets + (size_t)bq * E * 3;
    for (int e = tid; e < E; e += GD4D_WAVE * WAVES) {
      s_pt[3 * e + 0] = (rp[0] * p.rng_scale[0] + p.rng_lo[0]) + offs[3 * e + 0];
      s_pt[3 * e + 1] = (rp[1] * p.rng_scale[1] + p.rng_lo[1]) + offs[3 * e + 1];
      s_pt[3 * e + 2] = (rp[2] * p.rng_scale[2] + p.rng_lo[2]) + offs[3 * e + 2];
    }
  }
  __syncthreads();

  // softmax weights of this lane's head (B > 1: of the logit class of each camera, recomputed per camera)
  float aw[LP];
  auto softmax_of = [&](int bb) {
    const float* lg = p.attn_logits + (((size_t)bb * p.Q + q) * HH + h) * LP;
    float mx = lg[0];
#pragma unroll
    for (int i = 1; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { aw[i] = expf(lg[i] - mx); sum += aw[i]; }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < LP; ++i) aw[i] *= inv;
  };
  if (!BMULTI) softmax_of(b);
  const float4 g = *reinterpret_cast<const float4*>(p.grad_out + (size_t)bq * kBC + lane * 4);
  // the same row again, lane-contiguous: channel 64 i + lane.  The atomic adds into grad_value use THIS mapping, so one
  // wave instruction covers whole 128-byte lines (64 consecutive channels of a pixel row) instead of 8 dwords in each
  // of 8 lines: a quarter of the L2 atomic requests.
  float gl4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gl4[i] = p.grad_out[(size_t)bq * kBC + 64 * i + lane];

  float ga[LP];                                    // dL/d a (before the softmax backward), this head
  float gpt[kBP][3];                               // dL/d (metre point) of this head's points
#pragma unroll
  for (int i = 0; i < LP; ++i) ga[i] = 0.f;
#pragma unroll
  for (int k = 0; k < kBP; ++k) { gpt[k][0] = gpt[k][1] = gpt[k][2] = 0.f; }

  for (int n = wave; n < p.N; n += WAVES) {        // cameras dealt to waves (all cameras: cheap skip below)
    const int row = b * p.N + n;
    const float* m = p.lidar2img + (size_t)row * 16;
    float u[kBP], v[kBP], cxv[kBP], cyv[kBP], czv[kBP];
    bool vis[kBP];
    bool any = false;
#pragma unroll
    for (int k = 0; k < kBP; ++k) {
      const float X = s_pt[3 * (h * kBP + k) + 0], Y = s_pt[3 * (h * kBP + k) + 1], Z = s_pt[3 * (h * kBP + k) + 2];
      const float eps = 1e-5f;
      cxv[k] = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
      cyv[k] = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
      czv[k] = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
      const float zc = fmaxf(czv[k], eps);
      u[k] = (cxv[k] / zc) / p.img_w;
      v[k] = (cyv[k] / zc) / p.img_h;
      vis[k] = czv[k] > eps && u[k] > 0.f && u[k] < 1.f && v[k] > 0.f && v[k] < 1.f;
      any |= vis[k];
    }
    if (!__any(any)) {                              // nobody in this wave sees camera n
      if (lane == 0) p.grad_cam_logits[(size_t)b * p.Q * p.N + (size_t)n * p.Q + q] = 0.f;
      continue;
    }
    const float cl = p.cam_logits[(size_t)b * p.Q * p.N + (size_t)n * p.Q + q];
    const float cw = p.raw_cam ? cl : 1.0f / (1.0f + expf(-cl));
    const int bb = BMULTI ? row % p.B : 0;           // logit class of this value row (:277)
    if (BMULTI) {
      softmax_of(bb);
#pragma unroll
      for (int i = 0; i < LP; ++i) ga[i] = 0.f;      // per camera; parked per class below
    }
    const float* vrow = p.value + (size_t)row * p.S * kBC;
    float* gvrow = p.grad_value + (size_t)row * p.S * kBC;
    float cam_acc = 0.f;                             // sum_{l,p} a * T of this head (group-uniform)

#pragma unroll
    for (int k = 0; k < kBP; ++k) {
      if (!__any(vis[k])) continue;                  // wave-uniform; heads that do not see the point run with ok = false
      const bool vk = vis[k];
      const float uk = vk ? u[k] : 0.5f, vvk = vk ? v[k] : 0.5f;
      float gu = 0.f, gv = 0.f;                      // dL/du, dL/dv (before cw), accumulated over levels
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int W = p.lvl_w[l], H = p.lvl_h[l];
        const float x = fmaf(uk, (float)W, -0.5f);
        const float y = fmaf(vvk, (float)H, -0.5f);
        const float xf = floorf(x), yf = floorf(y);
        const float dx = x - xf, dy = y - yf;
        const int x0 = (int)xf, y0 = (int)yf;
        const bool x0ok = x0 >= 0, x1ok = x0 + 1 < W, y0ok = y0 >= 0, y1ok = y0 + 1 < H;
        const float a = aw[l * kBP + k];
        const float wq = cw * a;                     // weight of this (camera, level, point) on out
        float d[4] = {0.f, 0.f, 0.f, 0.f};            // <g_h, v_corner>
        const bool ok[4] = {vk && x0ok && y0ok, vk && x1ok && y0ok, vk && x0ok && y1ok, vk && x1ok && y1ok};
        const float bw[4] = {(1.f - dx) * (1.f - dy), dx * (1.f - dy), (1.f - dx) * dy, dx * dy};
        // the four corners' value rows requested together (a load under `if (ok[c])`, each followed by the corner's atomics,
        // was one round trip per corner); corners outside the map read the level's first pixel and are dropped by the select
        int pixc[4];
        float4 valc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          pixc[c] = ok[c] ? p.lvl_start[l] + (y0 + (c >> 1)) * W + (x0 + (c & 1)) : -1;   // head-group uniform
          valc[c] = *reinterpret_cast<const float4*>(vrow + (size_t)max(pixc[c], p.lvl_start[l]) * kBC + lane * 4);
        }
        asm volatile("" : "+v"(valc[0].x), "+v"(valc[0].y), "+v"(valc[0].z), "+v"(valc[0].w), "+v"(valc[1].x), "+v"(valc[1].y),
                          "+v"(valc[1].z), "+v"(valc[1].w), "+v"(valc[2].x), "+v"(valc[2].y), "+v"(valc[2].z), "+v"(valc[2].w),
                          "+v"(valc[3].x), "+v"(valc[3].y), "+v"(valc[3].z), "+v"(valc[3].w));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int pix = pixc[c];
          const float4 val = valc[c];
          d[c] = ok[c] ? (g.x * val.x + g.y * val.y) + (g.z * val.z + g.w * val.w) : 0.f;
          const float s = wq * bw[c];
          // scatter: instruction i adds channels [64 i, 64 i + 64) of this corner's pixel rows; lane -> channel
          // 64 i + lane, whose head's pixel and weight come from that head's lane group
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int src = ((64 * i + lane) / DH) * LPH;
            const int hp = __shfl(pix, src);
            const float hs = __shfl(s, src);
            if (hp >= 0) atomicAdd(gvrow + (size_t)hp * kBC + 64 * i + lane, hs * gl4[i]);
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = group_sum<LPH>(d[c]);
        const float T = (bw[0] * d[0] + bw[1] * d[1]) + (bw[2] * d[2] + bw[3] * d[3]);
        ga[l * kBP + k] += cw * T;
        cam_acc += a * T;
        // d/dx, d/dy of the bilinear weights (zero-padded corners have d = 0)
        const float dTdx = (1.f - dy) * (d[1] - d[0]) + dy * (d[3] - d[2]);
        const float dTdy = (1.f - dx) * (d[2] - d[0]) + dx * (d[3] - d[1]);
        gu += a * dTdx * (float)W;
        gv += a * dTdy * (float)H;
      }
      gu *= cw; gv *= cw;
      // u = cx / (cz * Wimg), v = cy / (cz * Himg)   (cz > eps for visible points; gu = gv = 0 for the others)
      const float iz = vk ? 1.0f / czv[k] : 0.f;
      const float gcx = gu * iz / p.img_w;
      const float gcy = gv * iz / p.img_h;
      const float gcz = -(gu * cxv[k] * iz * iz / p.img_w + gv * cyv[k] * iz * iz / p.img_h);
      gpt[k][0] += gcx * m[0] + gcy * m[4] + gcz * m[8];
      gpt[k][1] += gcx * m[1] + gcy * m[5] + gcz * m[9];
      gpt[k][2] += gcx * m[2] + gcy * m[6] + gcz * m[10];
    }
    if (BMULTI && sub == 0) {                        // one lane per (wave, head): no race
      float* dst = s_gab + (((size_t)wave * p.B + bb) * HH + h) * LP;
#pragma unroll
      for (int i = 0; i < LP; ++i) dst[i] += ga[i];
    }
    // camera-logit gradient: sum over heads of (a * T), one lane per head then across the groups
    float cs = (sub == 0) ? cam_acc : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
    if (lane == 0) p.grad_cam_logits[(size_t)b * p.Q * p.N + (size_t)n * p.Q + q] = (p.raw_cam ? 1.f : cw * (1.f - cw)) * cs;
  }

  // ---- combine the waves' partial sums (fixed order), softmax backward, write ----
  if (sub == 0) {
    float* dst = s_part + ((size_t)wave * HH + h) * PSTRIDE;
#pragma unroll
    for (int i = 0; i < LP; ++i) dst[i] = ga[i];
#pragma unroll
    for (int k = 0; k < kBP; ++k) { dst[LP + 3 * k] = gpt[k][0]; dst[LP + 3 * k + 1] = gpt[k][1]; dst[LP + 3 * k + 2] = gpt[k][2]; }
  }
  __syncthreads();
  if (tid < HH) {
    const int hh = tid;
    float tot[PSTRIDE];
#pragma unroll
    for (int i = 0; i < PSTRIDE; ++i) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) t += s_part[((size_t)w * HH + hh) * PSTRIDE + i];
      tot[i] = t;
    }
    if (BMULTI) {                                    // partial dL/d a of every class -> workspace; softmax backward later
      for (int cls = 0; cls < p.B; ++cls) {
        float* dstp = p.ga_part + ((((size_t)b * p.B + cls) * p.Q + q) * HH + hh) * LP;
#pragma unroll
        for (int i = 0; i < LP; ++i) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) t += s_gab[(((size_t)w * p.B + cls) * HH + hh) * LP + i];
          dstp[i] = t;
        }
      }
    }
    // softmax backward: dlogit_i = a_i (ga_i - sum_j a_j ga_j)
    const float* lg = p.attn_logits + ((size_t)bq * HH + hh) * LP;
    float a2[LP];
    float mx = lg[0];
#pragma unroll
    for (int i = 1; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { a2[i] = expf(lg[i] - mx); sum += a2[i]; }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { a2[i] /= sum; dot += a2[i] * tot[i]; }
    if (!BMULTI) {
      float* gl = p.grad_attn_logits + ((size_t)bq * HH + hh) * LP;
#pragma unroll
      for (int i = 0; i < LP; ++i) gl[i] = a2[i] * (tot[i] - dot);
    }
    float* go = p.grad_offsets + ((size_t)bq * HH + hh) * kBP * 3;
#pragma unroll
    for (int i = 0; i < kBP * 3; ++i) go[i] = tot[LP + i];
    // re-park the per-head point gradient sums for the reference-point reduction
#pragma unroll
    for (int d3 = 0; d3 < 3; ++d3) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < kBP; ++k) t += tot[LP + 3 * k + d3];
      s_pt[3 * hh + d3] = t;
    }
  }
  __syncthreads();
  if (tid < 3) {
    float t = 0.f;
    for (int hh = 0; hh < HH; ++hh) t += s_pt[3 * hh + tid];
    p.grad_ref[(size_t)bq * 3 + tid] = t * p.rng_scale[tid];
  }
}

// B > 1: grad_attn_logits[bb, q, h, :] from the partial rows of all geometry owners b (fixed order) - softmax backward.
template <int LP>
__global__ __launch_bounds__(64) void cross_attn_bwd_logits_kernel(const CrossAttnBwdParams p, int HH) {
  const int bbq = blockIdx.x;                      // bb * Q + q
  const int bb = bbq / p.Q, q = bbq - bb * p.Q;
  const int hh = threadIdx.x;
  if (hh >= HH) return;
  float tot[LP];
#pragma unroll
  for (int i = 0; i < LP; ++i) tot[i] = 0.f;
  for (int b = 0; b < p.B; ++b) {
    const float* src = p.ga_part + ((((size_t)b * p.B + bb) * p.Q + q) * HH + hh) * LP;
#pragma unroll
    for (int i = 0; i < LP; ++i) tot[i] += src[i];
  }
  const float* lg = p.attn_logits + ((size_t)bbq * HH + hh) * LP;
  float a2[LP];
  float mx = lg[0];
#pragma unroll
  for (int i = 1; i < LP; ++i) mx = fmaxf(mx, lg[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) { a2[i] = expf(lg[i] - mx); sum += a2[i]; }
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) { a2[i] /= sum; dot += a2[i] * tot[i]; }
  float* gl = p.grad_attn_logits + ((size_t)bbq * HH + hh) * LP;
#pragma unroll
  for (int i = 0; i < LP; ++i) gl[i] = a2[i] * (tot[i] - dot);
}

template <int HH>
static int launch_bwd(const CrossAttnBwdParams& p, hipStream_t s) {
  constexpr int WAVES = 4;
  const dim3 grid(p.order ? ((p.B * p.Q + 7) / 8) * 8 : p.B * p.Q);
  const bool multi = p.B > 1;
  auto lds_for = [&](int L) {
    return (size_t)WAVES * HH * (L * kBP + 3 * kBP) * sizeof(float) + 64 * sizeof(int) + (size_t)HH * kBP * 3 * sizeof(float) +
           (multi ? (size_t)WAVES * p.B * HH * L * kBP * sizeof(float) : 0);
  };
#define GD4D_BWD_GO(L_)                                                                                                       \
  if (multi) {                                                                                                                \
    hipLaunchKernelGGL((cross_attn_bwd_block<HH, L_, WAVES, true>), grid, dim3(GD4D_WAVE * WAVES), lds_for(L_), s, p);        \
    if (int rc = check_launch()) return rc;                                                                                   \
    hipLaunchKernelGGL((cross_attn_bwd_logits_kernel<L_ * kBP>), dim3(p.B * p.Q), dim3(64), 0, s, p, HH);                     \
  } else {                                                                                                                    \
    hipLaunchKernelGGL((cross_attn_bwd_block<HH, L_, WAVES, false>), grid, dim3(GD4D_WAVE * WAVES), lds_for(L_), s, p);       \
  }
  switch (p.L) {
    case 1: GD4D_BWD_GO(1) break;
    case 2: GD4D_BWD_GO(2) break;
    case 3: GD4D_BWD_GO(3) break;
    case 4: GD4D_BWD_GO(4) break;
    default: return GD4D_EUNSUPPORTED;
  }
#undef GD4D_BWD_GO
  return check_launch();
}

}  // namespace gd4d

extern "C" size_t gd4d_cross_attn_bwd_workspace_bytes(int B, int Q, int Hh, int L, int P) {
  if (B <= 1 || Q <= 0 || Hh <= 0 || L <= 0 || P <= 0) return 0;
  return (size_t)B * B * Q * Hh * L * P * sizeof(float);
}

extern "C" int gd4d_cross_attn_bwd(const void* value, const int32_t* level_hw, const float* ref,
                                   const float* offsets, const float* attn_logits, const float* cam_logits,
                                   const float* lidar2img, const double* pc_range, float img_h, float img_w,
                                   const float* grad_out, void* grad_value, float* grad_ref,
                                   float* grad_offsets, float* grad_attn_logits, float* grad_cam_logits,
                                   int B, int N, int Q, int Hh, int Dh, int L, int P, int value_dtype,
                                   int value_layout, int flags, const int32_t* query_order, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  using namespace gd4d;
  if (!value || !level_hw || !ref || !offsets || !attn_logits || !cam_logits || !lidar2img || !pc_range ||
      !grad_out || !grad_value || !grad_ref || !grad_offsets || !grad_attn_logits || !grad_cam_logits)
    return GD4D_EINVAL;
  if (B <= 0 || N <= 0 || Q <= 0 || Hh <= 0 || Dh <= 0 || L <= 0 || !(img_h > 0.f) || !(img_w > 0.f)) return GD4D_EINVAL;
  if (B > 8 || Hh * Dh != kBC || P != kBP || L > 4 || N > 64 || value_dtype != GD4D_F32 ||
      value_layout != GD4D_LAYOUT_PIXEL_MAJOR)
    return GD4D_EUNSUPPORTED;
  if (B > 1 && (!workspace || workspace_bytes < gd4d_cross_attn_bwd_workspace_bytes(B, Q, Hh, L, P))) return GD4D_EWORKSPACE;
  if (!aligned16(value) || !aligned16(grad_out) || !aligned16(grad_value)) return GD4D_EALIGN;
  CrossAttnBwdParams p{};
  p.value = static_cast<const float*>(value); p.ref = ref; p.offsets = offsets; p.attn_logits = attn_logits;
  p.cam_logits = cam_logits; p.lidar2img = lidar2img; p.grad_out = grad_out;
  p.grad_value = static_cast<float*>(grad_value); p.grad_ref = grad_ref; p.grad_offsets = grad_offsets;
  p.grad_attn_logits = grad_attn_logits; p.grad_cam_logits = grad_cam_logits; p.order = query_order;
  p.ga_part = static_cast<float*>(workspace);
  p.B = B; p.N = N; p.Q = Q; p.L = L;
  p.raw_cam = (flags & GD4D_CA_RAW_CAM_WEIGHTS) ? 1 : 0;
  int start = 0;
  for (int l = 0; l < L; ++l) {
    const int h = level_hw[2 * l], w = level_hw[2 * l + 1];
    if (h <= 0 || w <= 0) return GD4D_EINVAL;
    p.lvl_h[l] = h; p.lvl_w[l] = w; p.lvl_start[l] = start;
    start += h * w;
  }
  p.S = start;
  for (int k = 0; k < 3; ++k) {
    p.rng_scale[k] = static_cast<float>(pc_range[k + 3] - pc_range[k]);
    p.rng_lo[k] = static_cast<float>(pc_range[k]);
  }
  p.img_h = img_h; p.img_w = img_w;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (Hh) {
    case 4: return launch_bwd<4>(p, s);
    case 8: return launch_bwd<8>(p, s);
    case 16: return launch_bwd<16>(p, s);
    default: return GD4D_EUNSUPPORTED;
  }
}

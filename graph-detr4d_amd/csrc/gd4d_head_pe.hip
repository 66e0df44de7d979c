// Feature position embedding of Detr3DHeadPE - the step that rewrites the feature maps the fused sample-aggregate
// kernel gathers from.  Reference: projects/mmdet3d_plugin/models/dense_heads/detr3d_head_pe.py
// :427-491 (position_embeding), :525-557 (forward), models/utils/positional_encoding.py:58-100.
//
// The dense 1x1 convolutions of that stage are plain GEMMs and stay with the library; what is here are the three
// bandwidth-bound pieces the reference builds out of dozens of elementwise ops and multi-GB temporaries:
//   gd4d_frustum_pe_input_fwd  pixel x depth-bin frustum -> lidar frame -> pc_range units -> inverse_sigmoid, written
//                              straight in the (B*N, 3*D, H, W) layout position_encoder's first conv reads, plus
//                              the "most depth bins outside the range" flag.  The reference materialises a
//                              (B,N,W,H,D,4,4) repeated-matrix tensor (2.3 GB at level 0, N = 24) on the way.
//   gd4d_sine_pe3d_fwd         (camera, row, column) cumulative-sum embeddings -> 3 x 128 sin / cos channels.
//   gd4d_se_fuse_fwd           out = feat + (pe * sigmoid(gate) + sine_embed): SELayer's gating and the two adds.
#include "gd4d_common.h"

namespace gd4d {

struct FrustumParams {
  const float* img2lidar;   // (R, 16)
  float* out;               // (R, 3*D, H, W)
  uint8_t* outside;         // (R, H, W)
  int R, H, W, D;
  int chlast, S, start;     // chlast: out is (R, S, 3*D) rows, this level occupying pixels [start, start + H*W) of each row
  float pad_h, pad_w, depth_start, bin_size;
  float lo[3], span[3];
};

__global__ __launch_bounds__(256) void frustum_pe_input_kernel(const FrustumParams p) {
  const int hw = p.H * p.W;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.R * hw) return;
  const int r = (int)(idx / hw);
  const int pix = (int)(idx - (long long)r * hw);
  const int y = pix / p.W, x = pix - y * p.W;
  const float* m = p.img2lidar + (size_t)r * 16;
  // torch.arange(H).float() * pad_h / H  (:440-441): multiply, then divide, in fp32
  const float ch = ((float)y * p.pad_h) / (float)p.H;
  const float cw = ((float)x * p.pad_w) / (float)p.W;
  float* o = p.out + (size_t)r * 3 * p.D * hw + pix;
  const size_t cstride = (size_t)hw;
  const float eps = 1e-5f;
  int n_out = 0;
  for (int d = 0; d < p.D; ++d) {
    const float fi = (float)d;
    const float depth = p.depth_start + (p.bin_size * fi) * (fi + 1.0f);      // :450-453
    const float s = fmaxf(depth, eps);
    const float px = cw * s, py = ch * s;                                     // :458
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v = ((m[4 * k] * px + m[4 * k + 1] * py) + m[4 * k + 2] * depth) + m[4 * k + 3];   // :469
      c[k] = (v - p.lo[k]) / p.span[k];                                       // :470-475
      n_out += (c[k] > 1.0f || c[k] < 0.0f) ? 1 : 0;                          // :477
      o[(size_t)(3 * d + k) * cstride] = inv_sigmoid(c[k]);                   // :480-481 layout, channel = 3 d + axis
    }
  }
  p.outside[idx] = (float)n_out > (float)p.D * 0.5f ? 1 : 0;                  // :478
}

// Channels-last form: one wave per pixel, lane = depth bin, so a wave writes its pixel's 3*D consecutive floats as one
// contiguous run (the per-pixel form above would scatter 4-byte stores 3*D floats apart).
//
// Round 6: the kernel was bound by its ARITHMETIC, not by its 568 MB of stores (0.33 ms = 142 M elements x ~90 instructions: an
// IEEE division and libm's logf per element inside inv_sigmoid).  The projection and the normalisation keep the reference's fp32
// operation sequence (c, which decides the clamps, is bit-identical); inv_sigmoid of the clamped value is taken on the
// transcendental unit - log(a / b) = (log2 a - log2 b) ln 2, v_log_f32 at ~1 ulp, |error| <~ 2e-6 on a value of up to 11.5 that
// enters a 192 -> 1024 -> 256 MLP (the module's 2e-4 / fp64 tests bound the result) - and the 3 D floats of a pixel leave as 16-byte
// stores (through a 768-byte LDS patch per wave) instead of three 4-byte stores per lane.
__device__ __forceinline__ float inv_sigmoid_fast(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float a = fminf(fmaxf(x, 1e-5f), 1.f), b = fminf(fmaxf(1.f - x, 1e-5f), 1.f);
  return (__builtin_amdgcn_logf(a) - __builtin_amdgcn_logf(b)) * 0.69314718055994530942f;
}

constexpr int FR_PIX = 1;       // pixels per wave (16 per wave measured 65 against 58 us per launch: the kernel is not bound by the rate waves are launched at)

__global__ __launch_bounds__(256) void frustum_pe_chlast_kernel(const FrustumParams p) {
  __shared__ __attribute__((aligned(16))) float s_row[4][3 * 64 + 4];
  const int hw = p.H * p.W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long total = (long long)p.R * hw;
  const long long first = ((long long)blockIdx.x * 4 + wave) * FR_PIX;
  const float eps = 1e-5f;
  const bool fast_store = p.D == 64;                           // (the head's depth_num; any other: the plain stores)
  for (int it = 0; it < FR_PIX; ++it) {
    const long long idx = first + it;
    if (idx >= total) return;                                  // (wave-uniform; no workgroup barrier below)
    const int r = (int)(idx / hw);
    const int pix = (int)(idx - (long long)r * hw);
    const int y = pix / p.W, x = pix - y * p.W;
    const float* m = p.img2lidar + (size_t)r * 16;
    const float ch = ((float)y * p.pad_h) / (float)p.H;
    const float cw = ((float)x * p.pad_w) / (float)p.W;
    float* o = p.out + ((size_t)r * p.S + p.start + pix) * 3 * p.D;
    int n_out = 0;
    if (fast_store && it > 0) __builtin_amdgcn_wave_barrier();   // the previous pixel's patch has been read
    for (int d = lane; d < p.D; d += 64) {
      const float fi = (float)d;
      const float depth = p.depth_start + (p.bin_size * fi) * (fi + 1.0f);
      const float s = fmaxf(depth, eps);
      const float px = cw * s, py = ch * s;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float v = ((m[4 * k] * px + m[4 * k + 1] * py) + m[4 * k + 2] * depth) + m[4 * k + 3];
        const float c = (v - p.lo[k]) / p.span[k];
        n_out += (c > 1.0f || c < 0.0f) ? 1 : 0;
        const float e = inv_sigmoid_fast(c);
        if (fast_store) s_row[wave][3 * d + k] = e;
        else o[3 * d + k] = e;
      }
    }
    if (fast_store) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // wave-private patch
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < 48) *reinterpret_cast<float4*>(o + 4 * lane) = *reinterpret_cast<const float4*>(&s_row[wave][4 * lane]);
    }
    if (p.outside) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) n_out += __shfl_xor(n_out, off);
      if (lane == 0) p.outside[idx] = (float)n_out > (float)p.D * 0.5f ? 1 : 0;
    }
  }
}

struct SineParams {
  const float* embed[3];    // n, y, x embeddings, each (R, H, W) (already normalised)
  const float* dim_t;       // (F)
  float* out;               // (R, 3*F, H, W), or channels-last (R, S, 3*F) when S > 0 (this level at pixels [start, start + HW))
  int R, HW, F, S, start;
};

__global__ __launch_bounds__(256) void sine_pe3d_kernel(const SineParams p) {
  // one thread per (r, channel, pixel); the fastest index is the one that is contiguous in the output layout
  const long long total = (long long)p.R * 3 * p.F * p.HW;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int pix, ch, r;
  if (p.S > 0) {                                    // channels-last: channel fastest
    ch = (int)(idx % (3 * p.F));
    const long long t = idx / (3 * p.F);
    pix = (int)(t % p.HW);
    r = (int)(t / p.HW);
  } else {
    pix = (int)(idx % p.HW);
    const long long t = idx / p.HW;
    ch = (int)(t % (3 * p.F));
    r = (int)(t / (3 * p.F));
  }
  const int part = ch / p.F, f = ch - part * p.F;
  const float* e = part == 0 ? p.embed[0] : part == 1 ? p.embed[1] : p.embed[2];
  // positional_encoding.py:90-98 stacks (pos[..., 0::2].sin(), pos[..., 1::2].cos()) on dim 4 of a 5-D tensor, i.e.
  // BEFORE the feature axis: the first F/2 channels are the sines of the even features, the last F/2 the cosines of
  // the odd ones (not DETR's interleaving).
  const int half = p.F / 2;
  const bool is_cos = f >= half;
  const int src = is_cos ? 2 * (f - half) + 1 : 2 * f;
  const float v = e[(size_t)r * p.HW + pix] / p.dim_t[src];
  const float o = is_cos ? cosf(v) : sinf(v);
  if (p.S > 0) p.out[((size_t)r * p.S + p.start + pix) * 3 * p.F + ch] = o;
  else p.out[idx] = o;
}

__global__ __launch_bounds__(256) void se_fuse_kernel(const float4* __restrict__ feat, const float4* __restrict__ gate,
                                                      const float4* __restrict__ pe, const float4* __restrict__ sine,
                                                      float4* __restrict__ out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 f = feat[i], g = gate[i], q = pe[i], s = sine[i];
  float4 o;
  o.x = f.x + (q.x * (1.0f / (1.0f + expf(-g.x))) + s.x);
  o.y = f.y + (q.y * (1.0f / (1.0f + expf(-g.y))) + s.y);
  o.z = f.z + (q.z * (1.0f / (1.0f + expf(-g.z))) + s.z);
  o.w = f.w + (q.w * (1.0f / (1.0f + expf(-g.w))) + s.w);
  out[i] = o;
}

// out[r, c, pix] = feat[r, c, pix] + (pe[r, start + pix, c] * sigmoid(gate[r, start + pix, c]) + sine[r, c, pix]):
// the channels-last products of the GEMM path meet the NCHW maps; 32 x 32 (channel, pixel) tiles through LDS so that
// both sides are read and written coalesced.  OUT_CHLAST: the result is stored (R, H, W, C) - the layout the decoder's
// gathers read in place (no per-sample copy there): then it is `feat` that goes through the tile, the sum is the same
// expression, so both layouts hold the same bits.
template <bool OUT_CHLAST>
__global__ __launch_bounds__(256) void se_fuse_chlast_kernel(const float* __restrict__ feat, const float* __restrict__ gate,
                                                             const float* __restrict__ pe, const float* __restrict__ sine,
                                                             float* __restrict__ out, int C, int HW, int S, int start,
                                                             int sine_chlast) {
  __shared__ float tile[32][33];
  const int r = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
  if (OUT_CHLAST) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                     // read NCHW: tx = pixel, rows = channels
      const int c = c0 + ty + 8 * i, pix = p0 + tx;
      tile[ty + 8 * i][tx] = pix < HW ? feat[((size_t)r * C + c) * HW + pix] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                     // channels-last: tx = channel, rows = pixels
      const int pix = p0 + ty + 8 * i, c = c0 + tx;
      if (pix < HW) {
        const size_t o = ((size_t)r * S + start + pix) * C + c;
        const float v = pe[o] * (1.0f / (1.0f + expf(-gate[o]))) + sine[o];
        out[((size_t)r * HW + pix) * C + c] = tile[tx][ty + 8 * i] + v;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                       // read channels-last: tx = channel, rows = pixels
    const int pix = p0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (pix < HW) {
      const size_t o = ((size_t)r * S + start + pix) * C + c;
      v = pe[o] * (1.0f / (1.0f + expf(-gate[o])));
      if (sine_chlast) v += sine[o];
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                       // write NCHW: tx = pixel, rows = channels
    const int c = c0 + ty + 8 * i, pix = p0 + tx;
    if (pix < HW) {
      const size_t o = ((size_t)r * C + c) * HW + pix;
      out[o] = feat[o] + (sine_chlast ? tile[tx][ty + 8 * i] : tile[tx][ty + 8 * i] + sine[o]);
    }
  }
}

// Backward of se_fuse_chlast_kernel for one level.  grad_out arrives NCHW; everything downstream (the weight-gradient
// contractions over pixels) wants channels-last rows, so this is the transposing pass in the other direction:
//   grad_sine[o] = g,   grad_pe[o] = g sigmoid(gate[o]),   grad_gate[o] = g pe[o] sigmoid'(gate[o])     (o = channels-last)
// grad_pe / grad_gate may be the pe / gate buffers themselves (each element is read, then written, by one thread).
__global__ __launch_bounds__(256) void se_fuse_chlast_bwd_kernel(const float* __restrict__ grad_out, const float* gate,
                                                                 const float* pe, float* grad_gate, float* grad_pe,
                                                                 float* __restrict__ grad_sine, int C, int HW, int S,
                                                                 int start) {
  __shared__ float tile[32][33];
  const int r = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                       // read NCHW: tx = pixel, rows = channels
    const int c = c0 + ty + 8 * i, pix = p0 + tx;
    tile[ty + 8 * i][tx] = pix < HW ? grad_out[((size_t)r * C + c) * HW + pix] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                       // write channels-last: tx = channel, rows = pixels
    const int pix = p0 + ty + 8 * i, c = c0 + tx;
    if (pix < HW) {
      const size_t o = ((size_t)r * S + start + pix) * C + c;
      const float g = tile[tx][ty + 8 * i];
      const float s = 1.0f / (1.0f + expf(-gate[o]));
      const float pv = pe[o];
      grad_sine[o] = g;
      grad_pe[o] = g * s;
      grad_gate[o] = g * pv * (s * (1.0f - s));
    }
  }
}

}  // namespace gd4d

extern "C" int gd4d_frustum_pe_input_fwd(const float* img2lidar, float* out, uint8_t* outside, int R, int H, int W, int D,
                                         float pad_h, float pad_w, float depth_start, const double* pc_range,
                                         int row_pixels, int row_start, void* stream) {
  using namespace gd4d;
  if (!img2lidar || !out || !outside || !pc_range || R <= 0 || H <= 0 || W <= 0 || D <= 0) return GD4D_EINVAL;
  if (row_pixels < 0 || (row_pixels > 0 && (row_start < 0 || row_start + H * W > row_pixels))) return GD4D_EINVAL;
  FrustumParams p{};
  p.img2lidar = img2lidar; p.out = out; p.outside = outside; p.R = R; p.H = H; p.W = W; p.D = D;
  p.chlast = row_pixels > 0 ? 1 : 0; p.S = row_pixels; p.start = row_start;
  p.pad_h = pad_h; p.pad_w = pad_w; p.depth_start = depth_start;
  // Python-float arithmetic of the reference (:452), rounded to fp32 when it meets the tensor
  p.bin_size = (float)((pc_range[3] - (double)depth_start) / ((double)D * (1.0 + (double)D)));
  for (int k = 0; k < 3; ++k) { p.lo[k] = (float)pc_range[k]; p.span[k] = (float)(pc_range[k + 3] - pc_range[k]); }
  const long long total = (long long)R * H * W;
  if (p.chlast)
    hipLaunchKernelGGL(frustum_pe_chlast_kernel, dim3((unsigned)((total + 4 * FR_PIX - 1) / (4 * FR_PIX))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(frustum_pe_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_sine_pe3d_fwd(const float* n_embed, const float* y_embed, const float* x_embed, const float* dim_t,
                                  float* out, int R, int HW, int F, int row_pixels, int row_start, void* stream) {
  using namespace gd4d;
  if (!n_embed || !y_embed || !x_embed || !dim_t || !out || R <= 0 || HW <= 0 || F <= 0) return GD4D_EINVAL;
  if (row_pixels < 0 || (row_pixels > 0 && (row_start < 0 || row_start + HW > row_pixels))) return GD4D_EINVAL;
  SineParams p{};
  p.S = row_pixels; p.start = row_start;
  p.embed[0] = n_embed; p.embed[1] = y_embed; p.embed[2] = x_embed; p.dim_t = dim_t; p.out = out;
  p.R = R; p.HW = HW; p.F = F;
  const long long total = (long long)R * 3 * F * HW;
  hipLaunchKernelGGL(sine_pe3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch();
}

extern "C" int gd4d_se_fuse_fwd(const float* feat, const float* gate, const float* pe, const float* sine, float* out,
                                size_t n, void* stream) {
  using namespace gd4d;
  if (!feat || !gate || !pe || !sine || !out || n == 0) return GD4D_EINVAL;
  if (n % 4 != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(feat) || !aligned16(gate) || !aligned16(pe) || !aligned16(sine) || !aligned16(out)) return GD4D_EALIGN;
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(se_fuse_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(feat), reinterpret_cast<const float4*>(gate),
                     reinterpret_cast<const float4*>(pe), reinterpret_cast<const float4*>(sine),
                     reinterpret_cast<float4*>(out), n4);
  return check_launch();
}

extern "C" int gd4d_se_fuse_chlast_fwd(const float* feat, const float* gate, const float* pe, const float* sine, float* out,
                                       int R, int C, int HW, int row_pixels, int row_start, int sine_chlast, int out_chlast,
                                       void* stream) {
  using namespace gd4d;
  if (!feat || !gate || !pe || !sine || !out || R <= 0 || C <= 0 || HW <= 0) return GD4D_EINVAL;
  if (row_start < 0 || row_start + HW > row_pixels) return GD4D_EINVAL;
  if (C % 32 != 0 || R > 65535 || (out_chlast && !sine_chlast)) return GD4D_EUNSUPPORTED;
  const dim3 grid((HW + 31) / 32, C / 32, R);
  if (out_chlast)
    hipLaunchKernelGGL(se_fuse_chlast_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), feat, gate, pe, sine, out,
                       C, HW, row_pixels, row_start, 1);
  else
    hipLaunchKernelGGL(se_fuse_chlast_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), feat, gate, pe, sine, out,
                       C, HW, row_pixels, row_start, sine_chlast ? 1 : 0);
  return check_launch();
}

extern "C" int gd4d_se_fuse_chlast_bwd(const float* grad_out, const float* gate, const float* pe, float* grad_gate,
                                       float* grad_pe, float* grad_sine, int R, int C, int HW, int row_pixels, int row_start,
                                       void* stream) {
  using namespace gd4d;
  if (!grad_out || !gate || !pe || !grad_gate || !grad_pe || !grad_sine || R <= 0 || C <= 0 || HW <= 0) return GD4D_EINVAL;
  if (row_start < 0 || row_start + HW > row_pixels) return GD4D_EINVAL;
  if (C % 32 != 0 || R > 65535) return GD4D_EUNSUPPORTED;
  hipLaunchKernelGGL(se_fuse_chlast_bwd_kernel, dim3((HW + 31) / 32, C / 32, R), dim3(256), 0,
                     static_cast<hipStream_t>(stream), grad_out, gate, pe, grad_gate, grad_pe, grad_sine, C, HW, row_pixels,
                     row_start);
  return check_launch();
}

// Backward of gd4d_value_proj_fwd (training): the two 739 800 x 256 x 256 contractions per decoder layer that a library
// fp32 GEMM runs at 0.9-1.5 ms each (dW has M = N = 256 and K = 739 800, a shape rocBLAS has no good tile for, and dX
// needs a transposed copy back to NCHW).  Both run on the bf16 MFMA with split operands like the forward
// (x = hi + lo in bf16; hi*hi + hi*lo + lo*hi accumulated in fp32: fp32-class, ~2^-17 relative per product) and read /
// write the tensors in the layouts the callers hold: grad_out (R, S, C) channels-last as gd4d_cross_attn_bwd leaves it,
// the pyramid and its gradient NCHW per level.
//
//   gd4d_value_proj_bwd_input   gin_l[r, ci, pix] (+)= sum_co gout[r, start_l + pix, co] * W[co, ci]
//   gd4d_value_proj_bwd_weight  gw[co, ci] = sum_{r, l, pix} gout[r, start_l + pix, co] * x_l[r, ci, pix];  gb[co] = sum gout
//
// Reference: autograd of Deform3DCrossAttn.forward's `self.value_proj(value_flatten)` with the flatten / transpose /
// cat in front of it (deform3d_cross_attn.py:264-280).
#include <stdlib.h>

#include "gd4d_common.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int VB_C = 256;

struct VpBwdParams {
  const float* gout;                    // (R, S, C)
  const float* weight;                  // (C, C) [co][ci]
  float* gin[GD4D_MAX_LEVELS];          // level l: (R, C, HW_l)                 (input kernel)
  const float* x[GD4D_MAX_LEVELS];      // level l: (R, C, HW_l)                 (weight kernel)
  float* ws;                            // [workgroups][C*C + C] partial sums    (weight kernel)
  int hw[GD4D_MAX_LEVELS];
  int start[GD4D_MAX_LEVELS];
  int tiles[GD4D_MAX_LEVELS];
  int tile_base[GD4D_MAX_LEVELS + 1];
  int R, L, S;
  int dbg;                              // dev ablation bits (GD4D_VW_DBG): 1 = no MFMAs, 2 = no global loads, 4 = no conversion / LDS writes, 8 = no pyramid loads, 16 = no grad_out loads
};

__device__ __forceinline__ unsigned vb_cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}

// 8 floats -> 16 bytes of bf16 "hi" halves and 16 bytes of bf16 residuals
__device__ __forceinline__ void vb_split8(const float* v, u32x4& h, u32x4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = vb_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    const float ra = v[2 * i] - __uint_as_float(hh[i] << 16);            // exact: hi is a rounding of the value
    const float rb = v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u);
    ll[i] = vb_cvt_pk_bf16(ra, rb);
  }
  h = u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}

__device__ __forceinline__ bf16x8 vb_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// byte offset of 16-byte chunk `chunk` (8 bf16) of row `row` in a [rows][256] bf16 LDS image (XOR swizzle: the
// b128 writes of 32 consecutive chunks and the b128 fragment reads of 32 consecutive rows are both conflict free)
__device__ __forceinline__ int vb_lds_off(int row, int chunk) {
  return row * (VB_C * 2) + ((chunk ^ (row & 15)) << 4);
}

#define VB_SGPR(x) __builtin_amdgcn_readfirstlane(x)

// Workgroup-uniform cursor over the (level, camera row, tile in row) sequence; level parameters live in an LDS table
// (dynamic indexing of the by-value kernarg struct would be copied to scratch) and are re-read at level boundaries only.
struct VbCursor {
  int lvl, row, tin, hw, tiles, start;
  uintptr_t ptr;                                    // p.gin[lvl] or p.x[lvl]
};

__device__ __forceinline__ void vb_fill_table(int* tab, const VpBwdParams& p, bool use_x) {
#pragma unroll
  for (int l = 0; l < GD4D_MAX_LEVELS; ++l) {
    const uintptr_t a = use_x ? reinterpret_cast<uintptr_t>(p.x[l]) : reinterpret_cast<uintptr_t>(p.gin[l]);
    tab[6 * l + 0] = p.hw[l]; tab[6 * l + 1] = p.tiles[l]; tab[6 * l + 2] = p.start[l];
    tab[6 * l + 3] = p.tile_base[l]; tab[6 * l + 4] = (int)(unsigned)(a & 0xffffffffu); tab[6 * l + 5] = (int)(unsigned)(a >> 32);
  }
}

__device__ __forceinline__ void vb_set_level(VbCursor& c, const int* tab, int lvl) {
  c.lvl = lvl;
  c.hw = VB_SGPR(tab[6 * lvl + 0]); c.tiles = VB_SGPR(tab[6 * lvl + 1]); c.start = VB_SGPR(tab[6 * lvl + 2]);
  const unsigned lo = (unsigned)VB_SGPR(tab[6 * lvl + 4]), hi = (unsigned)VB_SGPR(tab[6 * lvl + 5]);
  c.ptr = ((uintptr_t)hi << 32) | lo;
}

__device__ __forceinline__ void vb_seek(VbCursor& c, const int* tab, const VpBwdParams& p, int t) {
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
    if (l < p.L && t >= p.tile_base[l]) lvl = l;
  vb_set_level(c, tab, VB_SGPR(lvl));
  const int rel = t - VB_SGPR(tab[6 * c.lvl + 3]);
  c.row = VB_SGPR(rel / c.tiles);
  c.tin = VB_SGPR(rel - c.row * c.tiles);
}

__device__ __forceinline__ void vb_advance(VbCursor& c, const int* tab, const VpBwdParams& p, int step) {
  c.tin += step;
  while (c.tin >= c.tiles && c.lvl < p.L) {
    c.tin -= c.tiles;
    if (++c.row == p.R) {
      c.row = 0;
      if (c.lvl + 1 < p.L) vb_set_level(c, tab, c.lvl + 1); else c.lvl = p.L;
    }
  }
  c.tin = VB_SGPR(c.tin); c.row = VB_SGPR(c.row); c.lvl = VB_SGPR(c.lvl);
}

// ---------------------------------------------------------------------------------------------------------------------
// d(pyramid).  One persistent 512-thread workgroup per CU; wave w owns input channels [32w, 32w + 32) and keeps its
// W^T fragments (hi, lo; all K = 256 output channels) in registers.  Per tile of 64 pixels: the (64, 256) fp32 block of
// grad_out - one contiguous 64 KB run - is loaded with 32 B per lane, split to bf16 hi / lo and parked in LDS as
// [pix][co]; the MFMA computes D[ci][pix] = W^T gout^T so that a lane's accumulator column is a pixel and the stores
// are pixel-contiguous 128 B runs of the NCHW gradient.  ACCUM: the accumulators start from the tensor's current values
// (the gradient of the shared pyramid is the sum over the decoder layers).
template <bool ACCUM>
__global__ __launch_bounds__(512, 2) void value_proj_bwd_input_kernel(const VpBwdParams p) {
  constexpr int BM = 64, SUB = BM / 32, KSTEPS = VB_C / 16;
  constexpr int IMG = BM * VB_C * 2;                              // one bf16 [64][256] image: 32 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 buffers][hi, lo] + level table
  int* const tab = reinterpret_cast<int*>(smem + 4 * IMG);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = VB_SGPR(tid >> 6);
  const int col = lane & 31, kg = lane >> 5;

  // A fragments: m = ci = 32 wave + col, k = co = 16 s + 8 kg + j
  bf16x8 whi[KSTEPS], wlo[KSTEPS];
  {
    const float* wcol = p.weight + (size_t)(8 * kg) * VB_C + 32 * wave + col;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = wcol[(size_t)(16 * s + j) * VB_C];
      u32x4 h, l;
      vb_split8(v, h, l);
      whi[s] = vb_frag(h);
      wlo[s] = vb_frag(l);
    }
  }

  const int total = p.tile_base[GD4D_MAX_LEVELS];
  const int slot = blockIdx.x, slots = gridDim.x;
  if (slot >= total) return;
  if (tid == 0) vb_fill_table(tab, p, false);
  __syncthreads();

  VbCursor cur, nxt;                 // tile being multiplied / tile being loaded
  vb_seek(cur, tab, p, slot);
  nxt = cur;

  // staging: chunk q = tid + 512 i of the tile's 2048 16-byte-of-bf16 chunks; pixel q / 32, channels 8 (q % 32) .. +8
  float4 stage[4][2];
  auto issue_loads = [&](const VbCursor& c) {
    const int pix0 = c.tin * BM;
    const float* src = p.gout + ((size_t)c.row * p.S + c.start + pix0) * VB_C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 512 * i;
      const int pix = min(pix0 + (q >> 5), c.hw - 1) - pix0;      // tail rows re-read the last valid row (zeroed in park)
      const float4* g = reinterpret_cast<const float4*>(src + (size_t)pix * VB_C + 8 * (q & 31));
      stage[i][0] = g[0];
      stage[i][1] = g[1];
    }
  };
  auto park = [&](const VbCursor& c, int buf) {
    char* hi_img = smem + buf * 2 * IMG;
    char* lo_img = hi_img + IMG;
    const int rem = c.hw - c.tin * BM;                            // valid pixel rows in this tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 512 * i;
      const bool in = (q >> 5) < rem;
      const float v[8] = {in ? stage[i][0].x : 0.f, in ? stage[i][0].y : 0.f, in ? stage[i][0].z : 0.f,
                          in ? stage[i][0].w : 0.f, in ? stage[i][1].x : 0.f, in ? stage[i][1].y : 0.f,
                          in ? stage[i][1].z : 0.f, in ? stage[i][1].w : 0.f};
      u32x4 h, l;
      vb_split8(v, h, l);
      const int off = vb_lds_off(q >> 5, q & 31);
      *reinterpret_cast<u32x4*>(hi_img + off) = h;
      *reinterpret_cast<u32x4*>(lo_img + off) = l;
    }
  };

  f32x16 acc[SUB];
  // accumulator r of sub-tile m <-> gin[row][32 wave + (r & 3) + 8 (r >> 2) + 4 kg][pix0 + 32 m + col]
  auto out_ptr = [&](const VbCursor& c) {
    return reinterpret_cast<float*>(c.ptr) + ((size_t)c.row * VB_C + 32 * wave + 4 * kg) * c.hw + c.tin * BM + col;
  };
  auto init_acc = [&](const VbCursor& c) {
    if (ACCUM) {
      const float* o = out_ptr(c);
      const int rem = c.hw - c.tin * BM;
#pragma unroll
      for (int m = 0; m < SUB; ++m) {
        const int dp = min(32 * m + col, rem - 1) - col;          // clamped: never stored back for tail pixels
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = o[(size_t)((r & 3) + 8 * (r >> 2)) * c.hw + dp];
      }
    } else {
#pragma unroll
      for (int m = 0; m < SUB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    }
  };

  // `nxt` is the tile whose grad_out rows sit in the staging registers: its loads are issued as soon as the previous
  // tile has been parked (before the barrier), so they are in flight across the barrier and the whole MFMA phase.
  issue_loads(cur);
  init_acc(cur);
  park(cur, 0);
  if (slot + slots < total) { vb_advance(nxt, tab, p, slots); issue_loads(nxt); }
  __syncthreads();

  int buf = 0;
  for (int t = slot; t < total; t += slots) {
    const bool has_next = t + slots < total;
    const char* hi_img = smem + buf * 2 * IMG;
    const char* lo_img = hi_img + IMG;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
      for (int m = 0; m < SUB; ++m) {
        const int off = vb_lds_off(32 * m + col, 2 * s + kg);
        const bf16x8 bhi = vb_frag(*reinterpret_cast<const u32x4*>(hi_img + off));
        const bf16x8 blo = vb_frag(*reinterpret_cast<const u32x4*>(lo_img + off));
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[s], bhi, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], blo, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], bhi, acc[m], 0, 0, 0);
      }
    }
    {
      float* o = out_ptr(cur);
      const int rem = cur.hw - cur.tin * BM;
#pragma unroll
      for (int m = 0; m < SUB; ++m) {
        if (32 * m + col < rem) {
#pragma unroll
          for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * cur.hw + 32 * m] = acc[m][r];
        }
      }
    }
    cur = nxt;
    if (has_next) {
      init_acc(cur);
      park(cur, buf ^ 1);                                      // the other buffer: its readers finished before the last barrier
      if (t + 2 * slots < total) { vb_advance(nxt, tab, p, slots); issue_loads(nxt); }
    }
    __syncthreads();
    buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// d(weight), d(bias).  K = all pixels of all cameras.  One persistent 1024-thread workgroup per CU takes every
// gridDim.x-th 32-pixel tile and keeps the whole 256 x 256 result in its accumulators (16 waves, 4 x 4, each 64 x 64 =
// 2 x 2 MFMA 32x32x16 tiles: the arrangement of gd4d_gemm_bf16x3_fwd), then writes its partial to the workspace; a
// second small kernel adds the partials in a fixed order (deterministic, no atomics).
//
// The contraction index is the pixel, which is the contiguous index of the NCHW pyramid but the row index of grad_out.
// MFMA fragments want 8 consecutive k per lane, but a sum over k does not care which 8 as long as both operands agree:
// chunk g (0..3) of a tile holds pixels 8 g .. 8 g + 7 for BOTH operands; x gets them with two 16-byte loads per lane
// (dword loads on levels whose rows are not 16-byte aligned), grad_out with 8 row-strided dword loads, lanes along
// channels.  LDS stage: [chunk g][row 0..255][16 B] for gout_hi, gout_lo, x_hi, x_lo, double buffered (128 KB).
constexpr int VW_BK = 32, VW_THREADS = 1024;
constexpr int VW_ARR = 4 * VB_C * 16;                  // one [4][256][16 B] array: 16 KB
constexpr int VW_STAGE = 4 * VW_ARR;

__global__ __launch_bounds__(VW_THREADS) void value_proj_bwd_weight_kernel(const VpBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages + level table + bias partials
  int* const tab = reinterpret_cast<int*>(smem + 2 * VW_STAGE);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l32 = lane & 31, kg = lane >> 5;

  // tiles blockIdx.x, blockIdx.x + gridDim.x, ...: the workgroups running at the same time read NEIGHBOURING 128-byte
  // pieces of every pyramid row (one DRAM page serves many of them); contiguous per-workgroup ranges made the pyramid
  // read the slowest part of the kernel (2.5 TB/s for that operand alone)
  const int total = p.tile_base[GD4D_MAX_LEVELS];
  const int stride = gridDim.x;
  const int t0 = blockIdx.x;
  const int steps = t0 < total ? (total - t0 + stride - 1) / stride : 0;

  if (tid == 0) vb_fill_table(tab, p, true);
  __syncthreads();
  VbCursor c;
  vb_seek(c, tab, p, min(t0, total - 1));

  // staging roles: x chunk (ci = tid / 4, g = tid % 4); gout chunk (co = tid % 256, g = tid / 256)
  const int xci = tid >> 2, xg = tid & 3;
  const int yco = tid & 255, yg = tid >> 8;
  float xs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ys[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  auto issue = [&]() {
    if (p.dbg & 2) return;
    const int pix0 = c.tin * VW_BK;
    const float* xrow = reinterpret_cast<const float*>(c.ptr) + ((size_t)c.row * VB_C + xci) * c.hw;
    const int px = pix0 + 8 * xg;
    const bool wide = (c.hw % 4 == 0) && ((c.ptr & 15u) == 0);    // workgroup-uniform
    if (p.dbg & 8) {
    } else if (wide) {
      // a quad that starts inside the row ends inside it; out-of-row quads re-read the last one and are zeroed in park
      const float4 a = *reinterpret_cast<const float4*>(xrow + min(px, c.hw - 4));
      const float4 b = *reinterpret_cast<const float4*>(xrow + min(px + 4, c.hw - 4));
      xs[0] = a.x; xs[1] = a.y; xs[2] = a.z; xs[3] = a.w; xs[4] = b.x; xs[5] = b.y; xs[6] = b.z; xs[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xs[j] = xrow[min(px + j, c.hw - 1)];
    }
    const float* yrow = p.gout + ((size_t)c.row * p.S + c.start) * VB_C + yco;
    const int py = pix0 + 8 * yg;
    if (p.dbg & 16) {
    } else if (pix0 + VW_BK <= c.hw) {                        // full tile (workgroup-uniform): no clamps, one base address
      const float* y0 = yrow + (size_t)py * VB_C;
#pragma unroll
      for (int j = 0; j < 8; ++j) ys[j] = y0[j * VB_C];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) ys[j] = yrow[(size_t)min(py + j, c.hw - 1) * VB_C];
    }
  };
  auto park = [&](int stage, int rem) {            // rem: valid pixels of the tile the registers hold
    if (p.dbg & 4) return;
    char* base = smem + stage * VW_STAGE;
    u32x4 h, l;
    if (rem < VW_BK) {                                 // tail tile (workgroup-uniform): zero what lies beyond the row
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ys[j] = 8 * yg + j < rem ? ys[j] : 0.f;
        xs[j] = 8 * xg + j < rem ? xs[j] : 0.f;
      }
    }
    bsum += ((ys[0] + ys[1]) + (ys[2] + ys[3])) + ((ys[4] + ys[5]) + (ys[6] + ys[7]));
    vb_split8(ys, h, l);
    const int yoff = (yg * VB_C + yco) * 16;
    *reinterpret_cast<u32x4*>(base + yoff) = h;
    *reinterpret_cast<u32x4*>(base + VW_ARR + yoff) = l;
    vb_split8(xs, h, l);
    const int xoff = (xg * VB_C + xci) * 16;
    *reinterpret_cast<u32x4*>(base + 2 * VW_ARR + xoff) = h;
    *reinterpret_cast<u32x4*>(base + 3 * VW_ARR + xoff) = l;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // The loads of tile s + 1 are issued as soon as the registers are free - right after tile s has been converted and
  // parked, BEFORE the barrier - so they are in flight across the barrier wait and the whole MFMA phase of step s.
  int rem_next = 0;                                   // valid pixels of the tile the staging registers hold
  if (steps > 0) {
    issue();
    park(0, c.hw - c.tin * VW_BK);
    if (steps > 1) {
      vb_advance(c, tab, p, stride);
      issue();
      rem_next = c.hw - c.tin * VW_BK;
    }
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int cur = s & 1;
    const bool has_next = s + 1 < steps;
    const char* base = smem + cur * VW_STAGE;
    if (!(p.dbg & 1))
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int aoff = ((2 * ks + kg) * VB_C + 64 * wm + 32 * i + l32) * 16;
        const int boff = ((2 * ks + kg) * VB_C + 64 * wn + 32 * i + l32) * 16;
        ah[i] = vb_frag(*reinterpret_cast<const u32x4*>(base + aoff));
        al[i] = vb_frag(*reinterpret_cast<const u32x4*>(base + VW_ARR + aoff));
        bh[i] = vb_frag(*reinterpret_cast<const u32x4*>(base + 2 * VW_ARR + boff));
        bl[i] = vb_frag(*reinterpret_cast<const u32x4*>(base + 3 * VW_ARR + boff));
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    if (has_next) {
      park(cur ^ 1, rem_next);
      if (s + 2 < steps) {                              // uniform branch
        vb_advance(c, tab, p, stride);
        issue();
        rem_next = c.hw - c.tin * VW_BK;
      }
    }
    __syncthreads();
  }

  // partial results: ws[wg][co][ci], then the bias partial.  C/D: column n = ci = lane & 31, rows m = co.
  float* ws = p.ws + (size_t)blockIdx.x * (VB_C * VB_C + VB_C);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 64 * wm + 32 * mi + 4 * kg + (r & 3) + 8 * (r >> 2);
        ws[co * VB_C + 64 * wn + 32 * ni + l32] = acc[mi][ni][r];
      }
  float* bpart = reinterpret_cast<float*>(smem);      // stage memory is free now (last barrier passed)
  bpart[yg * VB_C + yco] = bsum;
  __syncthreads();
  if (tid < VB_C) ws[VB_C * VB_C + tid] = (bpart[tid] + bpart[VB_C + tid]) + (bpart[2 * VB_C + tid] + bpart[3 * VB_C + tid]);
}

// out[i] = sum over workgroups of ws[wg][i], fixed order; i < C*C -> grad_w, else grad_b
__global__ __launch_bounds__(256) void value_proj_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw,
                                                                    float* __restrict__ gb, int parts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  constexpr int N = VB_C * VB_C + VB_C;
  if (i >= N) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 4 <= parts; b += 4) {
    s0 += ws[(size_t)b * N + i];
    s1 += ws[(size_t)(b + 1) * N + i];
    s2 += ws[(size_t)(b + 2) * N + i];
    s3 += ws[(size_t)(b + 3) * N + i];
  }
  for (; b < parts; ++b) s0 += ws[(size_t)b * N + i];
  const float s = (s0 + s1) + (s2 + s3);
  if (i < VB_C * VB_C) gw[i] = s;
  else if (gb) gb[i - VB_C * VB_C] = s;
}

static int vb_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return cus;
}

constexpr int VW_MAX_PARTS = 512;

static int vb_fill_levels(VpBwdParams& p, const int32_t* level_hw, int R, int L, int bm) {
  int s = 0, base = 0;
  for (int l = 0; l < L; ++l) {
    if (level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    const int hw = level_hw[2 * l] * level_hw[2 * l + 1];
    p.hw[l] = hw;
    p.start[l] = s;
    p.tiles[l] = (hw + bm - 1) / bm;
    p.tile_base[l] = base;
    s += hw;
    base += R * p.tiles[l];
  }
  for (int l = L; l <= GD4D_MAX_LEVELS; ++l) p.tile_base[l] = base;
  p.S = s; p.R = R; p.L = L;
  return GD4D_OK;
}

}  // namespace gd4d

extern "C" int gd4d_value_proj_bwd_input(const float* grad_out, const float* weight, float* const* grad_feats,
                                         const int32_t* level_hw, int R, int C, int L, int accumulate, void* stream) {
  using namespace gd4d;
  if (!grad_out || !weight || !grad_feats || !level_hw || R <= 0 || C <= 0 || L <= 0) return GD4D_EINVAL;
  if (C != VB_C || L > GD4D_MAX_LEVELS) return GD4D_EUNSUPPORTED;
  if (!aligned16(grad_out)) return GD4D_EALIGN;
  VpBwdParams p{};
  if (int rc = vb_fill_levels(p, level_hw, R, L, 64)) return rc;
  for (int l = 0; l < L; ++l) {
    if (!grad_feats[l]) return GD4D_EINVAL;
    p.gin[l] = grad_feats[l];
  }
  p.gout = grad_out; p.weight = weight;
  const int total = p.tile_base[L];
  const int grid = total < vb_cus() ? total : vb_cus();
  const size_t lds = 4 * (size_t)64 * VB_C * 2 + 256;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto go = [&](auto kern) {
    (void)allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
  };
  if (accumulate) go(value_proj_bwd_input_kernel<true>); else go(value_proj_bwd_input_kernel<false>);
  return check_launch();
}

extern "C" size_t gd4d_value_proj_bwd_weight_workspace_bytes(void) {
  return (size_t)gd4d::VW_MAX_PARTS * (gd4d::VB_C * gd4d::VB_C + gd4d::VB_C) * sizeof(float);
}

extern "C" int gd4d_value_proj_bwd_weight(const float* grad_out, const void* const* feats, const int32_t* level_hw,
                                          float* grad_weight, float* grad_bias, void* workspace, size_t workspace_bytes,
                                          int R, int C, int L, void* stream) {
  using namespace gd4d;
  if (!grad_out || !feats || !level_hw || !grad_weight || !workspace || R <= 0 || C <= 0 || L <= 0) return GD4D_EINVAL;
  if (C != VB_C || L > GD4D_MAX_LEVELS) return GD4D_EUNSUPPORTED;
  if (workspace_bytes < gd4d_value_proj_bwd_weight_workspace_bytes()) return GD4D_EINVAL;
  if (!aligned16(workspace)) return GD4D_EALIGN;
  VpBwdParams p{};
  if (int rc = vb_fill_levels(p, level_hw, R, L, VW_BK)) return rc;
  for (int l = 0; l < L; ++l) {
    if (!feats[l]) return GD4D_EINVAL;
    p.x[l] = static_cast<const float*>(feats[l]);
  }
  p.gout = grad_out;
  p.ws = static_cast<float*>(workspace);
  p.dbg = 0;
  const int total = p.tile_base[L];
  int grid = vb_cus();
  if (grid > VW_MAX_PARTS) grid = VW_MAX_PARTS;
  if (grid > total) grid = total;
  const size_t lds = 2 * (size_t)VW_STAGE + 256;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!allow_dynamic_lds(reinterpret_cast<const void*>(value_proj_bwd_weight_kernel), (int)lds)) return GD4D_ELAUNCH;
  hipLaunchKernelGGL(value_proj_bwd_weight_kernel, dim3(grid), dim3(VW_THREADS), lds, st, p);
  if (int rc = check_launch()) return rc;
  constexpr int N = VB_C * VB_C + VB_C;
  hipLaunchKernelGGL(value_proj_bwd_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, st, p.ws, grad_weight, grad_bias,
                     grid);
  return check_launch();
}

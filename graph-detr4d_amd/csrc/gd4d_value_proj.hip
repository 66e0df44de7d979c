// gd4d_value_proj_fwd: value_proj over the multi-camera FPN pyramid for gfx950.
//
// Reference: Deform3DCrossAttn.forward, deform3d_cross_attn.py:264-280 - every level
// (B*N, C, H, W) is flattened + transposed to channels-last, the levels are concatenated and a
// Linear(256 -> 256) is applied (a 739 800 x 256 x 256 GEMM at N = 24, the dominant dense
// contraction of a decoder layer), then viewed (B*N, S, Hh, Dh).
//
// This kernel reads the NCHW maps exactly as the caller holds them and writes the channels-last
// head-major value tensor in ONE pass (the reference makes a transposed copy, a concatenated copy
// and the GEMM output): HBM traffic = read pyramid once + write value once.
//
//   out[r, start_l + pix, co] = sum_ci in_l[r, ci, pix] * W[co, ci] + bias[co]
//
// Arithmetic: split-bf16 ("bf16x3") on the MFMA pipe.  x = hi + lo with hi = bf16(x),
// lo = bf16(x - hi); a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulate.  Dropped terms are
// <= ~2^-17 relative per product, i.e. fp32-class results (measured <= 2e-5 abs against an fp32
// GEMM on N(0,1) features) at 3/16 of the cost of the f32-input MFMA.  The f32-input MFMA would make
// this layer MFMA-bound at ~0.65 ms (157 TF peak); bf16x3 puts it under the HBM time (~0.28 ms).
//
// Structure (per CU: one persistent 512-thread workgroup = 8 waves, 2 per SIMD):
//   * wave w owns output channels [32w, 32w+32) and keeps its W_hi / W_lo MFMA B-fragments for
//     the whole K = 256 in registers (2 x 64 VGPRs), loaded once per launch;
//   * per tile of BM pixels: all 512 threads load the fp32 (ci, pix) block with pixel-contiguous
//     (coalesced) dword loads, split to hi/lo bf16 with v_cvt_pk_bf16_f32 and park it in LDS as
//     [pix][ci] rows (the transpose happens in registers: a lane gathers 8 ci of ONE pixel and
//     issues one ds_write_b128), 16-byte chunks XOR-swizzled by (pix & 15) so that both the
//     writes and the ds_read_b128 A-fragment reads are bank-conflict free;
//   * each wave then runs K/16 x 3 MFMA 32x32x16 per 32-pixel sub-tile and stores its 32 channels
//     (128 B per pixel row, one full L2 line) straight from the accumulators;
//   * next tile's global loads are issued before the MFMA phase (register prefetch) and LDS is
//     double buffered, so HBM reads overlap the matrix work.
#include <stdlib.h>

#include <atomic>

#include "gd4d_common.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int VP_C = 256;          // in = out channels (embed_dims of every reference config)
constexpr int VP_THREADS = 512;
constexpr int VP_KSTEPS = VP_C / 16;

struct ValueProjParams {
  const void* in[GD4D_MAX_LEVELS];   // level l: (R, C, HW_l)
  int hw[GD4D_MAX_LEVELS];
  int start[GD4D_MAX_LEVELS];        // pixel offset of level l inside a row of `out`
  int tiles[GD4D_MAX_LEVELS];        // tiles per camera-row at level l
  int tile_base[GD4D_MAX_LEVELS + 1];  // prefix over levels of R * tiles[l]
  const float* weight[GD4D_MAX_LAYERS];   // per decoder layer: (C, C) row-major [co][ci]
  const float* bias[GD4D_MAX_LAYERS];     // (C) or null
  void* out[GD4D_MAX_LAYERS];             // (R, S, C)
  int R, L, S, NL;
  int head_major, Hh, Dh;                 // output layout: (R, S, C) or (R, Hh, S, Dh)
  int xcd_groups;                         // 1: layer groups co-located per XCD (grid % (8*NL) == 0)
  int single_product;                     // 1: bf16-class single product (GD4D_VP_PRECISION_BF16)
  int dbg;                                // dev ablation bits (GD4D_VP_DBG): 1 = skip stores, 2 = skip MFMAs
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}

// split two floats into packed (hi, hi) and (lo, lo) bf16 pairs
struct HiLo { unsigned hi, lo; };
__device__ __forceinline__ HiLo split2(float a, float b) {
  HiLo r;
  r.hi = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(r.hi << 16);          // exact: hi is a rounding of a
  const float rb = b - __uint_as_float(r.hi & 0xffff0000u);
  r.lo = cvt_pk_bf16(ra, rb);
  return r;
}

// 8 consecutive floats -> one 16-byte chunk of hi halves and one of lo halves
__device__ __forceinline__ void split8(const float* v, u32x4& h, u32x4& l) {
  const HiLo a = split2(v[0], v[1]), b = split2(v[2], v[3]), c = split2(v[4], v[5]), d = split2(v[6], v[7]);
  h = u32x4{a.hi, b.hi, c.hi, d.hi};
  l = u32x4{a.lo, b.lo, c.lo, d.lo};
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
  return __builtin_bit_cast(bf16x8, v);
}

// byte offset of 16-byte chunk `chunk` (8 bf16) of row `row` in a [rows][256] bf16 LDS image
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * (VP_C * 2) + ((chunk ^ (row & 15)) << 4);
}

template <int BM, bool OUT_BF16>
__global__ __launch_bounds__(VP_THREADS, 2) void value_proj_kernel(const ValueProjParams p) {
  constexpr int PIX_GROUPS = VP_THREADS / BM;          // threads sharing one pixel column
  constexpr int CPT = VP_C / PIX_GROUPS;               // input channels per thread per tile
  constexpr int SUB = BM / 32;                         // 32-pixel MFMA sub-tiles per tile
  constexpr int IMG = BM * VP_C * 2;                   // bytes of one bf16 [BM][256] image
  static_assert(CPT % 8 == 0, "a thread packs whole 16-byte chunks");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][hi, lo][BM][256] bf16

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31;           // MFMA column (output channel within the wave's block) / A row
  const int kg = lane >> 5;            // which 8 of the 16 k of a step this lane holds

  // Workgroup -> (layer, slot).  Consecutive workgroups serve different layers of the SAME tile
  // sequence, so the NL layer groups sweep the pyramid in step and all but the first reader of a
  // tile are served from L2 / Infinity Cache instead of HBM.
  const int layer = blockIdx.x % p.NL;
  const int slot = blockIdx.x / p.NL;
  const int slots = gridDim.x / p.NL;
  const float* __restrict__ weight = p.weight[layer];
  void* __restrict__ outp = p.out[layer];

  // ---- W fragments for this wave: co = 32*wave + col, k = 16*s + 8*kg .. +8 ----
  bf16x8 whi[VP_KSTEPS], wlo[VP_KSTEPS];
  {
    const float* wrow = weight + (size_t)(32 * wave + col) * VP_C + 8 * kg;
#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(wrow + 16 * s);
      const float4 b = *reinterpret_cast<const float4*>(wrow + 16 * s + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      u32x4 h, l;
      split8(v, h, l);
      whi[s] = as_bf16x8(h);
      wlo[s] = as_bf16x8(l);
    }
  }
  const float bias = p.bias[layer] ? p.bias[layer][32 * wave + col] : 0.f;

  // staging role of this thread: pixel column `spix` of the tile, channels [sc0, sc0 + CPT)
  const int spix = tid % BM;
  const int sc0 = (tid / BM) * CPT;
  float stage[CPT];

  const int total = p.tile_base[p.L];
  auto decode = [&](int t, int& lvl, int& row, int& pix0) {
    lvl = 0;
#pragma unroll
    for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
      if (l < p.L && t >= p.tile_base[l]) lvl = l;
    const int rel = t - p.tile_base[lvl];
    row = rel / p.tiles[lvl];
    pix0 = (rel - row * p.tiles[lvl]) * BM;
  };
  auto issue_loads = [&](int t) {
    int lvl, row, pix0;
    decode(t, lvl, row, pix0);
    const int hw = p.hw[lvl];
    const int pix = pix0 + spix;
    const float* src = static_cast<const float*>(p.in[lvl]) + ((size_t)row * VP_C + sc0) * hw + pix;
    if (pix < hw) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) stage[k] = src[(size_t)k * hw];
    } else {
#pragma unroll
      for (int k = 0; k < CPT; ++k) stage[k] = 0.f;
    }
  };
  auto park = [&](int buf) {
    char* hi_img = smem + buf * 2 * IMG;
    char* lo_img = hi_img + IMG;
#pragma unroll
    for (int c = 0; c < CPT / 8; ++c) {
      u32x4 h, l;
      split8(stage + 8 * c, h, l);
      const int off = lds_off(spix, sc0 / 8 + c);
      *reinterpret_cast<u32x4*>(hi_img + off) = h;
      *reinterpret_cast<u32x4*>(lo_img + off) = l;
    }
  };

  int t = slot;
  if (t >= total) return;
  issue_loads(t);
  park(0);
  __syncthreads();

  int buf = 0;
  for (; t < total; t += slots) {
    const int tn = t + slots;
    const bool has_next = tn < total;
    if (has_next) issue_loads(tn);            // register prefetch: in flight during the MFMA phase

    int lvl, row, pix0;
    decode(t, lvl, row, pix0);
    const char* hi_img = smem + buf * 2 * IMG;
    const char* lo_img = hi_img + IMG;

    f32x16 acc[SUB];
#pragma unroll
    for (int m = 0; m < SUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = bias;

#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
#pragma unroll
      for (int m = 0; m < SUB; ++m) {
        const int off = lds_off(32 * m + col, 2 * s + kg);
        const bf16x8 ahi = as_bf16x8(*reinterpret_cast<const u32x4*>(hi_img + off));
        const bf16x8 alo = as_bf16x8(*reinterpret_cast<const u32x4*>(lo_img + off));
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, whi[s], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wlo[s], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, whi[s], acc[m], 0, 0, 0);
      }
    }

    // ---- epilogue: C/D layout col = lane&31 (co), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pix) ----
    {
      const int hw = p.hw[lvl];
      const size_t obase = ((size_t)row * p.S + p.start[lvl] + pix0 + 4 * kg) * VP_C + 32 * wave + col;
      const bool full = pix0 + BM <= hw;                     // workgroup-uniform
#pragma unroll
      for (int m = 0; m < SUB; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dp = 32 * m + (r & 3) + 8 * (r >> 2);    // pixel offset inside the tile (minus 4*kg)
          if (full || pix0 + dp + 4 * kg < hw) {
            if (OUT_BF16)
              static_cast<uint16_t*>(outp)[obase + (size_t)dp * VP_C] = f32_to_bf16(acc[m][r]);
            else
              static_cast<float*>(outp)[obase + (size_t)dp * VP_C] = acc[m][r];
          }
        }
      }
    }

    if (has_next) park(buf ^ 1);              // other buffer: nobody reads it during this iteration
    __syncthreads();
    buf ^= 1;
  }
}

}  // namespace gd4d

static std::atomic<int> g_cu_limit{0};

extern "C" size_t gd4d_value_proj_workspace_bytes(void) { return 0; }

extern "C" int gd4d_value_proj_set_cu_limit(int cus) { return g_cu_limit.exchange(cus < 0 ? 0 : cus); }

namespace gd4d {

// ---------------------------------------------------------------------------------------------
// v2: software-pipelined variant (BM = 32).  Counters on the kernel above showed the MFMA pipe only
// ~38 % busy and the waves parked ~44 % of the time: every tile ran load -> MFMA -> store -> convert
// as workgroup-wide phases.  Here the phases of THREE consecutive tiles overlap inside one barrier
// interval:
//     tile t+2 : HBM -> LDS raw fp32 image by LDS-DMA (global_load_lds, no VGPR staging)
//     tile t+1 : raw fp32 -> hi/lo bf16 [pix][ci] image, sliced into the 16 k-steps of ...
//     tile t   : ... the MFMA loop, so VALU/LDS conversion work issues under the matrix pipe
//     tile t-1 : accumulators are stored at the top of the interval (they drain during the MFMAs)
// One __syncthreads per tile.  LDS: 2 raw images (2 x 32 KB) + 2 bf16 hi/lo images (2 x 32 KB).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

#define GD4D_SGPR(x) __builtin_amdgcn_readfirstlane(x)

// SINGLE: one bf16 product a_hi*w_hi (bf16-class accuracy, for bf16 value storage) instead of the
// fp32-class three-product split: 1/3 of the MFMAs, no lo image (half the LDS traffic).
template <bool OUT_BF16, int DBG, bool HEAD_MAJOR = false, bool SINGLE = false>   // DBG: compile-time ablation bits (dev only; production = 0)
__global__ __launch_bounds__(VP_THREADS, 2) void value_proj_pipe_kernel(const ValueProjParams p) {
  constexpr int BM = 32;
  constexpr int RAW = VP_C * BM * 4;               // bytes of one raw [256 ci][32 pix] fp32 image
  constexpr int IMG = BM * VP_C * 2;               // bytes of one bf16 [32 pix][256 ci] image
  extern __shared__ __attribute__((aligned(16))) char smem[];   // raw[2] | {hi, lo}[2]
  char* const raw_base = smem;
  char* const img_base = smem + 2 * RAW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31;
  const int kg = lane >> 5;

  // Workgroup -> (layer, slot).  Observed dispatch places workgroup b on XCD b % 8 (speed only): the
  // NL workgroups that serve the same slot are put on ONE XCD so they sweep the same tiles through
  // the same L2 - the pyramid is then fetched once per slot instead of once per layer.
  int layer, slot, slots;
  {
    const int per_xcd = gridDim.x / 8;             // host guarantees gridDim.x % 8 == 0 when p.xcd_groups
    if (p.xcd_groups) {
      const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
      layer = idx % p.NL;
      slot = xcd * (per_xcd / p.NL) + idx / p.NL;
      slots = 8 * (per_xcd / p.NL);
    } else {
      layer = blockIdx.x % p.NL;
      slot = blockIdx.x / p.NL;
      slots = gridDim.x / p.NL;
    }
  }
  const float* __restrict__ weight = p.weight[layer];
  void* __restrict__ outp = p.out[layer];

  bf16x8 whi[VP_KSTEPS], wlo[VP_KSTEPS];
  {
    const float* wrow = weight + (size_t)(32 * wave + col) * VP_C + 8 * kg;
#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(wrow + 16 * s);
      const float4 b = *reinterpret_cast<const float4*>(wrow + 16 * s + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      u32x4 h, l;
      split8(v, h, l);
      whi[s] = as_bf16x8(h);
      wlo[s] = as_bf16x8(l);
    }
  }
  const float bias = p.bias[layer] ? p.bias[layer][32 * wave + col] : 0.f;

  const int total = p.tile_base[p.L];
  const int ntile = slot < total ? (total - slot + slots - 1) / slots : 0;
  if (ntile == 0) return;

  // Level table in the tail of LDS (dynamic indexing of the by-value kernarg struct would make the
  // compiler copy it to scratch).  Written with static indices; read only at level boundaries.
  int* const tab = reinterpret_cast<int*>(smem + 2 * RAW + 4 * IMG);     // [L][6]: hw, tiles, start, base, in.lo, in.hi
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < GD4D_MAX_LEVELS; ++l) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(p.in[l]);
      tab[6 * l + 0] = p.hw[l]; tab[6 * l + 1] = p.tiles[l]; tab[6 * l + 2] = p.start[l];
      tab[6 * l + 3] = p.tile_base[l]; tab[6 * l + 4] = (int)(unsigned)(a & 0xffffffffu); tab[6 * l + 5] = (int)(unsigned)(a >> 32);
    }
  }
  __syncthreads();

  // Tile cursor (DMA side, runs two tiles ahead): plain workgroup-uniform scalars, advanced without
  // divisions; level parameters change only at level boundaries.
  int c_lvl, c_row, c_tin, c_hw, c_tiles, c_start;
  const float* c_in;
  auto set_level = [&](int lvl) {
    c_lvl = lvl;
    c_hw = GD4D_SGPR(tab[6 * lvl + 0]); c_tiles = GD4D_SGPR(tab[6 * lvl + 1]); c_start = GD4D_SGPR(tab[6 * lvl + 2]);
    const unsigned lo = (unsigned)GD4D_SGPR(tab[6 * lvl + 4]), hi = (unsigned)GD4D_SGPR(tab[6 * lvl + 5]);
    c_in = reinterpret_cast<const float*>(((uintptr_t)hi << 32) | lo);
  };
  {
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
      if (l < p.L && slot >= p.tile_base[l]) lvl = l;
    set_level(GD4D_SGPR(lvl));
    const int rel = slot - GD4D_SGPR(tab[6 * c_lvl + 3]);
    c_row = GD4D_SGPR(rel / c_tiles);                 // the only division, once per workgroup
    c_tin = GD4D_SGPR(rel - c_row * c_tiles);
  }
  auto advance = [&]() {
    c_tin += slots;
    while (c_tin >= c_tiles && c_lvl < p.L) {
      c_tin -= c_tiles;
      if (++c_row == p.R) {
        c_row = 0;
        if (c_lvl + 1 < p.L) set_level(c_lvl + 1); else c_lvl = p.L;
      }
    }
    c_tin = GD4D_SGPR(c_tin); c_row = GD4D_SGPR(c_row); c_lvl = GD4D_SGPR(c_lvl);
  };

  // LDS-DMA of the tile under the cursor into raw image `rb`.  Wave w fills ci rows [32w, 32w+32).
  auto issue_dma = [&](int rb) {
    if (DBG & 4) return;
    const int hw = c_hw;
    const int pix0 = c_tin * BM;
    const float* src = c_in + (size_t)c_row * VP_C * hw;
    char* dst = raw_base + rb * RAW + (32 * wave) * (BM * 4);
    const bool wide = (hw % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);   // wave-uniform
    if (wide) {
      // one instruction = 8 ci rows x 128 B; a quad that starts inside the row also ends inside it
      int pq = pix0 + 4 * (lane & 7);
      if (pq >= hw) pq = hw - 4;                    // tail lanes re-read the last quad (never stored)
      const float* g = src + (size_t)(32 * wave + (lane >> 3)) * hw + pq;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (size_t)(8 * i) * hw),
                                         (lds_void_t*)(dst + i * 8 * (BM * 4)), 16, 0, 0);
    } else {
      // one instruction = 2 ci rows x 128 B, one pixel per lane (clamped exactly at the row end)
      const int px = min(pix0 + (lane & 31), hw - 1);
      const float* g = src + (size_t)(32 * wave + (lane >> 5)) * hw + px;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (size_t)(2 * i) * hw),
                                         (lds_void_t*)(dst + i * 2 * (BM * 4)), 4, 0, 0);
    }
  };

  // conversion role of this thread: pixel spix, channels [16*scg, 16*scg + 16)
  const int spix = tid & 31;
  const int scg = tid >> 5;

  // FIFO of (first output pixel row, valid pixel rows) for tiles k, k+1, k+2 - filled at DMA time
  int orow0 = 0, orem0 = 0, orow1 = 0, orem1 = 0, orow2 = 0, orem2 = 0;
  auto tile_info = [&](int& orow, int& orem) {
    orow = GD4D_SGPR(c_row * p.S + c_start + c_tin * BM);
    orem = GD4D_SGPR(c_hw - c_tin * BM);
  };

  // ---- prologue: tiles 0 and 1 in flight, tile 0 converted ----
  issue_dma(0);
  tile_info(orow0, orem0);
  advance();
  if (ntile > 1) { issue_dma(1); tile_info(orow1, orem1); advance(); }
  __syncthreads();                                 // (drains the DMAs)
  {
    const float* rawf = reinterpret_cast<const float*>(raw_base) + (16 * scg) * BM + spix;
    float cv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) cv[j] = rawf[j * BM];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 h, l;
      split8(cv + 8 * c, h, l);
      const int off = lds_off(spix, 2 * scg + c);
      *reinterpret_cast<u32x4*>(img_base + off) = h;
      if (!SINGLE) *reinterpret_cast<u32x4*>(img_base + IMG + off) = l;
    }
  }
  __syncthreads();

  f32x16 prev;                                     // accumulators of the previous tile, stored one interval later
  int prow = 0, prem = 0;                          // its first output pixel row / valid pixel rows
  // head-major: element offset of this lane's (head, channel) inside a camera row's planes
  const size_t hm_base = (size_t)((32 * wave + col) / p.Dh) * p.S * p.Dh + (32 * wave + col) % p.Dh;
  int pcam = 0, ppix = 0;                          // camera row / first pixel (in-row) of the previous tile
  // store accumulator row r of the previous tile (one store instruction)
  auto store_one = [&](int r, bool full) {
    if (DBG & 1) { asm volatile("" ::"v"(prev[r])); return; }
    const int dp = (r & 3) + 8 * (r >> 2);
    size_t o;
    if (HEAD_MAJOR) {
      // prow = cam_row * S + pixel-in-row; plane (cam_row, head) holds S x Dh elements
      o = hm_base + (size_t)(pcam * p.Hh) * p.S * p.Dh + (size_t)(ppix + 4 * kg + dp) * p.Dh;
    } else {
      o = ((size_t)prow + 4 * kg + dp) * VP_C + 32 * wave + col;
    }
    if (full || dp + 4 * kg < prem) {
      if (OUT_BF16) static_cast<uint16_t*>(outp)[o] = f32_to_bf16(prev[r]);
      else static_cast<float*>(outp)[o] = prev[r];
    }
  };

  for (int k = 0; k < ntile; ++k) {
    // VMEM program order of this interval: [DMA of tile k+2] then [<= 16 stores of tile k-1].
    // The counted wait at the bottom lets the stores stay in flight across the barrier.
    if (k + 2 < ntile) { issue_dma(k & 1); tile_info(orow2, orem2); advance(); }
    // The 16 stores of tile k-1 are spread over the 16 k-steps below (one each), so every store has
    // ~96 cycles of MFMA work to drain behind instead of backing up the wave's issue.
    const bool has_prev = k > 0;
    const bool full_prev = prem >= BM;              // workgroup-uniform; tail tiles take the guarded path
    const int ib = k & 1;
    const char* hi_img = img_base + ib * 2 * IMG;
    const char* lo_img = hi_img + IMG;
    char* nhi_img = img_base + (ib ^ 1) * 2 * IMG;  // tile k+1 is converted into the other image ...
    char* nlo_img = nhi_img + IMG;
    const float* rawf = reinterpret_cast<const float*>(raw_base + (ib ^ 1) * RAW) + (16 * scg) * BM + spix;  // ... from raw[(k+1)&1]

    f32x16 acc0, acc1;                              // two independent MFMA chains
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = bias; acc1[r] = 0.f; }
    float cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // A fragments through a 3-deep register ring: the reads for k-step s+2 are issued before the
    // MFMAs of step s, so LDS latency is covered by matrix work instead of alternating with it.
    u32x4 fh[3], fl[3];
    auto frag_load = [&](int s2, u32x4& h, u32x4& l) {
      if (DBG & 8) { h = u32x4{0u, 0u, 0u, 0u}; l = h; return; }
      const int off = lds_off(col, 2 * s2 + kg);
      h = *reinterpret_cast<const u32x4*>(hi_img + off);
      l = *reinterpret_cast<const u32x4*>(lo_img + off);
    };
    frag_load(0, fh[0], fl[0]);
    frag_load(1, fh[1], fl[1]);
#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
      if (s + 2 < VP_KSTEPS) frag_load(s + 2, fh[(s + 2) % 3], fl[(s + 2) % 3]);
      // 1/16 of the next tile's conversion per k-step (harmless garbage after the last tile)
      if (!(DBG & 16)) cv[s & 7] = rawf[s * BM];
      if (has_prev) store_one(s, full_prev);
      __builtin_amdgcn_sched_barrier(0);            // keep the prefetch ABOVE this step's MFMAs
      const bf16x8 ahi = as_bf16x8(fh[s % 3]);
      const bf16x8 alo = as_bf16x8(fl[s % 3]);
      if (!(DBG & 2)) {
        if (DBG & 32) __builtin_amdgcn_s_setprio(1);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, whi[s], acc0, 0, 0, 0);
        if (!SINGLE) {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, whi[s], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wlo[s], acc1, 0, 0, 0);
        }
        if (DBG & 32) __builtin_amdgcn_s_setprio(0);
      } else {
        asm volatile("" ::"v"(ahi), "v"(alo));
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((s & 7) == 7 && !(DBG & 16)) {
        u32x4 h, l;
        split8(cv, h, l);
        const int woff = lds_off(spix, 2 * scg + (s >> 3));
        *reinterpret_cast<u32x4*>(nhi_img + woff) = h;
        if (!SINGLE) *reinterpret_cast<u32x4*>(nlo_img + woff) = l;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[r] = SINGLE ? acc0[r] : acc0[r] + acc1[r];
    prow = orow0; prem = orem0;
    if (HEAD_MAJOR) { pcam = GD4D_SGPR(prow / p.S); ppix = GD4D_SGPR(prow - pcam * p.S); }   // once per tile
    orow0 = orow1; orem0 = orem1; orow1 = orow2; orem1 = orem2;
    // Raw barrier + explicit waits (a __syncthreads() here makes the compiler drain vmcnt(0), i.e. wait
    // for every store of the previous tile).  Needed before the barrier: this wave's LDS writes done
    // (lgkmcnt) and the DMA of tile k+2 landed - it was issued BEFORE the <= 16 stores, and VMEM ops
    // retire in order, so "at most 16 outstanding" implies the DMA is complete.
    // (k = 0 issues no stores, so nothing sits behind the DMA in the queue: full drain there too)
    if (has_prev && full_prev) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) store_one(r, prem >= BM);
}

// ---------------------------------------------------------------------------------------------
// v3: one wave per SIMD.  4 waves per workgroup (one workgroup per CU), wave w owns output channels
// [64w, 64w+64) and keeps W_hi / W_lo fragments for both 32-channel blocks in registers (256 of the
// 512 VGPR+AGPR entries a lone wave may use).  Compared with the 8-wave kernels: every A fragment
// pair read from LDS feeds 6 MFMAs instead of 3 (half the LDS fragment traffic per tile), the matrix
// pipe of a SIMD belongs to one wave (no two-wave arbitration), and the wave's other work (DMA issue,
// conversion slices, stores of the previous tile) is interleaved into its own MFMA stream.
template <bool OUT_BF16, bool HEAD_MAJOR>
__global__ __launch_bounds__(256, 1) void value_proj_w4_kernel(const ValueProjParams p) {
  constexpr int BM = 32;
  constexpr int RAW = VP_C * BM * 4;
  constexpr int IMG = BM * VP_C * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // raw[2] | {hi, lo}[2]
  char* const raw_base = smem;
  char* const img_base = smem + 2 * RAW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31;
  const int kg = lane >> 5;

  int layer, slot, slots;
  {
    const int per_xcd = gridDim.x / 8;
    if (p.xcd_groups) {
      const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
      layer = idx % p.NL;
      slot = xcd * (per_xcd / p.NL) + idx / p.NL;
      slots = 8 * (per_xcd / p.NL);
    } else {
      layer = blockIdx.x % p.NL;
      slot = blockIdx.x / p.NL;
      slots = gridDim.x / p.NL;
    }
  }
  const float* __restrict__ weight = p.weight[layer];
  void* __restrict__ outp = p.out[layer];

  bf16x8 whi[2][VP_KSTEPS], wlo[2][VP_KSTEPS];
  float bias[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int co = 64 * wave + 32 * cb + col;
    const float* wrow = weight + (size_t)co * VP_C + 8 * kg;
#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(wrow + 16 * s);
      const float4 b = *reinterpret_cast<const float4*>(wrow + 16 * s + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      u32x4 h, l;
      split8(v, h, l);
      whi[cb][s] = as_bf16x8(h);
      wlo[cb][s] = as_bf16x8(l);
    }
    bias[cb] = p.bias[layer] ? p.bias[layer][co] : 0.f;
  }

  const int total = p.tile_base[p.L];
  const int ntile = slot < total ? (total - slot + slots - 1) / slots : 0;
  if (ntile == 0) return;

  // level table in the tail of LDS (see value_proj_pipe_kernel)
  int* const tab = reinterpret_cast<int*>(smem + 2 * RAW + 4 * IMG);
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < GD4D_MAX_LEVELS; ++l) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(p.in[l]);
      tab[6 * l + 0] = p.hw[l]; tab[6 * l + 1] = p.tiles[l]; tab[6 * l + 2] = p.start[l];
      tab[6 * l + 3] = p.tile_base[l]; tab[6 * l + 4] = (int)(unsigned)(a & 0xffffffffu); tab[6 * l + 5] = (int)(unsigned)(a >> 32);
    }
  }
  __syncthreads();
  int c_lvl, c_row, c_tin, c_hw, c_tiles, c_start;
  const float* c_in;
  auto set_level = [&](int lvl) {
    c_lvl = lvl;
    c_hw = GD4D_SGPR(tab[6 * lvl + 0]); c_tiles = GD4D_SGPR(tab[6 * lvl + 1]); c_start = GD4D_SGPR(tab[6 * lvl + 2]);
    const unsigned lo = (unsigned)GD4D_SGPR(tab[6 * lvl + 4]), hi = (unsigned)GD4D_SGPR(tab[6 * lvl + 5]);
    c_in = reinterpret_cast<const float*>(((uintptr_t)hi << 32) | lo);
  };
  {
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
      if (l < p.L && slot >= p.tile_base[l]) lvl = l;
    set_level(GD4D_SGPR(lvl));
    const int rel = slot - GD4D_SGPR(tab[6 * c_lvl + 3]);
    c_row = GD4D_SGPR(rel / c_tiles);
    c_tin = GD4D_SGPR(rel - c_row * c_tiles);
  }
  auto advance = [&]() {
    c_tin += slots;
    while (c_tin >= c_tiles && c_lvl < p.L) {
      c_tin -= c_tiles;
      if (++c_row == p.R) {
        c_row = 0;
        if (c_lvl + 1 < p.L) set_level(c_lvl + 1); else c_lvl = p.L;
      }
    }
    c_tin = GD4D_SGPR(c_tin); c_row = GD4D_SGPR(c_row); c_lvl = GD4D_SGPR(c_lvl);
  };

  // LDS-DMA: wave w fills ci rows [64w, 64w+64) of the raw image
  auto issue_dma = [&](int rb) {
    const int hw = c_hw;
    const int pix0 = c_tin * BM;
    const float* src = c_in + (size_t)c_row * VP_C * hw;
    char* dst = raw_base + rb * RAW + (64 * wave) * (BM * 4);
    const bool wide = (hw % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    if (wide) {
      int pq = pix0 + 4 * (lane & 7);
      if (pq >= hw) pq = hw - 4;
      const float* g = src + (size_t)(64 * wave + (lane >> 3)) * hw + pq;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (size_t)(8 * i) * hw),
                                         (lds_void_t*)(dst + i * 8 * (BM * 4)), 16, 0, 0);
    } else {
      const int px = min(pix0 + (lane & 31), hw - 1);
      const float* g = src + (size_t)(64 * wave + (lane >> 5)) * hw + px;
#pragma unroll
      for (int i = 0; i < 32; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (size_t)(2 * i) * hw),
                                         (lds_void_t*)(dst + i * 2 * (BM * 4)), 4, 0, 0);
    }
  };

  // conversion role: pixel spix, channels [32*scg, 32*scg + 32)
  const int spix = tid & 31;
  const int scg = tid >> 5;                         // 0..7

  int orow0 = 0, orem0 = 0, orow1 = 0, orem1 = 0, orow2 = 0, orem2 = 0;
  auto tile_info = [&](int& orow, int& orem) {
    orow = GD4D_SGPR(c_row * p.S + c_start + c_tin * BM);
    orem = GD4D_SGPR(c_hw - c_tin * BM);
  };

  issue_dma(0);
  tile_info(orow0, orem0);
  advance();
  if (ntile > 1) { issue_dma(1); tile_info(orow1, orem1); advance(); }
  __syncthreads();
  {
    const float* rawf = reinterpret_cast<const float*>(raw_base) + (32 * scg) * BM + spix;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float cv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) cv[j] = rawf[(8 * c + j) * BM];
      u32x4 h, l;
      split8(cv, h, l);
      const int off = lds_off(spix, 4 * scg + c);
      *reinterpret_cast<u32x4*>(img_base + off) = h;
      *reinterpret_cast<u32x4*>(img_base + IMG + off) = l;
    }
  }
  __syncthreads();

  f32x16 prev[2];
  int prow = 0, prem = 0, pcam = 0, ppix = 0;
  const size_t hm_base0 = (size_t)((64 * wave + col) / p.Dh) * p.S * p.Dh + (64 * wave + col) % p.Dh;
  const size_t hm_base1 = (size_t)((64 * wave + 32 + col) / p.Dh) * p.S * p.Dh + (64 * wave + 32 + col) % p.Dh;
  auto store_one = [&](int cb, int r, bool full) {
    const int dp = (r & 3) + 8 * (r >> 2);
    size_t o;
    if (HEAD_MAJOR) o = (cb ? hm_base1 : hm_base0) + (size_t)(pcam * p.Hh) * p.S * p.Dh + (size_t)(ppix + 4 * kg + dp) * p.Dh;
    else o = ((size_t)prow + 4 * kg + dp) * VP_C + 64 * wave + 32 * cb + col;
    if (full || dp + 4 * kg < prem) {
      if (OUT_BF16) static_cast<uint16_t*>(outp)[o] = f32_to_bf16(prev[cb][r]);
      else static_cast<float*>(outp)[o] = prev[cb][r];
    }
  };

  for (int k = 0; k < ntile; ++k) {
    if (k + 2 < ntile) { issue_dma(k & 1); tile_info(orow2, orem2); advance(); }
    const bool has_prev = k > 0;
    const bool full_prev = prem >= BM;
    const int ib = k & 1;
    const char* hi_img = img_base + ib * 2 * IMG;
    const char* lo_img = hi_img + IMG;
    char* nhi_img = img_base + (ib ^ 1) * 2 * IMG;
    char* nlo_img = nhi_img + IMG;
    const float* rawf = reinterpret_cast<const float*>(raw_base + (ib ^ 1) * RAW) + (32 * scg) * BM + spix;

    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = bias[cb];
    float cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 fh[2], fl[2];
    {
      const int off = lds_off(col, kg);
      fh[0] = *reinterpret_cast<const u32x4*>(hi_img + off);
      fl[0] = *reinterpret_cast<const u32x4*>(lo_img + off);
    }
#pragma unroll
    for (int s = 0; s < VP_KSTEPS; ++s) {
      if (s + 1 < VP_KSTEPS) {                       // prefetch the next step's A fragments
        const int off = lds_off(col, 2 * (s + 1) + kg);
        fh[(s + 1) & 1] = *reinterpret_cast<const u32x4*>(hi_img + off);
        fl[(s + 1) & 1] = *reinterpret_cast<const u32x4*>(lo_img + off);
      }
      // 2/32 of the next tile's conversion and 2/32 of the previous tile's stores per k-step
      cv[(2 * s) & 7] = rawf[(2 * s) * BM];
      cv[(2 * s + 1) & 7] = rawf[(2 * s + 1) * BM];
      if (has_prev) { store_one(0, s, full_prev); store_one(1, s, full_prev); }
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 ahi = as_bf16x8(fh[s & 1]);
      const bf16x8 alo = as_bf16x8(fl[s & 1]);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, whi[cb][s], acc[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wlo[cb][s], acc[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, whi[cb][s], acc[cb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((s & 3) == 3) {                            // 8 channels gathered: one chunk of the next image
        u32x4 h, l;
        split8(cv, h, l);
        const int woff = lds_off(spix, 4 * scg + (s >> 2));
        *reinterpret_cast<u32x4*>(nhi_img + woff) = h;
        *reinterpret_cast<u32x4*>(nlo_img + woff) = l;
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[cb][r] = acc[cb][r];
    prow = orow0; prem = orem0;
    if (HEAD_MAJOR) { pcam = GD4D_SGPR(prow / p.S); ppix = GD4D_SGPR(prow - pcam * p.S); }
    orow0 = orow1; orem0 = orem1; orow1 = orow2; orem1 = orem2;
    // DMA of tile k+2 was issued before this interval's 32 stores (see value_proj_pipe_kernel)
    if (has_prev && full_prev) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { store_one(0, r, prem >= BM); store_one(1, r, prem >= BM); }
}

static int vp_variant() {
  static int v = 0;
  if (!v) {
    const char* e = getenv("GD4D_VP_VARIANT");        // dev A/B switch: 1 = phase kernel, 2 = pipelined
    v = e ? atoi(e) : 2;
    if (v < 1 || v > 3) v = 2;
  }
  return v;
}

static int vp_tile_pixels() {
  static int bm = 0;
  if (!bm) {
    const char* e = getenv("GD4D_VP_BM");            // dev A/B switch
    bm = (e && atoi(e) == 64) ? 64 : 32;
  }
  return bm;
}

template <int BM>
static int vp_launch(ValueProjParams& p, const int32_t* level_hw, int R, int L, int NL, int out_dtype,
                     hipStream_t st) {
  int s = 0, base = 0;
  for (int l = 0; l < L; ++l) {
    const int hw = level_hw[2 * l] * level_hw[2 * l + 1];
    p.hw[l] = hw;
    p.start[l] = s;
    p.tiles[l] = (hw + BM - 1) / BM;
    p.tile_base[l] = base;
    s += hw;
    base += R * p.tiles[l];
  }
  for (int l = L; l <= GD4D_MAX_LEVELS; ++l) p.tile_base[l] = base;
  p.S = s;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  {
    // leave CUs free for kernels of another stream (gd4d_value_proj_set_cu_limit; GD4D_VP_CUS overrides, dev switch)
    static int env_limit = -1;
    if (env_limit < 0) { const char* e = getenv("GD4D_VP_CUS"); env_limit = e ? atoi(e) : 0; }
    const int limit = env_limit > 0 ? env_limit : g_cu_limit.load(std::memory_order_relaxed);
    if (limit >= 8 && limit < cus) cus = limit - limit % 8;
  }
  int slots = cus / NL;                                // persistent: <= one workgroup per CU
  if (slots < 1) slots = 1;
  if (slots > base) slots = base;
  const int grid = slots * NL;
  if (BM == 32 && (vp_variant() >= 2 || p.head_major || p.single_product)) {
    // co-locate the NL workgroups of a slot on one XCD: grid = 8 XCDs x (cus/8 rounded down to a multiple of NL)
    int g2 = grid;
    p.xcd_groups = 0;
    if (NL > 1 && cus % 8 == 0 && (cus / 8) >= NL) {
      const int per_xcd = ((cus / 8) / NL) * NL;
      if (8 * (per_xcd / NL) <= base) { g2 = 8 * per_xcd; p.xcd_groups = 1; }
    }
    const int grid = g2;
    size_t lds2 = 2 * (size_t)VP_C * 32 * 4 + 2 * 2 * (size_t)32 * VP_C * 2 + 256;   // raw[2] + {hi,lo}[2] = 128 KB, + level table
    {
      // GD4D_VP_LDS_FENCE=1: claim the CU's whole LDS so that kernels of another stream that use LDS (the query-side
      // linears, LayerNorms, attention core) cannot co-reside on this kernel's CUs and are dispatched to the CUs
      // GD4D_VP_CUS leaves free instead of fighting the persistent waves for issue slots.
      static int fence = -1;
      if (fence < 0) { const char* e = getenv("GD4D_VP_LDS_FENCE"); fence = e ? atoi(e) : 0; }
      if (fence) lds2 = 160 * 1024;
    }
    const bool ob = out_dtype == GD4D_BF16;
    if (vp_variant() == 3 && !p.single_product && p.dbg == 0) {     // one wave per SIMD
      auto go4 = [&](auto kern) {
        (void)allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds2);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds2, st, p);
      };
      if (p.head_major) { if (ob) go4(value_proj_w4_kernel<true, true>); else go4(value_proj_w4_kernel<false, true>); }
      else { if (ob) go4(value_proj_w4_kernel<true, false>); else go4(value_proj_w4_kernel<false, false>); }
      return check_launch();
    }
    auto go = [&](auto kern) {
      (void)allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds2);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(VP_THREADS), lds2, st, p);
    };
    switch (p.dbg) {                                 // ablation builds exist for fp32 output only
      case 1: go(value_proj_pipe_kernel<false, 1>); break;
      case 2: go(value_proj_pipe_kernel<false, 2>); break;
      case 3: go(value_proj_pipe_kernel<false, 3>); break;
      case 7: go(value_proj_pipe_kernel<false, 7>); break;
      case 31: go(value_proj_pipe_kernel<false, 31>); break;
      case 32: go(value_proj_pipe_kernel<false, 32>); break;
      default:
        if (p.single_product) {                      // bf16 output only (validated by the caller)
          if (p.head_major) go(value_proj_pipe_kernel<true, 0, true, true>); else go(value_proj_pipe_kernel<true, 0, false, true>);
        } else if (p.head_major) { if (ob) go(value_proj_pipe_kernel<true, 0, true>); else go(value_proj_pipe_kernel<false, 0, true>); }
        else { if (ob) go(value_proj_pipe_kernel<true, 0>); else go(value_proj_pipe_kernel<false, 0>); }
        break;
    }
    return check_launch();
  }
  const size_t lds = 2 * 2 * (size_t)BM * VP_C * 2;    // 2 buffers x (hi, lo) x [BM][256] bf16
  if (out_dtype == GD4D_BF16) {
    (void)allow_dynamic_lds(reinterpret_cast<const void*>(value_proj_kernel<BM, true>), (int)lds);
    hipLaunchKernelGGL((value_proj_kernel<BM, true>), dim3(grid), dim3(VP_THREADS), lds, st, p);
  } else {
    (void)allow_dynamic_lds(reinterpret_cast<const void*>(value_proj_kernel<BM, false>), (int)lds);
    hipLaunchKernelGGL((value_proj_kernel<BM, false>), dim3(grid), dim3(VP_THREADS), lds, st, p);
  }
  return check_launch();
}

}  // namespace gd4d

extern "C" int gd4d_value_proj_multi_fwd(const void* const* feats, const int32_t* level_hw,
                                         const float* const* weights, const float* const* biases,
                                         void* const* outs, int R, int C, int L, int NL, int Hh,
                                         int in_dtype, int out_dtype, int out_layout, int precision,
                                         void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !weights || !outs) return GD4D_EINVAL;
  if (R <= 0 || C <= 0 || L <= 0 || NL <= 0) return GD4D_EINVAL;
  if (C != VP_C || L > GD4D_MAX_LEVELS || NL > GD4D_MAX_LAYERS || in_dtype != GD4D_F32) return GD4D_EUNSUPPORTED;
  if (out_dtype != GD4D_F32 && out_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  if (out_layout != GD4D_LAYOUT_PIXEL_MAJOR && out_layout != GD4D_LAYOUT_HEAD_MAJOR) return GD4D_EUNSUPPORTED;
  if (Hh <= 0 || C % Hh != 0) return GD4D_EINVAL;
  if (precision != GD4D_VP_PRECISION_F32 && precision != GD4D_VP_PRECISION_BF16) return GD4D_EUNSUPPORTED;
  if (precision == GD4D_VP_PRECISION_BF16 && out_dtype != GD4D_BF16) return GD4D_EUNSUPPORTED;
  ValueProjParams p{};
  for (int l = 0; l < L; ++l) {
    if (!feats[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0) return GD4D_EINVAL;
    p.in[l] = feats[l];
  }
  for (int i = 0; i < NL; ++i) {
    if (!weights[i] || !outs[i]) return GD4D_EINVAL;
    if (!aligned16(weights[i])) return GD4D_EALIGN;
    p.weight[i] = weights[i];
    p.bias[i] = biases ? biases[i] : nullptr;
    p.out[i] = outs[i];
  }
  p.R = R; p.L = L; p.NL = NL;
  p.head_major = out_layout == GD4D_LAYOUT_HEAD_MAJOR; p.Hh = Hh; p.Dh = C / Hh;
  p.single_product = precision == GD4D_VP_PRECISION_BF16;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("GD4D_VP_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return (vp_tile_pixels() == 64 && !p.head_major && !p.single_product) ? vp_launch<64>(p, level_hw, R, L, NL, out_dtype, st)
                                                   : vp_launch<32>(p, level_hw, R, L, NL, out_dtype, st);
}

extern "C" int gd4d_value_proj_fwd(const void* const* feats, const int32_t* level_hw, const float* weight,
                                   const float* bias, void* out, int R, int C, int L, int Hh, int in_dtype,
                                   int out_dtype, int out_layout, int precision, void* stream) {
  if (!weight || !out) return GD4D_EINVAL;
  const float* ws[1] = {weight};
  const float* bs[1] = {bias};
  void* os[1] = {out};
  return gd4d_value_proj_multi_fwd(feats, level_hw, ws, bs, os, R, C, L, 1, Hh, in_dtype, out_dtype, out_layout,
                                   precision, stream);
}

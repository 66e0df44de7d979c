// gd4d_mlp2_bf16x3_fwd: out = relu(X W1^T + b1) W2^T + b2 in ONE kernel - the head's position-embedding MLPs, two 1x1 convolutions
// with a ReLU between them over every pixel of every camera (dense_heads/detr3d_head_pe.py:380-390 `position_encoder`, applied at
// :543-553; 192 -> 1024 -> 256 over 739 800 pixels at 24 cameras).  As two GEMMs the 739 800 x 1024 hidden activation (3 GB fp32)
// made a round trip through HBM; here it never leaves the registers.
//
// Arithmetic: gd4d_gemm_bf16x3_fwd's - both operands of both products split into bf16 hi + lo, the three products hi hi + hi lo +
// lo hi accumulated in fp32 on the bf16 MFMA (~2^-16 relative per product).
//
// A wave owns 32 rows of X and ALL 256 output columns (8 tiles of v_mfma_f32_32x32x16_bf16, 128 accumulator registers).  The
// hidden axis is walked in chunks of 32:
//   phase A  H^T[32 hidden, 32 rows] = W1c X^T  - the TRANSPOSED product, so that the accumulator layout (a lane = one X row,
//            16 hidden units) is already the A-operand layout of the next product: bias, ReLU and the hi / lo split happen in
//            registers; hidden unit <-> k-slot is a fixed permutation inside a chunk, baked into W2's image
//   phase B  out[32 rows, 256] += H[32 rows, 32 hidden] W2c^T
// The weights are static: gd4d_mlp2_image lays both out as MFMA fragments (hi / lo planes, 1 KB per fragment and wave), a chunk's
// 24 + 32 KB (+ its 32 entries of b1) are fetched by LDS-DMA into a double-buffered stage while the previous chunk computes; the
// waves of a workgroup share them.
// X stays in REGISTERS, split once per tile (K1 / 16 steps x 8 registers: 96 at K1 = 192).  A first version re-read it per chunk
// from memory: 32 x 568 MB, and 32 workgroups x 196 KB of X per XCD do not fit its 4-MB L2 - 3.6 ms, slower than the two GEMMs it
// replaces (docs/measurements_r05.md).  With 128 accumulators, 96 - 128 registers of X and the fragments in flight a wave needs
// ~300 registers: ONE wave per SIMD (4 waves = 128 rows per workgroup, 512 registers each), the instruction stream itself keeps the
// matrix pipe busy (phase B: 48 MFMAs against 32 LDS reads).  One barrier per chunk.  114 KB of LDS, one workgroup per compute unit.
#include "gd4d_common.h"

// Measured and left off (docs/measurements_r05.md section 5; 2.16 ms with neither): the next stage's DMA pieces issued during phase A as
// well (2.23 - 2.27 ms), two alternating accumulators in phase A (2.23), both (2.31); a branch-free stage (padded piece count) with
// the requests pinned in front of the MFMAs by sched_group_barrier and the DMA as one burst between the phases (2.40), at the chunk's
// start (2.25 - 2.28) or a piece at the head of every group (2.29).  Counters of this version (profiles/r05_pmc_mlp2.txt): matrix
// pipe busy 37 %, waves parked at a counter / the barrier 33 % of their cycles, stalled at issue 38 %, active 29 %.
// Round 6, what bounds it (timing builds, position MLP at 24 cameras, same box): as shipped 2.03 ms; without the stage DMA, its waits and
// the barriers 1.77; additionally without the LDS fragment reads - the bare stream of 84 MFMAs + the ReLU / split per chunk - 1.69 ms,
// i.e. 1.2 PFLOP/s of bf16 products is what this instruction stream reaches on the device under load; the kernel is at 83 % of that.
// Two output tiles per phase-B group with their MFMAs alternating (the dependent-accumulate distance 2 instead of 1): 2.10 (slower;
// 1.73 against 1.70 in the bare stream) - dependent accumulation is forwarded and costs nothing.
#ifndef ML_DMA_IN_A
#define ML_DMA_IN_A 0
#endif
#ifndef ML_TWO_ACC
#define ML_TWO_ACC 0
#endif
#ifndef ML_AHEAD
#define ML_AHEAD 1           // (ml_phase) groups of phase B whose fragments are in flight (2, 3: the same 2.94-2.97 ms)
#endif
#ifndef ML_SPLIT_WAIT
#define ML_SPLIT_WAIT 1      // same box: position MLP 2.12-2.15 -> 2.08 ms, SE gate + fuse 1.30 -> 1.22 ms, the one-kernel form 2.89 = 2.89
#endif

namespace gd4d {

typedef __attribute__((ext_vector_type(8))) __bf16 ml_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ml_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned ml_u32x4;

constexpr int ML_WAVES = 4, ML_THREADS = 64 * ML_WAVES, ML_BM = 32 * ML_WAVES, ML_HC = 32, ML_N2 = 256, ML_NT = ML_N2 / 32;
constexpr int ML_S2 = ML_NT * 2 * 2 * 1024;          // bytes of a chunk of W2's image: [tile][step][plane][1 KB]

struct Mlp2Params {
  const float* x;
  const char* w1img;
  const char* w2img;
  const float* b2;
  float* out;
  int M, K1, H, ldx, ldo;
};

// The SE form (gd4d_mlp2_se_fuse_fwd): the rows are the pixels of NCHW feature levels laid side by side - row m = camera r,
// level l, pixel pix with m = r S + start[l] + pix - X is read from the maps themselves and the epilogue is the head's fuse:
//   out[r, pix, n] = feat[r, n, pix] + (pe[m, n] * sigmoid(acc[m, n] + b2[n]) + sine[m, n]),     out stored (R, HW_l, 256) per level.
struct Mlp2Se {
  const float* feat[4];     // (R, 256, HW_l)
  float* out[4];            // (R, HW_l, 256)
  const float* pe;          // (R S, 256) channels-last rows
  const float* sine;        // (R S, 256)
  int hw[4];
  int start[5];             // first row of level l inside a camera's S rows; start[l >= L] = S
  int S;
};

// sigmoid on the transcendental unit (v_exp_f32, v_rcp_f32: ~1e-7 absolute - the library's expf and an IEEE division are ~45
// instructions per element, and with one wave per SIMD the epilogue's arithmetic is not hidden by anything)
__device__ __forceinline__ float ml_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

struct Mlp2Row { const float* f; float* o; int pix, hw; };     // feat / out of the row's camera and level; its pixel

// The FRUSTUM form (gd4d_mlp2_frustum_fwd): X is not read at all.  Row m = camera r, level l, pixel (y, x); its 3 D = 192 inputs are the
// head's frustum coordinates (dense_heads/detr3d_head_pe.py:427-491: pixel centre x depth bin -> lidar frame by the camera's img2lidar
// matrix -> pc_range units -> inverse_sigmoid) - a function of 12 matrix entries and the pixel index, generated in registers when the
// tile's rows would have been loaded (gd4d_frustum_pe_input_fwd's arithmetic, operation for operation; the (R, S, 192) tensor it wrote
// and this kernel re-read - 568 MB each way at 24 cameras - does not exist).  Lane (row, kg) holds input channels 96 kg + j,
// j = 8 step + e: depth bin 32 kg + j / 3, axis j % 3 - so which axis a register holds is known at compile time; W1's image is made
// from W1 with its columns in that order (ops.mlp2_frustum_image).
struct Mlp2Fr {
  const float* i2l;         // (R, 16) img2lidar
  int w[4], h[4], hw[4];
  int start[5];
  int S;
  float pad_h, pad_w, depth_start, bin_size;
  float lo[3], span[3];
};

__device__ __forceinline__ float ml_inv_sigmoid_fast(float x) {      // gd4d_head_pe.hip's inv_sigmoid_fast
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float a = fminf(fmaxf(x, 1e-5f), 1.f), b = fminf(fmaxf(1.f - x, 1e-5f), 1.f);
  return (__builtin_amdgcn_logf(a) - __builtin_amdgcn_logf(b)) * 0.69314718055994530942f;
}

__device__ __forceinline__ Mlp2Row ml_row_of(const Mlp2Se& q, int m) {
  const int r = m / q.S;
  const int rem = m - r * q.S;
  const int l = (rem >= q.start[1]) + (rem >= q.start[2]) + (rem >= q.start[3]);
  const int hw = q.hw[l];
  return Mlp2Row{q.feat[l] + (size_t)r * ML_N2 * hw, q.out[l] + (size_t)r * hw * ML_N2, rem - q.start[l], hw};
}

__device__ __forceinline__ unsigned ml_cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}
// 8 floats -> 16 bytes of bf16 "hi" halves and 16 bytes of bf16 "lo" (residual) halves
__device__ __forceinline__ void ml_split8(const float (&v)[8], ml_u32x4& h, ml_u32x4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = ml_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    const float ra = v[2 * i] - __uint_as_float(hh[i] << 16);              // exact: hi is a rounding of the value
    const float rb = v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u);
    ll[i] = ml_cvt_pk_bf16(ra, rb);
  }
  h = ml_u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = ml_u32x4{ll[0], ll[1], ll[2], ll[3]};
}
__device__ __forceinline__ ml_bf16x8 ml_frag(const ml_u32x4& v) { return __builtin_bit_cast(ml_bf16x8, v); }

// hidden unit (inside its chunk of 32) that accumulator register r of a lane with k-group kg holds after phase A: the C / D row
// of v_mfma_f32_32x32x16 - and, read as r = 8 s + e, the k-slot e of phase B's step s
__host__ __device__ __forceinline__ int ml_hidden_of(int kg, int r) { return 4 * kg + (r & 3) + 8 * (r >> 2); }

template <int STEPS1, bool SE = false, bool FR = false>
__global__ __launch_bounds__(ML_THREADS, 1) void mlp2_kernel(const Mlp2Params p, const Mlp2Se q, const Mlp2Fr fr) {
  static_assert(!FR || (STEPS1 == 12 && !SE), "the frustum form: 3 x 64 depth bins");
  extern __shared__ __attribute__((aligned(16))) char ml_smem[];
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef const __attribute__((address_space(1))) void glb_void_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, kg = lane >> 5;
  constexpr int S1 = STEPS1 * 2048 + 1024;             // bytes of a chunk of W1's image: [step][plane][1 KB], then 1 KB holding b1's 32 entries
  constexpr int STAGE = S1 + ML_S2;
  const int nchunks = p.H / ML_HC;
  // A workgroup walks tiles blockIdx, blockIdx + grid, ..: the next tile's rows of X are requested when this tile's products start
  // and the weight stages form a ring across tiles (the stage after a tile's last chunk is the next tile's chunk 0).  Launched with
  // one workgroup per tile (see mlp2_grid).
  const int ntiles = (p.M + ML_BM - 1) / ML_BM;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;

  // A stage = a chunk's fragments, 1 KB each: a wave instruction (LDS-DMA) moves one.  Wave w moves pieces w, w + 4, ..; issuing
  // one costs 60 - 185 cycles, so the next chunk's pieces go out one at a time BETWEEN the MFMA groups of this chunk's phase B.
  constexpr int n1 = S1 >> 10, npieces = STAGE >> 10;
  constexpr int PER_WAVE = (npieces + ML_WAVES - 1) / ML_WAVES;
  auto stage_piece = [&](int c, int buf, int k) {        // k-th piece of this wave
    const int i = wave + ML_WAVES * k;
    if (i >= npieces) return;
    const char* src = i < n1 ? p.w1img + (size_t)c * S1 + ((size_t)i << 10) : p.w2img + (size_t)c * ML_S2 + ((size_t)(i - n1) << 10);
    __builtin_amdgcn_global_load_lds((glb_void_t*)(src + lane * 16), (lds_void_t*)(ml_smem + buf * STAGE + (i << 10)), 16, 0, 0);
  };
  for (int k = 0; k < PER_WAVE; ++k) stage_piece(0, 0, k);
  int cc = 0;                                          // chunks this workgroup has walked: chunk cc's stage is buffer cc & 1

  // this lane's row of X: step st = channels 16 st + 8 kg .. + 7 (requested a tile ahead, split into bf16 hi / lo when its tile starts)
  float xr[STEPS1][8];
  auto load_x = [&](int tl) {
    const int m = min(tl * ML_BM + wave * 32 + l32, p.M - 1);
    if (FR) {
      const int r = m / fr.S;
      const int rem = m - r * fr.S;
      const int l = (rem >= fr.start[1]) + (rem >= fr.start[2]) + (rem >= fr.start[3]);
      const int pix = rem - fr.start[l];
      const int W = fr.w[l], H = fr.h[l];
      const int y = pix / W, x = pix - y * W;
      const float* mt = fr.i2l + (size_t)r * 16;
      float mm[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) mm[i] = mt[i];
      const float ch = ((float)y * fr.pad_h) / (float)H;          // torch.arange(H).float() * pad_h / H (:440-441)
      const float cw = ((float)x * fr.pad_w) / (float)W;
      const float fi0 = kg ? 32.0f : 0.0f;
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) {
        const float fi = fi0 + (float)dd;
        const float depth = fr.depth_start + (fr.bin_size * fi) * (fi + 1.0f);      // :450-453
        const float sdep = fmaxf(depth, 1e-5f);
        const float px = cw * sdep, py = ch * sdep;                                  // :458
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float v = ((mm[4 * k] * px + mm[4 * k + 1] * py) + mm[4 * k + 2] * depth) + mm[4 * k + 3];   // :469
          const float c = (v - fr.lo[k]) / fr.span[k];                               // :470-475
          const int j = 3 * dd + k;
          xr[j >> 3][j & 7] = ml_inv_sigmoid_fast(c);                               // :480-481
        }
      }
    } else if (SE) {
      // NCHW maps: channel c of this lane's pixel is hw floats further - a wave instruction reads 32 consecutive pixels of two channels
      const Mlp2Row g = ml_row_of(q, m);
      const float* xc = g.f + (size_t)(8 * kg) * g.hw + g.pix;
#pragma unroll
      for (int st = 0; st < STEPS1; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[st][e] = xc[(size_t)(16 * st + e) * g.hw];
    } else {
      const float* xrow = p.x + (size_t)m * p.ldx + 8 * kg;
#pragma unroll
      for (int st = 0; st < STEPS1; ++st) {
        const float4 a = *reinterpret_cast<const float4*>(xrow + 16 * st), b = *reinterpret_cast<const float4*>(xrow + 16 * st + 4);
        xr[st][0] = a.x; xr[st][1] = a.y; xr[st][2] = a.z; xr[st][3] = a.w;
        xr[st][4] = b.x; xr[st][5] = b.y; xr[st][6] = b.z; xr[st][7] = b.w;
      }
    }
  };
  load_x(tile);

  for (;;) {
  const int m0 = tile * ML_BM + wave * 32;
  ml_u32x4 xh[STEPS1], xl[STEPS1];
#pragma unroll
  for (int st = 0; st < STEPS1; ++st) ml_split8(xr[st], xh[st], xl[st]);
  const int next_tile = tile + (int)gridDim.x;
  const bool last_tile = next_tile >= ntiles;
  if (!last_tile) load_x(next_tile);

  ml_f32x16 acc[ML_NT];
#pragma unroll
  for (int t = 0; t < ML_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int c = 0; c < nchunks; ++c, ++cc) {
    // A wave's pieces of a stage go out W1's first (7 - 9 of them), then W2's (8 per wave: 32 pieces, 4 waves) and complete in that
    // order: the first product only needs W1's - waiting for all of them here exposed the round trip of pieces issued two or three
    // MFMA groups ago (ML_SPLIT_WAIT=0: the single wait).
    if (ML_SPLIT_WAIT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of this chunk's stage has landed
    __syncthreads();                                   // ... everybody's has; and everybody is done with the other buffer
    const bool more = !(last_tile && c + 1 == nchunks);
    const int cn = c + 1 < nchunks ? c + 1 : 0;        // (the ring: behind a tile's last chunk comes the next tile's first)
    const char* s1 = ml_smem + (cc & 1) * STAGE;
    const char* s2 = s1 + S1;
    // ---- phase A: H^T = W1c X^T (A operand = W1 fragment, B operand = X fragment) ----
    // Two accumulators, even and odd steps: an MFMA that follows a gap in the instruction stream (the LDS requests, a DMA piece)
    // then depends on a result two steps old, not on the one just issued (a dependent MFMA behind a gap waits out the full latency).
    ml_f32x16 h, h_odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h[r] = 0.f; h_odd[r] = 0.f; }
    // (one wave per SIMD: nobody else hides an LDS round trip, so a step's fragments are requested a step ahead and the matrix
    //  pipe works on step st while they travel; the next chunk's stage goes out a piece per step - the other buffer is free
    //  since the barrier above)
    const char* f1 = s1 + lane * 16;
    ml_u32x4 wh = *reinterpret_cast<const ml_u32x4*>(f1), wl = *reinterpret_cast<const ml_u32x4*>(f1 + 1024);
#pragma unroll
    for (int st = 0; st < STEPS1; ++st) {
      ml_u32x4 nh = wh, nl = wl;
      if (st + 1 < STEPS1) {
        nh = *reinterpret_cast<const ml_u32x4*>(f1 + (st + 1) * 2048);
        nl = *reinterpret_cast<const ml_u32x4*>(f1 + (st + 1) * 2048 + 1024);
      }
      if (ML_DMA_IN_A && more && st < PER_WAVE) stage_piece(cn, (cc + 1) & 1, st);
      if (ML_TWO_ACC && (st & 1)) {
        h_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wl), ml_frag(xh[st]), h_odd, 0, 0, 0);
        h_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wh), ml_frag(xl[st]), h_odd, 0, 0, 0);
        h_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wh), ml_frag(xh[st]), h_odd, 0, 0, 0);
      } else {
        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wl), ml_frag(xh[st]), h, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wh), ml_frag(xl[st]), h, 0, 0, 0);
        h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wh), ml_frag(xh[st]), h, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);               // (keep the requests in front of the MFMAs they travel under)
      wh = nh; wl = nl;
    }
    if (STEPS1 > 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) h[r] += h_odd[r];
    }
    // bias, ReLU, hi / lo split: registers 8 s .. 8 s + 7 are the A operand of phase B's step s; b1's entries of registers
    // 4 j .. 4 j + 3 are the consecutive hidden units 4 kg + 8 j + (0 .. 3)
    ml_u32x4 ah[2], al[2];
    {
      const float* b1c = reinterpret_cast<const float*>(s1 + STEPS1 * 2048);
      float4 bq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const float4*>(b1c + 4 * kg + 8 * j);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 ba = bq[2 * s], bb = bq[2 * s + 1];
        const float v[8] = {fmaxf(h[8 * s] + ba.x, 0.f), fmaxf(h[8 * s + 1] + ba.y, 0.f), fmaxf(h[8 * s + 2] + ba.z, 0.f),
                            fmaxf(h[8 * s + 3] + ba.w, 0.f), fmaxf(h[8 * s + 4] + bb.x, 0.f), fmaxf(h[8 * s + 5] + bb.y, 0.f),
                            fmaxf(h[8 * s + 6] + bb.z, 0.f), fmaxf(h[8 * s + 7] + bb.w, 0.f)};
        ml_split8(v, ah[s], al[s]);
      }
    }
    // ---- phase B: out += H W2c^T (step-major: the first step's MFMAs run while the second step's operand is still being split) ----
    if (ML_SPLIT_WAIT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // W2's pieces of this chunk
      __syncthreads();
    }
    constexpr int GROUPS = 2 * ML_NT;
    const char* f2 = s2 + lane * 16;
    ml_u32x4 vh = *reinterpret_cast<const ml_u32x4*>(f2), vl = *reinterpret_cast<const ml_u32x4*>(f2 + 1024);
#pragma unroll
    for (int grp = 0; grp < GROUPS; ++grp) {
      const int s = grp / ML_NT, t = grp % ML_NT;
      ml_u32x4 nh = vh, nl = vl;
      if (grp + 1 < GROUPS) {                          // the next group's fragments: [tile][step][plane]
        const int s_n = (grp + 1) / ML_NT, t_n = (grp + 1) % ML_NT;
        nh = *reinterpret_cast<const ml_u32x4*>(f2 + ((t_n * 2 + s_n) * 2) * 1024);
        nl = *reinterpret_cast<const ml_u32x4*>(f2 + ((t_n * 2 + s_n) * 2) * 1024 + 1024);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(al[s]), ml_frag(vh), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(ah[s]), ml_frag(vl), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(ah[s]), ml_frag(vh), acc[t], 0, 0, 0);
      // the next chunk's stage, a piece per MFMA group (the other buffer: nobody reads it until the next barrier)
      if (more && (ML_DMA_IN_A ? STEPS1 : 0) + grp < PER_WAVE) stage_piece(cn, (cc + 1) & 1, (ML_DMA_IN_A ? STEPS1 : 0) + grp);
      __builtin_amdgcn_sched_barrier(0);
      vh = nh; vl = nl;
    }
    if (more)
      for (int k = (ML_DMA_IN_A ? STEPS1 : 0) + GROUPS; k < PER_WAVE; ++k) stage_piece(cn, (cc + 1) & 1, k);
  }
  // C / D of 32x32x16: column n = lane & 31, rows 4 (lane >> 5) + (r & 3) + 8 (r >> 2)
  if (SE) {
    // a lane holds, per tile, 4 groups of 4 consecutive rows (pixels) of ONE channel: the channels-last operands and the result
    // are coalesced over the lanes (consecutive channels), the map's four pixels are one 16-byte read per lane (any 4-byte
    // alignment; a group that crosses the end of a level or of the rows goes pixel by pixel)
    typedef float ml_f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mj = m0 + 4 * kg + 8 * j;
      if (mj >= p.M) continue;
      const Mlp2Row g = ml_row_of(q, mj);
      const bool whole = g.pix + 3 < g.hw && mj + 3 < p.M;
      if (whole) {
        // every read of the group's 8 tiles is requested before the first is used (one wave per SIMD: nothing else hides a round trip)
        ml_f4u f4[ML_NT];
        float pv[ML_NT][4], sv[ML_NT][4];
#pragma unroll
        for (int t = 0; t < ML_NT; ++t) {
          const int n = 32 * t + l32;
          f4[t] = *reinterpret_cast<const ml_f4u*>(g.f + (size_t)n * g.hw + g.pix);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const size_t o = (size_t)(mj + i) * ML_N2 + n;
            pv[t][i] = q.pe[o]; sv[t][i] = q.sine[o];
          }
        }
#pragma unroll
        for (int t = 0; t < ML_NT; ++t) {
          const int n = 32 * t + l32;
          const float bv = p.b2 ? p.b2[n] : 0.f;
          const float fv[4] = {f4[t].x, f4[t].y, f4[t].z, f4[t].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float gate = acc[t][4 * j + i] + bv;
            g.o[(size_t)(g.pix + i) * ML_N2 + n] = fv[i] + (pv[t][i] * ml_sigmoid(gate) + sv[t][i]);
          }
        }
        continue;
      }
      for (int i = 0; i < 4; ++i) {                       // a group across the end of a level / camera / the rows: pixel by pixel
        const int m = mj + i;
        if (m >= p.M) break;
        const Mlp2Row gi = ml_row_of(q, m);
#pragma unroll
        for (int t = 0; t < ML_NT; ++t) {
          const int n = 32 * t + l32;
          const size_t o = (size_t)m * ML_N2 + n;
          float a = 0.f;                                   // acc[t][4 j + i] with a runtime i: a select, not an indexed register
#pragma unroll
          for (int k = 0; k < 4; ++k) a = i == k ? acc[t][4 * j + k] : a;
          const float gate = a + (p.b2 ? p.b2[n] : 0.f);
          gi.o[(size_t)gi.pix * ML_N2 + n] = gi.f[(size_t)n * gi.hw + gi.pix] + (q.pe[o] * ml_sigmoid(gate) + q.sine[o]);
        }
      }
    }
  } else {
#pragma unroll
  for (int t = 0; t < ML_NT; ++t) {
    const int n = 32 * t + l32;
    const float bv = p.b2 ? p.b2[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + ml_hidden_of(kg, r);
      if (m < p.M) p.out[(size_t)m * p.ldo + n] = acc[t][r] + bv;
    }
  }
  }
  if (last_tile) break;
  tile = next_tile;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// gd4d_mlp2_pe_se_fwd: BOTH MLPs of the head's position embedding and the fuse in one kernel, for cameras whose embedding is not kept
// (the past-frame cameras of the temporal pattern - their matrices carry the ego motion and change with every sample - or every
// camera): per tile of 128 pixels
//     pe   = position_encoder(frustum coordinates)          192 -> 1024 -> 256, the inputs generated in registers (Mlp2Fr)
//     gate = conv_expand(relu(conv_reduce(feat)))           256 ->  256 -> 256, the inputs read from the NCHW maps (Mlp2Se)
//     out  = feat + (pe * sigmoid(gate) + sine)             channels-last, the layout the decoder gathers in place
// The two products share the accumulator layout (a lane = one output channel of a tile of 32, 16 rows), so `pe` never leaves the
// registers: 128 of them wait while the second MLP runs.  Against gd4d_mlp2_frustum_fwd + gd4d_mlp2_se_fuse_fwd the (R S, 256)
// embedding between them - 757 MB written and read at 24 cameras - does not exist.  Registers: 128 (pe) + 128 (gate) + 128 (the
// maps' rows, bf16 hi / lo) + fragments: one wave per SIMD, as both kernels it is made of.  LDS: the larger of the two stage pairs.
// A phase = one MLP's walk over its hidden chunks (mlp2_kernel's chunk body, one tile per workgroup, no ring across tiles).
template <int STEPS1>
__device__ __forceinline__ void ml_phase(const char* __restrict__ w1img, const char* __restrict__ w2img, const int nchunks, char* smem,
                                         const ml_u32x4 (&xh)[STEPS1], const ml_u32x4 (&xl)[STEPS1], ml_f32x16 (&acc)[ML_NT],
                                         const int wave, const int lane) {
  typedef __attribute__((address_space(3))) void lds_void_t;
  typedef const __attribute__((address_space(1))) void glb_void_t;
  constexpr int S1 = STEPS1 * 2048 + 1024;
  constexpr int STAGE = S1 + ML_S2;
  constexpr int n1 = S1 >> 10, npieces = STAGE >> 10;
  constexpr int PER_WAVE = (npieces + ML_WAVES - 1) / ML_WAVES;
  const int kg = lane >> 5;
  auto stage_piece = [&](int c, int buf, int k) {
    const int i = wave + ML_WAVES * k;
    if (i >= npieces) return;
    const char* src = i < n1 ? w1img + (size_t)c * S1 + ((size_t)i << 10) : w2img + (size_t)c * ML_S2 + ((size_t)(i - n1) << 10);
    __builtin_amdgcn_global_load_lds((glb_void_t*)(src + lane * 16), (lds_void_t*)(smem + buf * STAGE + (i << 10)), 16, 0, 0);
  };
  __syncthreads();                                     // (the previous phase's stages are done with)
  for (int k = 0; k < PER_WAVE; ++k) stage_piece(0, 0, k);
  for (int c = 0; c < nchunks; ++c) {
    if (ML_SPLIT_WAIT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (W1's pieces: see mlp2_kernel)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = c + 1 < nchunks;
    const char* s1 = smem + (c & 1) * STAGE;
    const char* s2 = s1 + S1;
    ml_f32x16 h;
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = 0.f;
    const char* f1 = s1 + lane * 16;
    ml_u32x4 wh = *reinterpret_cast<const ml_u32x4*>(f1), wl = *reinterpret_cast<const ml_u32x4*>(f1 + 1024);
#pragma unroll
    for (int st = 0; st < STEPS1; ++st) {
      ml_u32x4 nh = wh, nl = wl;
      if (st + 1 < STEPS1) {
        nh = *reinterpret_cast<const ml_u32x4*>(f1 + (st + 1) * 2048);
        nl = *reinterpret_cast<const ml_u32x4*>(f1 + (st + 1) * 2048 + 1024);
      }
      h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wl), ml_frag(xh[st]), h, 0, 0, 0);
      h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wh), ml_frag(xl[st]), h, 0, 0, 0);
      h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(wh), ml_frag(xh[st]), h, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      wh = nh; wl = nl;
    }
    ml_u32x4 ah[2], al[2];
    {
      const float* b1c = reinterpret_cast<const float*>(s1 + STEPS1 * 2048);
      float4 bq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const float4*>(b1c + 4 * kg + 8 * j);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 ba = bq[2 * s], bb = bq[2 * s + 1];
        const float v[8] = {fmaxf(h[8 * s] + ba.x, 0.f), fmaxf(h[8 * s + 1] + ba.y, 0.f), fmaxf(h[8 * s + 2] + ba.z, 0.f),
                            fmaxf(h[8 * s + 3] + ba.w, 0.f), fmaxf(h[8 * s + 4] + bb.x, 0.f), fmaxf(h[8 * s + 5] + bb.y, 0.f),
                            fmaxf(h[8 * s + 6] + bb.z, 0.f), fmaxf(h[8 * s + 7] + bb.w, 0.f)};
        ml_split8(v, ah[s], al[s]);
      }
    }
    if (ML_SPLIT_WAIT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    constexpr int GROUPS = 2 * ML_NT;
    const char* f2 = s2 + lane * 16;
    // group g's fragments: [tile][step][plane]; requested ML_AHEAD groups before their MFMAs
    auto frag_at = [&](int g) -> const char* { return f2 + (((g % ML_NT) * 2 + g / ML_NT) * 2) * 1024; };
    ml_u32x4 qh[ML_AHEAD + 1], ql[ML_AHEAD + 1];
#pragma unroll
    for (int a = 0; a < ML_AHEAD; ++a) {
      qh[a] = *reinterpret_cast<const ml_u32x4*>(frag_at(a));
      ql[a] = *reinterpret_cast<const ml_u32x4*>(frag_at(a) + 1024);
    }
#pragma unroll
    for (int grp = 0; grp < GROUPS; ++grp) {
      const int s = grp / ML_NT, t = grp % ML_NT;
      if (grp + ML_AHEAD < GROUPS) {
        qh[ML_AHEAD] = *reinterpret_cast<const ml_u32x4*>(frag_at(grp + ML_AHEAD));
        ql[ML_AHEAD] = *reinterpret_cast<const ml_u32x4*>(frag_at(grp + ML_AHEAD) + 1024);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(al[s]), ml_frag(qh[0]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(ah[s]), ml_frag(ql[0]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml_frag(ah[s]), ml_frag(qh[0]), acc[t], 0, 0, 0);
      if (more && grp < PER_WAVE) stage_piece(c + 1, (c + 1) & 1, grp);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < ML_AHEAD; ++a) { qh[a] = qh[a + 1]; ql[a] = ql[a + 1]; }
    }
    if (more)
      for (int k = GROUPS; k < PER_WAVE; ++k) stage_piece(c + 1, (c + 1) & 1, k);
  }
}

struct Mlp2PeSe {
  const char* pe_w1; const char* pe_w2; const float* pe_b2; int pe_h;       // position_encoder: gd4d_mlp2_image of (W1 permuted, b1, W2)
  const char* se_w1; const char* se_w2; const float* se_b2; int se_h;       // SELayer: conv_reduce / conv_expand
  float* pe_out;                                                            // (R S, 256) or NULL: the embedding, if somebody keeps it
  int M;
};

__global__ __launch_bounds__(ML_THREADS, 1) void mlp2_pe_se_kernel(const Mlp2PeSe p, const Mlp2Se q, const Mlp2Fr fr) {
  extern __shared__ __attribute__((aligned(16))) char ml_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, kg = lane >> 5;
  const int m0 = blockIdx.x * ML_BM + wave * 32;
  const int m = min(m0 + l32, p.M - 1);
  // ---- position_encoder(frustum): the inputs generated (mlp2_kernel's frustum form) ----
  ml_f32x16 pe[ML_NT];
#pragma unroll
  for (int t = 0; t < ML_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) pe[t][r] = 0.f;
  {
    ml_u32x4 fh[12], fl[12];
    {
      const int r = m / fr.S;
      const int rem = m - r * fr.S;
      const int l = (rem >= fr.start[1]) + (rem >= fr.start[2]) + (rem >= fr.start[3]);
      const int pix = rem - fr.start[l];
      const int W = fr.w[l], H = fr.h[l];
      const int y = pix / W, x = pix - y * W;
      const float* mt = fr.i2l + (size_t)r * 16;
      float mm[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) mm[i] = mt[i];
      const float ch = ((float)y * fr.pad_h) / (float)H;
      const float cw = ((float)x * fr.pad_w) / (float)W;
      const float fi0 = kg ? 32.0f : 0.0f;
      float xf[12][8];
#pragma unroll
      for (int dd = 0; dd < 32; ++dd) {
        const float fi = fi0 + (float)dd;
        const float depth = fr.depth_start + (fr.bin_size * fi) * (fi + 1.0f);
        const float sdep = fmaxf(depth, 1e-5f);
        const float px = cw * sdep, py = ch * sdep;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float v = ((mm[4 * k] * px + mm[4 * k + 1] * py) + mm[4 * k + 2] * depth) + mm[4 * k + 3];
          const float c = (v - fr.lo[k]) / fr.span[k];
          const int j = 3 * dd + k;
          xf[j >> 3][j & 7] = ml_inv_sigmoid_fast(c);
        }
      }
#pragma unroll
      for (int st = 0; st < 12; ++st) ml_split8(xf[st], fh[st], fl[st]);
    }
    ml_phase<12>(p.pe_w1, p.pe_w2, p.pe_h / ML_HC, ml_smem, fh, fl, pe, wave, lane);
  }
#pragma unroll
  for (int t = 0; t < ML_NT; ++t) {
    const float bv = p.pe_b2 ? p.pe_b2[32 * t + l32] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) pe[t][r] += bv;
  }
  if (p.pe_out) {
#pragma unroll
    for (int t = 0; t < ML_NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mr = m0 + ml_hidden_of(kg, r);
        if (mr < p.M) p.pe_out[(size_t)mr * ML_N2 + 32 * t + l32] = pe[t][r];
      }
  }
  // ---- the gate ----
  ml_f32x16 acc[ML_NT];
#pragma unroll
  for (int t = 0; t < ML_NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  {
    float xr[16][8];
    {
      const Mlp2Row g = ml_row_of(q, m);
      const float* xc = g.f + (size_t)(8 * kg) * g.hw + g.pix;
#pragma unroll
      for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[st][e] = xc[(size_t)(16 * st + e) * g.hw];
    }
    ml_u32x4 xh[16], xl[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) ml_split8(xr[st], xh[st], xl[st]);
    ml_phase<16>(p.se_w1, p.se_w2, p.se_h / ML_HC, ml_smem, xh, xl, acc, wave, lane);
  }
  // ---- out = feat + (pe * sigmoid(gate) + sine): mlp2_kernel's SE epilogue with pe in registers ----
  typedef float ml_f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int mj = m0 + 4 * kg + 8 * j;
    if (mj >= p.M) continue;
    const Mlp2Row g = ml_row_of(q, mj);
    const bool whole = g.pix + 3 < g.hw && mj + 3 < p.M;
    if (whole) {
      ml_f4u f4[ML_NT];
      float sv[ML_NT][4];
#pragma unroll
      for (int t = 0; t < ML_NT; ++t) {
        const int n = 32 * t + l32;
        f4[t] = *reinterpret_cast<const ml_f4u*>(g.f + (size_t)n * g.hw + g.pix);
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[t][i] = q.sine[(size_t)(mj + i) * ML_N2 + n];
      }
#pragma unroll
      for (int t = 0; t < ML_NT; ++t) {
        const int n = 32 * t + l32;
        const float bv = p.se_b2 ? p.se_b2[n] : 0.f;
        const float fv[4] = {f4[t].x, f4[t].y, f4[t].z, f4[t].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gate = acc[t][4 * j + i] + bv;
          g.o[(size_t)(g.pix + i) * ML_N2 + n] = fv[i] + (pe[t][4 * j + i] * ml_sigmoid(gate) + sv[t][i]);
        }
      }
      continue;
    }
    for (int i = 0; i < 4; ++i) {                         // a group across the end of a level / camera / the rows: pixel by pixel
      const int mi = mj + i;
      if (mi >= p.M) break;
      const Mlp2Row gi = ml_row_of(q, mi);
#pragma unroll
      for (int t = 0; t < ML_NT; ++t) {
        const int n = 32 * t + l32;
        float a = 0.f, pv = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a = i == k ? acc[t][4 * j + k] : a; pv = i == k ? pe[t][4 * j + k] : pv; }
        const float gate = a + (p.se_b2 ? p.se_b2[n] : 0.f);
        gi.o[(size_t)gi.pix * ML_N2 + n] = gi.f[(size_t)n * gi.hw + gi.pix] + (pv * ml_sigmoid(gate) + q.sine[(size_t)mi * ML_N2 + n]);
      }
    }
  }
}

// One thread per 16-byte piece of the two images.
//   W1 image  [chunk]{[step of 16 inputs][plane hi / lo][lane 64][8 bf16], 1 KB: b1[32 chunk .. + 32]}:   lane (l32, kg), element e = W1[32 chunk + l32][16 step + 8 kg + e]
//   W2 image  [chunk][tile of 32 outputs][step s][plane][lane 64][8 bf16]:  element e = W2[32 tile + l32][32 chunk + hidden_of(kg, 8 s + e)]
__global__ __launch_bounds__(256) void mlp2_image_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ b1,
                                                         char* __restrict__ img1, char* __restrict__ img2, int H, int K1) {
  const int steps1 = K1 / 16, nchunks = H / ML_HC;
  const size_t S1 = (size_t)steps1 * 2048 + 1024;
  const long long p1 = (long long)nchunks * steps1 * 64, p2 = (long long)nchunks * ML_NT * 2 * 64;   // pieces per plane pair
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float v[8];
  char* dst;
  if (i < p1) {
    const int lane = (int)(i & 63);
    const long long cs = i >> 6;                       // chunk * steps1 + step
    const int st = (int)(cs % steps1), c = (int)(cs / steps1);
    const int l32 = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w1[(size_t)(ML_HC * c + l32) * K1 + 16 * st + 8 * kg + e];
    dst = img1 + (size_t)c * S1 + (size_t)st * 2048 + lane * 16;
    if (st == 0 && lane < 8) {                         // the chunk's 32 entries of b1 behind its fragments
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b1) bv = *reinterpret_cast<const float4*>(b1 + ML_HC * c + 4 * lane);
      *reinterpret_cast<float4*>(img1 + (size_t)c * S1 + (size_t)steps1 * 2048 + lane * 16) = bv;
    }
  } else if (i < p1 + p2) {
    const long long j = i - p1;
    const int lane = (int)(j & 63);
    const long long cts = j >> 6;                      // (chunk * NT + tile) * 2 + s
    const int s = (int)(cts & 1), t = (int)((cts >> 1) % ML_NT), c = (int)((cts >> 1) / ML_NT);
    const int l32 = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w2[(size_t)(32 * t + l32) * H + ML_HC * c + ml_hidden_of(kg, 8 * s + e)];
    dst = img2 + (size_t)cts * 2048 + lane * 16;
  } else {
    return;
  }
  ml_u32x4 h, l;
  ml_split8(v, h, l);
  *reinterpret_cast<ml_u32x4*>(dst) = h;
  *reinterpret_cast<ml_u32x4*>(dst + 1024) = l;
}

// One workgroup per tile.  (The kernel walks tiles blockIdx, blockIdx + grid, .. with the next tile's rows requested a tile ahead and the
// weight stages as a ring across tiles: a grid of one workgroup per compute unit measured the same - 2.19 / 0.98 / 1.33 ms against
// 2.19 / 1.01 / 1.31 ms for the three shapes of tools/bench_mlp2.py - so the launch, the rows' round trip and the first stage are not
// what a tile waits for; the hardware's own hand-out of tiles is kept.)
static int mlp2_grid(int M) { return (M + ML_BM - 1) / ML_BM; }

static bool mlp2_shape_ok(int K1, int H, int N2) {          // (K1 / 16 is a template argument: the instantiated step counts)
  const int st = K1 / 16;
  return K1 > 0 && K1 % 16 == 0 && (st == 1 || st == 2 || st == 4 || st == 8 || st == 12 || st == 16) && H > 0 && H % ML_HC == 0 && N2 == ML_N2;
}

}  // namespace gd4d

extern "C" size_t gd4d_mlp2_image_bytes(int K1, int H, int N2) {
  if (!gd4d::mlp2_shape_ok(K1, H, N2)) return 0;
  return (size_t)H * K1 * 4 + (size_t)(H / gd4d::ML_HC) * 1024 + (size_t)N2 * H * 4;      // bf16 hi + lo of both weights, b1 per chunk
}

extern "C" int gd4d_mlp2_image(const float* w1, const float* b1, const float* w2, int K1, int H, int N2, void* image, void* stream) {
  using namespace gd4d;
  if (!w1 || !w2 || !image) return GD4D_EINVAL;
  if (!mlp2_shape_ok(K1, H, N2)) return GD4D_EUNSUPPORTED;
  if (!aligned16(image)) return GD4D_EALIGN;
  const long long pieces = (long long)(H / ML_HC) * (K1 / 16) * 64 + (long long)(H / ML_HC) * ML_NT * 2 * 64;
  char* img1 = static_cast<char*>(image);
  if (b1 && !aligned16(b1)) return GD4D_EALIGN;
  char* img2 = img1 + (size_t)H * K1 * 4 + (size_t)(H / ML_HC) * 1024;
  hipLaunchKernelGGL(mlp2_image_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w1, w2, b1,
                     img1, img2, H, K1);
  return check_launch();
}

extern "C" int gd4d_mlp2_bf16x3_fwd(const float* x, const void* image, const float* b2, float* out, int M, int K1,
                                    int H, int N2, int ldx, int ldo, void* stream) {
  using namespace gd4d;
  if (!x || !image || !out || M <= 0 || ldx < K1 || ldo < N2) return GD4D_EINVAL;
  if (!mlp2_shape_ok(K1, H, N2) || ldx % 4 != 0) return GD4D_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(image)) return GD4D_EALIGN;
  const char* img1 = static_cast<const char*>(image);
  Mlp2Params p{x, img1, img1 + (size_t)H * K1 * 4 + (size_t)(H / ML_HC) * 1024, b2, out, M, K1, H, ldx, ldo};
  const int lds = 2 * ((K1 / 16) * 2048 + 1024 + ML_S2);
  const Mlp2Se none{};
  const Mlp2Fr nofr{};
  auto go = [&](auto kern) -> int {
    if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return GD4D_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(mlp2_grid(M)), dim3(ML_THREADS), lds, static_cast<hipStream_t>(stream), p, none, nofr);
    return check_launch();
  };
  switch (K1 / 16) {
    case 1: return go(mlp2_kernel<1>);
    case 2: return go(mlp2_kernel<2>);
    case 4: return go(mlp2_kernel<4>);
    case 8: return go(mlp2_kernel<8>);
    case 12: return go(mlp2_kernel<12>);
    default: return go(mlp2_kernel<16>);
  }
}

extern "C" int gd4d_mlp2_se_fuse_fwd(const void* const* feats, const int32_t* level_hw, int L, int R, const void* image, const float* b2,
                                     const float* pe, const float* sine, void* const* outs, int C, int H, void* stream) {
  using namespace gd4d;
  if (!feats || !level_hw || !image || !pe || !sine || !outs || L <= 0 || R <= 0) return GD4D_EINVAL;
  if (L > 4 || C != ML_N2 || !mlp2_shape_ok(C, H, C)) return GD4D_EUNSUPPORTED;
  if (!aligned16(image)) return GD4D_EALIGN;
  Mlp2Se q{};
  long long S = 0;
  for (int l = 0; l < 4; ++l) {
    const int ll = l < L ? l : 0;
    if (l < L && (!feats[l] || !outs[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0)) return GD4D_EINVAL;
    q.feat[l] = static_cast<const float*>(feats[ll]); q.out[l] = static_cast<float*>(outs[ll]);
    q.hw[l] = level_hw[2 * ll] * level_hw[2 * ll + 1];
    q.start[l] = (int)S;
    if (l < L) S += q.hw[l];
  }
  for (int l = L; l <= 4; ++l) q.start[l] = (int)S;
  if (S * R >= (1ll << 31)) return GD4D_EUNSUPPORTED;
  q.S = (int)S; q.pe = pe; q.sine = sine;
  const int M = (int)(S * R), K1 = C;
  const char* img1 = static_cast<const char*>(image);
  Mlp2Params p{nullptr, img1, img1 + (size_t)H * K1 * 4 + (size_t)(H / ML_HC) * 1024, b2, nullptr, M, K1, H, 0, 0};
  const int lds = 2 * ((K1 / 16) * 2048 + 1024 + ML_S2);
  auto kern = mlp2_kernel<16, true>;
  if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return GD4D_ELAUNCH;
  const Mlp2Fr nofr{};
  hipLaunchKernelGGL(kern, dim3(mlp2_grid(M)), dim3(ML_THREADS), lds, static_cast<hipStream_t>(stream), p, q, nofr);
  return check_launch();
}

// out (R S, 256; row stride ldo) = position_encoder(frustum coordinates of every pixel of every level of every camera): see Mlp2Fr.
// `image`: gd4d_mlp2_image of (W1 with its columns permuted - column 16 st + 8 kg + e of the image's W1 = column 96 kg + 8 st + e of
// the module's -, b1, W2).  D must be 64 (K1 = 192).
extern "C" int gd4d_mlp2_frustum_fwd(const float* img2lidar, const int32_t* level_hw, int L, int R, float pad_h, float pad_w, int D,
                                     float depth_start, const double* pc_range, const void* image, const float* b2, float* out,
                                     int H, int ldo, void* stream) {
  using namespace gd4d;
  if (!img2lidar || !level_hw || !pc_range || !image || !out || L <= 0 || R <= 0 || ldo < ML_N2) return GD4D_EINVAL;
  const int K1 = 3 * D;
  if (L > 4 || D != 64 || !mlp2_shape_ok(K1, H, ML_N2)) return GD4D_EUNSUPPORTED;
  if (!aligned16(image)) return GD4D_EALIGN;
  Mlp2Fr fr{};
  long long S = 0;
  for (int l = 0; l < 4; ++l) {
    const int ll = l < L ? l : 0;
    if (l < L && (level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0)) return GD4D_EINVAL;
    fr.h[l] = level_hw[2 * ll]; fr.w[l] = level_hw[2 * ll + 1]; fr.hw[l] = fr.h[l] * fr.w[l];
    fr.start[l] = (int)S;
    if (l < L) S += fr.hw[l];
  }
  for (int l = L; l <= 4; ++l) fr.start[l] = (int)S;
  if (S * R >= (1ll << 31)) return GD4D_EUNSUPPORTED;
  fr.S = (int)S; fr.i2l = img2lidar; fr.pad_h = pad_h; fr.pad_w = pad_w; fr.depth_start = depth_start;
  // (the frustum kernel's constants: Python-float arithmetic of the reference (:452), rounded to fp32 when it meets the tensor)
  fr.bin_size = (float)((pc_range[3] - (double)depth_start) / ((double)D * (1.0 + (double)D)));
  for (int k = 0; k < 3; ++k) { fr.lo[k] = (float)pc_range[k]; fr.span[k] = (float)(pc_range[k + 3] - pc_range[k]); }
  const int M = (int)(S * R);
  const char* img1 = static_cast<const char*>(image);
  Mlp2Params p{nullptr, img1, img1 + (size_t)H * K1 * 4 + (size_t)(H / ML_HC) * 1024, b2, out, M, K1, H, 0, ldo};
  const int lds = 2 * ((K1 / 16) * 2048 + 1024 + ML_S2);
  const Mlp2Se none{};
  auto kern = mlp2_kernel<12, false, true>;
  if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return GD4D_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(mlp2_grid(M)), dim3(ML_THREADS), lds, static_cast<hipStream_t>(stream), p, none, fr);
  return check_launch();
}

// See mlp2_pe_se_kernel.  pe_image: gd4d_mlp2_image of (position_encoder[0]'s weight with its columns permuted as gd4d_mlp2_frustum_fwd
// takes it, its bias, position_encoder[2]'s weight); se_image: as gd4d_mlp2_se_fuse_fwd; the geometry arguments as gd4d_mlp2_frustum_fwd,
// feats / sine / outs as gd4d_mlp2_se_fuse_fwd.  pe_out: NULL, or (R S, 256) to keep the embedding as well.
extern "C" int gd4d_mlp2_pe_se_fwd(const float* img2lidar, const void* const* feats, const int32_t* level_hw, int L, int R, float pad_h,
                                   float pad_w, int D, float depth_start, const double* pc_range, const void* pe_image,
                                   const float* pe_b2, int pe_H, const void* se_image, const float* se_b2, int se_H, const float* sine,
                                   void* const* outs, float* pe_out, void* stream) {
  using namespace gd4d;
  if (!img2lidar || !feats || !level_hw || !pc_range || !pe_image || !se_image || !sine || !outs || L <= 0 || R <= 0) return GD4D_EINVAL;
  const int K1 = 3 * D, C = ML_N2;
  if (L > 4 || D != 64 || !mlp2_shape_ok(K1, pe_H, C) || !mlp2_shape_ok(C, se_H, C)) return GD4D_EUNSUPPORTED;
  if (!aligned16(pe_image) || !aligned16(se_image)) return GD4D_EALIGN;
  Mlp2Fr fr{};
  Mlp2Se q{};
  long long S = 0;
  for (int l = 0; l < 4; ++l) {
    const int ll = l < L ? l : 0;
    if (l < L && (!feats[l] || !outs[l] || level_hw[2 * l] <= 0 || level_hw[2 * l + 1] <= 0)) return GD4D_EINVAL;
    fr.h[l] = level_hw[2 * ll]; fr.w[l] = level_hw[2 * ll + 1]; fr.hw[l] = fr.h[l] * fr.w[l];
    q.feat[l] = static_cast<const float*>(feats[ll]); q.out[l] = static_cast<float*>(outs[ll]); q.hw[l] = fr.hw[l];
    fr.start[l] = q.start[l] = (int)S;
    if (l < L) S += fr.hw[l];
  }
  for (int l = L; l <= 4; ++l) fr.start[l] = q.start[l] = (int)S;
  if (S * R >= (1ll << 31)) return GD4D_EUNSUPPORTED;
  fr.S = q.S = (int)S; q.sine = sine; q.pe = nullptr;
  fr.i2l = img2lidar; fr.pad_h = pad_h; fr.pad_w = pad_w; fr.depth_start = depth_start;
  fr.bin_size = (float)((pc_range[3] - (double)depth_start) / ((double)D * (1.0 + (double)D)));
  for (int k = 0; k < 3; ++k) { fr.lo[k] = (float)pc_range[k]; fr.span[k] = (float)(pc_range[k + 3] - pc_range[k]); }
  const int M = (int)(S * R);
  const char* pi = static_cast<const char*>(pe_image);
  const char* si = static_cast<const char*>(se_image);
  Mlp2PeSe p{pi, pi + (size_t)pe_H * K1 * 4 + (size_t)(pe_H / ML_HC) * 1024, pe_b2, pe_H,
             si, si + (size_t)se_H * C * 4 + (size_t)(se_H / ML_HC) * 1024, se_b2, se_H, pe_out, M};
  const int lds = 2 * (16 * 2048 + 1024 + ML_S2);                  // the second MLP's stage pair (the larger)
  auto kern = mlp2_pe_se_kernel;
  if (!allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return GD4D_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(mlp2_grid(M)), dim3(ML_THREADS), lds, static_cast<hipStream_t>(stream), p, q, fr);
  return check_launch();
}

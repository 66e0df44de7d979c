// The A-stationary value_proj tile loop (gd4d_value_proj.hip explains the shape) as a device function: the kernel of
// gd4d_value_proj_multi_fwd and the GUEST workgroups of gd4d_row_chain_fwd (gd4d_rowchain.hip: one decoder layer's value_proj over
// the coarse pyramid levels, riding in the launch of the previous layer's row chain) run the same body.
#pragma once
#include "gd4d_common.h"

namespace gd4d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int VA_C = 256;
constexpr int VA_KSTEPS = VA_C / 16;
constexpr int VA_TILE = 32;                    // pixels per wave tile (= MFMA M)
constexpr int VA_RING = 2;                     // LDS ring slots: the DMA runs one chunk ahead

constexpr int va_chunk_bytes(bool single) { return VA_KSTEPS * (single ? 1 : 2) * 1024; }
// workspace image: NL * 8 chunks, then the bias table [NL][256] fp32
constexpr size_t va_image_bytes(int NL, bool single) { return (size_t)NL * 8 * va_chunk_bytes(single) + (size_t)NL * VA_C * 4; }

struct VpaParams {
  const void* in[GD4D_MAX_LEVELS];     // level l: (R, C, HW_l) fp32
  int hw[GD4D_MAX_LEVELS];
  int start[GD4D_MAX_LEVELS];          // pixel offset of level l inside a row of `out`
  int tiles[GD4D_MAX_LEVELS];          // tiles per camera-row at level l
  int tile_base[GD4D_MAX_LEVELS + 1];  // prefix over levels of R * tiles[l]
  void* out[GD4D_MAX_LAYERS];
  const char* wimg;                    // workspace: NL * 8 chunks
  const float* bias_table;             // [NL][256] fp32; nullptr: behind the chunks (va_image_bytes)
  int in_chlast;                       // the levels are stored (R, HW, C) - channels-last rows - instead of (R, C, HW)
  int R, L, S, NL, Hh, Dh, total;
  unsigned long long* trace;           // dev (DBG & 16): per wave of the first 8 workgroups, cycles spent per phase segment
};

__device__ __forceinline__ unsigned va_cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}

// 8 consecutive floats -> one 16-byte chunk of bf16 hi halves and one of bf16 lo halves; x ~= hi + lo to ~2^-17 relative
__device__ __forceinline__ void va_split8(const float* v, u32x4& h, u32x4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = va_cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    const float ra = v[2 * i] - __uint_as_float(hh[i] << 16);              // exact: hi is a rounding of the input
    const float rb = v[2 * i + 1] - __uint_as_float(hh[i] & 0xffff0000u);
    ll[i] = va_cvt_pk_bf16(ra, rb);
  }
  h = u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}

__device__ __forceinline__ bf16x8 va_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// gd4d_chain_guest -> the kernel's parameters (one layer, pixel-major fp32 rows): gd4d_row_chain_guest_fwd and gd4d_value_proj_guest_fwd
inline int va_guest_params(const gd4d_chain_guest* guest, VpaParams& g) {
  if (!guest || !guest->image || !guest->out || guest->R <= 0 || guest->L <= 0 || guest->L > GD4D_MAX_LEVELS) return GD4D_EINVAL;
  if (!aligned16(guest->image) || !aligned16(guest->out)) return GD4D_EALIGN;
  g = VpaParams{};
  int st = 0, base = 0;
  for (int l = 0; l < guest->L; ++l) {
    const int h = guest->level_hw[2 * l], w = guest->level_hw[2 * l + 1];
    if (!guest->feats[l] || h <= 0 || w <= 0) return GD4D_EINVAL;
    if (guest->chlast && !aligned16(guest->feats[l])) return GD4D_EALIGN;
    g.in[l] = guest->feats[l]; g.hw[l] = h * w; g.start[l] = st; g.tiles[l] = (h * w + VA_TILE - 1) / VA_TILE; g.tile_base[l] = base;
    st += h * w;
    base += guest->R * g.tiles[l];
  }
  for (int l = guest->L; l <= GD4D_MAX_LEVELS; ++l) g.tile_base[l] = base;
  for (int l = guest->L; l < GD4D_MAX_LEVELS; ++l) { g.tiles[l] = 1; g.hw[l] = 1; }
  g.S = st; g.R = guest->R; g.L = guest->L; g.NL = 1; g.Hh = 8; g.Dh = VA_C / 8; g.total = base;
  g.out[0] = guest->out;
  g.wimg = static_cast<const char*>(guest->image);
  g.in_chlast = guest->chlast ? 1 : 0;
  return GD4D_OK;
}

// dynamic LDS a workgroup of `waves` waves needs: ring, bias table, transposing patches
constexpr size_t va_lds_bytes(int NL, bool single, int waves) { return VA_RING * (size_t)va_chunk_bytes(single) + (size_t)NL * VA_C * 4 + (size_t)waves * VA_TILE * 144; }

// ---------------------------------------------------------------------------------------------------------------
// WAVES waves per workgroup, one workgroup per CU.  SINGLE: one bf16 product a_hi*w_hi (bf16-class, for bf16 value storage).
// DBG: compile-time ablation bits (dev only, production = 0): 1 no stores, 2 no MFMAs, 4 no fragment reads, 8 no DMA,
// 16 s_memtime trace of the phase segments (wait + barrier / k-loop)
//
// One PHASE = one chunk: [counted vmcnt: my pieces of this chunk have landed] [s_barrier] [k-loop].  Everything else
// rides inside the k-loop, behind MFMAs, one memory instruction every other k-step (the two waves that share a SIMD use
// opposite k-step parities): first the DMA pieces of the NEXT chunk (the other ring slot was read in the previous phase:
// every wave has passed this phase's barrier, hence finished reading it), then the stores of the PREVIOUS chunk's
// result, which sits in the other of two accumulator sets (no copy, no add between two phases).
//
// The product is computed TRANSPOSED, D[channel][pixel] = W_frag (A operand, from LDS) x x_frag (B operand, the tile's
// registers): a lane ends up with 4 x 4 consecutive channels of ITS pixel.  Before it leaves, the 32 x 32 result is
// turned through a private 4.5 KB LDS patch (4 ds_write_b128 + 4 ds_read_b128 per wave, pitch 144 B: conflict-free)
// so that 8 consecutive lanes hold one pixel's 32 channels: every store instruction then writes 8 FULL 128-byte lines.
// This is what the CU's store path wants (tools/microbench/write_probe.hip, one CU alone): full lines 16 B per lane
// 63 B/clk, dword stores of two lines 33 B/clk, 16-byte pieces of 32 different lines 16 B/clk - the path costs 2 cycles
// per line TOUCHED, and at 32 KB of results per 3072 MFMA cycles the partial-line forms took 1000-2000 of them.
// block / nblocks: this workgroup's index among the workgroups that share the job (a launch of its own: its index in the grid;
// guest workgroups of another kernel's launch - gd4d_row_chain_fwd - : their index among the guests); smem: >= va_lds_bytes.
// IN_CHLAST: the levels are stored (R, HW, C) - a compile-time form, so that the tile's loads are two 16-byte loads per k-step (as a
// run-time choice the compiler merges both forms into dword loads with a variable stride: 64 lines touched per instruction).
template <int WAVES, bool OUT_BF16, bool HEAD_MAJOR, bool SINGLE, int DBG = 0, bool IN_CHLAST = false>
__device__ __forceinline__ void value_proj_astat_body(const VpaParams& p, const int block, const int nblocks, char* const smem) {
  constexpr int PARTS = SINGLE ? 1 : 2;
  constexpr int CHUNK = VA_KSTEPS * PARTS * 1024;
  constexpr int PIECES = CHUNK / 1024 / WAVES;                // 16-byte DMA instructions per wave and chunk
  static_assert(PIECES * WAVES * 1024 == CHUNK, "pieces divide evenly");
  static_assert(VA_RING == 2, "the phase loop below is unrolled by 2 (8 * NL chunks per pass)");
  constexpr int NOPS = PIECES + 4;                            // memory instructions per wave and phase
  static_assert(2 * NOPS <= VA_KSTEPS, "one memory instruction every other k-step");
  constexpr int PITCH = 144;                                  // bytes per pixel row of the transposing patch
  // smem: ring[2][CHUNK] | bias[NL][256] | patch[WAVES] - ONE object
  char* const bias_lds = smem + VA_RING * CHUNK;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31;           // MFMA column = this lane's pixel inside the tile
  const int kg = lane >> 5;            // which 8 of the 16 k of a step this lane holds
  const int burst_s = (wave & 3) * 2 + (wave >> 2) * 8;      // dev (DBG & 128): the k-step of this wave's store burst
  const bool odd_wave = wave >= WAVES / 2;   // waves w and w + WAVES/2 share a SIMD: they get opposite k-step parities
  // LDS byte addresses inside this wave's patch: where this lane parks its quads, and where it picks up a line piece
  // (LDS byte addresses for the inline assembly below: the dynamic segment need not start at 0 - the host kernel may own static LDS)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned patch = lds0 + (unsigned)(VA_RING * CHUNK + p.NL * VA_C * 4 + wave * (VA_TILE * PITCH));
  const unsigned st_w = patch + col * PITCH + kg * 16;               // + 32 i: channels 8 i + 4 kg .. of pixel col
  const unsigned st_r = patch + (lane >> 3) * PITCH + (lane & 7) * 16;   // + 8 m * PITCH: pixel 8 m + lane/8, channels 4 (lane & 7) ..

  const int slots = nblocks * WAVES;
  const int P = 8 * p.NL;                                      // chunks per pass (even)
  const int full_passes = p.total / slots;
  const int npass = full_passes + (p.total - full_passes * slots ? 1 : 0);

  const unsigned lane16 = lane * 16;
  // DMA piece i (0 .. PIECES - 1) of chunk `c` into ring slot `b`: wave w moves the 1-KB pieces w, w + WAVES, ...
  auto issue_piece = [&](int c, int b, int i) {
    if (DBG & 8) return;
    const int piece = wave + i * WAVES;
    unsigned lo = lane16;
    asm volatile("" : "+v"(lo));                               // opaque: keeps hipcc from hoisting (and spilling) one
    __builtin_amdgcn_global_load_lds((glb_void_t*)(p.wimg + (size_t)c * CHUNK + piece * 1024 + lo),   // 64-bit address per piece
                                     (lds_void_t*)(smem + b * CHUNK + piece * 1024), 16, 0, 0);
  };

  f32x16 accA, accB;                   // phases in ring slot 0 accumulate into accA, slot 1 into accB
  char* prev_out = nullptr;            // where the result of the previous phase goes: base of (layer, camera row, tile, chunk)
  int prev_valid = 0;                  // its number of existing pixel rows (>= 32: all)
  int prev_head_stride = 0;            // HEAD_MAJOR with Dh < 32: bytes between the chunk's two heads
  bool have_prev = false;
  int steady = 0;                      // the previous phase issued exactly PIECES pieces, then 4 stores
  unsigned long long seg[4] = {0, 0, 0, 0};

  // The previous phase's accumulators -> the patch -> back, line-major: afterwards quad m of `acc` holds channels
  // 4 (lane & 7) .. + 4 of pixel 8 m + lane / 8.  Inline asm on purpose: hipcc would put "s_waitcnt vmcnt(0)" in front of
  // a visible LDS store while LDS-DMA is in flight (it cannot tell the patch from the ring).
  auto transpose_prev = [&](f32x16& acc) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 q0 = {acc[0], acc[1], acc[2], acc[3]}, q1 = {acc[4], acc[5], acc[6], acc[7]};
    f32x4 q2 = {acc[8], acc[9], acc[10], acc[11]}, q3 = {acc[12], acc[13], acc[14], acc[15]};
    // LDS operations of a wave execute in order: the reads below see the writes, and may land in the registers the
    // writes were issued from
    asm volatile("ds_write_b128 %4, %0\n\tds_write_b128 %4, %1 offset:32\n\tds_write_b128 %4, %2 offset:64\n\t"
                 "ds_write_b128 %4, %3 offset:96\n\t"
                 "ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %5 offset:%7\n\t"
                 "ds_read_b128 %3, %5 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                 : "v"(st_w), "v"(st_r), "n"(8 * PITCH), "n"(16 * PITCH), "n"(24 * PITCH) : "memory");
    acc[0] = q0[0]; acc[1] = q0[1]; acc[2] = q0[2]; acc[3] = q0[3];
    acc[4] = q1[0]; acc[5] = q1[1]; acc[6] = q1[2]; acc[7] = q1[3];
    acc[8] = q2[0]; acc[9] = q2[1]; acc[10] = q2[2]; acc[11] = q2[3];
    acc[12] = q3[0]; acc[13] = q3[1]; acc[14] = q3[2]; acc[15] = q3[3];
  };
  // store quad m of the (transposed) previous result: pixel 8 m + lane / 8, channels 4 (lane & 7) ..
  const int lpix = lane >> 3, lgrp = lane & 7;
  auto store_group = [&](const f32x16& acc, int m) {
    if (DBG & 1) { asm volatile("" ::"v"(acc[4 * m]), "v"(acc[4 * m + 1]), "v"(acc[4 * m + 2]), "v"(acc[4 * m + 3])); return; }
    int q = 8 * m + lpix;
    asm volatile("" : "+v"(q));                                // opaque (see issue_piece)
    constexpr int ES = OUT_BF16 ? 2 : 4;
    size_t o;                                                  // byte offset from prev_out
    if (!HEAD_MAJOR) o = ((size_t)q * VA_C + 4 * lgrp) * ES;
    else if (p.Dh >= 32) o = ((size_t)q * p.Dh + 4 * lgrp) * ES;
    else o = (size_t)(4 * lgrp / p.Dh) * prev_head_stride + ((size_t)q * p.Dh + (4 * lgrp) % p.Dh) * ES;
    if (q < prev_valid) {
      if (OUT_BF16) {
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(acc[4 * m]) | ((unsigned)f32_to_bf16(acc[4 * m + 1]) << 16);
        pk.y = (unsigned)f32_to_bf16(acc[4 * m + 2]) | ((unsigned)f32_to_bf16(acc[4 * m + 3]) << 16);
        *reinterpret_cast<uint2*>(prev_out + o) = pk;
      } else {
        // (as guests of a chain launch, `sc1` and `nt` stores measured 669 / 676 against 676 samples/s for plain ones: the rows are
        //  not what slows the chain beside them)
        *reinterpret_cast<float4*>(prev_out + o) = make_float4(acc[4 * m], acc[4 * m + 1], acc[4 * m + 2], acc[4 * m + 3]);
      }
    }
  };

  // prologue: the bias table into LDS; chunk 0 of the first pass on its way
  {
    const float* table = p.bias_table ? p.bias_table : reinterpret_cast<const float*>(p.wimg + (size_t)P * CHUNK);
    for (int i = threadIdx.x; i < p.NL * VA_C; i += 64 * WAVES) reinterpret_cast<float*>(bias_lds)[i] = table[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PIECES; ++i) issue_piece(0, 0, i);

  for (int pass = 0; pass < npass; ++pass) {
    // Tile of this wave.  Full passes give a workgroup WAVES consecutive tiles (contiguous output); the last, partial
    // pass is dealt wave-major so that its tiles spread over all workgroups instead of filling the first few.
    const int t = pass < full_passes ? pass * slots + block * WAVES + wave
                                     : full_passes * slots + wave * nblocks + block;
    const bool active = t < p.total;                           // wave-uniform
    const float* src = static_cast<const float*>(p.in[0]);
    int hw = p.hw[0], ostart = p.start[0], tiles = p.tiles[0], tbase = 0;
#pragma unroll
    for (int l = 1; l < GD4D_MAX_LEVELS; ++l)
      if (l < p.L && t >= p.tile_base[l]) {
        src = static_cast<const float*>(p.in[l]); hw = p.hw[l]; ostart = p.start[l]; tiles = p.tiles[l]; tbase = p.tile_base[l];
      }
    const int rel = active ? t - tbase : 0;
    const int row = __builtin_amdgcn_readfirstlane(rel / tiles);
    const int pix0 = __builtin_amdgcn_readfirstlane((rel - row * tiles) * VA_TILE);
    const int valid = __builtin_amdgcn_readfirstlane(hw - pix0);       // pixel rows of the tile that exist (>= 32: full)

    // ---- the tile's fragments (B operand): ci = 16*s + 8*kg + j at pixel pix0 + col ----
    u32x4 ahi[VA_KSTEPS], alo[VA_KSTEPS];
    if (active) {
      const int pix = min(pix0 + col, hw - 1);                 // tail lanes re-read the last pixel (never stored)
      const float* gp = IN_CHLAST ? src + ((size_t)row * hw + pix) * VA_C + 8 * kg : src + ((size_t)row * VA_C + 8 * kg) * hw + pix;
#pragma unroll
      for (int s4 = 0; s4 < VA_KSTEPS; s4 += 4) {              // 4 k-steps (32 loads) per batch bounds the staging registers
        float v[4][8];
        if (IN_CHLAST) {                                       // (R, HW, C) rows: the lane's 8 channels of a k-step are 32 contiguous bytes
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(gp + 16 * (s4 + s)), b = *reinterpret_cast<const float4*>(gp + 16 * (s4 + s) + 4);
            v[s][0] = a.x; v[s][1] = a.y; v[s][2] = a.z; v[s][3] = a.w; v[s][4] = b.x; v[s][5] = b.y; v[s][6] = b.z; v[s][7] = b.w;
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[s][j] = gp[(size_t)(16 * (s4 + s) + j) * hw];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          u32x4 h, l;
          va_split8(v[s], h, l);
          ahi[s4 + s] = h;
          alo[s4 + s] = l;
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < VA_KSTEPS; ++s) { ahi[s] = u32x4{0u, 0u, 0u, 0u}; alo[s] = ahi[s]; }
    }
    // wave-uniform element offset of the tile's first pixel inside a layer's value tensor (+ channel block per chunk)
    const size_t tile_elem = HEAD_MAJOR ? ((size_t)row * p.Hh * p.S + ostart + pix0) * p.Dh
                                        : ((size_t)row * p.S + ostart + pix0) * VA_C;

    // one phase; SLOT = ring slot = accumulator set (compile-time: the ring offsets become immediates);
    // STAG = 1: this wave's memory instructions sit in the odd k-steps
    auto phase = [&](auto slot_c, auto stag_c, int c) {
      constexpr int SLOT = decltype(slot_c)::value;
      constexpr int STAG = decltype(stag_c)::value;
      f32x16& cur = SLOT ? accB : accA;
      f32x16& old = SLOT ? accA : accB;
      unsigned long long tt0 = 0, tt1 = 0;
      if (DBG & 16) tt0 = __builtin_amdgcn_s_memtime();
      // my pieces of this chunk were issued at the start of the previous phase; a steady wave issued exactly 4 stores since
      if (DBG & 128) { if (steady && wave >= WAVES / 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else if (DBG & 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // dev: free-running waves (no barrier)
      else if ((DBG & 32) && steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (steady) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(DBG & 64)) __builtin_amdgcn_s_barrier();           // chunk complete in SLOT; the other slot is free for the DMA
      asm volatile("" ::: "memory");
      if (DBG & 16) tt1 = __builtin_amdgcn_s_memtime();
      const int nc = c + 1 == P ? 0 : c + 1;                   // next chunk of the (periodic) stream
      const bool more = pass < npass - 1 || c + 1 < P;         // none behind the end of the last pass
      if (!active) {                                           // no tile in the last, partial pass: feed the ring, drain
        if (more) {
#pragma unroll
          for (int i = 0; i < PIECES; ++i) issue_piece(nc, SLOT ^ 1, i);
        }
        if (have_prev) {
          transpose_prev(old);
#pragma unroll
          for (int m = 0; m < 4; ++m) store_group(old, m);
        }
        have_prev = false;
        steady = 0;
        return;
      }
      if (DBG & 32) {
        // dev experiment: ROLE SPLIT.  Waves 0 .. WAVES/2-1 (one per SIMD) run the matrix work of TWO tiles and never
        // touch global memory; their SIMD partners issue all of the workgroup's DMA pieces and stores (same bytes, same
        // instruction counts per workgroup and phase as the production kernel; results are garbage).
        const char* wb2 = smem + SLOT * CHUNK;
        if (wave < WAVES / 2) {
          for (int rep = 0; rep < 2; ++rep) {
            u32x4 xh[2], xl[2];
            xh[0] = *reinterpret_cast<const u32x4*>(wb2 + lane * 16);
            xl[0] = *reinterpret_cast<const u32x4*>(wb2 + 1024 + lane * 16);
#pragma unroll
            for (int s = 0; s < VA_KSTEPS; ++s) {
              if (s + 1 < VA_KSTEPS) {
                xh[(s + 1) & 1] = *reinterpret_cast<const u32x4*>(wb2 + ((s + 1) * PARTS) * 1024 + lane * 16);
                xl[(s + 1) & 1] = *reinterpret_cast<const u32x4*>(wb2 + ((s + 1) * PARTS + 1) * 1024 + lane * 16);
              }
              cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_frag(xh[s & 1]), va_frag(ahi[s]), cur, 0, 0, 0);
              cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_frag(xh[s & 1]), va_frag(alo[s]), cur, 0, 0, 0);
              cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_frag(xl[s & 1]), va_frag(ahi[s]), cur, 0, 0, 0);
            }
            typedef __attribute__((ext_vector_type(4))) float f32x4;
            f32x4 q0 = {cur[0], cur[1], cur[2], cur[3]}, q1 = {cur[4], cur[5], cur[6], cur[7]};
            f32x4 q2 = {cur[8], cur[9], cur[10], cur[11]}, q3 = {cur[12], cur[13], cur[14], cur[15]};
            const unsigned wa = st_w + rep * (WAVES / 2) * (VA_TILE * PITCH);     // own patch, then the partner's
            asm volatile("ds_write_b128 %4, %0\n\tds_write_b128 %4, %1 offset:32\n\tds_write_b128 %4, %2 offset:64\n\t"
                         "ds_write_b128 %4, %3 offset:96" :: "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(wa) : "memory");
          }
        } else {
          if (more) {
#pragma unroll
            for (int i = 0; i < 2 * PIECES; ++i) {
              const int piece = (wave - WAVES / 2) + i * (WAVES / 2);
              unsigned lo = lane16;
              asm volatile("" : "+v"(lo));
              __builtin_amdgcn_global_load_lds((glb_void_t*)(p.wimg + (size_t)nc * CHUNK + piece * 1024 + lo),
                                               (lds_void_t*)(smem + (SLOT ^ 1) * CHUNK + piece * 1024), 16, 0, 0);
            }
          }
          if (have_prev) {
            for (int rep = 0; rep < 2; ++rep) {
              typedef __attribute__((ext_vector_type(4))) float f32x4;
              f32x4 q0, q1, q2, q3;
              const unsigned ra = st_r - rep * (WAVES / 2) * (VA_TILE * PITCH);
              asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\t"
                           "ds_read_b128 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                           : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3)
                           : "v"(ra), "n"(8 * PITCH), "n"(16 * PITCH), "n"(24 * PITCH) : "memory");
              old[0] = q0[0]; old[1] = q0[1]; old[2] = q0[2]; old[3] = q0[3];
              old[4] = q1[0]; old[5] = q1[1]; old[6] = q1[2]; old[7] = q1[3];
              old[8] = q2[0]; old[9] = q2[1]; old[10] = q2[2]; old[11] = q2[3];
              old[12] = q3[0]; old[13] = q3[1]; old[14] = q3[2]; old[15] = q3[3];
              char* keep = prev_out;
              if (rep) prev_out -= (size_t)(WAVES / 2) * VA_TILE * VA_C * 4;
#pragma unroll
              for (int m = 0; m < 4; ++m) store_group(old, m);
              prev_out = keep;
            }
          }
        }
        steady = 0;
        if (wave >= WAVES / 2) steady = more && have_prev && prev_valid >= VA_TILE && !(DBG & 1);
        have_prev = true;
        {
          const int layer = c >> 3, cb = c & 7;
          void* outp = p.out[0];
#pragma unroll
          for (int l = 1; l < GD4D_MAX_LAYERS; ++l)
            if (l == layer) outp = p.out[l];
          prev_out = static_cast<char*>(outp) + (tile_elem + 32 * cb) * 4;
          prev_valid = valid;
        }
        return;
      }
      const char* wb = smem + SLOT * CHUNK;
      {
        const char* bp = bias_lds + (c * 32 + kg * 4) * 4;     // bias[layer][32 cb + 8 i + 4 kg .. +4]  (c = 8 layer + cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 bv = *reinterpret_cast<const u32x4*>(bp + 32 * i);
          cur[4 * i + 0] = __uint_as_float(bv[0]); cur[4 * i + 1] = __uint_as_float(bv[1]);
          cur[4 * i + 2] = __uint_as_float(bv[2]); cur[4 * i + 3] = __uint_as_float(bv[3]);
        }
      }
      u32x4 wh[2], wl[2];
      if (DBG & 4) { wh[0] = u32x4{1u, 2u, 3u, (unsigned)lane}; wl[0] = wh[0]; wh[1] = wh[0]; wl[1] = wh[0]; }
      else {
        wh[0] = *reinterpret_cast<const u32x4*>(wb + lane * 16);
        if (!SINGLE) wl[0] = *reinterpret_cast<const u32x4*>(wb + 1024 + lane * 16);
      }
      const bool stores_now = have_prev;
#pragma unroll
      for (int s = 0; s < VA_KSTEPS; ++s) {
        if (s + 1 < VA_KSTEPS && !(DBG & 4)) {                 // next k-step's W fragments while this step's MFMAs run
          wh[(s + 1) & 1] = *reinterpret_cast<const u32x4*>(wb + ((s + 1) * PARTS) * 1024 + lane * 16);
          if (!SINGLE) wl[(s + 1) & 1] = *reinterpret_cast<const u32x4*>(wb + ((s + 1) * PARTS + 1) * 1024 + lane * 16);
        }
        if (DBG & 128) {                                       // dev: one wave at a time pushes its 4 stores as a burst
          if (s >= STAG && ((s - STAG) & 1) == 0) {
            const int j = (s - STAG) >> 1;
            if (j < PIECES) { if (more) issue_piece(nc, SLOT ^ 1, j); }
          }
          if (s == burst_s && stores_now) {
            transpose_prev(old);
#pragma unroll
            for (int m = 0; m < 4; ++m) store_group(old, m);
          }
        } else {
        if (s == 2 * PIECES + STAG - 1) { if (stores_now) transpose_prev(old); }    // between the pieces and the stores
        if (s >= STAG && ((s - STAG) & 1) == 0) {              // this wave's memory instruction of the k-step, if any
          const int j = (s - STAG) >> 1;
          if (j < PIECES) { if (more) issue_piece(nc, SLOT ^ 1, j); }                // pieces FIRST in the queue,
          else if (j < NOPS) { if (stores_now) store_group(old, j - PIECES); }       // then the stores
        }
        }
        if (DBG & 2) { asm volatile("" ::"v"(wh[s & 1]), "v"(wl[s & 1]), "v"(ahi[s]), "v"(alo[s])); continue; }
        cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_frag(wh[s & 1]), va_frag(ahi[s]), cur, 0, 0, 0);
        if (!SINGLE) {
          cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_frag(wh[s & 1]), va_frag(alo[s]), cur, 0, 0, 0);
          cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_frag(wl[s & 1]), va_frag(ahi[s]), cur, 0, 0, 0);
        }
      }
      // a full tile issues exactly 4 store instructions (a partial one may skip instructions whose 8 pixels are all absent)
      steady = more && stores_now && prev_valid >= VA_TILE && !(DBG & 9);
      // this phase's result stays in `cur`; the next phase stores it
      have_prev = true;
      {
        const int layer = c >> 3, cb = c & 7;
        void* outp = p.out[0];
#pragma unroll
        for (int l = 1; l < GD4D_MAX_LAYERS; ++l)
          if (l == layer) outp = p.out[l];
        constexpr int ES = OUT_BF16 ? 2 : 4;
        size_t e = tile_elem;
        if (!HEAD_MAJOR) e += 32 * cb;
        else e += (size_t)((32 * cb) / p.Dh) * p.S * p.Dh + (32 * cb) % p.Dh;
        prev_out = static_cast<char*>(outp) + e * ES;
        prev_head_stride = p.S * p.Dh * ES;
        prev_valid = valid;
      }
      if (DBG & 16) {
        const unsigned long long tt2 = __builtin_amdgcn_s_memtime();
        seg[0] += tt1 - tt0; seg[1] += tt2 - tt1; seg[2] += 1;
      }
    };
    steady = 0;                        // (the tile's loads above drained the queue: counts restart)
    using std::integral_constant;
    if (odd_wave) {
      for (int c = 0; c < P; c += 2) {
        phase(integral_constant<int, 0>{}, integral_constant<int, 1>{}, c);
        phase(integral_constant<int, 1>{}, integral_constant<int, 1>{}, c + 1);
      }
    } else {
      for (int c = 0; c < P; c += 2) {
        phase(integral_constant<int, 0>{}, integral_constant<int, 0>{}, c);
        phase(integral_constant<int, 1>{}, integral_constant<int, 0>{}, c + 1);
      }
    }
  }
  // the last phase's result (P is even: it sits in accB)
  if (have_prev) {
    transpose_prev(accB);
#pragma unroll
    for (int m = 0; m < 4; ++m) store_group(accB, m);
  }
  if ((DBG & 16) && p.trace && block < 8 && lane == 0) {
    unsigned long long* t = p.trace + (block * WAVES + wave) * 8;
    for (int i = 0; i < 4; ++i) t[i] = seg[i];
  }
}

}  // namespace gd4d

// The pyramid-gradient bookkeeping's fill step as a device function: gd4d_cross_attn_sliced_bwd.hip launches it as a kernel of its
// own (gd4d_pyramid_grad_fill), gd4d_train.hip as guest workgroups of the attention backward (gd4d_mha_core_bwd_fill).
#pragma once
#include "gd4d_common.h"
#include "gd4d_cross_attn_sliced.h"

namespace gd4d {

// records[start[chunk] + slot] = {weight, pixel-in-chunk << 26 | table row}: no atomics
template <int FP = 4>
__device__ __forceinline__ void pyramid_grad_fill_body(const int* __restrict__ hdr, const uint2* __restrict__ pair,
                                                       const uint2* __restrict__ slots, int cap_t, int HH, int BQ,
                                                       const int* __restrict__ start, uint2* __restrict__ rec,
                                                       const int32_t* __restrict__ order, unsigned id_base, const int ph) {
  const int lane = threadIdx.x & 63;
  if (ph >= BQ * HH) return;
  const int pos = ph / HH, h = ph - pos * HH;
  const int M = hdr[pos * kPlanHdr + h];
  const int T = (M + 3) >> 2;
  const unsigned id = id_base + (unsigned)((order ? order[pos] : pos) * HH + h);
  const size_t prow = (size_t)ph * cap_t * 64 + lane;
  // four passes at a time: their slot and pair rows requested together, then the four chunk starts (which depend on the
  // slots), then the stores - two round trips per four passes (one pass per iteration was two dependent round trips per pass
  // on a wave that walks ~4 passes: 26 us per launch)
  // (FP = 8 for the guests of a training chain - gd4d_rowchain.hip: few waves, each walks several rows one after the other)
  for (int t = 0; t < T; t += FP) {
    uint2 sr[FP], pr[FP];
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const size_t at = prow + (size_t)min(t + i, T - 1) * 64;
      sr[i] = slots[at];
      pr[i] = pair[at];
    }
    int st[FP];
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const bool ok = t + i < T && sr[i].y != 0xffffffffu;
      st[i] = ok ? start[sr[i].x >> 6] : -1;
    }
#pragma unroll
    for (int i = 0; i < FP; ++i)
      if (st[i] >= 0) rec[(size_t)st[i] + sr[i].y] = make_uint2(pr[i].y, ((sr[i].x & 63u) << 26) | id);
  }
}

// Up to two layers' fills as guest workgroups of another launch: 8 (position, head) waves per workgroup, job 1 after job 0; groups
// of 8 guest workgroups spread evenly among the host's (the host renumbers its own: block - 8 * guests in front).
struct FillGuest {
  const int* hdr[2];
  const uint2* pair[2];
  const uint2* slots[2];
  const int32_t* order[2];
  unsigned id_base[2];
  int BQ[2];
  const int* start;
  uint2* rec;
  int cap_t, HH, wgs0;             // workgroups of job 0
  int guest_groups, total_groups;  // groups of 8 workgroups: guests among all
};

// true: this workgroup was a guest (done); false: `host` = its index among the host's workgroups
__device__ __forceinline__ bool fill_guest_or_host(const FillGuest& fg, int& host) {
  const long long grp = blockIdx.x >> 3;
  const int before = (int)(grp * fg.guest_groups / fg.total_groups);          // guest groups in front of this one
  if ((int)((grp + 1) * fg.guest_groups / fg.total_groups) > before) {
    const int gid = before * 8 + (int)(blockIdx.x & 7);
    const int job = gid >= fg.wgs0 ? 1 : 0;
    const int local = gid - (job ? fg.wgs0 : 0);
    if (fg.hdr[job])
      pyramid_grad_fill_body(fg.hdr[job], fg.pair[job], fg.slots[job], fg.cap_t, fg.HH, fg.BQ[job], fg.start, fg.rec, fg.order[job],
                             fg.id_base[job], local * 8 + (int)(threadIdx.x >> 6));
    return true;
  }
  host = (int)blockIdx.x - 8 * before;
  return false;
}

}  // namespace gd4d

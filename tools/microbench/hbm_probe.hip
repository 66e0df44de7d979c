// Dev probe (not product code): what read bandwidth does MI355X deliver for (a) a streaming read and (b) random
// gathers of SEG-byte segments (the fused sample-aggregate kernel reads 128-byte head-pixels), cache-cold
// (rotating over 6 x 757 MB buffers, more than L2 + Infinity Cache) and cache-warm (one buffer)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/hbm_probe.hip -o build/dbg/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void stream_read_u(const float4* __restrict__ src, size_t n4, float* sink) {
  float4 acc = make_float4(0, 0, 0, 0);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; i < n4; i += stride) { const float4 a = src[i]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) *sink = acc.x;
}

__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ src, size_t n4, float* sink) {
  float4 acc = make_float4(0, 0, 0, 0);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
    acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
  }
  for (; i < n4; i += stride) { const float4 a = src[i]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) *sink = acc.x;
}

// Each group of LPS = SEG/16 lanes reads one random SEG-byte segment per step; INFLIGHT independent loads per lane are
// issued before any is consumed.  `steps` batches per wave.
template <int SEG, int INFLIGHT>
__global__ __launch_bounds__(256) void gather_read(const float4* __restrict__ src, size_t nseg, int steps, float* sink,
                                                   unsigned seed) {
  constexpr int LPS = SEG / 16;
  const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned group = gtid / LPS, sub = gtid % LPS;
  float4 acc = make_float4(0, 0, 0, 0);
  unsigned long long state = (unsigned long long)(group + 1) * 0x9E3779B97F4A7C15ull + seed;
  for (int s = 0; s < steps; ++s) {
    float4 v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      const size_t seg = (size_t)((state >> 24) % nseg);
      v[k] = src[seg * LPS + sub];
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) *sink = acc.x;
}

template <typename F>
static float time_us(F&& launch, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 6; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main() {
  const size_t bytes = 24ull * 30825 * 256 * 4;          // one value tensor of the headline config: 757 MB
  const int NBUF = 6;
  std::vector<float4*> buf(NBUF);
  for (auto& b : buf) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
  float* sink; CK(hipMalloc(&sink, 4));
  const size_t n4 = bytes / 16;
  for (int rot : {NBUF, 1}) {
    const char* tag = rot == 1 ? "warm (1 buffer)" : "cold (6 buffers)";
    float us = time_us([&](int i) { hipLaunchKernelGGL(stream_read, dim3(256 * 8), dim3(256), 0, 0, buf[i % rot], n4, sink); }, 30);
    printf("stream read 757 MB, %s: %.1f us  %.2f TB/s\n", tag, us, bytes / us * 1e-6);
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
      us = time_us([&](int i) { hipLaunchKernelGGL(stream_read_u<8>, dim3(blocks), dim3(256), 0, 0, buf[i % rot], n4, sink); }, 30);
      printf("stream read x8 in flight, %5d blocks, %s: %.1f us  %.2f TB/s\n", blocks, tag, us, bytes / us * 1e-6);
    }
    us = time_us([&](int i) { hipLaunchKernelGGL(stream_read_u<16>, dim3(4096), dim3(256), 0, 0, buf[i % rot], n4, sink); }, 30);
    printf("stream read x16 in flight, 4096 blocks, %s: %.1f us  %.2f TB/s\n", tag, us, bytes / us * 1e-6);
    us = time_us([&](int i) { hipLaunchKernelGGL(stream_read_u<8>, dim3(4096), dim3(256), 0, 0, buf[0], (size_t)(128ull << 20) / 16, sink); }, 30);
    printf("stream read 128 MB resident (Infinity Cache): %.1f us  %.2f TB/s\n", us, (128ull << 20) / us * 1e-6);
    // gathers: 250 MB per launch like the fused kernel (121466 tuples x 16 level-corners x 128 B)
    const size_t total = 250ull << 20;
    auto run = [&](auto kern, int seg, int inflight, int blocks) {
      const size_t groups = (size_t)blocks * 256 / (seg / 16);
      const int steps = (int)(total / ((size_t)seg * inflight * groups));
      const size_t moved = (size_t)steps * seg * inflight * groups;
      float t = time_us([&](int i) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf[i % rot], bytes / seg, steps, sink, (unsigned)i * 7919u); }, 30);
      printf("gather %4d B segments, %2d loads in flight/lane, %5d blocks x %d steps, %s: %.1f us  %.2f TB/s\n", seg, inflight,
             blocks, steps, tag, t, moved / t * 1e-6);
    };
    run(gather_read<128, 16>, 128, 16, 900);
    run(gather_read<128, 16>, 128, 16, 768);
    run(gather_read<128, 16>, 128, 16, 3072);
    run(gather_read<128, 32>, 128, 32, 768);
    run(gather_read<128, 8>, 128, 8, 3072);
    run(gather_read<256, 16>, 256, 16, 768);
    run(gather_read<512, 16>, 512, 16, 768);
    run(gather_read<1024, 16>, 1024, 16, 768);
  }
  return 0;
}

// Dev probe: HBM WRITE bandwidth of MI355X for the store patterns value_proj can produce (4.5 GB per launch, rotated
// over 2 buffers so nothing is absorbed by the 256 MB Infinity Cache).
// build: hipcc --offload-arch=gfx950 -O3 -o write_probe write_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

// PATTERN 0: float4 per lane, fully contiguous per wave (1 KB per instruction, 8 full lines)
// PATTERN 1: dword per lane, lanes 0-31 / 32-63 = two 128-byte lines 4 KB apart (the accumulator-row store)
// PATTERN 2: float4 per lane, lane pairs write 32 contiguous bytes in each of 32 lines 1 KB apart (transposed product)
// PATTERN 3: as 0 with nontemporal stores
// transposed-product pattern with at most LIMIT stores of a wave in flight (s_waitcnt vmcnt after every block of 4)
template <int LIMIT>
__global__ __launch_bounds__(512) void wr_limited(float* out, size_t n_float4) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const size_t wave = tid >> 6, nwaves = nth >> 6;
  const float4 v = make_float4((float)tid, 1.f, 2.f, 3.f);
  const size_t nblk = n_float4 * 4 / (32 * 32);
  for (size_t blk = wave; blk < nblk; blk += nwaves) {
    const size_t prow = (blk / 8) * 32, cb = blk % 8;
    float* base = out + (prow + (lane & 31)) * 256 + cb * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(base + 8 * i) = v;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LIMIT) : "memory");
  }
}

template <int LIMIT> static void run_limited(float* a, float* b, size_t bytes) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(wr_limited<LIMIT>, dim3(256), dim3(512), 0, 0, a, bytes / 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(wr_limited<LIMIT>, dim3(256), dim3(512), 0, 0, (i & 1) ? a : b, bytes / 16);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 6;
  printf("8 waves/CU, <= %2d + 4 stores (1 KB each) in flight per wave: %8.1f us  %5.2f TB/s\n", LIMIT, ms * 1e3, bytes / ms / 1e9);
}

template <int PATTERN>
__global__ __launch_bounds__(256) void wr(float* out, size_t n_float4, int iters_unused) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const size_t wave = tid >> 6, nwaves = nth >> 6;
  const float4 v = make_float4((float)tid, 1.f, 2.f, 3.f);
  if (PATTERN == 0 || PATTERN == 3) {
    for (size_t i = tid; i < n_float4; i += nth) {
      typedef __attribute__((ext_vector_type(4))) float f4;
      const f4 vv = {v.x, v.y, v.z, v.w};
      if (PATTERN == 3) __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(out) + i);
      else reinterpret_cast<f4*>(out)[i] = vv;
    }
  } else if (PATTERN == 1) {
    // a wave owns blocks of 32 pixel rows x 32 channels inside (pixels, 256) fp32 rows: 16 stores of 2 rows each
    const size_t nblk = n_float4 * 4 / (32 * 32);
    for (size_t blk = wave; blk < nblk; blk += nwaves) {
      const size_t prow = (blk / 8) * 32, cb = blk % 8;
      float* base = out + (prow + 4 * (lane >> 5)) * 256 + cb * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) base[(size_t)((r & 3) + 8 * (r >> 2)) * 256] = v.x;
    }
  } else {
    const size_t nblk = n_float4 * 4 / (32 * 32);
    for (size_t blk = wave; blk < nblk; blk += nwaves) {
      const size_t prow = (blk / 8) * 32, cb = blk % 8;
      float* base = out + (prow + (lane & 31)) * 256 + cb * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(base + 8 * i) = v;
    }
  }
}

template <int PATTERN> static void run(const char* name, float* a, float* b, size_t bytes, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(wr<PATTERN>, dim3(grid), dim3(256), 0, 0, a, bytes / 16, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(wr<PATTERN>, dim3(grid), dim3(256), 0, 0, (i & 1) ? a : b, bytes / 16, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 6;
  printf("%-52s grid %5d: %8.1f us  %5.2f TB/s\n", name, grid, ms * 1e3, bytes / ms / 1e9);
}

template <int LIMIT> static void run_few(float* a, size_t bytes, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(wr_limited<LIMIT>, dim3(grid), dim3(512), 0, 0, a, bytes / 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(wr_limited<LIMIT>, dim3(grid), dim3(512), 0, 0, a, bytes / 16);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%3d workgroups x 8 waves (limit %2d): %8.1f us  %7.1f GB/s per workgroup = %5.1f B/clk @2.4GHz\n", grid, LIMIT, ms * 1e3,
         bytes / ms / 1e6 / grid, bytes / ms / 1e6 / grid / 2.4);
}

template <int PATTERN> static void run_few_pat(const char* name, float* a, size_t bytes, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(wr<PATTERN>, dim3(grid), dim3(256), 0, 0, a, bytes / 16, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(wr<PATTERN>, dim3(grid), dim3(256), 0, 0, a, bytes / 16, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %3d workgroups x 4 waves: %7.1f GB/s per workgroup = %5.1f B/clk @2.4GHz\n", name, grid,
         bytes / ms / 1e6 / grid, bytes / ms / 1e6 / grid / 2.4);
}

int main() {
  const size_t bytes = (size_t)4500 << 20;
  float *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  for (int grid : {512, 2048, 8192}) {
    run<0>("float4 contiguous", a, b, bytes, grid);
    run<3>("float4 contiguous, nontemporal", a, b, bytes, grid);
    run<1>("dword rows (2 full lines / instr, 16 instr)", a, b, bytes, grid);
    run<2>("float4 transposed (32 B in 32 lines / instr, 4 instr)", a, b, bytes, grid);
  }
  for (int g : {1, 8}) {
    run_few_pat<0>("float4 contiguous", a, (size_t)g * (64 << 20), g);
    run_few_pat<3>("float4 contiguous, nontemporal", a, (size_t)g * (64 << 20), g);
    run_few_pat<1>("dword rows (2 full lines / instr, 16 instr)", a, (size_t)g * (64 << 20), g);
    run_few_pat<2>("float4 transposed (32 B in 32 lines / instr)", a, (size_t)g * (64 << 20), g);
  }
  for (int g : {1, 2, 8, 32}) { run_few<48>(a, (size_t)g * (64 << 20), g); run_few<0>(a, (size_t)g * (64 << 20), g); }
  run_limited<0>(a, b, bytes); run_limited<4>(a, b, bytes); run_limited<8>(a, b, bytes); run_limited<12>(a, b, bytes);
  run_limited<16>(a, b, bytes); run_limited<24>(a, b, bytes); run_limited<32>(a, b, bytes); run_limited<48>(a, b, bytes);
  return 0;
}

// Dev probe: issue rate of v_mfma_f32_32x32x16_bf16 as a function of how many independent accumulator chains a wave
// interleaves (1..4) and of what sits between the MFMAs (nothing / ds_read_b128 pairs).  8 waves per CU, 256 CUs.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip ; run: ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int CHAINS, int PATTERN, bool LDS>
__global__ __launch_bounds__(512, 2) void probe(const u32x4* __restrict__ data, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[32768];
  const int lane = threadIdx.x & 63;
  u32x4 a[16], b0, b1;
  for (int s = 0; s < 16; ++s) a[s] = data[(threadIdx.x * 16 + s) % 4096];
  for (int i = threadIdx.x; i < 2048; i += 512) reinterpret_cast<u32x4*>(smem)[i] = data[(i * 7) % 4096];
  __syncthreads();
  b0 = data[lane]; b1 = data[lane + 64];
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (LDS) {
        b0 = *reinterpret_cast<const u32x4*>(smem + (s * 2) * 1024 + lane * 16);
        b1 = *reinterpret_cast<const u32x4*>(smem + (s * 2 + 1) * 1024 + lane * 16);
      }
      const bf16x8 ah = __builtin_bit_cast(bf16x8, a[s]), al = __builtin_bit_cast(bf16x8, a[(s + 5) & 15]);
      const bf16x8 wh = __builtin_bit_cast(bf16x8, b0), wl = __builtin_bit_cast(bf16x8, b1);
      if (PATTERN == 0) {         // three products round-robin over CHAINS accumulators
        acc[(3 * s + 0) % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[(3 * s + 0) % CHAINS], 0, 0, 0);
        acc[(3 * s + 1) % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[(3 * s + 1) % CHAINS], 0, 0, 0);
        acc[(3 * s + 2) % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[(3 * s + 2) % CHAINS], 0, 0, 0);
      } else {                    // the value_proj pattern: acc0, acc1, acc1
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[1], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <class K> static void run(const char* name, K kern, const u32x4* d, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double mfma = 256.0 * 8 * iters * 48;
  const double flops = mfma * 2 * 32 * 32 * 16;
  printf("%-34s %8.1f us  %6.1f cycles/MFMA/SIMD @2.4GHz  %7.0f TFLOP/s\n", name, ms * 1e3,
         ms * 1e-3 * 2.4e9 / (mfma / 1024), flops / ms / 1e9);
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && atoi(argv[1]) == 0;
  std::vector<unsigned> h(4096 * 4);
  for (auto& v : h) { const unsigned a = rand() & 0xffff, b = rand() & 0xffff; v = zeros ? 0u : (((0x3f00u | (a & 0x80ffu)) << 16) | (0x3f00u | (b & 0x80ffu))); }
  u32x4* d; float* out;
  hipMalloc(&d, h.size() * 4); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 400;
  run("1 chain", probe<1, 0, false>, d, out, iters);
  run("2 chains round-robin", probe<2, 0, false>, d, out, iters);
  run("3 chains round-robin", probe<3, 0, false>, d, out, iters);
  run("4 chains round-robin", probe<4, 0, false>, d, out, iters);
  run("acc0,acc1,acc1", probe<2, 1, false>, d, out, iters);
  run("1 chain + 2 ds_read/step", probe<1, 0, true>, d, out, iters);
  run("3 chains + 2 ds_read/step", probe<3, 0, true>, d, out, iters);
  run("acc0,acc1,acc1 + 2 ds_read/step", probe<2, 1, true>, d, out, iters);
  return 0;
}

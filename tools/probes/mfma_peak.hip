// Probe: sustained v_mfma_f32_32x32x16_f16 rate (no memory traffic), 4 waves/WG, 2 WGs per CU,
// same 4-accumulator x 3-term pattern as the f16x3 conv.  Random operands (DVFS depends on data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k(const _Float16* in, float* out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const f16x8*>(in + (threadIdx.x * 8 + i * 2048) % 16384);
    b[i] = *reinterpret_cast<const f16x8*>(in + (threadIdx.x * 8 + i * 2048 + 1024) % 16384);
  }
  f32x16 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2], b[j], acc[i * 2 + j], 0, 0, 0);
          acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j + 2], acc[i * 2 + j], 0, 0, 0);
          acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
        }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  std::vector<_Float16> h(16384);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.02f);
  _Float16* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 2048 * 256 * 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int blocks : {256, 512, 1024, 2048}) {
    const int iters = 4000;
    k<<<blocks, 256>>>(d, o, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<<<blocks, 256>>>(d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = 2.0 * 32 * 32 * 16 * 24.0 * iters * 4.0 * blocks;
    printf("blocks %4d (%.1f waves/SIMD): %.3f ms  %.1f TFLOP/s issued\n", blocks, blocks * 4 / 1024.0, ms, flop / ms / 1e9);
  }
  return 0;
}

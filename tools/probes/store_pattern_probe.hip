// Probe: does the CU's store path care whether a wave's two 16-byte-per-lane stores interleave inside 32-byte units
// (the conv epilogue's fp32 pattern: lane owns 8 consecutive floats, store 1 = floats 0..3, store 2 = floats 4..7) or
// each cover a contiguous 1 KB run?  And the S32 pattern: hi run / lo run halves of 128-byte blocks.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern_probe store_pattern_probe.hip && ./store_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* out, long per_block_floats, int iters) {
  float* base = out + (long)blockIdx.x * per_block_floats;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int it = 0; it < iters; ++it) {
    // one "pass" = every wave stores 2 KB (64 lanes x 32 B) as two instructions
    float* p = base + ((long)it * 4 + wave) * 512;
    if (MODE == 0) {          // interleaved: lane's 32-byte unit, halves by instruction
      *reinterpret_cast<f32x4*>(p + lane * 8) = v;
      *reinterpret_cast<f32x4*>(p + lane * 8 + 4) = v;
    } else if (MODE == 1) {   // contiguous: instruction 1 = first KB, instruction 2 = second KB
      *reinterpret_cast<f32x4*>(p + lane * 4) = v;
      *reinterpret_cast<f32x4*>(p + 256 + lane * 4) = v;
    } else {                  // S32-like: 128-byte blocks, instr 1 writes the 64-byte hi halves, instr 2 the lo halves
      const int blk = lane >> 2, q = lane & 3;
      *reinterpret_cast<f32x4*>(p + blk * 32 + q * 4) = v;
      *reinterpret_cast<f32x4*>(p + blk * 32 + 16 + q * 4) = v;
    }
  }
}

template <int MODE>
void run(const char* name, float* buf, int blocks, int iters) {
  const long per_block = (long)iters * 4 * 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, per_block, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(store_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, per_block, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 10.0 * blocks * per_block * 4;
  printf("%-28s %7.2f TB/s  (%.1f B/clk/CU at 2.1 GHz, 256 CUs)\n", name, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
  const int blocks = 2048, iters = 32;           // 64 KB per workgroup, like a 128x128 fp32 output tile
  float* buf;
  hipMalloc(&buf, (size_t)blocks * iters * 4 * 512 * 4);
  run<0>("interleaved 32-byte units", buf, blocks, iters);
  run<1>("contiguous 1 KB runs", buf, blocks, iters);
  run<2>("S32 hi/lo halves", buf, blocks, iters);
  run<0>("interleaved 32-byte units", buf, blocks, iters);
  run<1>("contiguous 1 KB runs", buf, blocks, iters);
  // a few workgroups only (one per 4 CUs at most): the per-CU store path without the HBM write limit of the whole chip
  printf("64 workgroups x 1 MB (rates below are per WORKGROUP: multiply the B/clk/CU column by 256/64 = 4):\n");
  run<0>("interleaved, 64 workgroups", buf, 64, 512);
  run<1>("contiguous, 64 workgroups", buf, 64, 512);
  run<2>("S32 halves, 64 workgroups", buf, 64, 512);
  return 0;
}

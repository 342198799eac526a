// Probe for the round-4 finding (hn_amd/build.py EXTRA_FLAGS, profiles/NOTEBOOK.md): packed-fp32 VALU instructions whose op_sel
// takes the LOW result from the HIGH register of a source pair gave wrong results in lanes 48-63 of the tiled preprocess kernel,
// but only while a second process had work on the card.  This probe repeats each form on known operands and counts mismatches
// per 16-lane group; run it alone and next to a load (e.g. `python tools/diag/race_hunt.py load 60 &`).
//   hipcc --offload-arch=gfx950 -O2 -o pk_opsel_probe pk_opsel_probe.hip && ./pk_opsel_probe [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// FORM 0: v_pk_mul_f32 d, a, b                     (low = a.lo * b.lo, high = a.hi * b.hi)
// FORM 1: v_pk_mul_f32 d, a, b op_sel_hi:[0,1]     (low = a.lo * b.lo, high = a.lo * b.hi)     -- the form the j = 0 column used
// FORM 2: v_pk_mul_f32 d, a, b op_sel:[1,0]        (low = a.hi * b.lo, high = a.hi * b.hi)     -- the form the j = 1 column used
// FORM 3: v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]   (low = a.lo * b.hi + c.lo, high = a.hi * b.hi + c.hi)
// LDS = 1: the b operand comes from LDS through ds_read2st64_b32 right before the instruction (as in the kernel)
template <int FORM, int LDS>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) {
  __shared__ float tile[2][1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const long base = ((long)blockIdx.x * 256 + tid) * 6;
  f32x2 a = {in[base], in[base + 1]}, b = {in[base + 2], in[base + 3]}, c = {in[base + 4], in[base + 5]};
  tile[0][tid] = b.x;
  tile[1][tid] = b.y;
  __syncthreads();
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 bb = b, d;
    if (LDS) {
      const unsigned addr = (unsigned)(size_t)&tile[0][tid];
      asm volatile("ds_read2st64_b32 %0, %1 offset1:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(bb) : "v"(addr) : "memory");
    }
    f32x2 want;
    if (FORM == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(bb));
      want = f32x2{a.x * b.x, a.y * b.y};
    } else if (FORM == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(bb));
      want = f32x2{a.x * b.x, a.x * b.y};
    } else if (FORM == 2) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(bb));
      want = f32x2{a.y * b.x, a.y * b.y};
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(bb), "v"(c));
      want = f32x2{__builtin_fmaf(a.x, b.y, c.x), __builtin_fmaf(a.y, b.y, c.y)};
    }
    if (d.x != want.x) wrong |= 1u;
    if (d.y != want.y) wrong |= 2u;
    a.x += 0.f * d.x;   // keep the loop from being hoisted
  }
  if (wrong & 1u) atomicAdd(&bad[(lane >> 4) * 2 + 0], 1u);
  if (wrong & 2u) atomicAdd(&bad[(lane >> 4) * 2 + 1], 1u);
}

template <int FORM, int LDS>
void run(const float* din, unsigned* dbad, int launches) {
  (void)hipMemset(dbad, 0, 8 * sizeof(unsigned));
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((probe<FORM, LDS>), dim3(8192), dim3(256), 0, 0, din, dbad, 64);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  printf("form %d lds %d: %s; wrong lanes (low, high) per 16-lane group: [%u %u] [%u %u] [%u %u] [%u %u] of %ld lane-results\n", FORM,
         LDS, hipGetErrorString(e), h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], (long)launches * 8192 * 256 / 4);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  const long n = 8192L * 256 * 6;
  float* h = (float*)malloc(n * sizeof(float));
  unsigned s = 12345u;
  for (long i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
  }
  float* din;
  unsigned* dbad;
  (void)hipMalloc(&din, n * sizeof(float));
  (void)hipMalloc(&dbad, 8 * sizeof(unsigned));
  (void)hipMemcpy(din, h, n * sizeof(float), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>(din, dbad, launches);
    run<1, 0>(din, dbad, launches);
    run<2, 0>(din, dbad, launches);
    run<3, 0>(din, dbad, launches);
    run<0, 1>(din, dbad, launches);
    run<1, 1>(din, dbad, launches);
    run<2, 1>(din, dbad, launches);
    run<3, 1>(din, dbad, launches);
  }
  return 0;
}

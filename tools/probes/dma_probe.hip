// Probe: global_load_lds_dwordx4 with a per-lane (gathered) source; LDS image = wave-linear 1 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
__global__ void probe(const float* src, const int* row_of_lane, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 256];   // 4 waves x 1 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane L fetches 16 bytes: row row_of_lane[L], chunk (L & 7)  (rows are 128 B = 32 floats)
  const float* g = src + (long)row_of_lane[threadIdx.x] * 32 + (lane & 7) * 4;
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(lds + wave * 256), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) (and everything else)
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) out[i] = lds[i];
}
int main() {
  const int rows = 64;
  std::vector<float> h(rows * 32);
  for (int i = 0; i < rows * 32; ++i) h[i] = i;
  std::vector<int> rl(256);
  for (int t = 0; t < 256; ++t) rl[t] = (t * 7 + 3) % rows;   // arbitrary gather
  float *d, *o; int* r;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 1024 * 4); hipMalloc(&r, 256 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(r, rl.data(), 256 * 4, hipMemcpyHostToDevice);
  probe<<<1, 256>>>(d, r, o);
  std::vector<float> res(1024);
  hipMemcpy(res.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    const int wave = t >> 6, lane = t & 63;
    for (int e = 0; e < 4; ++e) {
      const float want = rl[t] * 32 + (lane & 7) * 4 + e;
      const float got = res[wave * 256 + lane * 4 + e];
      if (want != got) { if (bad < 5) printf("t=%d e=%d want %f got %f\n", t, e, want, got); ++bad; }
    }
  }
  printf("dma probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}

#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(const Big<N> b, int* out) {
  typedef __attribute__((address_space(4))) const Big<N> KA;
  KA* kp = (KA*)__builtin_amdgcn_kernarg_segment_ptr();
  int idx = blockIdx.x;                       // uniform dynamic index -> scalar load from the kernarg segment
  if (threadIdx.x == 0) out[blockIdx.x] = kp->v[(idx * 97) % N] + b.v[N - 1];
}
template <int N> void run() {
  Big<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
  int* d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
  hipLaunchKernelGGL(k<N>, dim3(4), dim3(64), 0, 0, b, d);
  hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
  int h[4] = {0,0,0,0}; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("N=%d bytes=%zu launch=%s sync=%s out=%d %d %d %d (expect %d %d)\n", N, sizeof(b), hipGetErrorString(e), hipGetErrorString(e2), h[0], h[1], h[2], h[3], 0 + N - 1, 97 % N + N - 1);
  hipFree(d);
}
int main() { run<512>(); run<1000>(); run<1020>(); run<2000>(); run<4000>(); run<8000>(); return 0; }

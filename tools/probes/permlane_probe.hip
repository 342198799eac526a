// v_permlane16_swap_b32 semantics on gfx950 (the v8 conv epilogue relies on them): prints, for every lane, which
// (operand, lane) each of the two results came from.   hipcc --offload-arch=gfx950 -O2 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned a = 0x100u + lane, b = 0x200u + lane;   // operand tag | source lane
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
}
int main() {
  unsigned* d;
  unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int row = 0; row < 4; ++row)
    printf("row %d: r[0] <- %c lane %2u..   r[1] <- %c lane %2u..\n", row, (h[row * 16] >> 8) == 1 ? 'a' : 'b', h[row * 16] & 0xff,
           (h[64 + row * 16] >> 8) == 1 ? 'a' : 'b', h[64 + row * 16] & 0xff);
  return 0;
}

// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor)
// write ZEROS to its LDS slot, or leave the slot untouched?  (conv v6 relies on zeros for padding taps.)
// Also checks: soffset takes part in the address but not in the range check; voffset bit 31 => out of range.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void probe(const float* src, int nbytes, int soff, float* out) {
  __shared__ __attribute__((aligned(1024))) float smem[64 * 4];
  const int lane = threadIdx.x;
  for (int e = 0; e < 4; ++e) smem[lane * 4 + e] = -777.f;  // sentinel
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // lanes 0..31: in range (offset lane*16); lanes 32..47: offset with bit 31 set; lanes 48..63: offset just past nbytes
  unsigned voff = lane * 16;
  if (lane >= 32 && lane < 48) voff |= 0x80000000u;
  if (lane >= 48) voff = (unsigned)nbytes + (lane - 48) * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)smem, 16, (int)voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = smem[lane * 4 + e];
}

int main() {
  const int n = 4096;
  float* h = (float*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
  float *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  float r[256];
  for (int soff : {0, 256}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 2048 /* records: first 512 floats */, soff, o);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("soffset %d: lane0 %.0f %.0f | lane31 %.0f | lane32 (bit31) %.0f %.0f | lane47 %.0f | lane48 (past end) %.0f | lane63 %.0f\n",
           soff, r[0], r[1], r[31 * 4], r[32 * 4], r[32 * 4 + 1], r[47 * 4], r[48 * 4], r[63 * 4]);
  }
  // in range with soffset pushing the address past num_records (range check ignores soffset?)
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 2048, 2048 + 1024, o);
  hipDeviceSynchronize();
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("soffset 3072 (address beyond records, voff in range): lane0 %.0f (expect 1768 if soffset is outside the check)\n", r[0]);
  return 0;
}

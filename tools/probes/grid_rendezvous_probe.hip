// What does a grid-wide rendezvous INSIDE a kernel cost on this part?  (The number a persistent kernel that walks the ~60
// launch-bound layers of the A2J trunk at batch 1 would pay instead of the 4.5 us launch-to-launch gap.)
// G resident workgroups (one per CU) run R rounds of: write a payload line with the device-coherent policy (sc1), wait for the
// stores (vmcnt(0)), one agent-scope atomic add on a counter, spin on the counter with agent-scope loads (BOUNDED: a wave that
// does not see the round complete within 2^22 polls gives up and raises a flag -- the probe cannot hang), then read the
// payload of another workgroup (another XCD: workgroup ids are dealt round-robin over the 8 XCDs) and check it.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/grp tools/probes/grid_rendezvous_probe.hip && /tmp/grp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// tree != 0: the workgroups of an XCD (ids congruent mod 8) arrive on their XCD's counter (64 B apart); the last of them arrives
// on the chip's counter, which everybody polls: 32 + 8 serialised atomics on the critical path instead of 256.
__global__ __launch_bounds__(256) void rendezvous_kernel(unsigned* counter, unsigned* payload, int rounds, int payload_words,
                                                         unsigned* bad, unsigned* gave_up, int tree) {
  const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
  // two payload buffers, by round parity: a workgroup that is through rendezvous r writes round r + 1 at once, while a slower one
  // may still be reading round r (the first version of this probe had ONE buffer and -- correctly -- reported wrong words)
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)payload, 0, 2 * G * payload_words * 4, 0x00020000);
  unsigned wrong = 0;
  for (int r = 0; r < rounds; ++r) {
    if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // somebody timed out: everybody leaves
    // payload of this round: payload_words words per workgroup
    for (int i = tid * 4; i < payload_words; i += 256 * 4) {
      const unsigned v = (unsigned)(r * 1315423911u) ^ (unsigned)(w * 2654435761u) ^ (unsigned)i;
      const u32x4 d = {v, v + 1, v + 2, v + 3};
      __builtin_amdgcn_raw_buffer_store_b128(d, rs, (((r & 1) * G + w) * payload_words + i) * 4, 0, 16 /* sc1 */);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    if (tid == 0) {
      unsigned want = (unsigned)(r + 1) * (unsigned)G;
      if (tree) {
        const int x = w & 7, members = (G - x + 7) / 8;   // workgroups of this XCD
        const unsigned t = __hip_atomic_fetch_add(counter + 16 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)(r + 1) * (unsigned)members - 1u)
          __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        want = (unsigned)(r + 1) * (unsigned)(G < 8 ? G : 8);
      } else {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      int polls = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++polls > (1 << 22)) {
          atomicAdd(gave_up, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    // the payload of the workgroup "across the chip"
    const int o = (w + G / 2 + 1) % G;
    for (int i = tid * 4; i < payload_words; i += 256 * 4) {
      const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (((r & 1) * G + o) * payload_words + i) * 4, 0, 16 /* sc1 */);
      const unsigned v = (unsigned)(r * 1315423911u) ^ (unsigned)(o * 2654435761u) ^ (unsigned)i;
      wrong += (d[0] != v) + (d[1] != v + 1) + (d[2] != v + 2) + (d[3] != v + 3);
    }
    // (buffer r & 1 is written again in round r + 2, i.e. after rendezvous r + 1, at which every workgroup has finished these reads)
  }
  if (wrong) atomicAdd(bad, wrong);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned *counter, *payload, *bad, *gave;
  const int max_words = 16384;
  CK(hipMalloc(&counter, 4 * 16 * 9)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&gave, 4));
  CK(hipMalloc(&payload, (size_t)2 * cus * max_words * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# %d CUs; one 256-thread workgroup per CU; us per round (payload write + rendezvous + read of a far workgroup's payload)\n", cus);
  for (int tree : {0, 1})
  for (int G : {64, 128, cus}) {
    for (int words : {0, 512, 1024, 16384}) {   // 0 / 2 KB / 4 KB / 64 KB per workgroup per round
      for (int rep = 0; rep < 2; ++rep) {
        const int rounds = 200;
        CK(hipMemset(counter, 0, 4 * 16 * 9)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(gave, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rendezvous_kernel, dim3(G), dim3(256), 0, 0, counter, payload, rounds, words, bad, gave, tree);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hb = 0, hg = 0;
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hg, gave, 4, hipMemcpyDeviceToHost));
        if (rep == 1)
          printf("%s G %3d  payload %6d B : %6.2f us per round   wrong words %u   waves that gave up %u\n", tree ? "tree" : "flat", G, words * 4,
                 1e3 * ms / rounds, hb, hg);
      }
    }
  }
  return 0;
}

// Could the launch-bound 11 x 11 stage of A2J at batch 1 (~45 dependent launches of ~10 us) run as a layer-walking kernel on the
// CUs of ONE XCD?  Round 5's chain prototype lost because activations that cross XCDs inside a kernel must bypass the L2 (sc1:
// 0.85 TB/s).  Inside one XCD the L2 IS the coherence point: plain stores (the L1 is write-through) + sc0 loads (miss the L1,
// hit the L2) + an atomic counter in that L2.  This probe measures the three primitives such a kernel would pay per layer:
//   (1) a rendezvous of the 32 workgroups of one XCD (workgroup ids congruent mod 8; the XCC_ID register is read to check it),
//       agent-scope atomics (there is no scope between workgroup and agent: an sc0 load may hit this CU's L1 and spin on a
//       stale counter forever -- the first version of this probe did);
//   (2) one layer's filter bank (0.5 - 4 MB, never re-used: a 64 MB ring) streamed by those 32 workgroups;
//   (3) every workgroup writing its slice of a 128 / 512 KB activation map with plain stores and, after the rendezvous, reading
//       the WHOLE map -- with sc1 loads, or with plain loads behind `buffer_inv sc1` (checked word by word: coherence, not
//       just time).
// Every spin is bounded (2^22 polls): the probe cannot hang.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/xcp tools/probes/xcd_chain_probe.hip && /tmp/xcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 load4_sc1(const unsigned* p) {   // agent scope: served past this XCD's L1 AND L2
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Args {
  unsigned* counter;        // one 64-byte line
  const unsigned* weights;  // ring of never-re-used filter banks
  size_t ring_bytes;
  int w_bytes;              // filter bytes per round (all members together); 0 = none
  unsigned* act;            // 2 x act_words (by round parity)
  int act_words;            // activation words per round (all members together); 0 = none
  int rounds, xcd, local;   // local != 0: the map is read with PLAIN loads behind `buffer_inv sc1` (drops this CU's L1 and the
                            // XCD's clean L2 lines; the writers' lines are dirty in the SAME L2 and stay); 0: with sc1 loads
  unsigned *bad, *gave_up, *xcc, *sink;
};

__global__ __launch_bounds__(256) void xcd_chain_kernel(const Args a) {
  if ((int)(blockIdx.x & 7) != a.xcd) return;
  const int m = blockIdx.x >> 3, M = gridDim.x >> 3, tid = threadIdx.x;
  if (tid == 0) a.xcc[m] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits 3:0
  unsigned wrong = 0, acc = 0;
  for (int r = 0; r < a.rounds; ++r) {
    // (2) this member's slice of the round's filter bank
    if (a.w_bytes) {
      const size_t slice = (size_t)a.w_bytes / M;
      const size_t base = ((size_t)r * a.w_bytes) % a.ring_bytes + (size_t)m * slice;
      const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.weights) + base);
      u32x4 s = {0, 0, 0, 0};
      for (size_t i = tid; i < slice / 16; i += 256) {
        const u32x4 d = src[i];
        s[0] ^= d[0]; s[1] ^= d[1]; s[2] ^= d[2]; s[3] ^= d[3];
      }
      acc ^= s[0] ^ s[1] ^ s[2] ^ s[3];
    }
    // (3a) this member's slice of the activation map: plain stores
    if (a.act_words) {
      const int per = a.act_words / M;
      unsigned* dst = a.act + (size_t)(r & 1) * a.act_words + (size_t)m * per;
      for (int i = tid * 4; i < per; i += 256 * 4) {
        const unsigned v = (unsigned)(r * 1315423911u) ^ (unsigned)((m * per + i) * 2654435761u);
        *reinterpret_cast<u32x4*>(dst + i) = u32x4{v, v + 1, v + 2, v + 3};
      }
    }
    // (1) rendezvous of the M members
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the stores have reached the L2 (write-through L1)
    __syncthreads();
    if (tid == 0) {
      const unsigned want = (unsigned)(r + 1) * (unsigned)M;
      __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int polls = 0;
      while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++polls > (1 << 22)) {
          atomicAdd(a.gave_up, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (__hip_atomic_load(a.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    // (3b) the WHOLE map (what the other members wrote must be there)
    if (a.act_words) {
      const unsigned* src = a.act + (size_t)(r & 1) * a.act_words;
      if (a.local) asm volatile("buffer_inv sc1" ::: "memory");
      for (int i = tid * 4; i < a.act_words; i += 256 * 4) {
        const u32x4 d = a.local ? *reinterpret_cast<const u32x4*>(src + i) : load4_sc1(src + i);
        const unsigned v = (unsigned)(r * 1315423911u) ^ (unsigned)(i * 2654435761u);
        wrong += (d[0] != v) + (d[1] != v + 1) + (d[2] != v + 2) + (d[3] != v + 3);
      }
    }
  }
  if (wrong) atomicAdd(a.bad, wrong);
  if (acc == 0x12345678u) a.sink[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  Args a{};
  a.ring_bytes = (size_t)256 << 20;
  unsigned* w;
  CK(hipMalloc(&w, a.ring_bytes));
  CK(hipMemset(w, 1, a.ring_bytes));
  a.weights = w;
  CK(hipMalloc(&a.counter, 64)); CK(hipMalloc(&a.bad, 4)); CK(hipMalloc(&a.gave_up, 4)); CK(hipMalloc(&a.sink, 4));
  CK(hipMalloc(&a.xcc, 4 * 64));
  CK(hipMalloc(&a.act, (size_t)2 * (1 << 20)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# %d CUs; grid %d x 256 threads, only the workgroups with id %% 8 == xcd work (%d members)\n", cus, cus, cus / 8);
  struct Case { int local, w_kb, act_kb; };
  const Case cases[] = {{0, 0, 0}, {1, 512, 0}, {1, 1024, 0}, {1, 2360, 0}, {1, 4096, 0}, {1, 16384, 0},
                        {1, 0, 128}, {0, 0, 128}, {1, 0, 512}, {0, 0, 512}, {1, 1024, 128}, {1, 2360, 128}, {0, 2360, 128},
                        {1, 2360, 512}, {0, 2360, 512}};
  for (int xcd : {0, 3})
    for (const Case& c : cases) {
      float best = 1e9f;
      unsigned hb = 0, hg = 0;
      std::vector<unsigned> xcc(64, 99);
      for (int rep = 0; rep < 3; ++rep) {
        a.rounds = 100;
        a.xcd = xcd; a.local = c.local; a.w_bytes = c.w_kb * 1024 / (cus / 8) / 16 * 16 * (cus / 8); a.act_words = c.act_kb * 256;
        CK(hipMemset(a.counter, 0, 64)); CK(hipMemset(a.bad, 0, 4)); CK(hipMemset(a.gave_up, 0, 4)); CK(hipMemset(a.xcc, 0xff, 4 * 64));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(xcd_chain_kernel, dim3(cus), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        unsigned b = 0, g = 0;
        CK(hipMemcpy(&b, a.bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&g, a.gave_up, 4, hipMemcpyDeviceToHost));
        hb += b; hg += g;
        CK(hipMemcpy(xcc.data(), a.xcc, 4 * (cus / 8), hipMemcpyDeviceToHost));
      }
      bool same = true;
      for (int i = 1; i < cus / 8; ++i) same = same && xcc[i] == xcc[0];
      const double us = 1e3 * best / a.rounds;
      printf("xcd %d (XCC_ID %u%s)  %s  filters %5d KB  map %4d KB : %6.2f us per round", xcd, xcc[0], same ? ", all members" : " MIXED",
             c.local ? "inv+plain" : "sc1 loads", c.w_kb, c.act_kb, us);
      if (c.w_kb) printf("  (%5.2f TB/s of filters if that were all)", c.w_kb * 1024.0 / us / 1e6);
      printf("   wrong words %u   gave up %u\n", hb, hg);
    }
  return 0;
}

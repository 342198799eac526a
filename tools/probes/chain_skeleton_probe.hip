// A LOWER BOUND for a layer-walking chain kernel over A2J's layer3 at batch 1: the 18 dependent launches of
// that stage take 183 us in the product (profiles/r06a_b1_timeline.txt; 10.2 us each).  Would one persistent launch of 256
// workgroups that walks the 18 layers -- grid-wide rendezvous between two layers, activations exchanged through the
// device-coherent path because the XCDs' L2s are not coherent inside a kernel -- be at least 40 us faster?
// This is the SKELETON of such a kernel: per layer every workgroup moves exactly its share of the bytes the real layer moves
// and issues its share of the real MFMAs, with every load of a layer in flight at once and nothing else in the way:
//   * the layer's filter bank, streamed from HBM (never re-used: a ring);
//   * the A operand through sc1 loads: M x K x 4 B (im2col volume) x the number of 64-column tiles that re-read it (the product's
//     32 x 64 tile form; split-K divides K between workgroups and adds no A traffic);
//   * split-K partial planes, written and read back with sc1 (the product's in-kernel reduction);
//   * the output (and the residual read of a block's last layer) through sc1;
//   * the layer's MFMAs (v_mfma_f32_16x16x32_f16, three per MAC tile), chained on the loaded data;
//   * the tree rendezvous of tools/probes/grid_rendezvous_probe.hip (bounded polls: the probe cannot hang).
// No prologue, no descriptor set-up, no epilogue arithmetic, no dependency of the k loop on its own loads: a real chain kernel
// is slower than this.  If the skeleton is not >= 40 us under 183 us, the chain is not worth building.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/csk tools/probes/chain_skeleton_probe.hip && /tmp/csk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Layer {
  int w_bytes;       // filter bank
  int a_bytes;       // A operand through the coherent path (all workgroups together)
  int plane_bytes;   // split-K partial planes: written AND read back (0: unsplit)
  int out_bytes;     // output (+ residual read when res != 0)
  int res;
  int mfmas;         // 16x16x32 MFMAs of the layer (three terms included)
};

struct Args {
  Layer layers[24];
  int count;
  const unsigned* weights; size_t ring_bytes;
  unsigned* act;     // activation / plane scratch, 16 MB
  unsigned* sync;    // chip counter + 8 XCD counters (64 B apart) + gave-up word
  unsigned* sink;
  int rounds;        // the chain is walked `rounds` times in one launch (amortises the launch itself)
  int coherent;      // 1: sc1 loads / stores for activations; 0: plain (NOT coherent: the upper bound of what sc1 costs)
};

__device__ __forceinline__ u32x4 ld(const __amdgpu_buffer_rsrc_t& rs, int off, int coherent) {
  return coherent ? __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
}
__device__ __forceinline__ void st(const u32x4& v, const __amdgpu_buffer_rsrc_t& rs, int off, int coherent) {
  if (coherent) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
  else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
}

__device__ __forceinline__ bool rendezvous(unsigned* sync, unsigned round, int G) {
  const int w = blockIdx.x;
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const int x = w & 7, members = (G - x + 7) / 8;
    const unsigned t = __hip_atomic_fetch_add(sync + 16 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (round + 1) * (unsigned)members - 1u) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = (round + 1) * (unsigned)(G < 8 ? G : 8);
    int polls = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      if (++polls > (1 << 22)) {
        __hip_atomic_store(sync + 16 * 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256, 1) void chain_skeleton_kernel(const Args a) {
  const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t act = __builtin_amdgcn_make_buffer_rsrc((void*)a.act, 0, 16 << 20, 0x00020000);
  unsigned acc_u = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned round = 0;
  for (int r = 0; r < a.rounds; ++r) {
    for (int l = 0; l < a.count; ++l) {
      const Layer L = a.layers[l];
      // this workgroup's share of everything the layer reads, all of it requested before anything is used
      const int w_share = L.w_bytes / G / 4096 * 4096 + 4096, a_share = L.a_bytes / G / 4096 * 4096 + 4096;
      const size_t w_base = ((size_t)(r * a.count + l) * (size_t)L.w_bytes) % (a.ring_bytes - (size_t)G * w_share) + (size_t)w * w_share;
      const u32x4* wsrc = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.weights) + w_base);
      u32x4 s = {0, 0, 0, 0};
      for (int i = tid; i < w_share / 16; i += 256) {
        const u32x4 d = wsrc[i];
        s[0] ^= d[0]; s[1] ^= d[1]; s[2] ^= d[2]; s[3] ^= d[3];
      }
      // (the activation buffer of a layer is at most 512 KB: every workgroup reads ITS share from inside the first MB)
      const int a_base = (int)(((size_t)w * a_share) & ((1 << 20) - 1));
      for (int i = tid; i < a_share / 16; i += 256) {
        const u32x4 d = ld(act, (a_base + i * 16) & ((1 << 20) - 1), a.coherent);
        s[0] ^= d[0]; s[1] ^= d[1]; s[2] ^= d[2]; s[3] ^= d[3];
      }
      if (L.res) {
        const int share = L.out_bytes / G / 1024 * 1024 + 1024;
        for (int i = tid; i < share / 16; i += 256) {
          const u32x4 d = ld(act, (2 << 20) + w * share + i * 16, a.coherent);
          s[0] ^= d[0]; s[1] ^= d[1];
        }
      }
      // the layer's MFMAs: this wave's share, chained on what it loaded
      const int mf = (L.mfmas + G * 4 - 1) / (G * 4);
      f16x8 fa, fb;
      for (int e = 0; e < 8; ++e) {
        fa[e] = (_Float16)(float)((s[e & 3] >> e) & 1);
        fb[e] = (_Float16)(float)((s[(e + 1) & 3] >> e) & 1);
      }
      for (int i = 0; i < mf; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
      const u32x4 o = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), s[2], s[3]};
      if (L.plane_bytes) {   // partial planes out, acknowledged, and back in (the in-kernel reduction's traffic; its tickets are left out)
        const int share = L.plane_bytes / G / 1024 * 1024 + 1024;
        for (int i = tid; i < share / 16; i += 256) st(o, act, (4 << 20) + w * share + i * 16, a.coherent);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int i = tid; i < share / 16; i += 256) {
          const u32x4 d = ld(act, (4 << 20) + ((w + G / 2) % G) * share + i * 16, a.coherent);
          acc_u ^= d[0] ^ d[1];
        }
      }
      {
        const int share = L.out_bytes / G / 1024 * 1024 + 1024;
        u32x4 oo = o;
        oo[0] ^= acc_u;
        for (int i = tid; i < share / 16; i += 256) st(oo, act, (8 << 20) + w * share + i * 16, a.coherent);
      }
      acc_u ^= s[0] ^ s[1] ^ s[2] ^ s[3];
      if (!rendezvous(a.sync, round, G)) return;
      ++round;
    }
  }
  if (acc_u == 0x12345678u) a.sink[0] = acc_u + (unsigned)acc[2];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static Layer conv(int M, int K, int N, int splits, int res) {
  Layer L;
  const int Mp = (M + 31) / 32 * 32;
  L.w_bytes = N * K * 4;
  L.a_bytes = Mp * K * 4 * ((N + 63) / 64);
  L.plane_bytes = splits > 1 ? splits * Mp * N * 4 : 0;
  L.out_bytes = Mp * N * 4;
  L.res = res;
  L.mfmas = (Mp / 16) * (N / 16) * (K / 32) * 3;
  return L;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  Args a{};
  // A2J layer3 at one crop (a2j/resnet.py:99-147: Bottleneck x 6, 1024 channels, 11 x 11 after the stride-2 block), as the 18
  // launches of the product: block 0 = [conv1 512 -> 256 on 22 x 22 beside the 1 x 1 / stride-2 downsample 512 -> 1024],
  // [3 x 3 / stride 2], [1 x 1 256 -> 1024 + residual]; blocks 1..5 = 1 x 1 1024 -> 256, 3 x 3, 1 x 1 + residual.
  // splits: the product's plan for these shapes (profiles/r04_splitk_sweep_b1.txt)
  int n = 0;
  Layer first = conv(484, 512, 256, 4, 0), ds = conv(121, 512, 1024, 1, 0);
  first.w_bytes += ds.w_bytes; first.a_bytes += ds.a_bytes; first.out_bytes += ds.out_bytes; first.mfmas += ds.mfmas;
  a.layers[n++] = first;
  a.layers[n++] = conv(121, 2304, 256, 16, 0);
  a.layers[n++] = conv(121, 256, 1024, 1, 1);
  for (int b = 1; b < 6; ++b) {
    a.layers[n++] = conv(121, 1024, 256, 8, 0);
    a.layers[n++] = conv(121, 2304, 256, 16, 0);
    a.layers[n++] = conv(121, 256, 1024, 1, 1);
  }
  a.count = n;
  a.ring_bytes = (size_t)512 << 20;
  unsigned* w;
  CK(hipMalloc(&w, a.ring_bytes));
  CK(hipMemset(w, 1, a.ring_bytes));
  a.weights = w;
  CK(hipMalloc(&a.act, 16 << 20));
  CK(hipMemset(a.act, 0, 16 << 20));
  CK(hipMalloc(&a.sync, 4 * 16 * 12));
  CK(hipMalloc(&a.sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double wsum = 0, asum = 0, psum = 0, osum = 0; long msum = 0;
  for (int l = 0; l < n; ++l) { wsum += a.layers[l].w_bytes; asum += a.layers[l].a_bytes; psum += 2.0 * a.layers[l].plane_bytes; osum += a.layers[l].out_bytes * (1 + a.layers[l].res); msum += a.layers[l].mfmas; }
  printf("# %d CUs, %d layers: filters %.1f MB, A operand through the coherent path %.1f MB, partial planes (out + back) %.1f MB, outputs + residuals %.1f MB, %ld MFMAs\n",
         G, n, wsum / 1e6, asum / 1e6, psum / 1e6, osum / 1e6, msum);
  printf("# the product: 18 separate launches = 183 us (profiles/r06a_b1_timeline.txt)\n");
  struct Case { const char* name; int coherent; int planes; int afrac; };
  // variants: everything; without the split-K planes (an unsplit chain: longer k loops instead); activations NOT coherent (what sc1 costs)
  const Case cases[] = {{"sc1 activations, split-K planes", 1, 1, 1}, {"sc1 activations, no planes (unsplit)", 1, 0, 1},
                        {"plain (incoherent) activations, planes: the price of sc1", 0, 1, 1}, {"rendezvous + filters only", 1, 0, 0}};
  for (const Case& c : cases) {
    Args b = a;
    for (int l = 0; l < n; ++l) {
      if (!c.planes) b.layers[l].plane_bytes = 0;
      if (!c.afrac) { b.layers[l].a_bytes = 0; b.layers[l].out_bytes = 0; b.layers[l].res = 0; b.layers[l].mfmas = 0; }
    }
    b.coherent = c.coherent;
    b.rounds = 50;
    float best = 1e9f;
    unsigned gave = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(b.sync, 0, 4 * 16 * 12));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(chain_skeleton_kernel, dim3(G), dim3(256), 0, 0, b);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      unsigned g = 0;
      CK(hipMemcpy(&g, b.sync + 16 * 10, 4, hipMemcpyDeviceToHost));
      gave += g;
    }
    printf("%-62s : %7.1f us per walk of the 18 layers (%5.2f us per layer)   gave up %u\n", c.name, 1e3 * best / b.rounds,
           1e3 * best / b.rounds / n, gave);
  }
  return 0;
}

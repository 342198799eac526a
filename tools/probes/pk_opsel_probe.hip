// Probe for the round-4 finding (hn_amd/build.py EXTRA_FLAGS, profiles/NOTEBOOK.md): a packed-fp32 VALU instruction whose op_sel
// takes the LOW result from the HIGH register of a source pair gave wrong results in lanes 48-63 of the tiled preprocess kernel,
// but only while a second process had work on the card.  This probe runs every packed-32-bit form the library's disassembly
// holds (and the op_sel permutations around the failing one) on known operands and counts wrong results per 16-lane group; run
// it alone and next to a load (e.g. `python tests/card_load.py 60 &`).
//   hipcc --offload-arch=gfx950 -O2 -o pk_opsel_probe pk_opsel_probe.hip && ./pk_opsel_probe [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// packed semantics: low result = op(src_i[o_i]), high result = op(src_i[h_i]); defaults o = 0, h = 1; neg_lo / neg_hi negate a source
struct Form {
  const char* text;
  int kind;                 // 0 mul, 1 add, 2 fma, 3 mov
  int o[3], h[3], nl[3], nh[3];
};
#define FORMS(X)                                                                                                                  \
  X(0, "v_pk_mul_f32 %0, %1, %2", 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                                           \
  X(1, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]", 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0)                                           \
  X(2, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]", 0, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                              \
  X(3, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", 0, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                              \
  X(4, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]", 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0)                              \
  X(5, "v_pk_add_f32 %0, %1, %2", 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                                           \
  X(6, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]", 1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0)                                           \
  X(7, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0]", 1, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                              \
  X(8, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", 1, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                              \
  X(9, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]", 1, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 0)                                 \
  X(10, "v_pk_fma_f32 %0, %1, %2, %3", 2, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                                      \
  X(11, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]", 2, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0)                                    \
  X(12, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]", 2, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0)                                    \
  X(13, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]", 2, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0)                                    \
  X(14, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]", 2, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                       \
  X(15, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", 2, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                       \
  X(16, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]", 2, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                       \
  X(17, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[0,0,0]", 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0)                     \
  X(18, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]", 2, 0, 0, 0, 1, 1, 1, 1, 0, 0, 1, 0, 0)                        \
  X(19, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]", 3, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)                                             \
  X(20, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]", 3, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0)
constexpr int kForms = 21;

#define X_TABLE(id, text, kind, o0, o1, o2, h0, h1, h2, nl0, nl1, nl2, nh0, nh1, nh2) \
  {text, kind, {o0, o1, o2}, {h0, h1, h2}, {nl0, nl1, nl2}, {nh0, nh1, nh2}},
static const Form kTable[kForms] = {FORMS(X_TABLE)};

template <int F>
__device__ __forceinline__ f32x2 issue(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
#define X_ASM(id, text, kind, ...)                                                      \
  if constexpr (F == id) asm volatile(text : "=v"(d) : "v"(a), "v"(b), "v"(c));
  FORMS(X_ASM)
  return d;
}

__device__ __forceinline__ float half_of(f32x2 v, int hi, int neg) {
  const float x = hi ? v.y : v.x;
  return neg ? -x : x;
}

// LDS = 1: the b operand comes from LDS through ds_read2st64_b32 right before the instruction (as in the kernel)
template <int F, int LDS>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, Form f) {
  __shared__ float tile[2][1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const long base = ((long)blockIdx.x * 256 + tid) * 6;
  f32x2 a = {in[base], in[base + 1]}, b = {in[base + 2], in[base + 3]}, c = {in[base + 4], in[base + 5]};
  tile[0][tid] = b.x;
  tile[1][tid] = b.y;
  __syncthreads();
  f32x2 want;
  if (f.kind == 3) {   // v_pk_mov_b32: low = src0[o0], high = src1[o1]
    want = f32x2{half_of(a, f.o[0], 0), half_of(b, f.o[1], 0)};
  } else {
    const float al = half_of(a, f.o[0], f.nl[0]), bl = half_of(b, f.o[1], f.nl[1]), cl = half_of(c, f.o[2], f.nl[2]);
    const float ah = half_of(a, f.h[0], f.nh[0]), bh = half_of(b, f.h[1], f.nh[1]), ch = half_of(c, f.h[2], f.nh[2]);
    want = f.kind == 0 ? f32x2{al * bl, ah * bh} : f.kind == 1 ? f32x2{al + bl, ah + bh}
                                                               : f32x2{__builtin_fmaf(al, bl, cl), __builtin_fmaf(ah, bh, ch)};
  }
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 bb = b;
    if (LDS) {
      const unsigned addr = (unsigned)(size_t)&tile[0][tid];
      asm volatile("ds_read2st64_b32 %0, %1 offset1:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(bb) : "v"(addr) : "memory");
    }
    const f32x2 d = issue<F>(a, bb, c);
    if (__float_as_uint(d.x) != __float_as_uint(want.x)) wrong |= 1u;
    if (__float_as_uint(d.y) != __float_as_uint(want.y)) wrong |= 2u;
  }
  if (wrong & 1u) atomicAdd(&bad[(lane >> 4) * 2 + 0], 1u);
  if (wrong & 2u) atomicAdd(&bad[(lane >> 4) * 2 + 1], 1u);
}

// `self` modes: the competing work comes from THIS process on a second stream, two workgroups per CU; the kind of work is
// selectable so that the trigger can be narrowed down: m = MFMA loop, v = plain VALU (v_fma_f32) loop, l = LDS read / write loop,
// g = global-memory read loop (L2-resident 64 KB), d = LDS-DMA loop (global_load_lds_dwordx4), t = transcendental (v_exp_f32) loop
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
template <int KIND>
__global__ __launch_bounds__(256) void burn(float* out, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  float acc = threadIdx.x * 0.001f;
  if constexpr (KIND == 0) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
      a[i] = (_Float16)(threadIdx.x * 0.001f + i);
      b[i] = (_Float16)(blockIdx.x * 0.002f - i);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int it = 0; it < iters; ++it) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc3, 0, 0, 0);
    }
    acc = acc0[0] + acc1[1] + acc2[2] + acc3[3];
  } else if constexpr (KIND == 1) {
    float x0 = acc, x1 = acc + 1.f, x2 = acc + 2.f, x3 = acc + 3.f;
    for (int it = 0; it < iters * 4; ++it) {
      x0 = __builtin_fmaf(x0, 1.0000001f, 0.5f);
      x1 = __builtin_fmaf(x1, 0.9999999f, 0.25f);
      x2 = __builtin_fmaf(x2, 1.0000002f, 0.125f);
      x3 = __builtin_fmaf(x3, 0.9999998f, 0.0625f);
    }
    acc = x0 + x1 + x2 + x3;
  } else if constexpr (KIND == 2) {
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    int idx = threadIdx.x;
    for (int it = 0; it < iters * 2; ++it) {
      const float v = lds[idx & 4095];
      lds[(idx + 1024) & 4095] = v + 1.f;
      idx = idx * 5 + 17;
      acc += v;
    }
  } else if constexpr (KIND == 3) {
    int idx = threadIdx.x + blockIdx.x * 7;
    for (int it = 0; it < iters; ++it) {
      acc += src[idx & 16383];
      idx = idx * 5 + 17;
    }
  } else if constexpr (KIND == 4) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 65536, 0x00020000);
    const int wave_base = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 64;
    for (int it = 0; it < iters; ++it) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)((char*)lds + wave_base * 16), 16, (int)((threadIdx.x & 63) * 16), (it & 15) * 4096, 0, 0);
      if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += lds[threadIdx.x];
  } else {
    float x = acc * 0.01f;
    for (int it = 0; it < iters * 2; ++it) x = __builtin_amdgcn_exp2f(x) * 0.5f;
    acc = x;
  }
  if (acc == 12345.678f) out[0] = 1.f;
}
static int g_kind = 0;
static float* g_burn_src = nullptr;
static void launch_burn(hipStream_t st, float* out) {
  const int it = 400000;
  switch (g_kind) {
    case 0: hipLaunchKernelGGL(burn<0>, dim3(512), dim3(256), 0, st, out, g_burn_src, it); break;
    case 1: hipLaunchKernelGGL(burn<1>, dim3(512), dim3(256), 0, st, out, g_burn_src, it); break;
    case 2: hipLaunchKernelGGL(burn<2>, dim3(512), dim3(256), 0, st, out, g_burn_src, it / 4); break;
    case 3: hipLaunchKernelGGL(burn<3>, dim3(512), dim3(256), 0, st, out, g_burn_src, it / 8); break;
    case 4: hipLaunchKernelGGL(burn<4>, dim3(512), dim3(256), 0, st, out, g_burn_src, it / 4); break;
    default: hipLaunchKernelGGL(burn<5>, dim3(512), dim3(256), 0, st, out, g_burn_src, it); break;
  }
}
static hipStream_t g_side = nullptr;
static float* g_burn_out = nullptr;

template <int F, int LDS>
void run(const float* din, unsigned* dbad, int launches) {
  (void)hipMemset(dbad, 0, 8 * sizeof(unsigned));
  (void)hipDeviceSynchronize();
  if (g_side) launch_burn(g_side, g_burn_out);   // competing work beside the probe
  for (int l = 0; l < launches; ++l)
    hipLaunchKernelGGL((probe<F, LDS>), dim3(8192), dim3(256), 0, 0, din, dbad, 64, kTable[F]);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[8];
  (void)hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned total = 0;
  for (int i = 0; i < 8; ++i) total += h[i];
  printf("%-68s lds %d: %s; wrong (low high) per 16-lane group: [%u %u] [%u %u] [%u %u] [%u %u]%s\n", kTable[F].text, LDS,
         hipGetErrorString(e), h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], total ? "   <-- WRONG" : "");
  fflush(stdout);
}

template <int F>
void run_all(const float* din, unsigned* dbad, int launches) {
  if constexpr (F < kForms) {
    run<F, 0>(din, dbad, launches);
    run<F, 1>(din, dbad, launches);
    run_all<F + 1>(din, dbad, launches);
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 50;
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
  if (argc > 2 && argv[2][0] == 's') {   // "self" [m|v|l|g|d|t]
    (void)hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking);
    (void)hipMalloc(&g_burn_out, 16);
    (void)hipMalloc(&g_burn_src, 65536 + 4096);
    (void)hipMemset(g_burn_src, 0, 65536 + 4096);
    const char kind = argc > 3 ? argv[3][0] : 'm';
    const char* kinds = "mvlgdt";
    for (int i = 0; kinds[i]; ++i)
      if (kinds[i] == kind) g_kind = i;
    const char* names[] = {"MFMA", "plain VALU", "LDS read/write", "global reads", "LDS-DMA", "transcendental"};
    printf("competing %s work from this process on a second stream\n", names[g_kind]);
  }
  printf("%d launches x 8192 workgroups x 256 lanes x 64 issues per form = %ld lane-results per 16-lane group and half\n", launches,
         (long)launches * 8192 * 256 / 4);
  run_all<0>(din, dbad, launches);
  return 0;
}

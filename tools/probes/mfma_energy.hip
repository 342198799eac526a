// Energy probe: register-only MFMA loops on RANDOM operands, 2 workgroups x 4 waves per CU,
//   f16 : v_mfma_f32_16x16x32_f16   (what the f16x3 conv issues, 3 per MAC)
//   i8  : v_mfma_i32_16x16x64_i8    (a 3-slice integer split would issue 6 per MAC at twice the rate)
//   f16w: v_mfma_f32_32x32x16_f16   (round 3: the wide tile reads half the operand registers per flop -- does it hold a
//                                    higher clock at the power cap?)
// Reports issued ops/s, the shader clock the chip holds meanwhile (s_memtime vs s_memrealtime inside the kernel) and, from
// the host, rocm-smi's socket power sampled while the loop runs: ops/s/W decides whether "fewer joules per MAC" exists.
//   hipcc --offload-arch=gfx950 -O2 mfma_energy.hip -o mfma_energy && ./mfma_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I8>   // 0: f16 16x16x32, 1: i8 16x16x64, 2: f16 32x32x16
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, int iters, float* mhz) {
  i32x4 a[4], b[4];   // 128-bit operands either way (8 halfs or 16 int8)
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const i32x4*>(in + (threadIdx.x * 4 + i * 1024) % 8192);
    b[i] = *reinterpret_cast<const i32x4*>(in + (threadIdx.x * 4 + i * 1024 + 512) % 8192);
  }
  f32x4 accf[4] = {};
  i32x4 acci[4] = {};
  f32x16 accw[4] = {};
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (I8 == 1)
          acci[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + r) & 3], b[(t + 2 * r) & 3], acci[t], 0, 0, 0);
        else if constexpr (I8 == 2)
          accw[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(t + r) & 3]),
                                                           __builtin_bit_cast(f16x8, b[(t + 2 * r) & 3]), accw[t], 0, 0, 0);
        else
          accf[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(t + r) & 3]),
                                                           __builtin_bit_cast(f16x8, b[(t + 2 * r) & 3]), accf[t], 0, 0, 0);
      }
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 4; ++r) s += accf[i][r] + (float)acci[i][r] + accw[i][r] + accw[i][r + 12];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) mhz[0] = (float)((double)(c1 - c0) / ((double)(r1 - r0) * 0.01));
}

static double smi_power() {
  FILE* f = popen("rocm-smi --showpower 2>/dev/null", "r");
  if (!f) return -1;
  char line[512];
  double w = -1;
  while (fgets(line, sizeof line, f)) {
    std::string s(line);
    if (s.find("Socket Graphics Package Power") != std::string::npos || s.find("Average Graphics Package Power") != std::string::npos)
      w = atof(s.substr(s.rfind(':') + 1).c_str());
  }
  pclose(f);
  return w;
}

template <int I8>
static void run(const char* name, const unsigned* d, float* o, float* mhz, int blocks) {
  const int iters = I8 == 2 ? 30000 : 60000;
  k<I8><<<blocks, 256>>>(d, o, 1000, mhz);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double watts = -1;
  std::thread sampler([&] { std::this_thread::sleep_for(std::chrono::milliseconds(1200)); watts = smi_power(); });
  hipEventRecord(e0);
  for (int r = 0; r < 300; ++r) k<I8><<<blocks, 256>>>(d, o, iters, mhz);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  sampler.join();
  float ms, clk;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 300;
  hipMemcpy(&clk, mhz, 4, hipMemcpyDeviceToHost);
  const double ops = 2.0 * (I8 == 2 ? 32 * 32 * 16 : 16 * 16 * (I8 ? 64 : 32)) * 24.0 * iters * 4.0 * blocks;   // 24 MFMAs per iteration per wave
  const double tops = ops / ms / 1e9;
  printf("%-4s blocks %4d: %8.2f ms  %7.1f T(FL)OP/s issued  clock %4.0f MHz  power %6.1f W  %6.3f TOP/s/W  (%.1f %% of the %s peak at that clock)\n",
         name, blocks, ms, tops, clk, watts, watts > 0 ? tops / watts : 0.0, 100.0 * tops / ((I8 == 1 ? 5000.0 : 2500.0) * clk / 2400.0),
         I8 == 1 ? "5.0 POP/s i8" : "2.5 PFLOP/s f16");
}

int main() {
  std::vector<unsigned> h(8192);
  for (auto& v : h) {   // random bytes for i8; for f16 keep the exponent small (|x| < 2): clear bit 14 of both halves
    v = ((unsigned)rand() << 16) ^ (unsigned)rand();
    v &= 0xBFFFBFFFu;
  }
  unsigned* d;
  float *o, *mhz;
  hipMalloc(&d, h.size() * 4);
  hipMalloc(&o, 2048 * 256 * 4);
  hipMalloc(&mhz, 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("f16", d, o, mhz, 512);
    run<2>("f16w", d, o, mhz, 512);
    run<1>("i8", d, o, mhz, 512);
  }
  return 0;
}

// mfma_chain_probe.hip -- how many independent accumulators does v_mfma_i32_32x32x32_i8 (and 16x16x64) need to issue
// at its pipe rate?  NACC accumulators used round-robin, W waves per SIMD; cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC, bool SMALL>
__global__ __launch_bounds__(1024) void k(const int *in, int *out, long long *cycles, int rounds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v4i a = {in[lane], in[lane + 64], in[lane + 128], in[lane + 192]};
  v16i acc[NACC];
  v4i acs[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    acs[i] = v4i{0, 0, 0, 0};
  }
  const long long t0 = clock64();
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (SMALL) acs[r % NACC] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, acs[r % NACC], 0, 0, 0);
      else acc[r % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, acc[r % NACC], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) s ^= acc[i][r];
    s ^= acs[i].x ^ acs[i].y ^ acs[i].z ^ acs[i].w;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool SMALL>
void run(int waves, int *d_in, int *d_out, long long *d_cyc) {
  const int rounds = 400, blocks = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(64 * waves), 0, 0, d_in, d_out, d_cyc, rounds);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(64 * waves), 0, 0, d_in, d_out, d_cyc, rounds);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> c(16);
  CK(hipMemcpy(c.data(), d_cyc, sizeof(long long) * 16, hipMemcpyDeviceToHost));
  long long mx = 0;
  for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
  const double n = (double)rounds * 16;
  const double ops = SMALL ? 2.0 * 16 * 16 * 64 : 2.0 * 32 * 32 * 32;
  printf("  %s acc %d waves/SIMD %d: %6.1f clock64 ticks per MFMA per wave, %6.1f per SIMD, %6.0f TOPS (%.3f ms)\n",
         SMALL ? "16x16x64" : "32x32x32", NACC, waves / 4, (double)mx / n, (double)mx / n / (waves / 4.0),
         ops * n * waves * blocks / (ms * 1e-3) * 1e-12, ms);
}

int main() {
  int *d_in, *d_out;
  long long *d_cyc;
  CK(hipMalloc(&d_in, 4096));
  CK(hipMemset(d_in, 1, 4096));
  CK(hipMalloc(&d_out, 4 * 1024 * 1024));
  CK(hipMalloc(&d_cyc, 8 * 16 * 4096));
  for (int waves : {4, 8, 16}) {
    run<1, false>(waves, d_in, d_out, d_cyc);
    run<2, false>(waves, d_in, d_out, d_cyc);
    run<4, false>(waves, d_in, d_out, d_cyc);
  }
  for (int waves : {4, 8, 16}) {
    run<1, true>(waves, d_in, d_out, d_cyc);
    run<2, true>(waves, d_in, d_out, d_cyc);
    run<4, true>(waves, d_in, d_out, d_cyc);
    run<8, true>(waves, d_in, d_out, d_cyc);
  }
  return 0;
}

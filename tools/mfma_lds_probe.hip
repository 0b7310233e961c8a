// mfma_lds_probe.hip -- gfx950 micro-measurements behind the MFMA accumulation kernel (k3m.hip.h):
//   1. ds_read_b128 at byte-misaligned LDS addresses: is it correct, what does it cost per wave-instruction
//      (aligned stream / misaligned stream / the im2col pattern of the kernel: lane = neighbour offset,
//      address = (y + cy) * pitch + 16 * half + cx);
//   2. v_mfma_i32_32x32x32_i8 with A = B = the same registers (a SYRK step): result layout and rate;
//   3. the kernel's inner loop (one misaligned 16-byte read + one MFMA per 32 samples): cycles per step
//      at 1, 2, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_probe mfma_lds_probe.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v4i_u __attribute__((ext_vector_type(4), aligned(1)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4acc __attribute__((ext_vector_type(4)));

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

constexpr int kLds = 48 * 1024;

// pattern -> LDS byte address of the lane's 16-byte read at step `it`
// 0: aligned stream  1: misaligned stream (+mis)  2: im2col, pitch given
__device__ __forceinline__ int lane_addr(int pattern, int lane, int mis, int pitch) {
  if (pattern == 0) return lane * 16;
  if (pattern == 1) return lane * 16 + mis;
  const int i = lane & 31, h = lane >> 5;
  const int ii = i < 25 ? i : 24;  // spare matrix rows repeat the last entry
  const int cy = ii / 7, cx = ii % 7;
  return cy * pitch + 16 * h + cx + mis;
}

template <int READS>
__global__ __launch_bounds__(1024) void k_read(const int *in, int *out, long long *cycles, int pattern, int mis, int pitch, int iters) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[kLds];
  for (int i = threadIdx.x; i < kLds / 4; i += blockDim.x) ((int *)tile)[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = lane_addr(pattern, lane, mis, pitch) + wave * 2048;
  v4i acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < READS; ++r) {
      const v4i_u *p = (const v4i_u *)(tile + base + r * (pattern == 2 ? pitch : 1024));
      v4i v = *p;
      acc ^= v;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = clock64();
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// correctness of a misaligned read: out[lane] = the 16 bytes at tile + lane + 16 * lane
__global__ void k_check(const uint8_t *in, uint8_t *out) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tile[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x;
  const v4i_u *p = (const v4i_u *)(tile + 17 * lane);
  v4i v = *p;
  *(v4i_u *)(out + 16 * lane) = v;
}

// one SYRK step: S = V V^T with V[i][k], lane l holds row i = l & 31, k-half l >> 5 (16 bytes)
__global__ void k_syrk(const int8_t *v, int *out) {
  const int lane = threadIdx.x;
  v4i a = *(const v4i_u *)(v + (lane & 31) * 32 + (lane >> 5) * 16);
  v16i acc = {0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}

// the loop: STEPS x (misaligned read + mfma), `rounds` times
template <int STEPS>
__global__ __launch_bounds__(1024) void k_loop(const int *in, int *out, long long *cycles, int pitch, int rounds, int mode) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[kLds];
  for (int i = threadIdx.x; i < kLds / 4; i += blockDim.x) ((int *)tile)[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = lane_addr(2, lane, 5, pitch) + (wave & 3) * 64;
  v16i acc = {0};
  v4i keep = *(const v4i_u *)(tile + base);
  const long long t0 = clock64();
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int r = 0; r < STEPS; ++r) {
      v4i v = keep;
      if (mode != 1) v = *(const v4i_u *)(tile + base + r * pitch);   // mode 1: MFMA only
      if (mode == 3) v &= keep;                                       // + window mask
      if (mode != 2) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v, v, acc, 0, 0, 0);
      else acc[r & 15] ^= v.x ^ v.y ^ v.z ^ v.w;                        // mode 2: reads only
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
  int s = 0;
  for (int r = 0; r < 16; ++r) s ^= acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  int *d_in, *d_out;
  long long *d_cyc;
  std::vector<int> h_in(kLds / 4);
  for (size_t i = 0; i < h_in.size(); ++i) h_in[i] = (int)(i * 2654435761u);
  CK(hipMalloc(&d_in, kLds));
  CK(hipMalloc(&d_out, 4 * 1024 * 1024));
  CK(hipMalloc(&d_cyc, 8 * 16 * 4096));
  CK(hipMemcpy(d_in, h_in.data(), kLds, hipMemcpyHostToDevice));

  // ---- 1a. correctness of misaligned b128 ----
  {
    std::vector<uint8_t> src(4096), got(1024);
    for (int i = 0; i < 4096; ++i) src[i] = (uint8_t)(i * 37 + (i >> 8));
    uint8_t *ds, *dd;
    CK(hipMalloc(&ds, 4096));
    CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(ds, src.data(), 4096, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, ds, dd);
    CK(hipMemcpy(got.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int b = 0; b < 16; ++b) bad += got[16 * l + b] != src[17 * l + b];
    printf("misaligned ds_read_b128 (address = 17 * lane): %s (%d wrong bytes)\n", bad ? "WRONG" : "correct", bad);
  }
  // ---- 2. SYRK layout ----
  {
    std::vector<int8_t> v(32 * 32);
    for (int i = 0; i < 32; ++i)
      for (int k = 0; k < 32; ++k) v[i * 32 + k] = (int8_t)(((i * 7 + k * 13 + (i * k) % 5) % 255) - 127);
    int8_t *dv;
    int *dout;
    CK(hipMalloc(&dv, 1024));
    CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(dv, v.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_syrk, dim3(1), dim3(64), 0, 0, dv, dout);
    std::vector<int> got(1024);
    CK(hipMemcpy(got.data(), dout, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r) {
        const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int ref = 0;
        for (int k = 0; k < 32; ++k) ref += (int)v[row * 32 + k] * (int)v[col * 32 + k];
        bad += ref != got[lane * 16 + r];
      }
    printf("mfma_i32_32x32x32_i8(A = B = V rows): S = V V^T at row=(r&3)+8*(r>>2)+4*(lane>>5), col=lane&31: %s (%d wrong)\n",
           bad ? "WRONG" : "correct", bad);
  }
  // ---- 1b. read cost ----
  auto run_read = [&](int waves, int pattern, int mis, int pitch) {
    const int iters = 2000;
    constexpr int READS = 16;
    hipLaunchKernelGGL(k_read<READS>, dim3(1), dim3(64 * waves), 0, 0, d_in, d_out, d_cyc, pattern, mis, pitch, iters);
    CK(hipDeviceSynchronize());
    std::vector<long long> c(16);
    CK(hipMemcpy(c.data(), d_cyc, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
    // LDS cycles per wave-instruction on the CU = wall cycles / (reads per wave * waves)
    return (double)mx / ((double)iters * READS * waves);
  };
  printf("\nds_read_b128 cost, one CU (wall cycles per wave-instruction; 4.0 = 256 B/clk):\n");
  printf("%-44s %8s %8s %8s\n", "pattern", "4 waves", "8 waves", "16 waves");
  struct P { const char *name; int pattern, mis, pitch; };
  const P ps[] = {
      {"aligned stream", 0, 0, 0},          {"stream +1 byte", 1, 1, 0},         {"stream +2 bytes", 1, 2, 0},
      {"stream +4 bytes", 1, 4, 0},         {"stream +8 bytes", 1, 8, 0},        {"stream +5 bytes", 1, 5, 0},
      {"im2col pitch 64, +0", 2, 0, 64},    {"im2col pitch 64, +5", 2, 5, 64},   {"im2col pitch 192, +5", 2, 5, 192},
      {"im2col pitch 144, +5", 2, 5, 144},  {"im2col pitch 320, +5", 2, 5, 320}, {"im2col pitch 160, +5", 2, 5, 160},
      {"im2col pitch 48, +5", 2, 5, 48},    {"im2col pitch 80, +5", 2, 5, 80},
  };
  for (const P &p : ps)
    printf("%-44s %8.2f %8.2f %8.2f\n", p.name, run_read(4, p.pattern, p.mis, p.pitch), run_read(8, p.pattern, p.mis, p.pitch),
           run_read(16, p.pattern, p.mis, p.pitch));

  // ---- 3. the loop ----
  auto run_loop = [&](int waves, int pitch, int mode, int blocks) {
    const int rounds = 200;
    constexpr int STEPS = 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_loop<STEPS>, dim3(blocks), dim3(64 * waves), 0, 0, d_in, d_out, d_cyc, pitch, rounds, mode);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_loop<STEPS>, dim3(blocks), dim3(64 * waves), 0, 0, d_in, d_out, d_cyc, pitch, rounds, mode);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(16);
    CK(hipMemcpy(c.data(), d_cyc, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
    const double per_simd = (double)mx / ((double)rounds * STEPS * (waves / 4.0));  // cycles per MFMA per SIMD
    const double tops = 2.0 * 32 * 32 * 32 * (double)rounds * STEPS * waves * blocks / (ms * 1e-3) * 1e-12;
    printf("  waves/CU %2d pitch %3d mode %d blocks %4d: %6.1f cycles per step per SIMD, kernel %.3f ms, %.0f TOPS\n", waves, pitch,
           mode, blocks, per_simd, ms, mode == 2 ? 0.0 : tops);
  };
  printf("\nloop of (16-byte im2col read + mfma 32x32x32 i8); mode 0 read+mfma, 1 mfma only, 2 read only, 3 read+and+mfma\n");
  for (int mode = 0; mode < 4; ++mode)
    for (int waves : {4, 8, 16}) run_loop(waves, 64, mode, 256);
  for (int waves : {4, 8, 16}) run_loop(waves, 192, 0, 256);
  for (int waves : {4, 8}) run_loop(waves, 64, 0, 1024);
  return 0;
}

// lds_direct_test.hip -- does global_load_lds_dwordx4 (gfx950) gather per lane and land lane-contiguous in LDS?
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_direct_test.hip -o tools/lds_direct_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void *gp;
typedef __attribute__((address_space(3))) void *lp;
__global__ void t16(const uint32_t *src, uint32_t *out, int stride) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[2][64 * 4];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((gp)(src + lane * stride + 1), (lp)buf[1], 16, 0, 0);  // 4-byte aligned source
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int k = 0; k < 4; ++k) out[lane * 4 + k] = buf[1][lane * 4 + k];
}
__global__ void t4(const uint32_t *src, uint32_t *out, int stride) {
  __shared__ uint32_t buf[64];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((gp)(src + lane * stride), (lp)buf, 4, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[lane] = buf[lane];
}
int main() {
  const int stride = 37, n = 64 * stride + 8;
  std::vector<uint32_t> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1000u + i;
  uint32_t *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> r(256);
  t16<<<1, 64>>>(d, o, stride);
  hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int k = 0; k < 4; ++k) bad += r[l * 4 + k] != 1000u + l * stride + 1 + k;
  printf("dwordx4: %s (%d mismatches) e.g. lane 5: %u %u %u %u\n", bad ? "DIFFERENT LAYOUT" : "ok: lane-contiguous 16 B", bad, r[20], r[21], r[22], r[23]);
  t4<<<1, 64>>>(d, o, stride);
  hipMemcpy(r.data(), o, 64 * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int l = 0; l < 64; ++l) bad += r[l] != 1000u + l * stride;
  printf("dword:   %s (%d mismatches)\n", bad ? "DIFFERENT LAYOUT" : "ok", bad);
  return 0;
}

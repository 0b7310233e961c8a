// tools/load_pattern_probe.hip -- how fast can a 4K 10-bit luma plane set (64 frames, 1 062 MB) be READ with k1_moments' lane layout and with
// a unit layout, arithmetic left out?  (hipcc --offload-arch=gfx950 -O3 -o tools/load_pattern_probe tools/load_pattern_probe.hip)
//   P0: lane = one row of a 32 x 32 block (64 bytes: four 16-byte loads), 8 blocks a 256-thread workgroup  (k1_moments)
//   P1: wave = 128 x 32 samples (4 blocks): lane = (row group rg = lane >> 4: rows 8 rg .. 8 rg + 7, word w = lane & 15): eight 16-byte loads,
//       an instruction = 4 rows x 256 contiguous bytes
//   P2: the same unit, lane = (rows (lane >> 4) + 4 k, word w): an instruction = 4 CONSECUTIVE rows x 256 bytes
//   P3: P1 with the non-temporal hint
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define GP __attribute__((address_space(1)))
constexpr int W = 3840, H = 2160, NBW = W / 32, NBH = (H + 31) / 32, STRIDE = W * 2;
__device__ __forceinline__ u4 ld(const uint8_t *p, bool nt) {
  const GP u4 *q = (const GP u4 *)(uintptr_t)p;
  return nt ? __builtin_nontemporal_load(q) : *q;
}
__global__ __launch_bounds__(256) void p0(const uint8_t *base, size_t frame_bytes, uint32_t *out) {
  const int frame = blockIdx.y, tid = threadIdx.x, yi = tid & 31, blk = blockIdx.x * 8 + (tid >> 5);
  if (blk >= NBW * NBH) return;
  const int bx = blk % NBW, by = blk / NBW;
  const int y = min(by * 32 + yi, H - 1);
  const uint8_t *p = base + frame * frame_bytes + (size_t)y * STRIDE + bx * 64;
  u4 a = ld(p, false), b = ld(p + 16, false), c = ld(p + 32, false), d = ld(p + 48, false);
  u4 x = a ^ b ^ c ^ d;
  if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345678u) out[0] = 1;
}
template <int MODE>
__global__ __launch_bounds__(256) void p1(const uint8_t *base, size_t frame_bytes, uint32_t *out) {
  const int frame = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int unit = blockIdx.x * 4 + wv;  // units of 4 blocks
  constexpr int GX = NBW / 4;
  if (unit >= GX * NBH) return;
  const int ux = unit % GX, by = unit / GX;
  const int w = lane & 15, rg = lane >> 4;
  const uint8_t *p = base + frame * frame_bytes + (size_t)(by * 32) * STRIDE + ux * 256 + w * 16;
  u4 x = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = MODE == 2 ? rg + 4 * r : 8 * rg + r;
    const int y = min(by * 32 + row, H - 1) - by * 32;
    x ^= ld(p + (size_t)y * STRIDE, MODE == 3);
  }
  if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345678u) out[0] = 1;
}
int main() {
  const int B = 64;
  const size_t fb = (size_t)STRIDE * H;
  uint8_t *d;
  uint32_t *o;
  hipMalloc(&d, fb * B);
  hipMalloc(&o, 4);
  hipMemset(d, 0x5a, fb * B);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch) {
    launch();
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 10; ++it) {
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
      sum += ms;
    }
    printf("%s: best %.1f us  mean %.1f us  -> %.2f TB/s (best)\n", name, best * 1e3, sum * 100, fb * B / (best * 1e-3) / 1e12);
  };
  time("P0 lane = block row (k1_moments)", [&] { hipLaunchKernelGGL(p0, dim3((NBW * NBH + 7) / 8, B), dim3(256), 0, 0, d, fb, o); });
  time("P1 unit, lane = 8 rows of a row group", [&] { hipLaunchKernelGGL(p1<1>, dim3((NBW / 4 * NBH + 3) / 4, B), dim3(256), 0, 0, d, fb, o); });
  time("P2 unit, instruction = 4 consecutive rows", [&] { hipLaunchKernelGGL(p1<2>, dim3((NBW / 4 * NBH + 3) / 4, B), dim3(256), 0, 0, d, fb, o); });
  time("P3 = P1 non-temporal", [&] { hipLaunchKernelGGL(p1<3>, dim3((NBW / 4 * NBH + 3) / 4, B), dim3(256), 0, 0, d, fb, o); });
  time("P0 again", [&] { hipLaunchKernelGGL(p0, dim3((NBW * NBH + 7) / 8, B), dim3(256), 0, 0, d, fb, o); });
  return 0;
}

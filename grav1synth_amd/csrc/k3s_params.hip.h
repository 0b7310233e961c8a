// k3s_params.hip.h -- what is left of round 2's fused accumulation pass (the k3f_fused kernel itself was removed in round 4): the launch
// parameters and the load / narrowing helpers k3s.hip.h (the stream chain, the wide chain's fallback) is built on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pixel_helpers.hip.h"
#include "k3m.hip.h"
#include "kernels.hip.h"

namespace g1s {

constexpr int kFWaves = 4, kFThreads = 64 * kFWaves;
#ifndef G1S_F_OCC
#define G1S_F_OCC 4  // waves per SIMD the kernel is compiled for (4 workgroups to a CU)
#endif

struct FParams {
  FrameTable ft;
  const uint32_t *units;  // [batch][nunits][kMUnitDwords]  (k3m_units)
  const uint32_t *unit_count;
  long long *partials;    // [batch][G][3][kMRec]
  int32_t *ustats;        // [batch][nunits][kMStatInts]  per-unit block statistics + deferral bits (k3m_finish)
  int nunits;
  uint8_t *lplane;        // [batch][lrows][lpitch]  L (sum of the co-located luma residuals) at chroma resolution, int8
  uint32_t lpitch, lframe_bytes;
  int frames, wgs;        // the launch: frames x workgroups per frame, as a 1-D grid (see the kernel)
  int wg_cap;             // workgroups per frame the partial systems are laid out for (>= wgs: the luma and the chroma launch may differ)
  int deal;               // units to workgroups: 0 round-robin, 1 contiguous runs
  int reuse;              // 1: (luma launch) the left halo word of a unit whose left neighbour was the unit before it in the run is not read
  int dbg;                  // timing experiments (G1S_S_DBG, k3s.hip.h): bit 0 no global loads, 1 no residual arithmetic, 2 no statistics atomics, 3 no copy writes, 4 no multiplies, 5 no statistics / L stores, 6 no barrier in the loop; wrong results
  long long *phase_cycles;  // profiling aid (built with -DG1S_F_PHASES, run with G1S_F_PHASES=1): [workgroup][wave][8] cycles: tile copies, barrier 2, multiply, barrier 1, wait for the words, residuals, requests, stores; or null
};

// BPS: bytes per sample known at compile time (1, 2), or 0: given at run time (mixed depths)
template <int BPS>
__device__ __forceinline__ int f_bps(int runtime_bps) { return BPS ? BPS : runtime_bps; }

// a raw 8-sample word -> packed 16-bit pairs 0x00vv00vv of the narrowed samples
template <int BPS>
__device__ __forceinline__ void f_narrow(const u32x4 &v, int rbps, int shift, uint32_t (&h)[4]) {
  if (f_bps<BPS>(rbps) == 2) {
    const u16x2 sh = {(unsigned short)shift, (unsigned short)shift};
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, w[k]) >> sh) & 0x00ff00ffu;
  } else {
    h[0] = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u);
    h[1] = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u);
    h[2] = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u);
    h[3] = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u);
  }
}
// the raw word at base + off: one 16-byte (8-byte) load; `ok` false reads as zero
template <int BPS>
__device__ __forceinline__ u32x4 f_load(const uint8_t *base, uint32_t off, int rbps, bool ok) {
  u32x4 r = {0u, 0u, 0u, 0u};
  // (opaque to the optimiser: it would otherwise keep a 64-bit copy of the offset across the unit loop and lose the
  //  scalar-base + 32-bit-offset form of the load)
  asm volatile("" : "+v"(off));
  if (ok) {
    gptr_u8 p = as_global(base) + off;
    if (f_bps<BPS>(rbps) == 2) {
      r = *(gptr_u4)p;
    } else {
      const u32x2 t = *(gptr_u2)p;
      r.x = t.x;
      r.y = t.y;
    }
  }
  return r;
}
// the same word sample by sample: words that straddle the right plane edge, planes whose rows are not 16-byte
// aligned (samples outside the plane read as zero; the result has the layout of the vector load)
__device__ __forceinline__ u32x4 f_load_slow(const uint8_t *plane, uint32_t stride, int bps, int X0, int Y, int pw, int ph) {
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  if (Y >= 0 && Y < ph) {
    gptr_u8 row = as_global(plane) + (size_t)Y * stride;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int X = X0 + k;
      if (X >= 0 && X < pw) {
        if (bps == 2) w[k >> 1] |= (uint32_t)((gptr_u16)row)[X] << (16 * (k & 1));
        else w[k >> 2] |= (uint32_t)row[X] << (8 * (k & 3));
      }
    }
  }
  return u32x4{w[0], w[1], w[2], w[3]};
}

constexpr int kFBiasY = 16 * 255, kFBiasC = 8 * 255;  // bias of a lane's sum of residuals: two rows / one row of 8 samples

// blocks whose tile holds word wd of a row (WB words to a block; the tile reaches one word into its neighbours)
__device__ __forceinline__ void f_flag_blocks(int *flags, int wd, int WB) {
  const int b = wd / WB;
  if (b < kMUnitBlocks) flags[b] = 1;
  if (wd - b * WB <= 1 && b >= 1) flags[b - 1] = 1;
}

}  // namespace g1s

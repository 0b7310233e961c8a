// estimate.hip -- N4: the single-source noise estimator of `grav1synth estimate` (feature "unstable").
//
// The command (/root/reference/src/main.rs:534-608) calls av1_grain::estimate_plane_noise(&frame.y_plane, bit_depth) on
// every frame and writes "filmgrn1" and one "{:.3}" line per frame (-1 for None).  The estimator is av1-grain's port of
// libaom's av1_estimate_noise_from_single_plane: over the interior pixels of the luma plane, a Sobel gradient decides
// whether the pixel is smooth (|Gx| + |Gy|, rounded down to 8-bit scale, below 50); the smooth pixels' |Laplacian|
// (rounded to 8-bit scale) is averaged: sigma = accum / (6 count) * sqrt(pi / 2), None when fewer than 16 pixels are
// smooth.  One pass over one plane, a 3x3 stencil and two integer sums: HBM bound (2 bytes per luma pixel at 10 bit).
//
// Kernel: a wave owns 62 x 8 output columns of a strip of rows and walks the strip top to bottom with the three rows
// of the stencil in registers (each row is loaded once per strip, as one 8-sample word per lane: 16-byte loads at
// 10 bit); lanes 0 and 63 only carry halo columns, the horizontal neighbours of a lane's word come from the adjacent
// lanes.  Exact integers: the two sums of a frame are the same for any strip / wave / batch partition; the host turns
// them into the f64 with the reference's three operations.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/g1s_diff.h"

namespace {

constexpr int kEdgeThreshold = 50;  // EDGE_THRESHOLD
constexpr int kStripRows = 32;      // output rows per wave strip (+ 2 halo rows)
constexpr int kWavesPerWg = 4;
constexpr int kColsPerWave = 62 * 8;

struct EstFrame {
  const uint8_t *y;
  uint32_t stride;  // bytes
};

struct EstParams {
  const EstFrame *frames;
  unsigned long long *sums;  // [frames][2]: accum, count
  int W, H, bps, shift;
  int col_strips, row_strips, strip_rows;
};

// the 8 samples of a word, widened; outside the plane: zeros (never used by an output pixel that counts)
template <int BPS>
__device__ __forceinline__ void est_load(const uint8_t *row, int x0, int W, bool row_ok, int (&p)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = 0;
  if (!row_ok || x0 >= W || x0 + 8 <= 0) return;
  if (x0 >= 0 && x0 + 8 <= W) {
    if (BPS == 2) {
      // (rows of a frame are at least 2-byte aligned; a word is read with the widest loads its address allows)
      const uint8_t *a = row + (size_t)x0 * 2;
      uint32_t w[4];
      if (((uintptr_t)a & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(a);
        w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
      } else if (((uintptr_t)a & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = reinterpret_cast<const uint32_t *>(a)[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (uint32_t)reinterpret_cast<const uint16_t *>(a)[2 * k] | ((uint32_t)reinterpret_cast<const uint16_t *>(a)[2 * k + 1] << 16);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        p[2 * k] = (int)(w[k] & 0xffffu);
        p[2 * k + 1] = (int)(w[k] >> 16);
      }
    } else {
      const uint8_t *a = row + x0;
      uint32_t w[2];
      if (((uintptr_t)a & 7) == 0) {
        const uint2 v = *reinterpret_cast<const uint2 *>(a);
        w[0] = v.x, w[1] = v.y;
      } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) w[k] = (uint32_t)a[4 * k] | ((uint32_t)a[4 * k + 1] << 8) | ((uint32_t)a[4 * k + 2] << 16) | ((uint32_t)a[4 * k + 3] << 24);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // a word over the right (or left) edge of the plane
    const int x = x0 + k;
    if (x >= 0 && x < W) p[k] = BPS == 2 ? (int)reinterpret_cast<const uint16_t *>(row)[x] : (int)row[x];
  }
}

template <int BPS>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_estimate(EstParams ep) {
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x * kWavesPerWg + wave;
  if (strip >= ep.col_strips * ep.row_strips) return;
  const int cs = strip % ep.col_strips, rs = strip / ep.col_strips;
  const EstFrame fr = ep.frames[frame];
  const int W = ep.W, H = ep.H, shift = ep.shift, half = shift ? 1 << (shift - 1) : 0;
  // output columns of the wave: [cs * 496, cs * 496 + 496) (less the plane's first and last column); lane l holds the
  // word at x0 = cs * 496 - 8 + 8 l -- a multiple of 8 samples: aligned 16-byte loads -- lanes 0 and 63 the halo words
  const int x0 = cs * kColsPerWave - 8 + 8 * lane;
  const int y_first = 1 + rs * kStripRows, y_last = min(y_first + kStripRows, H - 1);  // output rows [y_first, y_last)
  int a[8], b[8], c[8];  // rows y - 1, y, y + 1
  est_load<BPS>(fr.y + (size_t)(y_first - 1) * fr.stride, x0, W, true, a);
  est_load<BPS>(fr.y + (size_t)y_first * fr.stride, x0, W, y_first < H, b);
  uint32_t accum = 0, count = 0;
  const bool out_lane = lane >= 1 && lane <= 62;
  for (int y = y_first; y < y_last; ++y) {
    est_load<BPS>(fr.y + (size_t)(y + 1) * fr.stride, x0, W, y + 1 < H, c);
    // the column left of the word (the neighbour lane's last sample) and right of it (its first), for the three rows
    const int al = __shfl_up(a[7], 1), bl = __shfl_up(b[7], 1), cl = __shfl_up(c[7], 1);
    const int ar = __shfl_down(a[0], 1), br = __shfl_down(b[0], 1), cr = __shfl_down(c[0], 1);
    if (out_lane) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int x = x0 + k;
        const int m00 = k ? a[k - 1] : al, m01 = a[k], m02 = k < 7 ? a[k + 1] : ar;
        const int m10 = k ? b[k - 1] : bl, m11 = b[k], m12 = k < 7 ? b[k + 1] : br;
        const int m20 = k ? c[k - 1] : cl, m21 = c[k], m22 = k < 7 ? c[k + 1] : cr;
        const int gx = (m00 - m02) + (m20 - m22) + 2 * (m10 - m12);
        const int gy = (m00 - m20) + (m02 - m22) + 2 * (m01 - m21);
        const int ga = (abs(gx) + abs(gy) + half) >> shift;
        const int v = 4 * m11 - 2 * (m01 + m21 + m10 + m12) + (m00 + m02 + m20 + m22);
        const bool on = x >= 1 && x < W - 1 && ga < kEdgeThreshold;
        accum += on ? (uint32_t)((abs(v) + half) >> shift) : 0u;
        count += on ? 1u : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] = b[k];
      b[k] = c[k];
    }
  }
  // wave sums -> one pair of atomics per wave
  for (int o = 32; o; o >>= 1) {
    accum += __shfl_down(accum, o);
    count += __shfl_down(count, o);
  }
  if (lane == 0 && (accum | count)) {
    atomicAdd(&ep.sums[2 * frame], (unsigned long long)accum);
    atomicAdd(&ep.sums[2 * frame + 1], (unsigned long long)count);
  }
}

// ---------------------------------------------------------------------------------
// k_estimate_pk: the same sums in packed 16-bit arithmetic, for depths 8 .. 12 (every intermediate fits 16 bits:
// |Gx|, |Gy| <= 4 * 4095, their sum + rounding <= 32768 as u16, |Laplacian| <= 8 * 4095).  Only the simplest
// 32-bit VALU operations issue at two cycles a wave on gfx950 (v_sub, v_and, shifts); everything else -- v_max,
// three-operand forms, every v_pk_* -- takes four (tools/valu_rate.hip), so a packed operation is two pixels
// for the price of one.  Both stencils are separable and the horizontal parts are carried down the strip:
//   per input row    H = l - r,  S = l + 2 c + r,  T = l - 2 c + r        (l, r: the row shifted by one sample)
//   per output row   Gx = H0 + 2 H1 + H2,  Gy = S0 - S2,  Laplacian = T0 - 2 T1 + T2
// 27 packed operations per pixel pair and row instead of ~34 scalar ones per pixel.  The smooth-pixel mask is
// a 0/1 per half (sign bit of ga - 50), the masked sum is one v_dot2_u32_u16.  Exact integers: same sums as
// k_estimate (compared on every test case).  A lane holds one 8-sample word (4 dwords) of the row; the sample
// left / right of the word comes from the neighbouring lane (DPP wave shifts).
// ---------------------------------------------------------------------------------
// (global address space: a pointer read from memory is `flat` to the compiler otherwise, and flat loads wait on the LDS counter too)
#define EST_GLOBAL __attribute__((address_space(1)))
typedef uint32_t est_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t est_u32x4 __attribute__((ext_vector_type(4)));
typedef const EST_GLOBAL uint8_t *est_gptr;
__device__ __forceinline__ est_gptr est_global(const uint8_t *p) { return (est_gptr)(uintptr_t)p; }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ i16x2 as_i(uint32_t v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ uint32_t bits(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t bits(i16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_abs(uint32_t v) {
  const i16x2 a = as_i(v), z = {0, 0};
  return bits(__builtin_elementwise_max(a, (i16x2)(z - a)));
}

// a * k + c per 16-bit half, k an inline constant (one instruction where the compiler writes a shift and an add)
__device__ __forceinline__ uint32_t pk_mad2(uint32_t a, uint32_t c) {
  uint32_t r;
  asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c));
  return r;
}
__device__ __forceinline__ uint32_t pk_madm2(uint32_t a, uint32_t c) {
  uint32_t r;
  asm("v_pk_mad_i16 %0, %1, -2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c));
  return r;
}

struct EstRow {
  uint32_t H[4], S[4], T[4];
};

// the lane's word of row y as 4 dwords of packed 16-bit samples (zeros outside the plane)
template <int BPS>
__device__ __forceinline__ void est_load_pk(const EstFrame &fr, int y, int x0, int W, int H, bool fast, uint32_t (&w)[4]) {
  w[0] = w[1] = w[2] = w[3] = 0u;
  if (y < 0 || y >= H || x0 >= W || x0 + 8 <= 0) return;
  est_gptr row = est_global(fr.y) + (size_t)y * fr.stride;
  if (fast && x0 >= 0 && x0 + 8 <= W) {
    if (BPS == 2) {
      const est_u32x4 v = *(const EST_GLOBAL est_u32x4 *)(row + (size_t)x0 * 2);
      w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    } else {
      const est_u32x2 v = *(const EST_GLOBAL est_u32x2 *)(row + x0);
      w[0] = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u);
      w[1] = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u);
      w[2] = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u);
      w[3] = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // unaligned planes, words over an edge of the plane
    const int x = x0 + k;
    if (x >= 0 && x < W) {
      const uint32_t v = BPS == 2 ? (uint32_t)((const EST_GLOBAL uint16_t *)row)[x] : (uint32_t)row[x];
      w[k >> 1] |= v << (16 * (k & 1));
    }
  }
}

__device__ __forceinline__ void est_derive(const uint32_t (&c)[4], EstRow &r) {
  const uint32_t prev = (uint32_t)__builtin_amdgcn_mov_dpp((int)c[3], 0x138, 0xf, 0xf, true);  // wave_shr:1: lane - 1's last dword
  const uint32_t next = (uint32_t)__builtin_amdgcn_mov_dpp((int)c[0], 0x130, 0xf, 0xf, true);  // wave_shl:1: lane + 1's first
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t P = q ? c[q - 1] : prev, N = q < 3 ? c[q + 1] : next;
    const uint32_t l = __builtin_amdgcn_alignbyte(c[q], P, 2), rr = __builtin_amdgcn_alignbyte(N, c[q], 2);
    const uint32_t lr = bits((u16x2)(as_u(l) + as_u(rr)));
    r.H[q] = bits((u16x2)(as_u(l) - as_u(rr)));
    r.S[q] = pk_mad2(c[q], lr);
    r.T[q] = pk_madm2(c[q], lr);
  }
}

// Output rows per wave strip: chosen by the host so that the launch is a whole number of resident rounds (8 waves a SIMD:
// 8192 waves on the chip): with 64-row strips a 32-frame 4K launch was 8704 waves -- 1.06 rounds, the last 6 % of the waves
// alone on the chip for as long as the first 94 % took.
constexpr int kPkMaxStripRows = 256;  // (the two 16-bit pixel counters of a lane hold 4 * rows each)

// the fast form: the lane's word lies inside the plane's rows or wholly outside (then any in-plane word will do: every
// pixel it could reach is masked), the rows are 16-byte (8-byte) aligned: one unconditional vector load, nothing to wait for
// at the load -- the request for row y + 2 is in flight while row y is worked on
template <int BPS>
__device__ __forceinline__ void est_load_fast(est_gptr base, uint32_t stride, int y, uint32_t xoff, uint32_t (&w)[4]) {
  est_gptr p = base + (size_t)y * stride + xoff;
  if (BPS == 2) {
    const est_u32x4 v = *(const EST_GLOBAL est_u32x4 *)p;
    w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
  } else {
    const est_u32x2 v = *(const EST_GLOBAL est_u32x2 *)p;
    w[0] = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u);
    w[1] = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u);
    w[2] = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u);
    w[3] = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u);
  }
}

template <int BPS, bool FAST>
__device__ __forceinline__ void est_strip(const EstFrame &fr, int W, int H, int shift, int x0, int y_first, int y_last, int lane,
                                          uint32_t &accum, uint32_t &count) {
  const uint32_t halfpk = shift ? 0x00010001u << (shift - 1) : 0u;
  // smooth: (|Gx| + |Gy| + half) >> shift < 50  <=>  |Gx| + |Gy| < (50 << shift) - half
  const uint32_t thrpk = 0x00010001u * (uint32_t)((kEdgeThreshold << shift) - (shift ? 1 << (shift - 1) : 0));
  const u16x2 sh = {(unsigned short)shift, (unsigned short)shift};
  // output columns of this lane: 1 <= x < W - 1, lanes 1 .. 62 (lanes 0 and 63 carry the halo columns)
  uint32_t cm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = x0 + 2 * q;
    const bool ok = lane >= 1 && lane <= 62;
    cm[q] = (ok && x >= 1 && x < W - 1 ? 1u : 0u) | (ok && x + 1 >= 1 && x + 1 < W - 1 ? 0x10000u : 0u);
  }
  // (rows are wave-uniform: the row address is scalar arithmetic, the lane adds its column offset)
  const est_gptr base = est_global(fr.y);
  const uint32_t xoff = (uint32_t)(min(max(x0, 0), max(W - 8, 0)) * BPS);  // (FAST: W is a multiple of 8, >= 8)
  auto load = [&](int y, uint32_t (&w)[4]) __attribute__((always_inline)) {
    if (FAST) est_load_fast<BPS>(base, fr.stride, min(y, H - 1), xoff, w);
    else est_load_pk<BPS>(fr, y, x0, W, H, false, w);
  };
  uint32_t cnt = 0;
  const u16x2 fifteen = {15, 15};
  auto output = [&](const EstRow &a, const EstRow &b, const EstRow &c) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t gx = pk_mad2(b.H[q], bits((i16x2)(as_i(a.H[q]) + as_i(c.H[q]))));
      const uint32_t gy = bits((i16x2)(as_i(a.S[q]) - as_i(c.S[q])));
      const u16x2 g = as_u(pk_abs(gx)) + as_u(pk_abs(gy));  // <= 32760
      const uint32_t on = bits((u16x2)((u16x2)(g - as_u(thrpk)) >> fifteen)) & cm[q];
      const uint32_t v = pk_madm2(b.T[q], bits((i16x2)(as_i(a.T[q]) + as_i(c.T[q]))));
      const u16x2 lv = (u16x2)(as_u(pk_abs(v)) + as_u(halfpk)) >> sh;
      accum = __builtin_amdgcn_udot2(lv, as_u(on), accum, false);
      cnt += on;  // (two 16-bit counters: at most 4 * kPkMaxStripRows each)
    }
  };
  uint32_t w0[4], w1[4], w2[4];
  EstRow r0, r1, r2;
  load(y_first - 1, w0);
  est_derive(w0, r0);
  load(y_first, w0);
  est_derive(w0, r1);
  load(y_first + 1, w0);
  load(y_first + 2, w1);
  // three output rows a turn: the rows rotate through the names, not through registers; loads two rows ahead of their use
  // (the rows past the strip's last are requested and dropped)
  int y = y_first;
  for (; y + 3 <= y_last; y += 3) {
    load(y + 3, w2);
    est_derive(w0, r2);
    output(r0, r1, r2);
    load(y + 4, w0);
    est_derive(w1, r0);
    output(r1, r2, r0);
    load(y + 5, w1);
    est_derive(w2, r1);
    output(r2, r0, r1);
  }
  if (y < y_last) {
    est_derive(w0, r2);
    output(r0, r1, r2);
    if (y + 1 < y_last) {
      est_derive(w1, r0);
      output(r1, r2, r0);
    }
  }
  count = (cnt & 0xffffu) + (cnt >> 16);
}

template <int BPS>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_estimate_pk(EstParams ep) {
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int strip = blockIdx.x * kWavesPerWg + wave;
  if (strip >= ep.col_strips * ep.row_strips) return;
  // (column strips are the fast index: the waves of a workgroup read adjacent kilobytes of the same rows)
  const int cs = strip % ep.col_strips, rs = strip / ep.col_strips;
  const EstFrame fr = ep.frames[frame];
  const int W = ep.W, H = ep.H;
  const bool fast = (((uintptr_t)fr.y | fr.stride) & (BPS == 2 ? 15 : 7)) == 0 && (W & 7) == 0;  // (uniform)
  const int x0 = cs * kColsPerWave - 8 + 8 * lane;
  const int y_first = 1 + rs * ep.strip_rows, y_last = min(y_first + ep.strip_rows, H - 1);  // output rows [y_first, y_last)
  uint32_t accum = 0, count = 0;
  if (fast) est_strip<BPS, true>(fr, W, H, ep.shift, x0, y_first, y_last, lane, accum, count);
  else est_strip<BPS, false>(fr, W, H, ep.shift, x0, y_first, y_last, lane, accum, count);
  for (int o = 32; o; o >>= 1) {
    accum += __shfl_down(accum, o);
    count += __shfl_down(count, o);
  }
  if (lane == 0 && (accum | count)) {
    atomicAdd(&ep.sums[2 * frame], (unsigned long long)accum);
    atomicAdd(&ep.sums[2 * frame + 1], (unsigned long long)count);
  }
}

constexpr double kSqrtPiBy2 = 1.2533141373155003;  // SQRT_PI_BY_2

}  // namespace

struct g1s_estimate {
  int device = 0;
  uint32_t bit_depth = 8;
  uint32_t W = 0, H = 0, bps = 0;
  uint32_t batch = 32;
  hipStream_t stream = nullptr;
  std::vector<EstFrame> h_frames;  // the batch being filled
  EstFrame *d_frames = nullptr;
  unsigned long long *d_sums = nullptr, *h_sums = nullptr;
  uint8_t *d_stage = nullptr;      // device copies of host frames
  size_t stage_frame = 0;
  std::vector<double> estimates;   // one per frame: sigma, or -1 (None)
  std::string err;
  uint64_t frames_kernel = 0;
  double ms_kernel = 0;
  bool timing = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  int fail(int code, const std::string &m) {
    err = m;
    return code;
  }
  int flush();
};

#define EST_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(G1S_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_)); \
  } while (0)

int g1s_estimate::flush() {
  const uint32_t B = (uint32_t)h_frames.size();
  if (!B) return G1S_OK;
  EST_TRY(hipMemcpyAsync(d_frames, h_frames.data(), sizeof(EstFrame) * B, hipMemcpyHostToDevice, stream));
  EST_TRY(hipMemsetAsync(d_sums, 0, sizeof(unsigned long long) * 2 * B, stream));
  EstParams ep;
  ep.frames = d_frames;
  ep.sums = d_sums;
  ep.W = (int)W;
  ep.H = (int)H;
  ep.bps = (int)bps;
  ep.shift = (int)bit_depth - 8;
  // depths up to 12 bits: the packed 16-bit kernel; deeper samples (or G1S_ESTIMATE=wide, a test aid): 32-bit arithmetic
  const char *mode = getenv("G1S_ESTIMATE");
  const bool packed = bit_depth <= 12 && !(mode && std::strcmp(mode, "wide") == 0);
  ep.col_strips = ((int)W - 1 + kColsPerWave - 1) / kColsPerWave;
  int strip_rows = kStripRows;
  if (packed && H >= 3) {
    // whole resident rounds: k rounds of `resident` waves, the smallest k whose strips are at most kPkMaxStripRows rows
    int cus = 256, wgs_per_cu = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (bps == 2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs_per_cu, k_estimate_pk<2>, 64 * kWavesPerWg, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs_per_cu, k_estimate_pk<1>, 64 * kWavesPerWg, 0);
    if (wgs_per_cu < 1) wgs_per_cu = 5;
    const long resident = (long)cus * wgs_per_cu * kWavesPerWg, cols = (long)ep.col_strips * B, rows = (long)H - 2;
    for (long k = 1;; ++k) {
      const long rs = std::max(1L, k * resident / cols);  // row strips a frame
      strip_rows = (int)((rows + rs - 1) / rs);
      if (strip_rows <= kPkMaxStripRows) break;
    }
    strip_rows = std::max(strip_rows, 8);
  }
  ep.strip_rows = strip_rows;
  ep.row_strips = ((int)H - 2 + strip_rows - 1) / strip_rows;
  if (W >= 3 && H >= 3) {
    const dim3 grid((ep.col_strips * ep.row_strips + kWavesPerWg - 1) / kWavesPerWg, B);
    if (timing) EST_TRY(hipEventRecord(ev0, stream));
    if (packed && bps == 2) hipLaunchKernelGGL(k_estimate_pk<2>, grid, dim3(64 * kWavesPerWg), 0, stream, ep);
    else if (packed) hipLaunchKernelGGL(k_estimate_pk<1>, grid, dim3(64 * kWavesPerWg), 0, stream, ep);
    else if (bps == 2) hipLaunchKernelGGL(k_estimate<2>, grid, dim3(64 * kWavesPerWg), 0, stream, ep);
    else hipLaunchKernelGGL(k_estimate<1>, grid, dim3(64 * kWavesPerWg), 0, stream, ep);
    if (timing) EST_TRY(hipEventRecord(ev1, stream));
  }
  EST_TRY(hipMemcpyAsync(h_sums, d_sums, sizeof(unsigned long long) * 2 * B, hipMemcpyDeviceToHost, stream));
  EST_TRY(hipStreamSynchronize(stream));
  EST_TRY(hipGetLastError());
  if (timing && W >= 3 && H >= 3) {
    float ms = 0;
    EST_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    ms_kernel += ms;
    frames_kernel += B;
  }
  for (uint32_t i = 0; i < B; ++i) {
    const unsigned long long accum = h_sums[2 * i], count = h_sums[2 * i + 1];
    // (count < 16) ? None : accum as f64 / (6 * count) as f64 * SQRT_PI_BY_2
    estimates.push_back(count < 16 ? -1.0 : (double)accum / (double)(6 * count) * kSqrtPiBy2);
  }
  h_frames.clear();
  return G1S_OK;
}

extern "C" {

g1s_estimate_t *g1s_estimate_new(uint32_t bit_depth, int32_t device, uint32_t batch_frames) {
  if (bit_depth < 8 || bit_depth > 16) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;  // no CPU fallback
  g1s_estimate *e = new g1s_estimate;
  if (device < 0) (void)hipGetDevice(&device);
  e->device = device;
  e->bit_depth = bit_depth;
  e->bps = bit_depth > 8 ? 2 : 1;
  e->batch = batch_frames ? std::min(batch_frames, 256u) : 32u;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void **)&e->d_frames, sizeof(EstFrame) * e->batch) != hipSuccess ||
      hipMalloc((void **)&e->d_sums, sizeof(unsigned long long) * 2 * e->batch) != hipSuccess ||
      hipHostMalloc((void **)&e->h_sums, sizeof(unsigned long long) * 2 * e->batch, hipHostMallocDefault) != hipSuccess ||
      hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
    g1s_estimate_free(e);
    return nullptr;
  }
  return e;
}

int g1s_estimate_frame(g1s_estimate_t *e, const g1s_frame_t *f) {
  if (!e || !f || !f->data[0]) return G1S_ERR_INVALID;
  (void)hipSetDevice(e->device);
  if ((f->bytes_per_sample == 1) != (e->bit_depth == 8) || (f->bytes_per_sample != 1 && f->bytes_per_sample != 2))
    return e->fail(G1S_ERR_INVALID, "bytes_per_sample does not match the bit depth given to g1s_estimate_new");
  if (f->width < 1 || f->height < 1) return e->fail(G1S_ERR_INVALID, "empty frame");
  if (!e->W) {
    e->W = f->width;
    e->H = f->height;
    e->stage_frame = (((size_t)e->W * e->bps + 15) & ~size_t(15)) * e->H;
  } else if (e->W != f->width || e->H != f->height) {
    return e->fail(G1S_ERR_DIM_MISMATCH, "frame geometry changed mid-stream");
  }
  EstFrame ef;
  if (f->on_device == 1) {
    ef.y = static_cast<const uint8_t *>(f->data[0]);
    ef.stride = (uint32_t)f->stride_bytes[0];
  } else {  // host planes: copied before the call returns (the `&frame.y_plane` borrow)
    if (!e->d_stage && hipMalloc((void **)&e->d_stage, e->stage_frame * e->batch) != hipSuccess)
      return e->fail(G1S_ERR_HIP, "hipMalloc of the staging buffer failed");
    const size_t row = ((size_t)e->W * e->bps + 15) & ~size_t(15);
    uint8_t *dst = e->d_stage + e->stage_frame * e->h_frames.size();
    if (hipMemcpy2D(dst, row, f->data[0], f->stride_bytes[0], (size_t)e->W * e->bps, e->H, hipMemcpyHostToDevice) != hipSuccess)
      return e->fail(G1S_ERR_HIP, "hipMemcpy2D of a host frame failed");
    ef.y = dst;
    ef.stride = (uint32_t)row;
  }
  e->h_frames.push_back(ef);
  return e->h_frames.size() == e->batch ? e->flush() : G1S_OK;
}

int g1s_estimate_finish(g1s_estimate_t *e, double *out, size_t cap, size_t *n_out) {
  if (!e) return G1S_ERR_INVALID;
  (void)hipSetDevice(e->device);
  const int rc = e->flush();
  if (rc) return rc;
  if (n_out) *n_out = e->estimates.size();
  if (e->estimates.size() > cap || (!out && !e->estimates.empty())) return e->fail(G1S_ERR_CAPACITY, "estimate buffer too small");
  if (!e->estimates.empty()) std::memcpy(out, e->estimates.data(), sizeof(double) * e->estimates.size());
  return G1S_OK;
}

int g1s_estimate_set_timing(g1s_estimate_t *e, int enable, double *ms_kernel, uint64_t *frames) {
  if (!e) return G1S_ERR_INVALID;
  e->timing = enable != 0;
  if (ms_kernel) *ms_kernel = e->ms_kernel;
  if (frames) *frames = e->frames_kernel;
  return G1S_OK;
}

const char *g1s_estimate_last_error(const g1s_estimate_t *e) { return e ? e->err.c_str() : ""; }

void g1s_estimate_free(g1s_estimate_t *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->d_frames) (void)hipFree(e->d_frames);
  if (e->d_sums) (void)hipFree(e->d_sums);
  if (e->h_sums) (void)hipHostFree(e->h_sums);
  if (e->d_stage) (void)hipFree(e->d_stage);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

long g1s_format_estimates(const double *estimates, size_t n, char *buf, size_t cap) {
  // writeln!("filmgrn1"), then writeln!("{:.3}", estimate.unwrap_or(-1f64)) per frame (src/main.rs:597-600)
  std::string s = "filmgrn1\n";
  char line[64];
  for (size_t i = 0; i < n; ++i) {
    snprintf(line, sizeof(line), "%.3f\n", estimates[i]);
    s += line;
  }
  if (s.size() > cap) return G1S_ERR_CAPACITY;
  std::memcpy(buf, s.data(), s.size());
  return (long)s.size();
}

}  // extern "C"

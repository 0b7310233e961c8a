// pixel_helpers.hip.h -- shared helpers of the pixel kernels: packed 16-bit arithmetic, the block-window rule of
// add_block_observations, the flat-block finder's row moments (k1f.hip.h), the per-batch zero fill.
// (The streaming pixel pass K0 of rounds 1 and 2 -- int8 residual / L / window-bit planes in front of the accumulation -- and
// its window kernel were removed in round 4; the name of the file stayed.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"

namespace g1s {

constexpr int kQLag = 3;
constexpr int kPadX = 8, kPadY = 3;

// full-wave integer sum, all in the VALU (DPP): quad swaps, half-row / row mirrors, then
// the row broadcasts; the total lands in lane 63 and is read back as a scalar.
__device__ __forceinline__ int wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------------------------
// window of a block (libaom add_block_observations), in samples of its plane
// ---------------------------------------------------------------------------------
struct Win {
  int flat, xs, xe, ys, ye;
};
// (lag = the AR lag of the generator: the border a window keeps from a non-flat neighbour and from the right
//  plane edge; the lag-structured kernels run lag 1 and 2 with lag-3 tiles and these narrower borders)
__device__ __forceinline__ Win block_window(const uint8_t *mask, int nbw, int nbh, int bx, int by, int bw, int bh,
                                            int pw, int ph, int lag) {
  Win w{0, 0, 0, 0, 0};
  if (bx < 0 || by < 0 || bx >= nbw || by >= nbh) return w;
  if (!mask[by * nbw + bx]) return w;
  w.flat = 1;
  w.ys = (by > 0 && mask[(by - 1) * nbw + bx]) ? 0 : lag;
  w.xs = (bx > 0 && mask[by * nbw + bx - 1]) ? 0 : lag;
  w.ye = min(ph - by * bh, bh);
  w.xe = min(pw - bx * bw - lag, (bx + 1 < nbw && mask[by * nbw + bx + 1]) ? bw : (bw - lag));
  if (w.xe <= w.xs || w.ye <= w.ys) w.flat = 0;  // empty window
  return w;
}
__device__ __forceinline__ int window_at(const uint8_t *mask, int nbw, int nbh, int bw, int bh, int pw, int ph, int X,
                                         int Y, int lag) {
  if (X < 0 || Y < 0 || X >= pw || Y >= ph) return 0;
  const int bx = X / bw, by = Y / bh;
  const Win w = block_window(mask, nbw, nbh, bx, by, bw, bh, pw, ph, lag);
  const int lx = X - bx * bw, ly = Y - by * bh;
  return w.flat && lx >= w.xs && lx < w.xe && ly >= w.ys && ly < w.ye;
}
// bytes lx0 .. lx0+N-1 of a window row as 0xFF / 0x00 (N = 4 or 8, little endian)
__device__ __forceinline__ unsigned long long window_bytes(const Win &w, int lx0, int ly, int nbytes) {
  if (!w.flat || ly < w.ys || ly >= w.ye) return 0ull;
  const int lo = min(max(w.xs - lx0, 0), nbytes), hi = min(max(w.xe - lx0, 0), nbytes);
  if (hi <= lo) return 0ull;
  const unsigned long long upto_hi = hi >= 8 ? ~0ull : ((1ull << (8 * hi)) - 1ull);
  const unsigned long long upto_lo = lo >= 8 ? ~0ull : ((1ull << (8 * lo)) - 1ull);
  return upto_hi & ~upto_lo;
}

// bits lx0 .. lx0+7 of a window row (bit k = sample lx0 + k)
__device__ __forceinline__ uint32_t window_bits8(const Win &w, int lx0, int ly) {
  if (!w.flat || ly < w.ys || ly >= w.ye) return 0u;
  const int lo = min(max(w.xs - lx0, 0), 8), hi = min(max(w.xe - lx0, 0), 8);
  if (hi <= lo) return 0u;
  return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
}
// 4 window bits -> 4 bytes 0xFF / 0x00 (bit k -> byte k)
__device__ __forceinline__ uint32_t expand_bits4(uint32_t n) {
  const uint32_t m = (n * 0x00204081u) & 0x01010101u;  // n <= 15: a 24-bit multiply
  // bytes 0 / 1 -> 0x00 / 0xFF without a multiply by 255: 0x80 - {0,1} = {0x80,0x7F} never borrows across bytes
  return (0x80808080u - m) ^ 0x80808080u;
}

// ---- packed 16-bit arithmetic (two samples per dword) ------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// a.lo * b.lo + a.hi * b.hi + c on packed i16
__device__ __forceinline__ int pk_dot(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
__device__ __forceinline__ int pk_lo(uint32_t a) { return (int)(short)(a & 0xffffu); }
__device__ __forceinline__ int pk_hi(uint32_t a) { return (int)a >> 16; }
// running min / max of packed i16 -> does some value not fit int8 (symmetric: |v| > 127) ?
__device__ __forceinline__ bool range_bad(uint32_t mx, uint32_t mn) {
  return max(pk_lo(mx), pk_hi(mx)) > 127 || min(pk_lo(mn), pk_hi(mn)) < -127;
}
// two packed-i16 dwords (4 values) -> 4 bytes (low byte of each)
__device__ __forceinline__ uint32_t pk_bytes(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }

// N (4 or 8) consecutive samples of a row, narrowed to 8 bits, two per dword.  Samples
// outside the plane read as 0.
template <int BPS, int N>
__device__ __forceinline__ void load_narrow(const uint8_t *base, uint32_t stride, int shift, bool vec_ok, int X0, int Y,
                                            int pw, int ph, uint32_t (&h)[N / 2]) {
#pragma unroll
  for (int k = 0; k < N / 2; ++k) h[k] = 0;
  if (Y >= ph || X0 >= pw) return;
  if (vec_ok && X0 + N <= pw) {
    gptr_u8 p = as_global(base) + (size_t)Y * stride + (size_t)X0 * BPS;
    if (BPS == 2) {
      const u16x2 sh = {(unsigned short)shift, (unsigned short)shift};
      uint32_t w[N / 2];
      if constexpr (N == 8) {
        const u32x4 v = *(gptr_u4)p;
        w[0] = v.x;
        w[1] = v.y;
        w[N / 2 - 2] = v.z;
        w[N / 2 - 1] = v.w;
      } else {
        const u32x2 v = *(gptr_u2)p;
        w[0] = v.x;
        w[1] = v.y;
      }
#pragma unroll
      for (int k = 0; k < N / 2; ++k) h[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, w[k]) >> sh) & 0x00ff00ffu;
    } else {
      uint32_t b0, b1 = 0;
      if constexpr (N == 8) {
        const u32x2 v = *(gptr_u2)p;
        b0 = v.x;
        b1 = v.y;
      } else {
        b0 = *(const G1S_GLOBAL uint32_t *)p;
      }
      h[0] = __builtin_amdgcn_perm(0u, b0, 0x0c010c00u);
      h[1] = __builtin_amdgcn_perm(0u, b0, 0x0c030c02u);
      if constexpr (N == 8) {
        h[N / 2 - 2] = __builtin_amdgcn_perm(0u, b1, 0x0c010c00u);
        h[N / 2 - 1] = __builtin_amdgcn_perm(0u, b1, 0x0c030c02u);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int X = X0 + k;
      const uint32_t v = X < pw ? (uint32_t)load_px<BPS>(base, stride, shift, X, Y) : 0u;
      h[k >> 1] |= v << (16 * (k & 1));
    }
  }
}

// ---- the flat-block finder's integer moments of one block row (see k1f.hip.h) ----
constexpr int kMomInts = 16;  // ints per block in the moments buffer
enum {
  kM_S0 = 0, kM_SXU, kM_SYU, kM_SAX, kM_SAY,        // full block: sum p, sum p xi, sum p yi, sum p |xi-16|, sum p |yi-16|
  kM_I0, kM_IXU, kM_IYU, kM_IPP,                    // interior (1..30)^2: sum p, sum p xi, sum p yi, sum p^2
  kM_DXX, kM_DYY, kM_DXY, kM_DX, kM_DY,             // interior central differences
  kM_CLIP                                           // sum p over the block's pixels INSIDE the plane (no replication): get_block_mean's sum, the
                                                    // record's luma_sum (k1_certify writes it; the accumulation launches no longer form it)
};
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
__device__ __forceinline__ uint32_t sad4(uint32_t a, uint32_t c) { return __builtin_amdgcn_sad_u8(a, 0u, c); }
// pk / pu / pd: the 32 packed pixels of row yi and of the rows above / below it (pu, pd are only used
// for yi = 1 .. 30).  s = this row's share of the 14 block sums.  Sums over shifted copies of a row or
// of a neighbouring row are taken from the row itself where they telescope:
//   sum_{x=1..30} p(x+1)^2 + p(x-1)^2 = 2 sum_x p^2 - p0^2 - p1^2 - p30^2 - p31^2,
//   sum_{x=1..30} p(x+1) - p(x-1)     = p30 + p31 - p0 - p1,
//   sum_{yi=1..30} q(yi+1) + q(yi-1)  = sum_r q(r) ([r >= 2] + [r <= 29])   (q = interior-column sum of p^2),
//   sum_{yi=1..30} i0(yi+1) - i0(yi-1) = i0(30) + i0(31) - i0(0) - i0(1).
__device__ __forceinline__ void row_moments(const uint32_t (&pk)[8], const uint32_t (&pu)[8], const uint32_t (&pd)[8], int yi,
                                            int32_t (&s)[14]) {
  uint32_t rowsum = 0, sxu = 0, sax = 0, pp = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    rowsum = sad4(pk[k], rowsum);
    sxu = udot4(pk[k], 0x03020100u + 0x04040404u * (uint32_t)k, sxu);
    // |xi - 16| for xi = 4k .. 4k+3
    const int a0 = abs(4 * k - 16), a1 = abs(4 * k + 1 - 16), a2 = abs(4 * k + 2 - 16), a3 = abs(4 * k + 3 - 16);
    sax = udot4(pk[k], (uint32_t)a0 | ((uint32_t)a1 << 8) | ((uint32_t)a2 << 16) | ((uint32_t)a3 << 24), sax);
    pp = udot4(pk[k], pk[k], pp);
  }
  const uint32_t p0 = pk[0] & 0xffu, p1 = (pk[0] >> 8) & 0xffu, p30 = (pk[7] >> 16) & 0xffu, p31 = pk[7] >> 24;
  const uint32_t i0 = rowsum - p0 - p31;           // interior columns 1 .. 30
  const uint32_t q = pp - p0 * p0 - p31 * p31;     // sum of p^2 over them
  const bool inner = yi >= 1 && yi <= kBlock - 2;
  s[kM_S0] = (int32_t)rowsum;
  s[kM_SXU] = (int32_t)sxu;
  s[kM_SYU] = (int32_t)rowsum * yi;
  s[kM_SAX] = (int32_t)sax;
  s[kM_SAY] = (int32_t)rowsum * abs(yi - 16);
  s[kM_I0] = inner ? (int32_t)i0 : 0;
  s[kM_IXU] = inner ? (int32_t)(sxu - 31u * p31) : 0;
  s[kM_IYU] = inner ? (int32_t)i0 * yi : 0;
  s[kM_IPP] = inner ? (int32_t)q : 0;
  s[kM_DYY] = (int32_t)q * ((yi >= 2 ? 1 : 0) + (yi <= kBlock - 3 ? 1 : 0));
  s[kM_DY] = yi >= kBlock - 2 ? (int32_t)i0 : (yi <= 1 ? -(int32_t)i0 : 0);
  s[kM_DXX] = 0;
  s[kM_DXY] = 0;
  s[kM_DX] = 0;
  if (inner) {
    uint32_t rl = 0, du = 0, rd = 0, ru = 0, ld = 0, lu = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t m = k == 0 ? 0xffffff00u : (k == 7 ? 0x00ffffffu : 0xffffffffu);  // interior columns 1..30
      // p(xi + 1), p(xi - 1) for xi = 4k .. 4k+3
      const uint32_t pr = __builtin_amdgcn_alignbyte(k < 7 ? pk[k + 1] : 0u, pk[k], 1) & m;
      const uint32_t pl = __builtin_amdgcn_alignbyte(pk[k], k > 0 ? pk[k - 1] : 0u, 3) & m;
      rl = udot4(pr, pl, rl);
      du = udot4(pd[k] & m, pu[k], du);
      rd = udot4(pr, pd[k], rd);
      ru = udot4(pr, pu[k], ru);
      ld = udot4(pl, pd[k], ld);
      lu = udot4(pl, pu[k], lu);
    }
    s[kM_DXX] = (int32_t)(2u * pp - p0 * p0 - p1 * p1 - p30 * p30 - p31 * p31 - 2u * rl);
    s[kM_DYY] -= (int32_t)(2u * du);
    s[kM_DXY] = (int32_t)rd - (int32_t)ru - (int32_t)ld + (int32_t)lu;
    s[kM_DX] = (int32_t)(p30 + p31) - (int32_t)(p0 + p1);
  }
}

// the bytes of a packed 32-pixel row from column wv (1 .. 31) on := the byte of column wv - 1
__device__ __forceinline__ void replicate_columns(uint32_t (&p)[8], int wv) {
  const int lk = (wv - 1) >> 2, lb = (wv - 1) & 3;
  uint32_t last = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k == lk) last = (p[k] >> (8 * lb)) & 0xffu;
  const uint32_t fill = last * 0x01010101u;
  const uint32_t keep = lb == 3 ? 0xffffffffu : ((1u << (8 * (lb + 1))) - 1u);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k > lk) p[k] = fill;
    else if (k == lk) p[k] = (p[k] & keep) | (fill & ~keep);
  }
}

// sums over the two 32-lane halves of a wave, N values at a time (DPP stages batched across the values);
// totals in lanes 16 .. 31 and 48 .. 63
template <int N>
__device__ __forceinline__ void half_sums_dpp(int (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0xB1, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x4E, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x141, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x140, 0xf, 0xf, false);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x142, 0xa, 0xf, false);
}

// The same sums for 16 values, with the values SPLIT between the lanes in the first two stages -- a lane pair, then a quad,
// keeps half of the values each and hands the other half over: half the additions (the reduction was a quarter of
// k1_moments).  On return every lane holds, with c = lane & 3, in x[j], j < 4, its 32-lane half's total of v[4 j + c].
__device__ __forceinline__ void half_sums_split16(const int (&v)[16], int (&x)[4]) {
  const int lane = (int)(threadIdx.x & 63);
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  int w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
    w[i] = keep + __builtin_amdgcn_update_dpp(0, send, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]: the lane next door
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int keep = b1 ? w[2 * j + 1] : w[2 * j], send = b1 ? w[2 * j] : w[2 * j + 1];
    x[j] = keep + __builtin_amdgcn_update_dpp(0, send, 0x4E, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]
  }
  // the other lanes of the same class: four lanes apart inside a row of 16 (rotations), then the other row of the half
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] += __builtin_amdgcn_update_dpp(0, x[j], 0x124, 0xf, 0xf, false);  // row_ror:4
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] += __builtin_amdgcn_update_dpp(0, x[j], 0x128, 0xf, 0xf, false);  // row_ror:8
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] += __shfl_xor(x[j], 16, 64);
}

// ---------------------------------------------------------------------------------
// k_zero: every per-batch zero fill in ONE launch (records, accumulators, flags, counters);
// separate memset nodes each cost a dispatch gap in the launch chain.
// ---------------------------------------------------------------------------------
constexpr int kZeroBufs = 7;
struct ZeroJob {
  uint32_t *ptr[kZeroBufs];
  uint32_t ndw[kZeroBufs];  // dwords
};
__global__ __launch_bounds__(256) void k_zero(ZeroJob z) {
  const uint32_t stride = gridDim.x * 256u;
#pragma unroll
  for (int r = 0; r < kZeroBufs; ++r) {
    uint32_t *p = z.ptr[r];
    const uint32_t n = z.ndw[r];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) p[i] = 0u;
  }
}

}  // namespace g1s

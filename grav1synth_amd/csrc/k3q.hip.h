// k3q.hip.h -- K3 for lag 3 by the autocorrelation structure of the normal equations.
//
// add_block_observations (av1-grain diff/solver.rs == libaom noise_model.c) sums, over
// window samples p (w(p) = 1), the outer product of [d(p+c_0)..d(p+c_23), (L(p)), d(p)],
// d = src8 - den8, L = co-located luma residual sum.  Substituting q = p + c_i,
//     A[i][j] = sum_q w(q - c_i) * d(q) * d(q + c_j - c_i)          (i <= j)
//     b[i]    = sum_q w(q - c_i) * d(q) * d(q - c_i)
// and the sum over q is partitioned by the 32x32 block AREA that contains q, and inside
// an area by 4-sample GROUP.  A group is
//     FULL    if every w(q - c_i) it needs is 1: its 324 products collapse onto 46 lag
//             sums G(delta) = sum d(q) d(q + delta), delta = (0..6,0) or (-6..6,1..3);
//     EMPTY   if every such w is 0: contributes nothing;
//     PARTIAL otherwise: the 324 masked products are needed.
// An area is INT if all its groups are full (the block, its left/right/lower neighbours
// are flat with full windows), EXT if all are empty, else MIX.  All integers, exact, and
// independent of how areas / groups are distributed over workgroups.
// The chroma cross terms sum_p w(p) L(p) d(p+c_i), sum w L^2, sum w L d stay p-centric
// under the block's own window (a separate, consistent partition).
//
//   k3_classify          per (frame, kind, area): class + compacted INT / MIX lists
//   k3_lag<KIND, false>  INT areas: 46 (+53 chroma) v_dot4c_i32_i8 per group
//   k3_lag<KIND, true>   MIX areas: the same for their FULL groups (+ L terms, statistics)
//   k3_partial<KIND>     MIX areas: PARTIAL groups, compacted per area, 324 masked products
//   k3q_generic          deferred areas (|d| > 127) and oddballs, plain int32
//   k3q_reduce           all chunk partials -> record int64 S / Sb / nobs
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"

namespace g1s {

constexpr int kQLag = 3;
constexpr int kQN = 24;
constexpr int kNumLags = 46;            // distinct c_j - c_i (incl. 0) and -c_i
constexpr int kNumLTerms = 24 + 2;      // (i,L) for the 24 neighbours, L*L, L*y  (L as ONE int8: |L| <= 127 or the area is deferred)
constexpr int kMaxAreasPerWG = 128;     // int32 accumulators stay exact
enum : uint8_t { kClsExt = 0, kClsInt = 1, kClsMix = 2 };

// lag index: dy = 0: dx 0..6 -> 0..6 ; dy = 1..3: dx -6..6 -> 7 + (dy-1)*13 + (dx+6)
__host__ __device__ constexpr int lag_index(int dx, int dy) { return dy == 0 ? dx : 7 + (dy - 1) * 13 + (dx + 6); }
__host__ __device__ constexpr int coord_x(int k) { return k % 7 - 3; }
__host__ __device__ constexpr int coord_y(int k) { return k / 7 - 3; }

// lag-kernel partial per (frame, plane, chunk): [46 lag sums][53 L terms][nobs term]
constexpr int kQPart = kNumLags + kNumLTerms + 1;
// partial-group kernel: two halves of 162 products, split by anchor
constexpr int kPHalf = 162;
constexpr int kPPart = 2 * kPHalf;
// anchors of half 0: {0,3,4,7,8,11,12,15,16,19,20,23}: 25+22+21+18+17+14+13+10+9+6+5+2 = 162
__host__ __device__ constexpr bool p_in_half(int half, int i) {
  return (((i & 3) == 0 || (i & 3) == 3) ? 0 : 1) == half;
}

struct QParams {
  int nchunks;       // workgroups per frame of k3_lag<.., false>
  int nchunks_mix;   // workgroups per frame of k3_lag<.., true> and k3_partial
  int mixed_fast;    // 1: MIX areas by k3_lag<true> + k3_partial; 0: by k3q_generic (debug)
  long long *lagacc;   // [batch][3][kQPart]  int64 sums of all lag-kernel workgroups (zeroed per batch)
  long long *paracc;   // [batch][3][kPPart]  int64 sums of all k3_partial workgroups (zeroed per batch)
  uint8_t *cls;        // [batch][2][nblocks]  area class per kind (luma, chroma)
  uint8_t *todo;       // [batch][2][nblocks]  1 = area left to k3q_generic
  uint32_t *lists;     // [batch][2 kinds][3 (INT, MIX, GENERIC)][nblocks] compacted area indices
  uint32_t *counts;    // [batch][2][3] list lengths (zeroed before k3_classify)
  uint8_t *winbuf;     // [batch][2][nblocks][32] windows of the 6 blocks an area sees (flat,xs,xe,ys,ye) x 6
};
constexpr uint32_t kEntryFlat = 0x80000000u;      // list entry = block index | (block is flat ? bit 31 : 0)
constexpr uint32_t kEntryDeferred = 0x40000000u;  // set by k3_lag<.., true> on a MIX entry it deferred to k3q_generic
constexpr uint32_t kEntryNone = 0xffffffffu;
constexpr uint32_t kEntryIndex = 0x3fffffffu;

__device__ __forceinline__ int sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, int sh) {
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
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
template <int N>
__device__ __forceinline__ void wave_sum_all(int (&a)[N]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = __shfl_xor(a[i], o, 64);
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] += t[i];
  }
}

// ---------------------------------------------------------------------------------
// window of a block (libaom add_block_observations), in samples of its plane
// ---------------------------------------------------------------------------------
struct Win {
  int flat, xs, xe, ys, ye;
};
__device__ __forceinline__ Win block_window(const uint8_t *mask, int nbw, int nbh, int bx, int by, int bw, int bh,
                                            int pw, int ph) {
  Win w{0, 0, 0, 0, 0};
  if (bx < 0 || by < 0 || bx >= nbw || by >= nbh) return w;
  if (!mask[by * nbw + bx]) return w;
  w.flat = 1;
  w.ys = (by > 0 && mask[(by - 1) * nbw + bx]) ? 0 : kQLag;
  w.xs = (bx > 0 && mask[by * nbw + bx - 1]) ? 0 : kQLag;
  w.ye = min(ph - by * bh, bh);
  w.xe = min(pw - bx * bw - kQLag, (bx + 1 < nbw && mask[by * nbw + bx + 1]) ? bw : (bw - kQLag));
  if (w.xe <= w.xs || w.ye <= w.ys) w.flat = 0;  // empty window
  return w;
}
__device__ __forceinline__ int window_at(const uint8_t *mask, int nbw, int nbh, int bw, int bh, int pw, int ph, int X,
                                         int Y) {
  if (X < 0 || Y < 0 || X >= pw || Y >= ph) return 0;
  const int bx = X / bw, by = Y / bh;
  const Win w = block_window(mask, nbw, nbh, bx, by, bw, bh, pw, ph);
  const int lx = X - bx * bw, ly = Y - by * bh;
  return w.flat && lx >= w.xs && lx < w.xe && ly >= w.ys && ly < w.ye;
}

// ---------------------------------------------------------------------------------
// k3_classify: one thread per block area and plane kind.  The area of block (bx, by)
// needs w on rows by*bh .. by*bh+bh+2, cols bx*bw-3 .. bx*bw+bw+2.
// grid = (ceil(nblocks/256), kinds, batch), block = 256.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3_classify(Geom g, const uint8_t *__restrict__ records, QParams qp) {
  const int blk = blockIdx.x * 256 + threadIdx.x;
  const int kind = blockIdx.y, frame = blockIdx.z;
  const uint8_t *mask = records + (size_t)frame * g.rec_size + g.off_mask;
  uint8_t c = kClsExt;
  if (blk < g.nblocks) {
    const int sx = kind ? g.xdec : 0, sy = kind ? g.ydec : 0;
    const int bw = kBlock >> sx, bh = kBlock >> sy, pw = g.W >> sx, ph = g.H >> sy;
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int AX0 = bx * bw - kQLag, AX1 = bx * bw + bw + kQLag, AY0 = by * bh, AY1 = by * bh + bh + kQLag;
    bool all1 = !(AX0 < 0 || AX1 > pw || AY1 > ph);
    bool any1 = false;
    uint8_t *wb = qp.winbuf + (((size_t)frame * 2 + kind) * g.nblocks + blk) * 32;
    for (int dby = 0; dby <= 1; ++dby) {
      for (int dbx = -1; dbx <= 1; ++dbx) {
        const int Bx = bx + dbx, By = by + dby;
        const Win w = block_window(mask, g.nbw, g.nbh, Bx, By, bw, bh, pw, ph);
        {
          uint8_t *o = wb + (dby * 3 + dbx + 1) * 5;
          o[0] = (uint8_t)w.flat;
          o[1] = (uint8_t)w.xs;
          o[2] = (uint8_t)max(w.xe, 0);
          o[3] = (uint8_t)w.ys;
          o[4] = (uint8_t)max(w.ye, 0);
        }
        const int rx0 = max(AX0, Bx * bw), rx1 = min(AX1, Bx * bw + bw);
        const int ry0 = max(AY0, By * bh), ry1 = min(AY1, By * bh + bh);
        if (rx0 >= rx1 || ry0 >= ry1) continue;
        if (!w.flat) {
          all1 = false;
          continue;
        }
        const int ax0 = rx0 - Bx * bw, ax1 = rx1 - Bx * bw, ay0 = ry0 - By * bh, ay1 = ry1 - By * bh;
        if (max(ax0, w.xs) < min(ax1, w.xe) && max(ay0, w.ys) < min(ay1, w.ye)) any1 = true;
        if (!(w.xs <= ax0 && ax1 <= w.xe && w.ys <= ay0 && ay1 <= w.ye)) all1 = false;
      }
    }
    c = all1 ? kClsInt : (any1 ? kClsMix : kClsExt);
    const size_t o = ((size_t)frame * 2 + kind) * g.nblocks + blk;
    qp.cls[o] = c;
    // deferred-to-generic: MIX when the fast mixed path is off; a flat block whose own area
    // is EXT still needs its block statistics
    qp.todo[o] = ((c == kClsMix && !qp.mixed_fast) || (c == kClsExt && mask[blk])) ? 1 : 0;
  }
  // compacted lists, one atomic per wave and class (any order: the sums are exact integers)
  const int lane = threadIdx.x & 63;
  const bool gen = blk < g.nblocks && qp.todo[((size_t)frame * 2 + kind) * g.nblocks + blk] != 0;
  for (int which = 0; which < 3; ++which) {
    const bool mine = blk < g.nblocks &&
                      (which == 0 ? c == kClsInt : (which == 1 ? (c == kClsMix && qp.mixed_fast) : gen));
    const unsigned long long b = __ballot(mine);
    if (b == 0) continue;
    const size_t lo = ((size_t)frame * 2 + kind) * 3 + which;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&qp.counts[lo], (uint32_t)__popcll(b));
    base = __shfl(base, 0, 64);
    if (mine)
      qp.lists[lo * g.nblocks + base + __popcll(b & ((1ull << lane) - 1ull))] =
          (uint32_t)blk | (mask[blk] ? kEntryFlat : 0u);
  }
}

// XCD-aware contiguous slice of a list of n entries for workgroup b of N (N % 8 == 0):
// block b runs on XCD b % 8 (observed, speed only), so XCD x gets the list range
// [x*n/8, (x+1)*n/8) and its workgroups walk adjacent sub-slices: neighbouring areas
// (which share halo lines) stay within one L2.
__device__ __forceinline__ void list_slice(int b, int N, int n, int &begin, int &end) {
  const int cpx = N >> 3;
  const int w = (b & 7) * cpx + (b >> 3);
  const int per = (n + N - 1) / N;
  begin = min(n, w * per);
  end = min(n, begin + per);
}

// ---- 8 consecutive samples of a row (vector global load, narrowed later) ----
struct Px8 {
  // plain scalars (not HIP's uint4 wrapper): arrays of this struct must stay in VGPRs
  uint32_t x, y, z, w;  // u16: 8 samples; u8: x, y hold 8 samples
  int state;            // 0 = zero (outside the plane), 1 = raw valid, 2 = edge segment: per-sample loads later
};
__device__ __forceinline__ Px8 fetch8(const uint8_t *base, uint32_t stride, int bps, bool vec_ok, int X0, int Y,
                                      int pw, int ph) {
  Px8 r;
  r.x = r.y = r.z = r.w = 0;
  r.state = 0;
  if (Y < 0 || Y >= ph || X0 + 8 <= 0 || X0 >= pw) return r;
  if (X0 >= 0 && X0 + 8 <= pw && vec_ok) {
    gptr_u8 p = as_global(base) + (size_t)Y * stride + (size_t)X0 * bps;
    if (bps == 2) {
      const u32x4 v = *(gptr_u4)p;
      r.x = v.x;
      r.y = v.y;
      r.z = v.z;
      r.w = v.w;
    } else {
      const u32x2 v = *(gptr_u2)p;
      r.x = v.x;
      r.y = v.y;
    }
    r.state = 1;
  } else {
    r.state = 2;
  }
  return r;
}
__device__ __forceinline__ void unpack8(const Px8 &p, const uint8_t *base, uint32_t stride, int bps, int shift,
                                        int X0, int Y, int pw, int (&v)[8]) {
  if (p.state == 1) {
    if (bps == 2) {
      const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] = (int)(((w[k] & 0xffffu) >> shift) & 0xffu);
        v[2 * k + 1] = (int)(((w[k] >> 16) >> shift) & 0xffu);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = (int)((p.x >> (8 * k)) & 0xffu);
        v[4 + k] = (int)((p.y >> (8 * k)) & 0xffu);
      }
    }
  } else if (p.state == 2) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int X = X0 + k;
      v[k] = (X >= 0 && X < pw) ? load_px_rt(base, stride, bps, shift, X, Y) : 0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0;
  }
}

// ---- packed 16-bit staging arithmetic ---------------------------------------------
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
// 8 samples -> 4 dwords of two u16 each, already narrowed to 8 bits (frame_into_u8:
// `(v >> (bd - 8)) as u8`).  Edge segments (state 2) take the per-sample path.
__device__ __forceinline__ void to16(const Px8 &p, const uint8_t *base, uint32_t stride, int bps, int shift, int X0,
                                     int Y, int pw, uint32_t (&h)[4]) {
  if (p.state == 1) {
    if (bps == 2) {
      const uint32_t w[4] = {p.x, p.y, p.z, p.w};
      const u16x2 sh = {(unsigned short)shift, (unsigned short)shift};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        h[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, w[k]) >> sh) & 0x00ff00ffu;
    } else {
      h[0] = __builtin_amdgcn_perm(0u, p.x, 0x0c010c00u);
      h[1] = __builtin_amdgcn_perm(0u, p.x, 0x0c030c02u);
      h[2] = __builtin_amdgcn_perm(0u, p.y, 0x0c010c00u);
      h[3] = __builtin_amdgcn_perm(0u, p.y, 0x0c030c02u);
    }
  } else if (p.state == 2) {
    int v[8];
    unpack8(p, base, stride, bps, shift, X0, Y, pw, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = (uint32_t)v[2 * k] | ((uint32_t)v[2 * k + 1] << 16);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = 0;
  }
}
// running min / max of packed i16 -> any |d| > 127 ?
__device__ __forceinline__ bool range_bad(uint32_t mx, uint32_t mn) {
  const int mxa = max((int)(short)(mx & 0xffffu), (int)(short)(mx >> 16));
  const int mna = min((int)(short)(mn & 0xffffu), (int)(short)(mn >> 16));
  return mxa > 127 || mna < -127;
}

// KIND: 0 = luma; 1 = chroma 4:2:0; 2 = chroma 4:2:2; 3 = chroma 4:4:4.
template <int KIND>
struct QShape {
  static constexpr bool kChroma = KIND != 0;
  static constexpr int SX = (KIND == 1 || KIND == 2) ? 1 : 0;
  static constexpr int SY = (KIND == 1) ? 1 : 0;
  static constexpr int BW = kBlock >> SX, BH = kBlock >> SY;
  static constexpr int G = BW / 4;               // 4-sample groups per row
  static constexpr int NG = G * BH;              // groups per plane area
  static constexpr int ROWS_PER_STEP = 64 / G;   // rows covered by one wave step
  static constexpr int NS = BH / ROWS_PER_STEP;  // wave steps per plane area
  static constexpr int NPL = kChroma ? 2 : 1;
  static constexpr int WAVES = (NPL * NS) < 4 ? (NPL * NS) : 4;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int STEPS_PER_WAVE = NPL * NS / WAVES;
  static constexpr int PITCH_DW = 32 + G;  // conflict-free for the (group, row) lane map
  static constexpr int PITCH = PITCH_DW * 4;
  static constexpr int UP = kChroma ? kQLag : 0;  // rows above the area (chroma L terms are p-centric)
  static constexpr int TH = BH + kQLag + UP;      // tile rows: -UP .. BH+2
  static constexpr int SEGS = (BW + 16) / 8;      // 8-sample segments per tile row: x = -8 .. BW+7
  static constexpr int NTILE = TH * SEGS * NPL;
  static constexpr int LCH = 8 >> SX;  // chroma samples per L item (8 luma samples wide)
  static constexpr int LSEGS = BW / LCH;
  static constexpr int NL = kChroma ? BH * LSEGS : 0;
  static constexpr int LROWS = 1 << SY;
  static constexpr int NITEMS = NTILE + NL;
  static constexpr int SLOT = kChroma ? LROWS : 1;
  static constexpr int TILE_BYTES = TH * PITCH;
  static constexpr int LTILE_BYTES = BH * PITCH;
  static constexpr int DATA_BYTES = NPL * TILE_BYTES + (kChroma ? LTILE_BYTES : 0);
  static constexpr int WTILE_BYTES = (BH + kQLag) * PITCH;  // rows 0..BH+2
  static constexpr int NACC = kNumLags + (kChroma ? kNumLTerms : 0);
};

// ---------------------------------------------------------------------------------
// Stager: HBM -> registers (prefetch, vector loads) -> LDS tiles of one area.
// LDS layout: plane tile pl at lds + pl*TILE_BYTES, sample (x, y) (x in -8..BW+7,
// y in -UP..BH+2) at byte (y + UP) * PITCH + 8 + x, so group g (x = 4g) is dword g + 2;
// then La, Lb tiles (block proper, byte y*PITCH + x); then (WTILE) the window-indicator
// tile, rows 0..BH+2, same column layout, bytes 0xFF / 0x00.
// ---------------------------------------------------------------------------------
template <int KIND, int NT, bool WTILE, bool LTERMS>
struct Stager {
  using S = QShape<KIND>;
  static constexpr int NITEMS = S::NTILE + (LTERMS ? S::NL : 0);
  static constexpr int MAXIT = (NITEMS + NT - 1) / NT;
  Px8 ps[MAXIT][S::SLOT], pd[MAXIT][S::SLOT];

  __device__ __forceinline__ void fetch(const FramePlanes &fp, const Geom &g, int tid, int blk) {
    constexpr bool CHROMA = S::kChroma;
    const int pw = g.W >> S::SX, ph = g.H >> S::SY;
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * S::BW, y_o = by * S::BH;
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
      // ONE straight-line path per (k, q) slot, parameters chosen by selects: every slot is
      // written exactly once with a static index, so ps / pd stay in VGPRs.
      const int it = tid + k * NT;
      const bool tile = it < S::NTILE;
      const bool lit = LTERMS && !tile && it < NITEMS;
      const int pl = it / (S::TH * S::SEGS);
      const int r = it - pl * (S::TH * S::SEGS);
      const int ty = r / S::SEGS, sg = r - ty * S::SEGS;
      const int rl = it - S::NTILE;
      const int yl = rl / S::LSEGS, sgl = rl - yl * S::LSEGS;
      // (no runtime index into fp: that would push the frame table to scratch memory)
      const uint8_t *spt = CHROMA ? (pl ? fp.src[2] : fp.src[1]) : fp.src[0];
      const uint8_t *dpt = CHROMA ? (pl ? fp.den[2] : fp.den[1]) : fp.den[0];
      const uint32_t sstt = CHROMA ? (pl ? fp.src_stride[2] : fp.src_stride[1]) : fp.src_stride[0];
      const uint32_t dstt = CHROMA ? (pl ? fp.den_stride[2] : fp.den_stride[1]) : fp.den_stride[0];
      const int vst = CHROMA ? (pl ? (g.vec_mask >> 2) : (g.vec_mask >> 1)) : g.vec_mask;
      const int vdt = CHROMA ? (pl ? (g.vec_mask >> 5) : (g.vec_mask >> 4)) : (g.vec_mask >> 3);
      const uint8_t *sp = tile ? spt : fp.src[0];
      const uint8_t *dp = tile ? dpt : fp.den[0];
      const uint32_t sst = tile ? sstt : fp.src_stride[0];
      const uint32_t dst = tile ? dstt : fp.den_stride[0];
      const bool vs = ((tile ? vst : g.vec_mask) & 1) != 0, vd = ((tile ? vdt : (g.vec_mask >> 3)) & 1) != 0;
      const int X0 = tile ? (x_o - 8 + 8 * sg) : ((x_o + sgl * S::LCH) << S::SX);
      const int Yb = tile ? (y_o - S::UP + ty) : ((y_o + yl) << S::SY);
      const int pwq = tile ? pw : g.W, phq = tile ? ph : g.H;
#pragma unroll
      for (int q = 0; q < S::SLOT; ++q) {
        const bool valid = tile ? (q == 0) : lit;
        ps[k][q] = fetch8(sp, sst, g.src_bps, vs, valid ? X0 : -64, Yb + q, pwq, phq);  // X0 = -64: outside -> zero
        pd[k][q] = fetch8(dp, dst, g.den_bps, vd, valid ? X0 : -64, Yb + q, pwq, phq);
      }
    }
  }

  // narrow, subtract, range-check, write LDS.  Returns true if some |d| > 127.
  // window-indicator tile (rows 0..BH+2 of the area) from the six windows in s_win
  __device__ __forceinline__ void build_wtile(int tid, uint8_t *lds, const uint8_t *s_win) {
    constexpr int bw = S::BW, bh = S::BH;
    constexpr int NW = (bh + kQLag) * S::SEGS;
#pragma unroll
    for (int k = 0; k < (NW + NT - 1) / NT; ++k) {
      const int it = tid + k * NT;
      if (it < NW) {
        const int y = it / S::SEGS, sg = it - y * S::SEGS;
        const int dby = y >= bh ? 1 : 0, ly = y - dby * bh;
        uint32_t wlo = 0, whi = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int x = -8 + 8 * sg + q;
          const int dbx = x < 0 ? -1 : (x >= bw ? 1 : 0), lx = x - dbx * bw;
          const uint8_t *wn = s_win + (dby * 3 + dbx + 1) * 5;
          const bool in = wn[0] && lx >= wn[1] && lx < wn[2] && ly >= wn[3] && ly < wn[4];
          const uint32_t b = in ? 0xffu : 0u;
          if (q < 4) wlo |= b << (8 * q); else whi |= b << (8 * (q - 4));
        }
        *reinterpret_cast<uint2 *>(lds + S::DATA_BYTES + y * S::PITCH + 8 * sg) = make_uint2(wlo, whi);
      }
    }
  }

  __device__ __forceinline__ bool store(const FramePlanes &fp, const Geom &g, int tid, int blk, uint8_t *lds,
                                        int &lsum) {
    constexpr bool CHROMA = S::kChroma;
    constexpr int bw = S::BW, bh = S::BH;
    const int pw = g.W >> S::SX;
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * bw, y_o = by * bh;
    uint32_t mx = 0, mn = 0;  // packed running max / min of the residuals
    bool lbad = false;        // the luma residual sum L does not fit int8
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
      const int it = tid + k * NT;
      if (it < S::NTILE) {
        const int pl = it / (S::TH * S::SEGS);
        const int r = it - pl * (S::TH * S::SEGS);
        const int ty = r / S::SEGS, sg = r - ty * S::SEGS;
        const uint8_t *sp = CHROMA ? (pl ? fp.src[2] : fp.src[1]) : fp.src[0];
        const uint8_t *dp = CHROMA ? (pl ? fp.den[2] : fp.den[1]) : fp.den[0];
        const uint32_t sst = CHROMA ? (pl ? fp.src_stride[2] : fp.src_stride[1]) : fp.src_stride[0];
        const uint32_t dst = CHROMA ? (pl ? fp.den_stride[2] : fp.den_stride[1]) : fp.den_stride[0];
        const int X0 = x_o - 8 + 8 * sg, Y = y_o - S::UP + ty;
        uint32_t hs[4], hv[4], d[4];
        to16(ps[k][0], sp, sst, g.src_bps, g.src_shift, X0, Y, pw, hs);
        to16(pd[k][0], dp, dst, g.den_bps, g.den_shift, X0, Y, pw, hv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[q] = pk_sub(hs[q], hv[q]);
          mx = pk_max(mx, d[q]);
          mn = pk_min(mn, d[q]);
        }
        const uint32_t lo = __builtin_amdgcn_perm(d[1], d[0], 0x06040200u);
        const uint32_t hi = __builtin_amdgcn_perm(d[3], d[2], 0x06040200u);
        if (!CHROMA && sg >= 1 && sg <= 4 && ty < bh) {  // block proper -> luma sum of the source
          lsum = (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(hs[1], hs[0], 0x06040200u), 0u, (uint32_t)lsum);
          lsum = (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(hs[3], hs[2], 0x06040200u), 0u, (uint32_t)lsum);
        }
        *reinterpret_cast<uint2 *>(lds + pl * S::TILE_BYTES + ty * S::PITCH + 8 * sg) = make_uint2(lo, hi);
      } else if (LTERMS && it < NITEMS) {
        const int r = it - S::NTILE;
        const int y = r / S::LSEGS, sg = r - y * S::LSEGS;
        const int X0 = (x_o + sg * S::LCH) << S::SX;
        uint32_t dsum[4] = {0, 0, 0, 0};  // packed i16 residuals, summed over the LROWS luma rows
#pragma unroll
        for (int q = 0; q < S::LROWS; ++q) {
          const int Y = ((y_o + y) << S::SY) + q;
          uint32_t hs[4], hv[4];
          to16(ps[k][q], fp.src[0], fp.src_stride[0], g.src_bps, g.src_shift, X0, Y, g.W, hs);
          to16(pd[k][q], fp.den[0], fp.den_stride[0], g.den_bps, g.den_shift, X0, Y, g.W, hv);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t d = pk_sub(hs[e], hv[e]);
            mx = pk_max(mx, d);
            mn = pk_min(mn, d);
            dsum[e] = pk_add(dsum[e], d);
          }
        }
        uint8_t *ta = lds + S::NPL * S::TILE_BYTES + y * S::PITCH + sg * S::LCH;
        if (S::SX == 1) {
          // L = horizontal pair sums: 4 chroma samples per item; must fit int8 like d
          uint32_t a0 = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int L = (int)(short)(dsum[e] & 0xffffu) + (int)(short)(dsum[e] >> 16);
            lbad |= (L > 127) | (L < -127);
            a0 |= ((uint32_t)L & 0xffu) << (8 * e);
          }
          *reinterpret_cast<uint32_t *>(ta) = a0;
        } else {
          // L = the (row-summed) luma residual itself: 8 chroma samples per item
          uint32_t lmx = 0, lmn = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lmx = pk_max(lmx, dsum[e]);
            lmn = pk_min(lmn, dsum[e]);
          }
          lbad |= range_bad(lmx, lmn);
          *reinterpret_cast<uint2 *>(ta) = make_uint2(__builtin_amdgcn_perm(dsum[1], dsum[0], 0x06040200u),
                                                       __builtin_amdgcn_perm(dsum[3], dsum[2], 0x06040200u));
        }
      }
    }
    return range_bad(mx, mn) || lbad;
  }
};

// Group classification from the w tile: w32 points at dword (row*PITCH_DW + g) of the w
// tile; the group's own samples are dword +2.  FULL / EMPTY are decided on the bytes
// x-3 .. x+6 of rows 0..3 (a superset of what the 24 masks read; both kernels use this
// same predicate, so the partition into full / partial / empty groups is consistent).
template <int PITCH_DW>
__device__ __forceinline__ void group_state(const uint32_t *w32, bool &full, bool &empty) {
  uint32_t all_and = 0xffffffffu, all_or = 0;
#pragma unroll
  for (int dy = 0; dy <= 3; ++dy) {
    const uint32_t *rp = w32 + dy * PITCH_DW;
    const uint32_t q1 = rp[1], q2 = rp[2], q3 = rp[3];
    all_and &= (q1 | 0x000000ffu) & q2 & (q3 | 0xff000000u);
    all_or |= (q1 & 0xffffff00u) | q2 | (q3 & 0x00ffffffu);
  }
  full = all_and == 0xffffffffu;
  empty = all_or == 0;
}

// ---------------------------------------------------------------------------------
// k3_lag<KIND, MIXED>: 46 lag sums (+53 chroma L terms) per group.
//   MIXED = false: the INT list (every group full, own window = whole block)
//   MIXED = true : the MIX list; only FULL groups enter the lag sums; L terms, nobs and
//                  block statistics use the block's own window / flat flag.
// grid = (nchunks or nchunks_mix, 1, batch), block = QShape::THREADS.
// int32 safety: per step |sum| <= 4*127^2; <= 128 areas * STEPS_PER_WAVE(<=2); x64 lanes < 2^31.
// ---------------------------------------------------------------------------------
// WV = waves per workgroup.  WV = 1 makes a wave autonomous: it stages and multiplies whole
// areas alone, workgroup barriers degenerate, and a CU runs 8-12 independent area pipelines.
constexpr int kLagWaves = 0;  // 0 = QShape<KIND>::WAVES (4 luma, 2 chroma 4:2:0); 1 was measured slower and is invalid for chroma
template <int KIND, bool MIXED, int WV>
__global__ __launch_bounds__(64 * (WV ? WV : QShape<KIND>::WAVES)) void k3_lag(const FramePlanes *__restrict__ frames,
                                                                              Geom g, QParams qp,
                                                                              uint8_t *__restrict__ records) {
  using S = QShape<KIND>;
  constexpr bool CHROMA = S::kChroma;
  constexpr int WAVES = WV ? WV : S::WAVES;
  constexpr int NACC = S::NACC, NT = 64 * WAVES;
  constexpr int STEPS_PER_WAVE = S::NPL * S::NS / WAVES;
  static_assert(S::NPL * S::NS % WAVES == 0 && WAVES % S::NPL == 0, "a wave owns the accumulators of ONE plane");
  __shared__ __attribute__((aligned(16))) uint8_t lds[S::DATA_BYTES + (MIXED ? S::WTILE_BYTES : 0)];
  __shared__ int s_flag[2];
  __shared__ int s_stat[2][4][4];
  __shared__ __attribute__((aligned(16))) uint8_t s_win[32];  // windows of the six blocks the current MIX area sees

  const int frame = blockIdx.z, chunk = blockIdx.x;
  const int stride = MIXED ? qp.nchunks_mix : qp.nchunks;
  const FramePlanes fp = frames[frame];
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  const uint8_t *mask = rec + g.off_mask;
  const size_t lsel = ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * 3 + (MIXED ? 1 : 0);
  uint32_t *list = qp.lists + lsel * g.nblocks;
  const int nlist = (int)qp.counts[lsel];
  uint8_t *todo = qp.todo + ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * g.nblocks;
  const uint8_t *winbase = qp.winbuf + ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * g.nblocks * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lg = lane % S::G, lr = lane / S::G;

  int acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  int nobs = 0;  // INT: number of areas (x BW*BH in the reducer); MIX: window samples
  if (tid < 2) s_flag[tid] = 0;

  // Software pipeline over the list slice.  Per iteration k: the samples of area k were
  // requested one iteration ago, the list entry of area k+1 two iterations ago; no load that
  // is waited on is ever issued after the prefetch (vmcnt retires in order).
  Stager<KIND, NT, MIXED, CHROMA> st;
  int li, li_end;
  list_slice(chunk, stride, nlist, li, li_end);
  auto entry_at = [&](int pos) -> uint32_t { return pos < li_end ? list[pos] : kEntryNone; };
  uint32_t e_cur = entry_at(li), e_nxt = entry_at(li + 1);
  uint32_t wreg = 0;  // threads 0..7: one dword of the windows of the area being prefetched
  if (e_cur != kEntryNone) {
    const int b0 = (int)(e_cur & kEntryIndex);
    if (MIXED && tid < 8) wreg = reinterpret_cast<const uint32_t *>(winbase + (size_t)b0 * 32)[tid];
    st.fetch(fp, g, tid, b0);
  }
  __syncthreads();

  int iter = 0;
  for (; e_cur != kEntryNone; e_cur = e_nxt, e_nxt = entry_at(li + 1)) {
    const int blk = (int)(e_cur & kEntryIndex);
    const bool isflat = (e_cur & kEntryFlat) != 0;
    const int fl = iter & 1;
    ++iter;
    if (MIXED && tid < 8) reinterpret_cast<uint32_t *>(s_win)[tid] = wreg;
    int lsum = 0;
    const bool bad = st.store(fp, g, tid, blk, lds, lsum);
    if (bad) s_flag[fl] = 1;
    if (!CHROMA) {
      lsum = wave_sum(lsum);
      if (lane == 0) s_stat[fl][wave][3] = lsum;
    }
    __syncthreads();  // d tiles + s_win complete
    ++li;
    if (e_nxt != kEntryNone) {  // prefetch area k+1 (in flight during the products below)
      const int bn = (int)(e_nxt & kEntryIndex);
      if (MIXED && tid < 8) wreg = reinterpret_cast<const uint32_t *>(winbase + (size_t)bn * 32)[tid];
      st.fetch(fp, g, tid, bn);
    }
    const bool deferred = s_flag[fl] != 0;
    if (tid == 0) s_flag[fl ^ 1] = 0;
    if (deferred) {
      if (tid == 0) {  // the generic kernel redoes this area in int32
        todo[blk] = 1;
        const size_t lg3 = ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * 3 + 2;
        qp.lists[lg3 * g.nblocks + atomicAdd(&qp.counts[lg3], 1u)] = (uint32_t)blk;
        if (MIXED) list[li - 1] = e_cur | kEntryDeferred;  // k3_partial skips it as well
      }
      __syncthreads();
      continue;
    }
    if (MIXED) {
      st.build_wtile(tid, lds, s_win);
      __syncthreads();
    }
    if (!MIXED && tid == 0) ++nobs;

    int sd = 0, sd2 = 0;
#pragma unroll 1
    for (int s = 0; s < STEPS_PER_WAVE; ++s) {
      const int widx = wave * STEPS_PER_WAVE + s;
      const int pl = widx / S::NS, step = widx - pl * S::NS;
      const int row = step * S::ROWS_PER_STEP + lr;  // sample row of the area
      const uint32_t *t32 = reinterpret_cast<const uint32_t *>(lds + pl * S::TILE_BYTES) + (row + S::UP) * S::PITCH_DW + lg;
      const uint32_t c0 = t32[2], c1 = t32[3], c2 = t32[4];
      uint32_t D0 = c0, Wc = 0xffffffffu;
      if (MIXED) {
        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(lds + S::DATA_BYTES) + row * S::PITCH_DW + lg;
        bool full, empty;
        group_state<S::PITCH_DW>(w32, full, empty);
        Wc = w32[2];
        if (!full) D0 = 0;  // partial groups belong to k3_partial, empty ones to nobody
        if (pl == 0) nobs = sdot4((int)(Wc & 0x01010101u), 0x01010101, nobs);
      }
      acc[0] = sdot4((int)D0, (int)c0, acc[0]);
      acc[1] = sdot4((int)D0, (int)alignbyte(c1, c0, 1), acc[1]);
      acc[2] = sdot4((int)D0, (int)alignbyte(c1, c0, 2), acc[2]);
      acc[3] = sdot4((int)D0, (int)alignbyte(c1, c0, 3), acc[3]);
      acc[4] = sdot4((int)D0, (int)c1, acc[4]);
      acc[5] = sdot4((int)D0, (int)alignbyte(c2, c1, 1), acc[5]);
      acc[6] = sdot4((int)D0, (int)alignbyte(c2, c1, 2), acc[6]);
#pragma unroll
      for (int dy = 1; dy <= 3; ++dy) {
        const uint32_t *rp = t32 + dy * S::PITCH_DW;
        const uint32_t e0 = rp[0], e1 = rp[1], e2 = rp[2], e3 = rp[3], e4 = rp[4];
        const int b = 7 + (dy - 1) * 13;
        acc[b + 0] = sdot4((int)D0, (int)alignbyte(e1, e0, 2), acc[b + 0]);    // dx = -6
        acc[b + 1] = sdot4((int)D0, (int)alignbyte(e1, e0, 3), acc[b + 1]);    // -5
        acc[b + 2] = sdot4((int)D0, (int)e1, acc[b + 2]);                      // -4
        acc[b + 3] = sdot4((int)D0, (int)alignbyte(e2, e1, 1), acc[b + 3]);    // -3
        acc[b + 4] = sdot4((int)D0, (int)alignbyte(e2, e1, 2), acc[b + 4]);    // -2
        acc[b + 5] = sdot4((int)D0, (int)alignbyte(e2, e1, 3), acc[b + 5]);    // -1
        acc[b + 6] = sdot4((int)D0, (int)e2, acc[b + 6]);                      // 0
        acc[b + 7] = sdot4((int)D0, (int)alignbyte(e3, e2, 1), acc[b + 7]);    // +1
        acc[b + 8] = sdot4((int)D0, (int)alignbyte(e3, e2, 2), acc[b + 8]);    // +2
        acc[b + 9] = sdot4((int)D0, (int)alignbyte(e3, e2, 3), acc[b + 9]);    // +3
        acc[b + 10] = sdot4((int)D0, (int)e3, acc[b + 10]);                    // +4
        acc[b + 11] = sdot4((int)D0, (int)alignbyte(e4, e3, 1), acc[b + 11]);  // +5
        acc[b + 12] = sdot4((int)D0, (int)alignbyte(e4, e3, 2), acc[b + 12]);  // +6
      }
      sd = sdot4((int)c0, 0x01010101, sd);
      sd2 = sdot4((int)c0, (int)c0, sd2);
      if (CHROMA) {
        // p-centric L terms under the block's own window (Wc; all ones for INT areas)
        const uint32_t *ta = reinterpret_cast<const uint32_t *>(lds + S::NPL * S::TILE_BYTES);
        const uint32_t Lr = ta[row * S::PITCH_DW + lg];
        const uint32_t Lm = Lr & Wc;
        int *al = acc + kNumLags;
#pragma unroll
        for (int cy = -3; cy <= 0; ++cy) {
          const uint32_t *rp = t32 + cy * S::PITCH_DW;
          const uint32_t u1 = rp[1], u2 = rp[2], u3 = rp[3];
#pragma unroll
          for (int cx = -3; cx <= 3; ++cx) {
            if (cy == 0 && cx >= 0) continue;
            const int k = (cy + 3) * 7 + (cx + 3);
            uint32_t v;
            if (cx < 0) v = alignbyte(u2, u1, 4 + cx);
            else if (cx == 0) v = u2;
            else v = alignbyte(u3, u2, cx);
            al[k] = sdot4((int)Lm, (int)v, al[k]);
          }
        }
        al[24] = sdot4((int)Lm, (int)Lr, al[24]);
        al[25] = sdot4((int)Lm, (int)c0, al[25]);
      }
    }
    // block statistics (only meaningful / stored for flat blocks)
    sd = wave_sum(sd);
    sd2 = wave_sum(sd2);
    if (lane == 0) {
      s_stat[fl][wave][0] = sd;
      s_stat[fl][wave][1] = sd2;
    }
    __syncthreads();
    if (tid == 0 && (!MIXED || isflat)) {
      const int(*sp)[4] = s_stat[fl];
      int a[2] = {0, 0}, b[2] = {0, 0}, l = 0;
      for (int w = 0; w < WAVES; ++w) {
        const int pl = (w * STEPS_PER_WAVE) / S::NS;  // a wave's steps stay within one plane
        a[pl] += sp[w][0];
        b[pl] += sp[w][1];
        l += sp[w][3];
      }
      if (!CHROMA) {
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[0])[blk] = a[0];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[0])[blk] = (uint32_t)b[0];
        reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)l;
      } else {
        for (int pl = 0; pl < 2; ++pl) {
          reinterpret_cast<int32_t *>(rec + g.off_sum_d[1 + pl])[blk] = a[pl];
          reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[1 + pl])[blk] = (uint32_t)b[pl];
        }
      }
    }
  }

  // ---- wave reduction + partial store: waves of the same plane add up ----
  __syncthreads();
  int *red = reinterpret_cast<int *>(lds);
  static_assert(4 * (kQPart + 1) * 4 <= S::DATA_BYTES, "reduction scratch must fit");
  {
    constexpr int CH = 23;
#pragma unroll
    for (int b0 = 0; b0 < NACC; b0 += CH) {
      int tmp[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) tmp[i] = (b0 + i < NACC) ? acc[b0 + i] : 0;
      wave_sum_all<CH>(tmp);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
          if (b0 + i < NACC) red[wave * (kQPart + 1) + b0 + i] = tmp[i];
      }
    }
    if (MIXED) nobs = wave_sum(nobs);
    if (lane == 0) red[wave * (kQPart + 1) + kQPart] = nobs;
  }
  __syncthreads();
  for (int pl = 0; pl < S::NPL; ++pl) {
    unsigned long long *out =
        reinterpret_cast<unsigned long long *>(qp.lagacc) + ((size_t)frame * 3 + (CHROMA ? 1 + pl : 0)) * kQPart;
    for (int i = tid; i < NACC; i += NT) {
      int v = 0;
      for (int w = 0; w < WAVES; ++w)
        if ((w * STEPS_PER_WAVE) / S::NS == pl) v += red[w * (kQPart + 1) + i];
      if (v != 0) atomicAdd(&out[i], (unsigned long long)(long long)v);
    }
    if (tid == 0) {
      long long v = 0;
      if (MIXED) {
        for (int w = 0; w < WAVES; ++w)
          if ((w * STEPS_PER_WAVE) / S::NS == 0) v += red[w * (kQPart + 1) + kQPart];  // counted on plane 0
      } else {
        v = (long long)red[kQPart] * S::BW * S::BH;  // thread 0 counted the areas
      }
      if (v != 0) atomicAdd(&out[kQPart - 1], (unsigned long long)v);
    }
  }
}

// ---------------------------------------------------------------------------------
// k3_partial<KIND>: PARTIAL groups of MIX areas, compacted per area, 324 masked products
//     acc(i, j) += dot4(D(q) & W(q - c_i), D(q + c_j - c_i)),  acc(i, y) likewise.
// A wave owns one (plane, half) combo; waves sharing a combo interleave list steps.
// grid = (nchunks_mix, 1, batch), block = 256.
// ---------------------------------------------------------------------------------
template <int HALF, int PITCH_DW>
__device__ __forceinline__ void partial_products(int (&acc)[kPHalf], const uint32_t (&D)[kNumLags],
                                                 const uint32_t *w32) {
  // window dwords of rows 0..3: q1 = x-4.., q2 = x.., q3 = x+4..
  uint32_t q1[4], q2[4], q3[4];
#pragma unroll
  for (int dy = 0; dy <= 3; ++dy) {
    const uint32_t *rp = w32 + dy * PITCH_DW;
    q1[dy] = rp[1];
    q2[dy] = rp[2];
    q3[dy] = rp[3];
  }
  int idx = 0;
#pragma unroll
  for (int i = 0; i < kQN; ++i) {
    if (!p_in_half(HALF, i)) continue;
    const int dx = -coord_x(i), dy = -coord_y(i);  // W(q - c_i)
    uint32_t wi;
    if (dx < 0) wi = alignbyte(q2[dy], q1[dy], 4 + dx);
    else if (dx == 0) wi = q2[dy];
    else wi = alignbyte(q3[dy], q2[dy], dx);
    const uint32_t md = D[0] & wi;
#pragma unroll
    for (int j = i; j < kQN; ++j) {
      acc[idx] = sdot4((int)md, (int)D[lag_index(coord_x(j) - coord_x(i), coord_y(j) - coord_y(i))], acc[idx]);
      ++idx;
    }
    acc[idx] = sdot4((int)md, (int)D[lag_index(dx, dy)], acc[idx]);
    ++idx;
  }
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void k3_partial(const FramePlanes *__restrict__ frames, Geom g, QParams qp,
                                                     uint8_t *__restrict__ records) {
  using S = QShape<KIND>;
  constexpr bool CHROMA = S::kChroma;
  constexpr int NT = 256;
  constexpr int NC = S::NPL * 2;   // (plane, half) combos
  constexpr int WPC = 4 / NC;      // waves per combo
  __shared__ __attribute__((aligned(16))) uint8_t lds[S::DATA_BYTES + S::WTILE_BYTES];
  __shared__ __attribute__((aligned(16))) uint8_t s_win[32];
  __shared__ uint16_t s_plist[S::NG];
  __shared__ int s_wcount[4];

  const int frame = blockIdx.z, chunk = blockIdx.x;
  const FramePlanes fp = frames[frame];
  const size_t lsel = ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * 3 + 1;
  const uint32_t *list = qp.lists + lsel * g.nblocks;
  const int nlist = (int)qp.counts[lsel];
  const uint8_t *winbase = qp.winbuf + ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * g.nblocks * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int combo = wave % NC, sub = wave / NC;
  const int my_pl = combo >> 1, my_half = combo & 1;

  int acc[kPHalf];
#pragma unroll
  for (int i = 0; i < kPHalf; ++i) acc[i] = 0;

  // Same software pipeline as k3_lag.  Areas that k3_lag<.., true> (which ran before on this
  // stream) deferred to the generic kernel carry kEntryDeferred and are skipped: one decision,
  // taken once.
  int li, li_end;
  list_slice(chunk, qp.nchunks_mix, nlist, li, li_end);
  auto entry_at = [&](int pos) -> uint32_t { return pos < li_end ? list[pos] : kEntryNone; };
  Stager<KIND, NT, true, false> st;
  uint32_t e_cur = entry_at(li), e_nxt = entry_at(li + 1);
  uint32_t wreg = 0;
  if (e_cur != kEntryNone && !(e_cur & kEntryDeferred)) {
    const int b0 = (int)(e_cur & kEntryIndex);
    if (tid < 8) wreg = reinterpret_cast<const uint32_t *>(winbase + (size_t)b0 * 32)[tid];
    st.fetch(fp, g, tid, b0);
  }
  __syncthreads();

  for (; e_cur != kEntryNone; e_cur = e_nxt, e_nxt = entry_at(li + 1)) {
    const int blk = (int)(e_cur & kEntryIndex);
    const bool skip = (e_cur & kEntryDeferred) != 0;
    if (!skip) {
      if (tid < 8) reinterpret_cast<uint32_t *>(s_win)[tid] = wreg;
      int lsum = 0;
      (void)st.store(fp, g, tid, blk, lds, lsum);
    }
    __syncthreads();
    ++li;
    if (e_nxt != kEntryNone && !(e_nxt & kEntryDeferred)) {
      const int bn = (int)(e_nxt & kEntryIndex);
      if (tid < 8) wreg = reinterpret_cast<const uint32_t *>(winbase + (size_t)bn * 32)[tid];
      st.fetch(fp, g, tid, bn);
    }
    if (skip) continue;
    st.build_wtile(tid, lds, s_win);
    __syncthreads();
    // ---- compact the partial groups of this area ----
    const uint32_t *w32b = reinterpret_cast<const uint32_t *>(lds + S::DATA_BYTES);
    bool part = false;
    if (tid < S::NG) {
      const int gr = tid / S::G, gg = tid - gr * S::G;
      bool full, empty;
      group_state<S::PITCH_DW>(w32b + gr * S::PITCH_DW + gg, full, empty);
      part = !full && !empty;
    }
    const unsigned long long bal = __ballot(part);
    if (lane == 0) s_wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < wave) off += s_wcount[w];
      total += s_wcount[w];
    }
    if (part) s_plist[off + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    __syncthreads();

    // ---- masked products over the compacted list ----
    const uint32_t *t32b = reinterpret_cast<const uint32_t *>(lds + my_pl * S::TILE_BYTES);
    const int nsteps = (total + 63) >> 6;
#pragma unroll 1
    for (int s = sub; s < nsteps; s += WPC) {
      const int e = s * 64 + lane;
      if (e < total) {
        const int gi = s_plist[e];
        const int row = gi / S::G, gg = gi - row * S::G;
        const uint32_t *t32 = t32b + (row + S::UP) * S::PITCH_DW + gg;
        uint32_t D[kNumLags];
        {
          const uint32_t c0 = t32[2], c1 = t32[3], c2 = t32[4];
          D[0] = c0;
          D[1] = alignbyte(c1, c0, 1);
          D[2] = alignbyte(c1, c0, 2);
          D[3] = alignbyte(c1, c0, 3);
          D[4] = c1;
          D[5] = alignbyte(c2, c1, 1);
          D[6] = alignbyte(c2, c1, 2);
#pragma unroll
          for (int dy = 1; dy <= 3; ++dy) {
            const uint32_t *rp = t32 + dy * S::PITCH_DW;
            const uint32_t e0 = rp[0], e1 = rp[1], e2 = rp[2], e3 = rp[3], e4 = rp[4];
            const int b = 7 + (dy - 1) * 13;
            D[b + 0] = alignbyte(e1, e0, 2);
            D[b + 1] = alignbyte(e1, e0, 3);
            D[b + 2] = e1;
            D[b + 3] = alignbyte(e2, e1, 1);
            D[b + 4] = alignbyte(e2, e1, 2);
            D[b + 5] = alignbyte(e2, e1, 3);
            D[b + 6] = e2;
            D[b + 7] = alignbyte(e3, e2, 1);
            D[b + 8] = alignbyte(e3, e2, 2);
            D[b + 9] = alignbyte(e3, e2, 3);
            D[b + 10] = e3;
            D[b + 11] = alignbyte(e4, e3, 1);
            D[b + 12] = alignbyte(e4, e3, 2);
          }
        }
        const uint32_t *w32 = w32b + row * S::PITCH_DW + gg;
        if (my_half == 0)
          partial_products<0, S::PITCH_DW>(acc, D, w32);
        else
          partial_products<1, S::PITCH_DW>(acc, D, w32);
      }
    }
    __syncthreads();  // tiles and list are rewritten by the next iteration
  }

  // ---- wave reduction + partial store ----
  __syncthreads();
  int *red = reinterpret_cast<int *>(lds);
  static_assert(4 * kPHalf * 4 <= S::DATA_BYTES, "reduction scratch must fit");
  {
    constexpr int CH = 27;
#pragma unroll
    for (int b0 = 0; b0 < kPHalf; b0 += CH) {
      int tmp[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) tmp[i] = (b0 + i < kPHalf) ? acc[b0 + i] : 0;
      wave_sum_all<CH>(tmp);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
          if (b0 + i < kPHalf) red[wave * kPHalf + b0 + i] = tmp[i];
      }
    }
  }
  __syncthreads();
  for (int pl = 0; pl < S::NPL; ++pl) {
    unsigned long long *out =
        reinterpret_cast<unsigned long long *>(qp.paracc) + ((size_t)frame * 3 + (CHROMA ? 1 + pl : 0)) * kPPart;
    for (int i = tid; i < kPPart; i += NT) {
      const int h = i / kPHalf, k = i - h * kPHalf;
      int v = 0;
      for (int w = 0; w < 4; ++w)
        if (w % NC == pl * 2 + h) v += red[w * kPHalf + k];
      if (v != 0) atomicAdd(&out[i], (unsigned long long)(long long)v);
    }
  }
}

// ---------------------------------------------------------------------------------
// k3q_generic: areas marked in `todo` (deferred because |d| > 127, EXT-but-flat, or MIX
// when the fast mixed path is off), plain int32.  Thread t owns up to two of the 350
// products:
//   (i, j):  sum_q W(q - c_i) d(q) d(q + c_j - c_i)        (i, y): ... d(q - c_i)
//   (i, L):  sum_p w(p) L(p) d(p + c_i);  (L,L), (L,y) likewise      [chroma]
// Adds straight into the record with int64 atomics (few areas), writes the block stats.
// grid = (chunks, nplanes, batch), block = 256.
// ---------------------------------------------------------------------------------
constexpr int kGW = kBlock + 12, kGH = kBlock + 6;  // d tile: cols -6..37, rows -3..34
__global__ __launch_bounds__(256) void k3q_generic(const FramePlanes *__restrict__ frames, Geom g, QParams qp,
                                                   uint8_t *__restrict__ records) {
  __shared__ int dt[kGH * kGW];        // d(q), x in -6..bw+5, y in -3..bh+2
  __shared__ uint8_t wt[kGH * kGW];    // w at the same positions
  __shared__ int lt[kBlock * kBlock];  // L(p) on the block proper
  __shared__ int red[4];
  const int c = blockIdx.y, frame = blockIdx.z;
  const int kind = c > 0 ? 1 : 0;
  const size_t lsel = ((size_t)frame * 2 + kind) * 3 + 2;
  const int nlist = (int)qp.counts[lsel];
  if ((int)blockIdx.x >= nlist) return;  // the common case: nothing deferred
  const uint32_t *list = qp.lists + lsel * g.nblocks;
  const FramePlanes fp = frames[frame];
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  const uint8_t *mask = rec + g.off_mask;
  const int sx = c ? g.xdec : 0, sy = c ? g.ydec : 0;
  const int pw = g.W >> sx, ph = g.H >> sy, bw = kBlock >> sx, bh = kBlock >> sy;
  const int TW = bw + 12, TH = bh + 6;
  const int nc = kQN + (c > 0);
  const int ntri = nc * (nc + 1) / 2, npairs = ntri + nc;
  int kindp[2], oa[2], ob[2], om[2], out_idx[2];
  bool have[2];
  auto tile_off = [&](int dx, int dy) { return (dy + 3) * TW + (dx + 6); };
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int p = threadIdx.x + s * 256;
    have[s] = p < npairs;
    int i = 0, j = 0;
    if (p < ntri) {
      int rem = p;
      while (rem >= nc - i) {
        rem -= nc - i;
        ++i;
      }
      j = i + rem;
      out_idx[s] = i * nc + j;
    } else {
      i = p - ntri;
      j = nc;  // y
      out_idx[s] = nc * nc + i;
    }
    if (!have[s]) i = j = 0;
    const bool iL = (c > 0 && i == kQN), jL = (c > 0 && j == kQN), jY = (j == nc);
    if (!iL && !jL) {
      kindp[s] = 0;  // q-centric: d(q) * d(q + delta), mask W(q - c_i)
      const int cxi = coord_x(i), cyi = coord_y(i);
      const int dx = (jY ? 0 : coord_x(j)) - cxi, dy = (jY ? 0 : coord_y(j)) - cyi;
      oa[s] = tile_off(0, 0);
      ob[s] = tile_off(dx, dy);
      om[s] = tile_off(-cxi, -cyi);
    } else if (!iL && jL) {
      kindp[s] = 1;  // p-centric: L(p) * d(p + c_i), mask w(p)
      oa[s] = tile_off(coord_x(i), coord_y(i));
      ob[s] = 0;
      om[s] = tile_off(0, 0);
    } else if (iL && jL) {
      kindp[s] = 2;  // L(p)^2
      oa[s] = ob[s] = 0;
      om[s] = tile_off(0, 0);
    } else {
      kindp[s] = 3;  // L(p) * d(p)
      oa[s] = tile_off(0, 0);
      ob[s] = 0;
      om[s] = tile_off(0, 0);
    }
  }
  unsigned long long *ar = reinterpret_cast<unsigned long long *>(rec + g.off_ar[c]);
  const uint8_t *sp = fp.src[c], *dp = fp.den[c];
  const uint32_t sst = fp.src_stride[c], dst = fp.den_stride[c];

  for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
    const int blk = (int)list[li];
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * bw, y_o = by * bh;
    int s_d = 0, s_d2 = 0, s_l = 0, s_n = 0;
    for (int idx = threadIdx.x; idx < TW * TH; idx += 256) {
      const int ty = idx / TW, tx = idx - ty * TW;
      const int X = x_o - 6 + tx, Y = y_o - 3 + ty;
      int d = 0, wv = 0;
      if (X >= 0 && X < pw && Y >= 0 && Y < ph) {
        const int s = load_px_rt(sp, sst, g.src_bps, g.src_shift, X, Y);
        d = s - load_px_rt(dp, dst, g.den_bps, g.den_shift, X, Y);
        wv = window_at(mask, g.nbw, g.nbh, bw, bh, pw, ph, X, Y);
        if (tx >= 6 && tx < 6 + bw && ty >= 3 && ty < 3 + bh) {  // block proper
          s_d += d;
          s_d2 += d * d;
          s_n += wv;
          if (c == 0) s_l += s;
        }
      }
      dt[idx] = d;
      wt[idx] = (uint8_t)wv;
    }
    if (c > 0) {
      for (int idx = threadIdx.x; idx < bw * bh; idx += 256) {
        const int y = idx / bw, x = idx - y * bw;
        const int X = x_o + x, Y = y_o + y;
        int L = 0;
        if (X < pw && Y < ph) {
          for (int dy = 0; dy < (1 << sy); ++dy)
            for (int dx = 0; dx < (1 << sx); ++dx) {
              const int lx = (X << sx) + dx, ly = (Y << sy) + dy;
              L += load_px_rt(fp.src[0], fp.src_stride[0], g.src_bps, g.src_shift, lx, ly) -
                   load_px_rt(fp.den[0], fp.den_stride[0], g.den_bps, g.den_shift, lx, ly);
            }
        }
        lt[idx] = L;
      }
    }
    {
      const int td = block_reduce_sum(s_d, red);
      const int td2 = block_reduce_sum(s_d2, red);
      const int tn = block_reduce_sum(s_n, red);
      const int tl = c == 0 ? block_reduce_sum(s_l, red) : 0;
      if (threadIdx.x == 0) {
        if (mask[blk]) {
          reinterpret_cast<int32_t *>(rec + g.off_sum_d[c])[blk] = td;
          reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[c])[blk] = (uint32_t)td2;
          if (c == 0) reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)tl;
        }
        if (tn) atomicAdd(&ar[nc * nc + nc], (unsigned long long)tn);
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (!have[s]) continue;
      int acc = 0;
      for (int y = 0; y < bh; ++y) {
        const int ro = y * TW;
        if (kindp[s] == 0) {
          for (int x = 0; x < bw; ++x)
            if (wt[om[s] + ro + x]) acc += dt[oa[s] + ro + x] * dt[ob[s] + ro + x];
        } else if (kindp[s] == 1) {
          for (int x = 0; x < bw; ++x)
            if (wt[om[s] + ro + x]) acc += lt[y * bw + x] * dt[oa[s] + ro + x];
        } else if (kindp[s] == 2) {
          for (int x = 0; x < bw; ++x)
            if (wt[om[s] + ro + x]) acc += lt[y * bw + x] * lt[y * bw + x];
        } else {
          for (int x = 0; x < bw; ++x)
            if (wt[om[s] + ro + x]) acc += lt[y * bw + x] * dt[oa[s] + ro + x];
        }
      }
      if (acc != 0) atomicAdd(&ar[out_idx[s]], (unsigned long long)(long long)acc);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// k3q_reduce: every chunk partial of one (frame, plane) -> record (upper triangle).
//   S[i][j] += G(c_j - c_i);  Sb[i] += G(-c_i);  chroma: S[i][L] += 4 Xa_i + Xb_i, ...
//   + the 324 masked products of k3_partial.
// grid = (nplanes, batch), block = 256: 4 lanes-groups of 64 split the chunk range.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3q_reduce(Geom g, QParams qp, uint8_t *__restrict__ records) {
  const int c = blockIdx.x, frame = blockIdx.y;
  const bool chroma = c > 0;
  const int nc = kQN + (chroma ? 1 : 0);
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  long long *ar = reinterpret_cast<long long *>(rec + g.off_ar[c]);
  const long long *lag = qp.lagacc + ((size_t)frame * 3 + c) * kQPart;
  const long long *par = qp.paracc + ((size_t)frame * 3 + c) * kPPart;
  for (int p = threadIdx.x; p < kQN * kQN; p += 256) {
    const int i = p / kQN, j = p % kQN;
    if (j < i) continue;
    ar[i * nc + j] += lag[lag_index(coord_x(j) - coord_x(i), coord_y(j) - coord_y(i))];
  }
  __syncthreads();  // the same S entries get the masked products below
  if (threadIdx.x < kQN) {
    const int i = threadIdx.x;
    ar[nc * nc + i] += lag[lag_index(-coord_x(i), -coord_y(i))];  // Sb[i]
    if (chroma) ar[i * nc + kQN] += lag[kNumLags + i];
    // masked products of anchor i
    const int h = p_in_half(0, i) ? 0 : 1;
    int idx = 0;
    for (int a = 0; a < i; ++a)
      if (p_in_half(h, a)) idx += (kQN - a) + 1;
    const long long *t = par + h * kPHalf + idx;
    int k = 0;
    for (int j = i; j < kQN; ++j) ar[i * nc + j] += t[k++];
    ar[nc * nc + i] += t[k];
  }
  if (chroma && threadIdx.x == 32) {
    ar[kQN * nc + kQN] += lag[kNumLags + 24];  // S[L][L]
    ar[nc * nc + kQN] += lag[kNumLags + 25];   // Sb[L]
  }
  if (threadIdx.x == 64) ar[nc * nc + nc] += lag[kQPart - 1];
}

}  // namespace g1s

// k3_fast.hip.h -- K3 for lag 3: exact AR normal-equation sums with v_dot4_i32_i8.
//
// add_block_observations (av1-grain diff/solver.rs == libaom noise_model.c) adds,
// per window sample p, the outer product of the vector
//     v(p) = [ d(p+c_0) .. d(p+c_23),  (L(p) for chroma),  d(p) ]
// (d = src8 - den8, L = co-located luma residual sum).  All of it is integer.
//
// Mapping for gfx950 (no MFMA on this path: BASELINE.json north_star):
//  * a lane owns a GROUP of 4 horizontally adjacent samples; operand k of the
//    group is 4 int8 in one VGPR, cut out of an LDS halo tile with
//    v_alignbyte_b32 (the tile is laid out so every shift is a compile-time
//    constant), so one v_dot4_i32_i8 performs 4 exact multiply-adds;
//  * the 324 (luma) products per group are split in two HALVES by left operand
//    (162 accumulators each, VGPR-resident for the whole kernel); a 256-thread
//    workgroup = 4 waves = {half 0, half 1} x {2 row halves} for luma, or
//    {Cb, Cr} x {half 0, half 1} for chroma (both planes share the L tile);
//  * the window of a block is a per-lane byte mask on the LEFT operand only;
//  * HBM -> LDS staging is software-pipelined: while the waves multiply block k,
//    the 8-sample (16-byte for u16) vector loads of the next flat block are
//    already in flight into registers; they are narrowed (>> (bd-8)), subtracted
//    and written with ds_write_b64 after the barrier;
//  * per-lane int32 accumulators live across all blocks of the workgroup's chunk
//    (<= 128 blocks, so int32 cannot overflow), are reduced across the wave once
//    at the end and stored as int32 partials; k3_fast_reduce adds the chunks
//    into the frame record in int64;
//  * a block whose |d| exceeds 127 anywhere in its tile is deferred to the
//    generic int32 kernel (k3_ar_generic with the defer list).
// The chroma luma-sum operand L (|L| <= 4*127) is split as L = 4a + b,
// a = L >> 2, b = L & 3, each int8; sums are recombined exactly in the reducer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"

namespace g1s {

constexpr int kFastLag = 3;
constexpr int kFastN = 24;
constexpr int kHalfPairs = 162;                 // luma products per half
constexpr int kHalfPairsChroma = 162 + 24 + 5;  // + (i,La),(i,Lb) for 12 anchors, + 5 L-only terms
constexpr int kMaxBlocksPerWG = 128;
constexpr int kFastThreads = 256;

// left operands (anchors) of half 0; the others belong to half 1
__host__ __device__ constexpr bool in_half(int half, int i) {
  // {0,3,4,7,8,11,12,15,16,19,20,23}: 25+22+21+18+17+14+13+10+9+6+5+2 = 162
  return (((i & 3) == 0 || (i & 3) == 3) ? 0 : 1) == half;
}

// Partial layout per (frame, kind, chunk): kind 0 = luma, 1 = Cb, 2 = Cr.
//   [half0 accumulators][half1 accumulators][nobs]
constexpr int kPartChroma = 2 * kHalfPairsChroma + 1;
constexpr int kPartStride = kPartChroma;  // ints per (frame, kind, chunk) slot

struct FastParams {
  int nchunks;
  int32_t *partials;    // [batch][3][nchunks][kPartStride]
  uint8_t *defer;       // [batch][2][nblocks] (luma, chroma) 1 = block left to the generic kernel
  uint32_t *defer_any;  // [batch] nonzero if the frame has deferred blocks
};

__device__ __forceinline__ int sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, int sh) {
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// all accumulators at once: 6 butterfly stages, each with N independent shuffles in flight
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

// One group step: all products of this wave's half.  V[k]: operand k (k = (cy+3)*7 + cx+3),
// VY: the sample itself, mask: window byte mask.
template <int HALF, bool CHROMA>
__device__ __forceinline__ void accumulate_half(int (&acc)[CHROMA ? kHalfPairsChroma : kHalfPairs],
                                                const uint32_t (&V)[kFastN], uint32_t VY, uint32_t La,
                                                uint32_t Lb, uint32_t mask) {
  int idx = 0;
#pragma unroll
  for (int i = 0; i < kFastN; ++i) {
    if (!in_half(HALF, i)) continue;
    const uint32_t mv = V[i] & mask;
#pragma unroll
    for (int j = i; j < kFastN; ++j) {
      acc[idx] = sdot4((int)mv, (int)V[j], acc[idx]);
      ++idx;
    }
    acc[idx] = sdot4((int)mv, (int)VY, acc[idx]);
    ++idx;
    if (CHROMA) {
      acc[idx] = sdot4((int)mv, (int)La, acc[idx]);
      ++idx;
      acc[idx] = sdot4((int)mv, (int)Lb, acc[idx]);
      ++idx;
    }
  }
  if (CHROMA && HALF == 0) {
    const uint32_t ma = La & mask, mb = Lb & mask;
    acc[idx + 0] = sdot4((int)ma, (int)La, acc[idx + 0]);
    acc[idx + 1] = sdot4((int)ma, (int)Lb, acc[idx + 1]);
    acc[idx + 2] = sdot4((int)mb, (int)Lb, acc[idx + 2]);
    acc[idx + 3] = sdot4((int)ma, (int)VY, acc[idx + 3]);
    acc[idx + 4] = sdot4((int)mb, (int)VY, acc[idx + 4]);
  }
}

// ---- 8 consecutive samples of a row -----------------------------------------
struct Px8 {
  uint4 raw;  // u16: 8 samples; u8: .x,.y hold 8 samples
  int state;  // 0 = all zero (outside the plane), 1 = raw is valid, 2 = edge segment: per-sample loads later
};
__device__ __forceinline__ Px8 fetch8(const uint8_t *base, uint32_t stride, int bps, bool vec_ok, int X0, int Y,
                                      int pw, int ph) {
  Px8 r;
  r.raw = make_uint4(0, 0, 0, 0);
  r.state = 0;
  if (Y < 0 || Y >= ph || X0 + 8 <= 0 || X0 >= pw) return r;
  if (X0 >= 0 && X0 + 8 <= pw && vec_ok) {
    gptr_u8 p = as_global(base) + (size_t)Y * stride + (size_t)X0 * bps;
    if (bps == 2) {
      r.raw = gload4((gptr_u4)p);
    } else {
      const uint2 v = gload2((gptr_u2)p);
      r.raw.x = v.x;
      r.raw.y = v.y;
    }
    r.state = 1;
  } else {
    r.state = 2;
  }
  return r;
}
// narrow to 8-bit samples (frame_into_u8: truncating shift); v[k] for k = 0..7
__device__ __forceinline__ void unpack8(const Px8 &p, const uint8_t *base, uint32_t stride, int bps, int shift,
                                        int X0, int Y, int pw, int (&v)[8]) {
  if (p.state == 1) {
    if (bps == 2) {
      const uint32_t w[4] = {p.raw.x, p.raw.y, p.raw.z, p.raw.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] = (int)(((w[k] & 0xffffu) >> shift) & 0xffu);
        v[2 * k + 1] = (int)(((w[k] >> 16) >> shift) & 0xffu);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = (int)((p.raw.x >> (8 * k)) & 0xffu);
        v[4 + k] = (int)((p.raw.y >> (8 * k)) & 0xffu);
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

// KIND: 0 = luma; 1 = chroma 4:2:0; 2 = chroma 4:2:2; 3 = chroma 4:4:4.
template <int KIND>
struct FastShape {
  static constexpr bool kChroma = KIND != 0;
  static constexpr int SX = (KIND == 1 || KIND == 2) ? 1 : 0;
  static constexpr int SY = (KIND == 1) ? 1 : 0;
  static constexpr int BW = kBlock >> SX, BH = kBlock >> SY;
  static constexpr int G = BW / 4;              // groups per block row
  static constexpr int ROWS_PER_STEP = 64 / G;  // 8 or 16
  static constexpr int STEPS = BH / ROWS_PER_STEP;
  static constexpr int PITCH_DW = 32 + G;  // conflict-free for the (group, row) lane map
  static constexpr int PITCH = PITCH_DW * 4;
  static constexpr int TH = BH + 3;
  static constexpr int SEGS = (BW + 16) / 8;  // 8-sample segments per tile row: x = -8 .. BW+7
  static constexpr int NPL = kChroma ? 2 : 1;
  static constexpr int NTILE = TH * SEGS * NPL;
  static constexpr int LCH = 8 >> SX;  // chroma samples per L item (8 luma samples wide)
  static constexpr int LSEGS = BW / LCH;
  static constexpr int NL = kChroma ? BH * LSEGS : 0;
  static constexpr int LROWS = 1 << SY;
  static constexpr int NITEMS = NTILE + NL;
  static constexpr int MAXIT = (NITEMS + kFastThreads - 1) / kFastThreads;
  static constexpr int SLOT = kChroma ? LROWS : 1;  // Px8 pairs per item slot
  static constexpr int TILE_BYTES = TH * PITCH;
  static constexpr int LTILE_BYTES = BH * PITCH;
  static constexpr int LDS_BYTES = NPL * TILE_BYTES + (kChroma ? 2 * LTILE_BYTES : 0);
  static constexpr int NACC = kChroma ? kHalfPairsChroma : kHalfPairs;
};

// LDS tile geometry: sample (x, y) of the block (x in -8..BW+7, y in -3..BH-1)
// lives at byte (y + 3) * PITCH + 8 + x: group g (x = 4g) is dword g + 2.
template <int KIND>
__global__ __launch_bounds__(kFastThreads, 2) void k3_fast(const FramePlanes *__restrict__ frames, Geom g,
                                                           FastParams fpm, uint8_t *__restrict__ records) {
  using S = FastShape<KIND>;
  constexpr bool CHROMA = S::kChroma;
  constexpr int NACC = S::NACC;
  __shared__ __attribute__((aligned(16))) uint8_t lds[S::LDS_BYTES];
  __shared__ int s_flag[2];
  __shared__ int s_stat[2][4][4];  // double-buffered by iteration parity

  const int frame = blockIdx.z;
  const int chunk = blockIdx.x;
  const FramePlanes fp = frames[frame];
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  const uint8_t *mask = rec + g.off_mask;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pw = g.W >> S::SX, ph = g.H >> S::SY;
  constexpr int bw = S::BW, bh = S::BH, lag = kFastLag;

  // wave roles
  const int half = wave & 1;
  const int plane_sel = CHROMA ? (wave >> 1) : 0;  // 0 = Cb, 1 = Cr
  const int row_half = CHROMA ? 0 : (wave >> 1);
  constexpr int my_steps = CHROMA ? S::STEPS : S::STEPS / 2;
  const int lg = lane % S::G, lr = lane / S::G;

  int acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  int nobs = 0;
  if (tid < 2) s_flag[tid] = 0;

  // ---- staging items of this thread (fixed for the whole kernel) ----
  // item < NTILE: (plane pl, tile row ty, segment sg); else: L item (chroma row, segment)
  Px8 ps[S::MAXIT][S::SLOT], pd[S::MAXIT][S::SLOT];
  auto item_fetch = [&](int blk) {
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * bw, y_o = by * bh;
#pragma unroll
    for (int k = 0; k < S::MAXIT; ++k) {
      const int it = tid + k * kFastThreads;
      if (it < S::NTILE) {
        const int pl = it / (S::TH * S::SEGS);
        const int r = it - pl * (S::TH * S::SEGS);
        const int ty = r / S::SEGS, sg = r - ty * S::SEGS;
        const int c = CHROMA ? 1 + pl : 0;
        const int X0 = x_o - 8 + 8 * sg, Y = y_o - lag + ty;
        ps[k][0] = fetch8(fp.src[c], fp.src_stride[c], g.src_bps, (g.vec_mask >> c) & 1, X0, Y, pw, ph);
        pd[k][0] = fetch8(fp.den[c], fp.den_stride[c], g.den_bps, (g.vec_mask >> (3 + c)) & 1, X0, Y, pw, ph);
      } else if (CHROMA && it < S::NITEMS) {
        const int r = it - S::NTILE;
        const int y = r / S::LSEGS, sg = r - y * S::LSEGS;
        const int X0 = (x_o + sg * S::LCH) << S::SX;  // luma coordinates
#pragma unroll
        for (int q = 0; q < S::LROWS; ++q) {
          const int Y = ((y_o + y) << S::SY) + q;
          ps[k][q] = fetch8(fp.src[0], fp.src_stride[0], g.src_bps, g.vec_mask & 1, X0, Y, g.W, g.H);
          pd[k][q] = fetch8(fp.den[0], fp.den_stride[0], g.den_bps, (g.vec_mask >> 3) & 1, X0, Y, g.W, g.H);
        }
      }
    }
  };
  // narrow, subtract, range-check and write the staged block into LDS
  auto item_store = [&](int blk, int &lsum) -> bool {
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * bw, y_o = by * bh;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < S::MAXIT; ++k) {
      const int it = tid + k * kFastThreads;
      if (it < S::NTILE) {
        const int pl = it / (S::TH * S::SEGS);
        const int r = it - pl * (S::TH * S::SEGS);
        const int ty = r / S::SEGS, sg = r - ty * S::SEGS;
        const int c = CHROMA ? 1 + pl : 0;
        const int X0 = x_o - 8 + 8 * sg, Y = y_o - lag + ty;
        int sv[8], dv[8];
        unpack8(ps[k][0], fp.src[c], fp.src_stride[c], g.src_bps, g.src_shift, X0, Y, pw, sv);
        unpack8(pd[k][0], fp.den[c], fp.den_stride[c], g.den_bps, g.den_shift, X0, Y, pw, dv);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int d = sv[q] - dv[q];
          bad |= (d > 127) | (d < -127);
          const uint32_t b = (uint32_t)d & 0xffu;
          if (q < 4) lo |= b << (8 * q); else hi |= b << (8 * (q - 4));
        }
        if (!CHROMA && sg >= 1 && sg <= 4 && ty >= lag) {  // block proper -> luma sum of the source
#pragma unroll
          for (int q = 0; q < 8; ++q) lsum += sv[q];
        }
        *reinterpret_cast<uint2 *>(lds + pl * S::TILE_BYTES + ty * S::PITCH + 8 * sg) = make_uint2(lo, hi);
      } else if (CHROMA && it < S::NITEMS) {
        const int r = it - S::NTILE;
        const int y = r / S::LSEGS, sg = r - y * S::LSEGS;
        const int X0 = (x_o + sg * S::LCH) << S::SX;
        int L[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) L[q] = 0;
#pragma unroll
        for (int q = 0; q < S::LROWS; ++q) {
          const int Y = ((y_o + y) << S::SY) + q;
          int sv[8], dv[8];
          unpack8(ps[k][q], fp.src[0], fp.src_stride[0], g.src_bps, g.src_shift, X0, Y, g.W, sv);
          unpack8(pd[k][q], fp.den[0], fp.den_stride[0], g.den_bps, g.den_shift, X0, Y, g.W, dv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int d = sv[e] - dv[e];
            bad |= (d > 127) | (d < -127);
            L[e >> S::SX] += d;
          }
        }
        uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
        for (int e = 0; e < S::LCH; ++e) {
          const uint32_t av = (uint32_t)(L[e] >> 2) & 0xffu, bv = (uint32_t)(L[e] & 3);
          if (e < 4) {
            a0 |= av << (8 * e);
            b0 |= bv << (8 * e);
          } else {
            a1 |= av << (8 * (e - 4));
            b1 |= bv << (8 * (e - 4));
          }
        }
        uint8_t *ta = lds + S::NPL * S::TILE_BYTES + y * S::PITCH + sg * S::LCH;
        uint8_t *tb = ta + S::LTILE_BYTES;
        if (S::LCH == 8) {
          *reinterpret_cast<uint2 *>(ta) = make_uint2(a0, a1);
          *reinterpret_cast<uint2 *>(tb) = make_uint2(b0, b1);
        } else {
          *reinterpret_cast<uint32_t *>(ta) = a0;
          *reinterpret_cast<uint32_t *>(tb) = b0;
        }
      }
    }
    return bad;
  };

  auto next_flat = [&](int blk) {
    while (blk < g.nblocks && !mask[blk]) blk += fpm.nchunks;
    return blk;
  };

  int cur = next_flat(chunk);
  if (cur < g.nblocks) item_fetch(cur);
  __syncthreads();

  int iter = 0;
  while (cur < g.nblocks) {
    const int blk = cur;
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * bw, y_o = by * bh;
    const int fl = iter & 1;
    ++iter;

    // ---------------- staged registers -> LDS ----------------
    int lsum = 0;
    const bool bad = item_store(blk, lsum);
    // window of this block + the next flat block: every mask byte is read BEFORE the
    // prefetch is issued (vmcnt retires in order: a later load waited on would drain it)
    const int y_start = (by > 0 && mask[(by - 1) * g.nbw + bx]) ? 0 : lag;
    const int x_start = (bx > 0 && mask[by * g.nbw + bx - 1]) ? 0 : lag;
    const int y_end = min(ph - y_o, bh);
    const int x_end = min(pw - x_o - lag, (bx + 1 < g.nbw && mask[by * g.nbw + bx + 1]) ? bw : (bw - lag));
    cur = next_flat(blk + fpm.nchunks);
    // ---------------- prefetch the next flat block: in flight during the products below ----------------
    if (cur < g.nblocks) item_fetch(cur);
    if (bad) s_flag[fl] = 1;
    if (!CHROMA) {
      lsum = wave_sum(lsum);
      if (lane == 0) s_stat[fl][wave][3] = lsum;
    }
    __syncthreads();
    const bool deferred = s_flag[fl] != 0;
    if (tid == 0) s_flag[fl ^ 1] = 0;
    if (deferred) {
      if (tid == 0) {
        fpm.defer[((size_t)frame * 2 + (CHROMA ? 1 : 0)) * g.nblocks + blk] = 1;
        fpm.defer_any[frame] = 1;
      }
      __syncthreads();
      continue;
    }

    if (tid == 0 && x_end > x_start && y_end > y_start) nobs += (x_end - x_start) * (y_end - y_start);

    // ------------------------------ products ------------------------------
    const uint32_t *t32 = reinterpret_cast<const uint32_t *>(lds + (CHROMA ? plane_sel * S::TILE_BYTES : 0));
    int sd = 0, sd2 = 0;
#pragma unroll 1
    for (int s = 0; s < my_steps; ++s) {
      const int row = (row_half * my_steps + s) * S::ROWS_PER_STEP + lr;  // sample row in the block
      uint32_t wm = 0;  // window byte mask of this group
      if (row >= y_start && row < y_end) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int x = 4 * lg + k;
          if (x >= x_start && x < x_end) wm |= 0xffu << (8 * k);
        }
      }
      uint32_t V[kFastN];
#pragma unroll
      for (int cy = -3; cy <= -1; ++cy) {
        const uint32_t *rp = t32 + (row + cy + 3) * S::PITCH_DW + lg + 1;
        const uint32_t d0 = rp[0], d1 = rp[1], d2 = rp[2];
        const int k0 = (cy + 3) * 7;
        V[k0 + 0] = alignbyte(d1, d0, 1);  // cx = -3
        V[k0 + 1] = alignbyte(d1, d0, 2);  // cx = -2
        V[k0 + 2] = alignbyte(d1, d0, 3);  // cx = -1
        V[k0 + 3] = d1;                    // cx = 0
        V[k0 + 4] = alignbyte(d2, d1, 1);  // cx = +1
        V[k0 + 5] = alignbyte(d2, d1, 2);  // cx = +2
        V[k0 + 6] = alignbyte(d2, d1, 3);  // cx = +3
      }
      uint32_t VY;
      {
        const uint32_t *rp = t32 + (row + 3) * S::PITCH_DW + lg + 1;
        const uint32_t d0 = rp[0], d1 = rp[1];
        V[21] = alignbyte(d1, d0, 1);
        V[22] = alignbyte(d1, d0, 2);
        V[23] = alignbyte(d1, d0, 3);
        VY = d1;
      }
      uint32_t La = 0, Lb = 0;
      if (CHROMA) {
        const uint32_t *ta = reinterpret_cast<const uint32_t *>(lds + S::NPL * S::TILE_BYTES);
        La = ta[row * S::PITCH_DW + lg];
        Lb = ta[S::LTILE_BYTES / 4 + row * S::PITCH_DW + lg];
      }
      if (half == 0) {
        accumulate_half<0, CHROMA>(acc, V, VY, La, Lb, wm);
        sd = sdot4((int)VY, 0x01010101, sd);
        sd2 = sdot4((int)VY, (int)VY, sd2);
      } else {
        accumulate_half<1, CHROMA>(acc, V, VY, La, Lb, wm);
      }
    }
    // block statistics: half-0 waves cover every row of their plane once
    if (half == 0) {
      sd = wave_sum(sd);
      sd2 = wave_sum(sd2);
      if (lane == 0) {
        s_stat[fl][wave][0] = sd;
        s_stat[fl][wave][1] = sd2;
      }
    }
    __syncthreads();
    if (tid == 0) {
      const int(*st)[4] = s_stat[fl];
      if (!CHROMA) {
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[0])[blk] = st[0][0] + st[2][0];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[0])[blk] = (uint32_t)(st[0][1] + st[2][1]);
        reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)(st[0][3] + st[1][3] + st[2][3] + st[3][3]);
      } else {
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[1])[blk] = st[0][0];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[1])[blk] = (uint32_t)st[0][1];
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[2])[blk] = st[2][0];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[2])[blk] = (uint32_t)st[2][1];
      }
    }
    // The tile is only rewritten after this barrier; s_stat[fl] is rewritten two
    // iterations later, i.e. after every thread passed the next iteration's barriers.
  }

  // ------------------------------ wave reduction + partial store ------------------------------
  __syncthreads();
  int *red = reinterpret_cast<int *>(lds);  // tiles are dead now
  static_assert(4 * NACC * 4 <= S::LDS_BYTES, "reduction scratch must fit in the tile LDS");
  // reduce in chunks of 27 accumulators: 27 independent ds_bpermute per stage
  {
    constexpr int CH = 27;
#pragma unroll
    for (int b0 = 0; b0 < NACC; b0 += CH) {
      int tmp[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) tmp[i] = (b0 + i < NACC) ? acc[b0 + i] : 0;
      wave_sum_all<CH>(tmp);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
          if (b0 + i < NACC) red[wave * NACC + b0 + i] = tmp[i];
      }
    }
  }
  __syncthreads();
  if (!CHROMA) {
    // two waves (row halves) hold the same half
    int32_t *out = fpm.partials + (((size_t)frame * 3 + 0) * fpm.nchunks + chunk) * kPartStride;
    for (int i = tid; i < 2 * NACC; i += kFastThreads) {
      const int h = i / NACC, k = i - h * NACC;
      out[i] = red[h * NACC + k] + red[(h + 2) * NACC + k];
    }
    if (tid == 0) out[2 * NACC] = nobs;
  } else {
    for (int pl = 0; pl < 2; ++pl) {
      int32_t *out = fpm.partials + (((size_t)frame * 3 + 1 + pl) * fpm.nchunks + chunk) * kPartStride;
      for (int i = tid; i < 2 * NACC; i += kFastThreads) out[i] = red[pl * 2 * NACC + i];
      if (tid == 0) out[2 * NACC] = nobs;
    }
  }
}

// ----------------------------------------------------------------------------
// Reducer: sums the chunk partials of one (frame, plane) into the record's
// int64 S / Sb / nobs (upper triangle; the host mirrors it).
// grid = (3 or 1, batch), block = 256.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3_fast_reduce(Geom g, FastParams fpm, uint8_t *__restrict__ records) {
  const int c = blockIdx.x, frame = blockIdx.y;
  const bool chroma = c > 0;
  const int NACC = chroma ? kHalfPairsChroma : kHalfPairs;
  const int nc = kFastN + (chroma ? 1 : 0);
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  long long *ar = reinterpret_cast<long long *>(rec + g.off_ar[c]);
  const int32_t *base = fpm.partials + ((size_t)frame * 3 + c) * fpm.nchunks * kPartStride;
  __shared__ long long tot[2 * kHalfPairsChroma + 1];
  for (int e = threadIdx.x; e < 2 * NACC + 1; e += 256) {
    long long s = 0;
    for (int ch = 0; ch < fpm.nchunks; ++ch) s += base[(size_t)ch * kPartStride + e];
    tot[e] = s;
  }
  __syncthreads();
  // scatter accumulator slots to (i, j); one thread per left operand
  if (threadIdx.x < kFastN) {
    const int i = threadIdx.x;
    const int h = in_half(0, i) ? 0 : 1;
    int idx = 0;
    for (int a = 0; a < i; ++a)
      if (in_half(h, a)) idx += (kFastN - a) + 1 + (chroma ? 2 : 0);
    const long long *t = tot + h * NACC + idx;
    int k = 0;
    for (int j = i; j < kFastN; ++j) ar[i * nc + j] += t[k++];
    ar[nc * nc + i] += t[k++];  // Sb[i]
    if (chroma) {
      const long long sa = t[k++], sb = t[k++];
      ar[i * nc + kFastN] += 4 * sa + sb;  // S[i][L]
    }
  }
  if (chroma && threadIdx.x == 32) {
    int idx = 0;
    for (int a = 0; a < kFastN; ++a)
      if (in_half(0, a)) idx += (kFastN - a) + 1 + 2;
    const long long *t = tot + idx;  // half 0 tail
    ar[kFastN * nc + kFastN] += 16 * t[0] + 8 * t[1] + t[2];  // S[L][L]
    ar[nc * nc + kFastN] += 4 * t[3] + t[4];                  // Sb[L]
  }
  if (threadIdx.x == 64) ar[nc * nc + nc] += tot[2 * NACC];  // nobs
}

}  // namespace g1s

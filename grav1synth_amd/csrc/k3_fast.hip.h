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
constexpr int kPartLuma = 2 * kHalfPairs + 1;
constexpr int kPartChroma = 2 * kHalfPairsChroma + 1;
constexpr int kPartStride = kPartChroma;  // ints per (frame, kind, chunk) slot

struct FastParams {
  int nchunks;
  int32_t *partials;  // [batch][3][nchunks][kPartStride]
  uint8_t *defer;     // [batch][2][nblocks] (luma, chroma) 1 = block left to the generic kernel
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

// LDS tile geometry: sample (x, y) of the block (x in -3..bw+2, y in -3..bh-1)
// lives at byte (y + 3) * pitch + 4 + x, so that group g (x = 4g) starts on a
// dword boundary and dword index of x=4g is g + 1.
template <bool CHROMA>
__global__ __launch_bounds__(kFastThreads, 2) void k3_fast(const FramePlanes *__restrict__ frames, Geom g,
                                                           FastParams fpm, uint8_t *__restrict__ records) {
  constexpr int NACC = CHROMA ? kHalfPairsChroma : kHalfPairs;
  constexpr int kMaxPitchDw = 40;
  constexpr int kTileBytes = (kBlock + 3) * kMaxPitchDw * 4;
  // chroma: tile 0 = Cb, tile 1 = Cr, then La, Lb (offset-0 only: no halo)
  __shared__ __attribute__((aligned(16))) uint8_t lds[CHROMA ? (2 * kTileBytes + 2 * kBlock * kMaxPitchDw * 4) : kTileBytes];
  __shared__ int s_flag[2];
  __shared__ int s_stat[2][4][4];  // double-buffered by iteration parity

  const int frame = blockIdx.z;
  const int chunk = blockIdx.x;
  const FramePlanes fp = frames[frame];
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  const uint8_t *mask = rec + g.off_mask;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

  const int sx = CHROMA ? g.xdec : 0, sy = CHROMA ? g.ydec : 0;
  const int pw = g.W >> sx, ph = g.H >> sy;
  const int bw = kBlock >> sx, bh = kBlock >> sy;
  const int G = bw >> 2;            // groups per block row (8 or 4)
  const int rows_per_step = 64 / G;  // 8 or 16
  const int steps_per_block = bh / rows_per_step;
  const int pitch_dw = 32 + G;  // conflict-free for the (group, row) lane map
  const int pitch = pitch_dw * 4;
  const int TW = bw + 6, TH = bh + 3;
  const int lag = kFastLag;

  // wave roles
  const int half = CHROMA ? (wave & 1) : (wave & 1);
  const int plane_sel = CHROMA ? (wave >> 1) : 0;  // 0 = Cb, 1 = Cr
  const int row_half = CHROMA ? 0 : (wave >> 1);
  const int my_steps = CHROMA ? steps_per_block : steps_per_block / 2;
  const int lg = lane % G, lr = lane / G;

  int acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  int nobs = 0;
  if (tid < 2) s_flag[tid] = 0;
  __syncthreads();

  int iter = 0;
  for (int blk = chunk; blk < g.nblocks; blk += fpm.nchunks) {
    if (!mask[blk]) continue;
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * bw, y_o = by * bh;
    const int fl = iter & 1;
    ++iter;

    // ------------------------------ stage the tile(s) ------------------------------
    int lsum = 0;
    bool bad = false;
    const int nplanes_here = CHROMA ? 2 : 1;
    for (int pl = 0; pl < nplanes_here; ++pl) {
      const int c = CHROMA ? 1 + pl : 0;
      const uint8_t *sp = fp.src[c], *dp = fp.den[c];
      const uint32_t sst = fp.src_stride[c], dst = fp.den_stride[c];
      uint8_t *tile = lds + pl * kTileBytes;
      for (int idx = tid; idx < TW * TH; idx += kFastThreads) {
        const int ty = idx / TW, tx = idx - ty * TW;
        const int X = x_o - lag + tx, Y = y_o - lag + ty;
        int d = 0;
        if (X >= 0 && X < pw && Y >= 0 && Y < ph) {
          const int s = load_px_rt(sp, sst, g.src_bps, g.src_shift, X, Y);
          d = s - load_px_rt(dp, dst, g.den_bps, g.den_shift, X, Y);
          if (!CHROMA && tx >= lag && tx < lag + bw && ty >= lag) lsum += s;
        }
        if (d > 127 || d < -127) bad = true;
        tile[ty * pitch + 1 + tx] = (uint8_t)(int8_t)d;  // x = tx - 3 -> byte 4 + x
      }
    }
    if (CHROMA) {
      uint8_t *ta = lds + 2 * kTileBytes, *tb = ta + kBlock * kMaxPitchDw * 4;
      for (int idx = tid; idx < bw * bh; idx += kFastThreads) {
        const int y = idx / bw, x = idx - y * bw;
        const int X = x_o + x, Y = y_o + y;
        int L = 0;
        if (X < pw && Y < ph) {
          for (int dy = 0; dy < (1 << sy); ++dy)
            for (int dx = 0; dx < (1 << sx); ++dx) {
              const int lx = (X << sx) + dx, ly = (Y << sy) + dy;
              const int dd = load_px_rt(fp.src[0], fp.src_stride[0], g.src_bps, g.src_shift, lx, ly) -
                             load_px_rt(fp.den[0], fp.den_stride[0], g.den_bps, g.den_shift, lx, ly);
              if (dd > 127 || dd < -127) bad = true;
              L += dd;
            }
        }
        ta[y * pitch + x] = (uint8_t)(int8_t)(L >> 2);
        tb[y * pitch + x] = (uint8_t)(L & 3);
      }
    }
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

    // ------------------------------ window of this block ------------------------------
    const int y_start = (by > 0 && mask[(by - 1) * g.nbw + bx]) ? 0 : lag;
    const int x_start = (bx > 0 && mask[by * g.nbw + bx - 1]) ? 0 : lag;
    const int y_end = min(ph - y_o, bh);
    const int x_end = min(pw - x_o - lag, (bx + 1 < g.nbw && mask[by * g.nbw + bx + 1]) ? bw : (bw - lag));
    if (tid == 0 && x_end > x_start && y_end > y_start) nobs += (x_end - x_start) * (y_end - y_start);

    // ------------------------------ products ------------------------------
    const uint8_t *tile = lds + (CHROMA ? plane_sel * kTileBytes : 0);
    int sd = 0, sd2 = 0;
    for (int s = 0; s < my_steps; ++s) {
      const int row = (row_half * my_steps + s) * rows_per_step + lr;  // sample row in the block
      // window byte mask of this group
      uint32_t wm = 0;
      if (row >= y_start && row < y_end) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int x = 4 * lg + k;
          if (x >= x_start && x < x_end) wm |= 0xffu << (8 * k);
        }
      }
      // operand rows: tile row (row + cy + 3), dwords lg .. lg+2
      uint32_t V[kFastN];
      const uint32_t *t32 = reinterpret_cast<const uint32_t *>(tile);
#pragma unroll
      for (int cy = -3; cy <= -1; ++cy) {
        const uint32_t *rp = t32 + (row + cy + 3) * pitch_dw + lg;
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
        const uint32_t *rp = t32 + (row + 3) * pitch_dw + lg;
        const uint32_t d0 = rp[0], d1 = rp[1];
        V[21] = alignbyte(d1, d0, 1);
        V[22] = alignbyte(d1, d0, 2);
        V[23] = alignbyte(d1, d0, 3);
        VY = d1;
      }
      uint32_t La = 0, Lb = 0;
      if (CHROMA) {
        const uint32_t *ta = reinterpret_cast<const uint32_t *>(lds + 2 * kTileBytes);
        const uint32_t *tb = ta + kBlock * kMaxPitchDw;
        La = ta[row * pitch_dw + lg];
        Lb = tb[row * pitch_dw + lg];
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
      if (!CHROMA) {
        const int(*st)[4] = s_stat[fl];
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[0])[blk] = st[0][0] + st[2][0];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[0])[blk] = (uint32_t)(st[0][1] + st[2][1]);
        reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)(st[0][3] + st[1][3] + st[2][3] + st[3][3]);
      } else {
        const int(*st)[4] = s_stat[fl];
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
  const int kinds = 3;
  (void)kinds;
  if (!CHROMA) {
    // luma: two waves (row halves) hold the same half -> both add into the slot via LDS
    __syncthreads();
    int *red = reinterpret_cast<int *>(lds);  // tiles are dead now
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int v = wave_sum(acc[i]);
      if (lane == 0) red[wave * NACC + i] = v;
    }
    __syncthreads();
    int32_t *out = fpm.partials + (((size_t)frame * 3 + 0) * fpm.nchunks + chunk) * kPartStride;
    for (int i = tid; i < 2 * NACC; i += kFastThreads) {
      const int h = i / NACC, k = i - h * NACC;
      out[i] = red[h * NACC + k] + red[(h + 2) * NACC + k];
    }
    if (tid == 0) out[2 * NACC] = nobs;
  } else {
    __syncthreads();
    int *red = reinterpret_cast<int *>(lds);
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int v = wave_sum(acc[i]);
      if (lane == 0) red[wave * NACC + i] = v;
    }
    __syncthreads();
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
  const int ns_scale = 1;  // L is stored pre-scaled by ns already (it is the SUM of luma residuals)
  (void)ns_scale;
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

// k3f.hip.h -- the fused accumulation pass: source / denoised planes -> residual tiles -> exact int8 SYRK.
//
// One kernel reads the 8/16-bit planes of the chunks that hold a flat block (and nothing else of the
// frame), narrows them (`(v >> (bd - 8)) as u8`, av1-grain util.rs frame_into_u8), forms d = src8 - den8 and
// the chroma regressor L (sum of the co-located luma residuals), takes the block statistics of
// get_block_mean / get_noise_var (exact integer sums), stages the 7 shifted int8 tile copies in LDS and
// multiplies them on the matrix cores (k3m.hip.h has the scheme: S = V V^T, v_mfma_i32_32x32x32_i8 with
// A = B).  No intermediate planes go through HBM: the pass reads (flat fraction) x (1 + halo) of the
// algorithmic bytes, the finder's luma-source pass (k1_moments) is the only other reader of the pixels.
//
// Workgroup = 4 waves, unit = 2 adjacent blocks of a block row (k3m_units): 36 KB of LDS, three to four
// workgroups to a CU, each in another phase.  Per unit:
//   staging   luma: waves 0-2 each take 6 row pairs, a lane one 8-sample word of both rows (the two rows
//             under a 4:2:0 chroma row: L needs no cross-lane traffic), source and denoised: four 16-byte
//             loads.  Chroma: every wave takes 10 (6) single rows of the two planes.  All loads are
//             requested ONE UNIT AHEAD into registers.  A residual (or L) outside int8 flags the blocks
//             whose tile holds it: they are left to the exact int32 kernel (k3_ar_generic, `only` list).
//   multiply  wave w takes rows 8w .. 8w+7 of every luma block and its share of the chroma steps.
// Two barriers per unit; accumulators (one 32x32 int32 per plane) stay in registers for the whole slice
// of the frame's unit list the workgroup walks; one partial system per workgroup (k3m_reduce).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k0.hip.h"
#include "k3m.hip.h"
#include "kernels.hip.h"

namespace g1s {

constexpr int kFWaves = 4, kFThreads = 64 * kFWaves;

struct FParams {
  FrameTable ft;
  uint8_t *records;
  uint8_t *only;          // [batch][2][nblocks]  flat blocks left to k3_ar_generic (zeroed per batch)
  uint32_t *only_any;     // [batch]
  const uint32_t *units;  // [batch][nunits][kMUnitDwords]  (k3m_units, windows without deferral)
  const uint32_t *unit_count;
  long long *partials;    // [batch][G][3][kMRec]
  int nunits;
};


// a raw 8-sample word -> packed 16-bit pairs 0x00vv00vv of the narrowed samples
__device__ __forceinline__ void f_narrow(const u32x4 &v, int bps, int shift, uint32_t (&h)[4]) {
  if (bps == 2) {
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
// the vector request of a word: inside the plane, rows 16-byte aligned; everything else reads as zero here
// (words that straddle the right plane edge or unaligned planes are fetched sample by sample at staging time)
__device__ __forceinline__ u32x4 f_request(const uint8_t *base, uint32_t stride, int bps, bool vec_ok, int X0, int Y, int pw, int ph) {
  u32x4 r = {0u, 0u, 0u, 0u};
  if (vec_ok && Y >= 0 && Y < ph && X0 >= 0 && X0 + 8 <= pw) {
    gptr_u8 p = as_global(base) + (size_t)Y * stride + (size_t)X0 * bps;
    if (bps == 2) {
      r = *(gptr_u4)p;
    } else {
      const u32x2 t = *(gptr_u2)p;
      r.x = t.x;
      r.y = t.y;
    }
  }
  return r;
}
__device__ __forceinline__ bool f_is_slow(bool vec_ok, int X0, int Y, int pw, int ph) {
  if (Y < 0 || Y >= ph || X0 + 8 <= 0 || X0 >= pw) return false;  // wholly outside: zeros
  return !vec_ok || X0 < 0 || X0 + 8 > pw;
}
__device__ __forceinline__ void f_slow_word(const uint8_t *base, uint32_t stride, int bps, int shift, int X0, int Y, int pw,
                                            uint32_t (&h)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) h[k] = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int X = X0 + k;
    if (X >= 0 && X < pw) h[k >> 1] |= (uint32_t)load_px_rt(base, stride, bps, shift, X, Y) << (16 * (k & 1));
  }
}

// blocks whose tile holds word wd of a row (WB words to a block; the tile reaches one word into its neighbours)
__device__ __forceinline__ void f_flag_blocks(int *flags, int wd, int WB) {
  const int b = wd / WB;
  if (b < kMUnitBlocks) flags[b] = 1;
  if (wd - b * WB <= 1 && b >= 1) flags[b - 1] = 1;
}

// ---------------------------------------------------------------------------------
// k3f_fused<CBW, CBH>: chroma block 32 >> xdec by 32 >> ydec (0, 0: luma only).
// grid = (G, 1, batch), block = 256, dynamic LDS = m_lds_bytes(CBW, CBH).
// ---------------------------------------------------------------------------------
template <int CBW, int CBH>
struct FShape {
  static constexpr bool CH = CBW != 0;
  static constexpr int CW_ = CH ? CBW : 16, CH_ = CH ? CBH : 16;
  // luma tile: rows -3 .. 31, samples -8 .. 71 of the chunk
  static constexpr int PY = m_pitch(32), WY = PY / 8, CSY = m_copy_stride(32, kBlock);
  static constexpr int PPJ = 64 / WY, PAIRS = (kBlock + 4) / 2;  // row pairs per luma job / per tile
  static_assert((PAIRS + PPJ - 1) / PPJ <= kFWaves - 1, "luma jobs: one per wave, the last wave has none");
  // chroma tiles: rows -3 .. CBH-1
  static constexpr int PC = m_pitch(CW_), WC = PC / 8, CSC = m_copy_stride(CW_, CH_);
  static constexpr int RC = CH_ + 3, RPW = 64 / WC;             // tile rows per plane / rows per wave and round
  static constexpr int CROUNDS = CH ? (2 * RC + kFWaves * RPW - 1) / (kFWaves * RPW) : 0;
};

// residuals of one row word: packed 16-bit pairs d16[4], running range (mx, mn), block statistics
__device__ __forceinline__ void f_residual(const uint32_t (&hs)[4], const uint32_t (&hv)[4], uint32_t (&d16)[4], uint32_t &mx, uint32_t &mn) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    d16[q] = pk_sub(hs[q], hv[q]);
    mx = pk_max(mx, d16[q]);
    mn = pk_min(mn, d16[q]);
  }
}

template <int CBW, int CBH>
__global__ __launch_bounds__(kFThreads, 3) void k3f_fused(Geom g, FParams fpar) {
  extern __shared__ __attribute__((aligned(16))) uint8_t m_smem[];
  using SH = FShape<CBW, CBH>;
  constexpr bool CH = SH::CH;
  constexpr int CW_ = SH::CW_, CH_ = SH::CH_, CROUNDS = SH::CROUNDS;
  constexpr int ZOFF = m_lds_tiles(CBW, CBH);
  constexpr int OFF_CB = m_tile_bytes(32, kBlock), OFF_CR = OFF_CB + m_tile_bytes(CW_, CH_);
  constexpr int OFF_L = OFF_CR + m_tile_bytes(CW_, CH_) + m_l_pad(CW_, CH_);
  __shared__ int s_sum[2][3][kMUnitBlocks][3];  // [unit parity][plane][block][sum d, sum d^2, sum src8 (luma)]
  __shared__ int s_bad[2][2][kMUnitBlocks];     // [unit parity][kind][block]

  const int frame = g.frame0 + (int)blockIdx.z;
  const int G = gridDim.x, wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t cnt = fpar.unit_count[frame];
  const uint32_t u0 = (uint32_t)((unsigned long long)cnt * wg / G), u1 = (uint32_t)((unsigned long long)cnt * (wg + 1) / G);
  const uint32_t *units = fpar.units + (size_t)frame * fpar.nunits * kMUnitDwords;
  const FramePlanes fp = fpar.ft.f[frame];
  uint8_t *rec = fpar.records + (size_t)frame * g.rec_size;
  const int sx = g.xdec, sy = g.ydec;
  const int cpw = g.W >> sx, cph = g.H >> sy;

  // ---- this lane's operand address inside a tile, per plane kind (k3m.hip.h) ----
  const int i = lane & 31, h = lane >> 5;
  int ea, ecxp, esp;
  m_entry(i, ea, ecxp, esp);
  const int base_luma = ecxp * SH::CSY + (3 - ea) * SH::PY + 16 * h;
  const int hoff_c = CW_ == 32 ? 16 * h : h * SH::PC;
  const int base_chroma = ecxp * SH::CSC + (3 - ea) * SH::PC + hoff_c;
  const int addr_cb = esp == 1 ? OFF_L + hoff_c : OFF_CB + base_chroma;
  const int addr_cr = esp == 1 ? OFF_L + hoff_c : OFF_CR + base_chroma;

  // ---- this lane's staging work ----
  // luma (waves 0 .. 2): pair ypair = rows 2 ypair - 4, 2 ypair - 3 of the block row = tile rows 2 ypair - 1, 2 ypair
  const int ypl = lane / SH::WY, ywd = lane - ypl * SH::WY;
  const int ypair = wave * SH::PPJ + ypl;
  const bool yon = ypl < SH::PPJ && ypair < SH::PAIRS;
  const int ytr0 = yon ? 2 * ypair - 1 : -9;
  // chroma: round k, tile row index rr = (4 k + wave) * RPW + lane / WC over the two planes' RC rows each
  const int cwd = lane % SH::WC;
  int cpl[CROUNDS > 0 ? CROUNDS : 1], ctr[CROUNDS > 0 ? CROUNDS : 1];  // plane (1, 2; 0: idle), tile row
#pragma unroll
  for (int k = 0; k < CROUNDS; ++k) {
    const int rr = (kFWaves * k + wave) * SH::RPW + lane / SH::WC;
    const bool on = lane / SH::WC < SH::RPW && rr < 2 * SH::RC;
    cpl[k] = on ? 1 + rr / SH::RC : 0;
    ctr[k] = on ? rr % SH::RC : 0;
  }

  v16i32 accY, accCb, accCr;
#pragma unroll
  for (int r = 0; r < 16; ++r) accY[r] = accCb[r] = accCr[r] = 0;
  long long nobs0 = 0, nobs1 = 0;

  if (tid < 4) reinterpret_cast<uint32_t *>(m_smem + ZOFF)[tid] = 0u;
  if (tid < 2 * 3 * kMUnitBlocks * 3) (&s_sum[0][0][0][0])[tid] = 0;
  if (tid < 2 * 2 * kMUnitBlocks) (&s_bad[0][0][0])[tid] = 0;

  // ---- the words of a unit, requested one unit ahead ----
  u32x4 ys_[2], yd_[2];                                         // luma: two rows, source and denoised
  u32x4 cs_[CROUNDS > 0 ? CROUNDS : 1], cd_[CROUNDS > 0 ? CROUNDS : 1];  // chroma: one row a round
  uint4 ent = make_uint4(0, 0, 0, 0);
  const bool vs0 = (g.vec_mask & 1) != 0, vd0 = (g.vec_mask & 8) != 0;
  auto request = [&](uint32_t u) {
    ent = *reinterpret_cast<const uint4 *>(units + (size_t)u * kMUnitDwords);
    const int bx0 = kMUnitBlocks * (int)(ent.x & 0xfffu), by = (int)((ent.x >> 12) & 0xfffu);
    {
      const int X0 = bx0 * 32 - 8 + 8 * ywd, Y0 = by * kBlock - 3 + ytr0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool on = ytr0 + r >= 0;  // (not: an idle lane, the row above the tile)
        ys_[r] = f_request(fp.src[0], fp.src_stride[0], g.src_bps, vs0 && on, X0, Y0 + r, g.W, g.H);
        yd_[r] = f_request(fp.den[0], fp.den_stride[0], g.den_bps, vd0 && on, X0, Y0 + r, g.W, g.H);
      }
    }
#pragma unroll
    for (int k = 0; k < CROUNDS; ++k) {
      const int c = cpl[k];
      const uint8_t *sp = c == 2 ? fp.src[2] : fp.src[1], *dp = c == 2 ? fp.den[2] : fp.den[1];
      const uint32_t sst = c == 2 ? fp.src_stride[2] : fp.src_stride[1], dst = c == 2 ? fp.den_stride[2] : fp.den_stride[1];
      const bool vs = ((g.vec_mask >> c) & 1) != 0 && c != 0, vd = ((g.vec_mask >> (3 + c)) & 1) != 0 && c != 0;
      const int X0 = bx0 * CW_ - 8 + 8 * cwd, Y = by * CH_ - 3 + ctr[k];
      cs_[k] = f_request(sp, sst, g.src_bps, vs, X0, Y, cpw, cph);
      cd_[k] = f_request(dp, dst, g.den_bps, vd, X0, Y, cpw, cph);
    }
  };
  if (u0 < u1) request(u0);

  for (uint32_t u = u0; u < u1; ++u) {
    const uint4 e0 = ent;
    const int par = (int)(u & 1u);
    const int bx0 = kMUnitBlocks * (int)(e0.x & 0xfffu), by = (int)((e0.x >> 12) & 0xfffu);
    const uint32_t fbits = e0.x >> 24;
    const uint32_t wins[4] = {e0.y & 0xffffu, e0.y >> 16, e0.z & 0xffffu, e0.z >> 16};  // luma block 0, 1; chroma block 0, 1
    __syncthreads();  // the previous unit's tiles are no longer read
    // ------------------------------- staging: luma -------------------------------
    if (wave < kFWaves - 1) {
      const int wd = ywd;
      const bool interior = wd >= 1 && wd <= SH::WY - 2;
      const int xw = 8 * (wd - 1), bq = (xw >> 5) & 1;  // sample of the chunk, block
      const int X0 = bx0 * 32 - 8 + 8 * wd, Y0 = by * kBlock - 3 + ytr0;
      const uint2 cm = interior ? m_colmask8(m_unpack(bq ? wins[1] : wins[0], g.lag), xw - 32 * bq) : make_uint2(0u, 0u);
      uint32_t mx = 0, mn = 0, keep16[4] = {0, 0, 0, 0};
      int sd = 0, sd2 = 0, ls = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int tr = ytr0 + r, Y = Y0 + r;
        uint32_t hs[4], hv[4], d16[4];
        f_narrow(ys_[r], g.src_bps, g.src_shift, hs);
        f_narrow(yd_[r], g.den_bps, g.den_shift, hv);
        if (tr >= 0 && f_is_slow(vs0, X0, Y, g.W, g.H)) f_slow_word(fp.src[0], fp.src_stride[0], g.src_bps, g.src_shift, X0, Y, g.W, hs);
        if (tr >= 0 && f_is_slow(vd0, X0, Y, g.W, g.H)) f_slow_word(fp.den[0], fp.den_stride[0], g.den_bps, g.den_shift, X0, Y, g.W, hv);
        f_residual(hs, hv, d16, mx, mn);
        if (tr >= 3 && interior) {  // the block proper: its statistics
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sd = pk_dot(d16[q], 0x00010001u, sd);
            sd2 = pk_dot(d16[q], d16[q], sd2);
          }
          ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[0], hs[1]), 0u, (uint32_t)ls);
          ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[2], hs[3]), 0u, (uint32_t)ls);
        }
        const uint32_t D0 = pk_bytes(d16[0], d16[1]), D1 = pk_bytes(d16[2], d16[3]);
        const uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)D1, 0x138, 0xf, 0xf, true);  // wave_shr:1
        const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)D0, 0x130, 0xf, 0xf, true);  // wave_shl:1
        if (tr >= 0 && (cm.x | cm.y)) m_write_copies(m_smem + tr * SH::PY + xw, SH::CSY, prev1, D0, D1, next0, cm);
        // ---- the chroma regressor L from the luma residuals (chroma resolution) ----
        if constexpr (CH) {
          uint32_t v[4] = {0, 0, 0, 0};
          bool have = false;
          int cy = 0;
          if (sy) {
            if (r == 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) keep16[q] = d16[q];
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = pk_add(keep16[q], d16[q]);
              have = tr >= 4;  // tile rows tr - 1, tr = block rows 2 cy, 2 cy + 1
              cy = (tr - 4) >> 1;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = d16[q];
            have = tr >= 3;
            cy = tr - 3;
          }
          if (have && interior) {
            const int xc = xw >> sx;  // first chroma sample under the word
            const uint2 lm = m_colmask8(m_unpack(bq ? wins[3] : wins[2], g.lag), xc - CW_ * bq);  // the co-located chroma block's window
            uint32_t lmx = 0, lmn = 0;
            if (sx) {
              const uint32_t p0 = ((uint32_t)pk_dot(v[0], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[1], 0x00010001u, 0) << 16);
              const uint32_t p1 = ((uint32_t)pk_dot(v[2], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[3], 0x00010001u, 0) << 16);
              lmx = pk_max(p0, p1);
              lmn = pk_min(p0, p1);
              if (lm.x) *reinterpret_cast<uint32_t *>(m_smem + OFF_L + cy * SH::PC + xc) = pk_bytes(p0, p1) & lm.x;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                lmx = pk_max(lmx, v[q]);
                lmn = pk_min(lmn, v[q]);
              }
              if (lm.x | lm.y)
                *reinterpret_cast<uint2 *>(m_smem + OFF_L + cy * SH::PC + xc) = make_uint2(pk_bytes(v[0], v[1]) & lm.x, pk_bytes(v[2], v[3]) & lm.y);
            }
            if (range_bad(lmx, lmn)) s_bad[par][1][bq] = 1;
          }
        }
      }
      if (yon) {
        if (interior && (sd | sd2 | ls)) {
          atomicAdd(&s_sum[par][0][bq][0], sd);
          atomicAdd(&s_sum[par][0][bq][1], sd2);
          atomicAdd(&s_sum[par][0][bq][2], ls);
        }
        if (range_bad(mx, mn)) f_flag_blocks(&s_bad[par][0][0], wd, 4);
      }
    }
    // ------------------------------- staging: chroma -------------------------------
#pragma unroll
    for (int k = 0; k < CROUNDS; ++k) {
      const int c = cpl[k], tr = ctr[k], wd = cwd;
      const uint8_t *sp = c == 2 ? fp.src[2] : fp.src[1], *dp = c == 2 ? fp.den[2] : fp.den[1];
      const uint32_t sst = c == 2 ? fp.src_stride[2] : fp.src_stride[1], dst = c == 2 ? fp.den_stride[2] : fp.den_stride[1];
      const bool vs = ((g.vec_mask >> c) & 1) != 0, vd = ((g.vec_mask >> (3 + c)) & 1) != 0;
      const bool interior = wd >= 1 && wd <= SH::WC - 2;
      const int xw = 8 * (wd - 1), bq = (xw / CW_) & 1;
      const int X0 = bx0 * CW_ - 8 + 8 * wd, Y = by * CH_ - 3 + tr;
      const uint2 cm = (interior && c) ? m_colmask8(m_unpack(bq ? wins[3] : wins[2], g.lag), xw - CW_ * bq) : make_uint2(0u, 0u);
      uint32_t hs[4], hv[4], d16[4], mx = 0, mn = 0;
      f_narrow(cs_[k], g.src_bps, g.src_shift, hs);
      f_narrow(cd_[k], g.den_bps, g.den_shift, hv);
      if (c && f_is_slow(vs, X0, Y, cpw, cph)) f_slow_word(sp, sst, g.src_bps, g.src_shift, X0, Y, cpw, hs);
      if (c && f_is_slow(vd, X0, Y, cpw, cph)) f_slow_word(dp, dst, g.den_bps, g.den_shift, X0, Y, cpw, hv);
      f_residual(hs, hv, d16, mx, mn);
      const uint32_t D0 = pk_bytes(d16[0], d16[1]), D1 = pk_bytes(d16[2], d16[3]);
      const uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)D1, 0x138, 0xf, 0xf, true);  // wave_shr:1
      const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)D0, 0x130, 0xf, 0xf, true);  // wave_shl:1
      if (cm.x | cm.y) m_write_copies(m_smem + (c == 2 ? OFF_CR : OFF_CB) + tr * SH::PC + xw, SH::CSC, prev1, D0, D1, next0, cm);
      if (c) {
        if (tr >= 3 && interior) {
          int sd = 0, sd2 = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sd = pk_dot(d16[q], 0x00010001u, sd);
            sd2 = pk_dot(d16[q], d16[q], sd2);
          }
          if (sd | sd2) {
            atomicAdd(&s_sum[par][c][bq][0], sd);
            atomicAdd(&s_sum[par][c][bq][1], sd2);
          }
        }
        if (range_bad(mx, mn)) f_flag_blocks(&s_bad[par][1][0], wd, CW_ / 8);
      }
    }
    if (u + 1 < u1) request(u + 1);
    __syncthreads();
    // ------------------------------- multiply -------------------------------
#pragma unroll
    for (int b = 0; b < kMUnitBlocks; ++b) {
      const MWin wy = m_unpack(wins[b], g.lag);
      if (wy.go) {
        if (__builtin_amdgcn_readfirstlane(s_bad[par][0][b])) {
          if (tid == 0) {
            fpar.only[((size_t)frame * 2 + 0) * g.nblocks + by * g.nbw + bx0 + b] = 1;
            fpar.only_any[frame] = 1u;
          }
        } else {
          constexpr int RPW = kBlock / kFWaves;
          m_rows_one<RPW, SH::PY>(accY, m_smem, base_luma + 32 * b + wave * RPW * SH::PY, wave * RPW, wy.ys, wy.ye, ZOFF);
          if (tid == 0) nobs0 += (long long)(wy.xe - wy.xs) * (wy.ye - wy.ys);
        }
      }
      if constexpr (CH) {
        const MWin wc = m_unpack(wins[kMUnitBlocks + b], g.lag);
        if (wc.go) {
          if (__builtin_amdgcn_readfirstlane(s_bad[par][1][b])) {
            if (tid == 0) {
              fpar.only[((size_t)frame * 2 + 1) * g.nblocks + by * g.nbw + bx0 + b] = 1;
              fpar.only_any[frame] = 1u;
            }
          } else {
            if constexpr (CW_ == 32) {
              constexpr int RPW = CH_ / kFWaves;
              const int o = CW_ * b + wave * RPW * SH::PC;
              m_rows_two<RPW, SH::PC>(accCb, accCr, m_smem, addr_cb + o, addr_cr + o, wave * RPW, wc.ys, wc.ye, ZOFF);
            } else {
              constexpr int SPW = CH_ / (2 * kFWaves);
              const int o = CW_ * b + 2 * wave * SPW * SH::PC;
              m_steps_two<SPW, SH::PC>(accCb, accCr, m_smem, addr_cb + o, addr_cr + o, wave * SPW, wc.ys, wc.ye, h, ZOFF);
            }
            if (tid == 0) nobs1 += (long long)(wc.xe - wc.xs) * (wc.ye - wc.ys);
          }
        }
      }
    }
    // ---- block statistics of this unit's flat blocks -> record; the other parity's flags and sums -> 0 ----
    if (tid < kMUnitBlocks) {
      const int b = tid, blk = by * g.nbw + bx0 + b;
      if ((fbits >> b) & 1u) {
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[0])[blk] = s_sum[par][0][b][0];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[0])[blk] = (uint32_t)s_sum[par][0][b][1];
        reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)s_sum[par][0][b][2];
        if (CH) {
          reinterpret_cast<int32_t *>(rec + g.off_sum_d[1])[blk] = s_sum[par][1][b][0];
          reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[1])[blk] = (uint32_t)s_sum[par][1][b][1];
          reinterpret_cast<int32_t *>(rec + g.off_sum_d[2])[blk] = s_sum[par][2][b][0];
          reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[2])[blk] = (uint32_t)s_sum[par][2][b][1];
        }
      }
    } else if (tid >= 64 && tid < 64 + 3 * kMUnitBlocks * 3) {
      (&s_sum[par ^ 1][0][0][0])[tid - 64] = 0;
    } else if (tid >= 128 && tid < 128 + 2 * kMUnitBlocks) {
      (&s_bad[par ^ 1][0][0])[tid - 128] = 0;
    }
  }

  // ---- the workgroup's partial systems: waves add into LDS (int64), one plain store per entry ----
  long long *s_S = reinterpret_cast<long long *>(m_smem);
  __syncthreads();
  for (int k = tid; k < 3 * kMRec; k += kFThreads) s_S[k] = 0;
  __syncthreads();
  auto flush = [&](const v16i32 &acc, int c, long long nobs) {
    const bool ch = c > 0;
    const int nc = g.n + (ch ? 1 : 0);
    const int ec = m_rec_index(i, g.lag, g.n, ch);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int er = m_rec_index(row, g.lag, g.n, ch);
      if (er < 0 || ec < 0 || er == nc) continue;
      int idx = -1;
      if (ec == nc) idx = nc * nc + er;
      else if (er <= ec) idx = er * nc + ec;
      if (idx >= 0 && acc[r] != 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(&s_S[c * kMRec + idx]), (unsigned long long)(long long)acc[r]);
    }
    if (tid == 0 && nobs) atomicAdd(reinterpret_cast<unsigned long long *>(&s_S[c * kMRec + nc * nc + nc]), (unsigned long long)nobs);
  };
  flush(accY, 0, nobs0);
  if (CH) {
    flush(accCb, 1, nobs1);
    flush(accCr, 2, nobs1);
  }
  __syncthreads();
  long long *out = fpar.partials + ((size_t)frame * G + wg) * 3 * kMRec;
  for (int k = tid; k < 3 * kMRec; k += kFThreads) out[k] = s_S[k];
}

}  // namespace g1s

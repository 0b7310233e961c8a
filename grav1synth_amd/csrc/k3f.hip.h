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
  const uint32_t *units;  // [batch][nunits][kMUnitDwords]  (k3m_units)
  const uint32_t *unit_count;
  long long *partials;    // [batch][G][3][kMRec]
  int32_t *ustats;        // [batch][nunits][kMStatInts]  per-unit block statistics + deferral bits (k3m_finish)
  int nunits;
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

// blocks whose tile holds word wd of a row (WB words to a block; the tile reaches one word into its neighbours)
__device__ __forceinline__ void f_flag_blocks(int *flags, int wd, int WB) {
  const int b = wd / WB;
  if (b < kMUnitBlocks) flags[b] = 1;
  if (wd - b * WB <= 1 && b >= 1) flags[b - 1] = 1;
}

// ---------------------------------------------------------------------------------
// k3f_fused<CBW, CBH, BPS>: chroma block 32 >> xdec by 32 >> ydec (0, 0: luma only).
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

__device__ __forceinline__ void f_residual(const uint32_t (&hs)[4], const uint32_t (&hv)[4], uint32_t (&d16)[4], uint32_t &mx, uint32_t &mn) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    d16[q] = pk_sub(hs[q], hv[q]);
    mx = pk_max(mx, d16[q]);
    mn = pk_min(mn, d16[q]);
  }
}

template <int CBW, int CBH, int BPS>
__global__ __launch_bounds__(kFThreads, 3) void k3f_fused(Geom g, FParams fpar) {
  extern __shared__ __attribute__((aligned(16))) uint8_t m_smem[];
  using SH = FShape<CBW, CBH>;
  constexpr bool CH = SH::CH;
  constexpr int CW_ = SH::CW_, CH_ = SH::CH_, CROUNDS = SH::CROUNDS, NCR = CROUNDS > 0 ? CROUNDS : 1;
  constexpr int ZOFF = m_lds_tiles(CBW, CBH);
  constexpr int OFF_CB = m_tile_bytes(32, kBlock), OFF_CR = OFF_CB + m_tile_bytes(CW_, CH_);
  constexpr int OFF_L = OFF_CR + m_tile_bytes(CW_, CH_) + m_l_pad(CW_, CH_);
  __shared__ int s_sum[2][kMStatInts];  // [unit parity][block * 7 + {luma: sum d, sum d^2, sum src8; Cb: sum d, sum d^2; Cr: ...}]
  __shared__ int s_bad[2][2][kMUnitBlocks];  // [unit parity][kind][block]

  const int frame = g.frame0 + (int)blockIdx.z;
  const int G = gridDim.x, wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t cnt = fpar.unit_count[frame];
  // Workgroup b runs on XCD b % 8 (observed; speed only).  An XCD owns a contiguous eighth of the frame's unit list and
  // deals it round-robin to its workgroups of this frame: horizontally adjacent units -- which share the 128-byte lines
  // their halo columns sit in -- are then read at about the same time through the same L2.
  const int xcd = wg & 7, jx = wg >> 3, nx = (G + 7 - xcd) >> 3;  // this workgroup's rank among the nx of its XCD
  const uint32_t c0 = (uint32_t)((unsigned long long)cnt * xcd / 8), c1 = (uint32_t)((unsigned long long)cnt * (xcd + 1) / 8);
  const uint32_t u0 = c0 + (uint32_t)jx, u1 = c1, ustep = (uint32_t)nx;
  const uint32_t *units = fpar.units + (size_t)frame * fpar.nunits * kMUnitDwords;
  int32_t *ustats = fpar.ustats + (size_t)frame * fpar.nunits * kMStatInts;
  const FramePlanes fp = fpar.ft.f[frame];
  const int sx = g.xdec, sy = g.ydec;
  const int cpw = g.W >> sx, cph = g.H >> sy;
  const int sbps = f_bps<BPS>(g.src_bps), dbps = f_bps<BPS>(g.den_bps);

  // ---- this lane's operand address inside a tile, per plane kind (k3m.hip.h) ----
  const int i = lane & 31, h = lane >> 5;
  int ea, ecxp, esp;
  m_entry(i, ea, ecxp, esp);
  const int base_luma = ecxp * SH::CSY + (3 - ea) * SH::PY + 16 * h + wave * (kBlock / kFWaves) * SH::PY;
  const int hoff_c = CW_ == 32 ? 16 * h : h * SH::PC;
  const int woff_c = wave * (CH_ / kFWaves) * SH::PC;  // this wave's first row (blocks 16 wide: first row pair)
  const int base_chroma = ecxp * SH::CSC + (3 - ea) * SH::PC + hoff_c + woff_c;
  const int addr_cb = esp == 1 ? OFF_L + hoff_c + woff_c : OFF_CB + base_chroma;
  const int addr_cr = esp == 1 ? OFF_L + hoff_c + woff_c : OFF_CR + base_chroma;

  // ---- this lane's staging work: offsets from the unit's origin (tile row 0, sample -8 of the chunk) ----
  // luma (waves 0 .. 2): pair ypair = tile rows 2 ypair - 1, 2 ypair (= block rows 2 ypair - 4, 2 ypair - 3)
  const int ypl = lane / SH::WY, ywd = lane - ypl * SH::WY;
  const int ypair = wave * SH::PPJ + ypl;
  const bool yon = wave < kFWaves - 1 && ypl < SH::PPJ && ypair < SH::PAIRS;
  const int ytr0 = yon ? 2 * ypair - 1 : -9;
  uint32_t yso[2], ydo[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    yso[r] = (uint32_t)max(ytr0 + r, 0) * fp.src_stride[0] + (uint32_t)(8 * ywd * sbps);
    ydo[r] = (uint32_t)max(ytr0 + r, 0) * fp.den_stride[0] + (uint32_t)(8 * ywd * dbps);
  }
  // chroma: round k, tile row index rr = (4 k + wave) * RPW + lane / WC over the two planes' RC rows each
  const int cwd = lane % SH::WC;
  int cpl[NCR], ctr[NCR];  // plane (1, 2; 0: idle), tile row
  uint32_t cso[NCR], cdo[NCR];
#pragma unroll
  for (int k = 0; k < CROUNDS; ++k) {
    const int rr = (kFWaves * k + wave) * SH::RPW + lane / SH::WC;
    const bool on = lane / SH::WC < SH::RPW && rr < 2 * SH::RC;
    cpl[k] = on ? 1 + rr / SH::RC : 0;
    ctr[k] = on ? rr % SH::RC : 0;
    cso[k] = (uint32_t)ctr[k] * (cpl[k] == 2 ? fp.src_stride[2] : fp.src_stride[1]) + (uint32_t)(8 * cwd * sbps);
    cdo[k] = (uint32_t)ctr[k] * (cpl[k] == 2 ? fp.den_stride[2] : fp.den_stride[1]) + (uint32_t)(8 * cwd * dbps);
  }
  // planes whose rows are 16-byte aligned take the vector loads; a chunk that reaches over the right plane edge
  // inside a word (W % 8 != 0) and unaligned planes go sample by sample
  const bool vec_all = (g.vec_mask & (CH ? 0x3f : 0x09)) == (CH ? 0x3f : 0x09);

  v16i32 accY, accCb, accCr;
#pragma unroll
  for (int r = 0; r < 16; ++r) accY[r] = accCb[r] = accCr[r] = 0;

  if (tid < 4) reinterpret_cast<uint32_t *>(m_smem + ZOFF)[tid] = 0u;
  if (tid < 2 * kMStatInts) (&s_sum[0][0])[tid] = 0;
  if (tid < 2 * 2 * kMUnitBlocks) (&s_bad[0][0][0])[tid] = 0;

  // ---- the words of a unit, requested one unit ahead ----
  u32x4 ys_[2], yd_[2];  // luma: two rows, source and denoised
  u32x4 cs_[NCR], cd_[NCR];  // chroma: one row a round
  uint4 ent = make_uint4(0, 0, 0, 0);
  auto request = [&](uint32_t u) {
    ent = *reinterpret_cast<const uint4 *>(units + (size_t)u * kMUnitDwords);
    const int bx0 = kMUnitBlocks * (int)(ent.x & 0xfffu), by = (int)((ent.x >> 12) & 0xfffu);
    const int X0y = bx0 * 32 - 8, Y0y = by * kBlock - 3, X0c = bx0 * CW_ - 8, Y0c = by * CH_ - 3;
    const bool slow = !vec_all || ((g.W & 7) != 0 && X0y + SH::PY > g.W) || (CH && (cpw & 7) != 0 && X0c + SH::PC > cpw);
    if (__builtin_expect(slow, 0)) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        ys_[r] = f_load_slow(fp.src[0], fp.src_stride[0], sbps, X0y + 8 * ywd, ytr0 + r >= 0 ? Y0y + ytr0 + r : -1, g.W, g.H);
        yd_[r] = f_load_slow(fp.den[0], fp.den_stride[0], dbps, X0y + 8 * ywd, ytr0 + r >= 0 ? Y0y + ytr0 + r : -1, g.W, g.H);
      }
#pragma unroll
      for (int k = 0; k < CROUNDS; ++k) {
        const int c = cpl[k];
        cs_[k] = f_load_slow(c == 2 ? fp.src[2] : fp.src[1], c == 2 ? fp.src_stride[2] : fp.src_stride[1], sbps, X0c + 8 * cwd,
                             c ? Y0c + ctr[k] : -1, cpw, cph);
        cd_[k] = f_load_slow(c == 2 ? fp.den[2] : fp.den[1], c == 2 ? fp.den_stride[2] : fp.den_stride[1], dbps, X0c + 8 * cwd,
                             c ? Y0c + ctr[k] : -1, cpw, cph);
      }
      return;
    }
    {
      // (pointers to the unit's origin: not dereferenced where the origin lies outside the plane)
      const uint8_t *sb = fp.src[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.src_stride[0] + (ptrdiff_t)X0y * sbps);
      const uint8_t *db = fp.den[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.den_stride[0] + (ptrdiff_t)X0y * dbps);
      const bool xok = X0y + 8 * ywd >= 0 && X0y + 8 * ywd + 8 <= g.W;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int Y = Y0y + ytr0 + r;
        const bool ok = xok && ytr0 + r >= 0 && Y >= 0 && Y < g.H;
        ys_[r] = f_load<BPS>(sb, yso[r], g.src_bps, ok);
        yd_[r] = f_load<BPS>(db, ydo[r], g.den_bps, ok);
      }
    }
    if constexpr (CH) {
      const ptrdiff_t o1s = (ptrdiff_t)Y0c * (ptrdiff_t)fp.src_stride[1] + (ptrdiff_t)X0c * sbps;
      const ptrdiff_t o2s = (ptrdiff_t)Y0c * (ptrdiff_t)fp.src_stride[2] + (ptrdiff_t)X0c * sbps;
      const ptrdiff_t o1d = (ptrdiff_t)Y0c * (ptrdiff_t)fp.den_stride[1] + (ptrdiff_t)X0c * dbps;
      const ptrdiff_t o2d = (ptrdiff_t)Y0c * (ptrdiff_t)fp.den_stride[2] + (ptrdiff_t)X0c * dbps;
      const bool xok = X0c + 8 * cwd >= 0 && X0c + 8 * cwd + 8 <= cpw;
#pragma unroll
      for (int k = 0; k < CROUNDS; ++k) {
        const int c = cpl[k], Y = Y0c + ctr[k];
        const bool ok = xok && c != 0 && Y >= 0 && Y < cph;
        cs_[k] = f_load<BPS>(c == 2 ? fp.src[2] + o2s : fp.src[1] + o1s, cso[k], g.src_bps, ok);
        cd_[k] = f_load<BPS>(c == 2 ? fp.den[2] + o2d : fp.den[1] + o1d, cdo[k], g.den_bps, ok);
      }
    }
  };
  if (u0 < u1) request(u0);

  for (uint32_t u = u0; u < u1; u += ustep) {
    const uint4 e0 = ent;
    const int par = (int)(((u - u0) / ustep) & 1u);
    const uint32_t wins[4] = {e0.y & 0xffffu, e0.y >> 16, e0.z & 0xffffu, e0.z >> 16};  // luma block 0, 1; chroma block 0, 1
    __syncthreads();  // the previous unit's tiles are no longer read
    // ------------------------------- staging: luma -------------------------------
    if (wave < kFWaves - 1) {
      const int wd = ywd;
      const bool interior = wd >= 1 && wd <= SH::WY - 2;
      const int xw = 8 * (wd - 1), bq = (xw >> 5) & 1;  // sample of the chunk, block
      const uint2 cm = interior ? m_colmask8(m_unpack(bq ? wins[1] : wins[0], g.lag), xw - 32 * bq) : make_uint2(0u, 0u);
      uint2 lm = make_uint2(0u, 0u);
      if (CH && interior) lm = m_colmask8(m_unpack(bq ? wins[3] : wins[2], g.lag), (xw >> sx) - CW_ * bq);  // the co-located chroma block's window
      uint32_t mx = 0, mn = 0, lmx = 0, lmn = 0, keep16[4] = {0, 0, 0, 0};
      int sd = 0, sd2 = 0, ls = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int tr = ytr0 + r;
        uint32_t hs[4], hv[4], d16[4];
        f_narrow<BPS>(ys_[r], g.src_bps, g.src_shift, hs);
        f_narrow<BPS>(yd_[r], g.den_bps, g.den_shift, hv);
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
          if (have && (lm.x | lm.y)) {
            const int xc = xw >> sx;  // first chroma sample under the word
            if (sx) {
              const uint32_t p0 = ((uint32_t)pk_dot(v[0], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[1], 0x00010001u, 0) << 16);
              const uint32_t p1 = ((uint32_t)pk_dot(v[2], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[3], 0x00010001u, 0) << 16);
              lmx = pk_max(lmx, pk_max(p0, p1));
              lmn = pk_min(lmn, pk_min(p0, p1));
              *reinterpret_cast<uint32_t *>(m_smem + OFF_L + cy * SH::PC + xc) = pk_bytes(p0, p1) & lm.x;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                lmx = pk_max(lmx, v[q]);
                lmn = pk_min(lmn, v[q]);
              }
              *reinterpret_cast<uint2 *>(m_smem + OFF_L + cy * SH::PC + xc) = make_uint2(pk_bytes(v[0], v[1]) & lm.x, pk_bytes(v[2], v[3]) & lm.y);
            }
          }
        }
      }
      if (interior && (sd | sd2 | ls)) {
        atomicAdd(&s_sum[par][7 * bq + 0], sd);
        atomicAdd(&s_sum[par][7 * bq + 1], sd2);
        atomicAdd(&s_sum[par][7 * bq + 2], ls);
      }
      if (yon && range_bad(mx, mn)) f_flag_blocks(&s_bad[par][0][0], wd, 4);
      if (CH && range_bad(lmx, lmn)) s_bad[par][1][bq] = 1;
    }
    // ------------------------------- staging: chroma -------------------------------
#pragma unroll
    for (int k = 0; k < CROUNDS; ++k) {
      const int c = cpl[k], tr = ctr[k], wd = cwd;
      const bool interior = wd >= 1 && wd <= SH::WC - 2;
      const int xw = 8 * (wd - 1), bq = (xw / CW_) & 1;
      const uint2 cm = (interior && c) ? m_colmask8(m_unpack(bq ? wins[3] : wins[2], g.lag), xw - CW_ * bq) : make_uint2(0u, 0u);
      uint32_t hs[4], hv[4], d16[4], mx = 0, mn = 0;
      f_narrow<BPS>(cs_[k], g.src_bps, g.src_shift, hs);
      f_narrow<BPS>(cd_[k], g.den_bps, g.den_shift, hv);
      f_residual(hs, hv, d16, mx, mn);
      const uint32_t D0 = pk_bytes(d16[0], d16[1]), D1 = pk_bytes(d16[2], d16[3]);
      const uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)D1, 0x138, 0xf, 0xf, true);  // wave_shr:1
      const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)D0, 0x130, 0xf, 0xf, true);  // wave_shl:1
      if (cm.x | cm.y) m_write_copies(m_smem + (c == 2 ? OFF_CR : OFF_CB) + tr * SH::PC + xw, SH::CSC, prev1, D0, D1, next0, cm);
      if (c && tr >= 3 && interior) {
        int sd = 0, sd2 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sd = pk_dot(d16[q], 0x00010001u, sd);
          sd2 = pk_dot(d16[q], d16[q], sd2);
        }
        if (sd | sd2) {
          atomicAdd(&s_sum[par][7 * bq + 1 + 2 * c], sd);
          atomicAdd(&s_sum[par][7 * bq + 2 + 2 * c], sd2);
        }
      }
      if (c && range_bad(mx, mn)) f_flag_blocks(&s_bad[par][1][0], wd, CW_ / 8);
    }
    if (u + ustep < u1) request(u + ustep);
    __syncthreads();
    // ------------------------------- multiply -------------------------------
    uint32_t defer = 0;
#pragma unroll
    for (int b = 0; b < kMUnitBlocks; ++b) {
      const MWin wy = m_unpack(wins[b], g.lag);
      if (wy.go) {
        if (__builtin_amdgcn_readfirstlane(s_bad[par][0][b])) {
          defer |= 1u << b;
        } else {
          constexpr int RPW = kBlock / kFWaves;
          m_rows_one<RPW, SH::PY>(accY, m_smem, base_luma + 32 * b, m_rowmask(wy.ys, wy.ye) >> (wave * RPW), ZOFF);
        }
      }
      if constexpr (CH) {
        const MWin wc = m_unpack(wins[kMUnitBlocks + b], g.lag);
        if (wc.go) {
          if (__builtin_amdgcn_readfirstlane(s_bad[par][1][b])) {
            defer |= 1u << (kMUnitBlocks + b);
          } else {
            constexpr int RPW = CH_ / kFWaves;
            const uint32_t rm = m_rowmask(wc.ys, wc.ye) >> (wave * RPW);
            if constexpr (CW_ == 32) m_rows_two<RPW, SH::PC>(accCb, accCr, m_smem, addr_cb + CW_ * b, addr_cr + CW_ * b, rm, ZOFF);
            else m_steps_two<RPW / 2, SH::PC>(accCb, accCr, m_smem, addr_cb + CW_ * b, addr_cr + CW_ * b, rm >> h, ZOFF);
          }
        }
      }
    }
    // ---- the unit's statistics record (k3m_finish scatters it); the other parity's flags and sums -> 0 ----
    if (tid < kMStatInts) {
      ustats[(size_t)u * kMStatInts + tid] = tid == 14 ? (int)defer : s_sum[par][tid];
    } else if (tid >= 64 && tid < 64 + kMStatInts) {
      s_sum[par ^ 1][tid - 64] = 0;
    } else if (tid >= 128 && tid < 128 + 2 * kMUnitBlocks) {
      (&s_bad[par ^ 1][0][0])[tid - 128] = 0;
    }
  }

  // ---- the workgroup's partial systems: waves add into LDS (int64), one plain store per entry ----
  long long *s_S = reinterpret_cast<long long *>(m_smem);
  __syncthreads();
  for (int k = tid; k < 3 * kMRec; k += kFThreads) s_S[k] = 0;
  __syncthreads();
  auto flush = [&](const v16i32 &acc, int c) {
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
  };
  flush(accY, 0);
  if (CH) {
    flush(accCb, 1);
    flush(accCr, 2);
  }
  __syncthreads();
  long long *out = fpar.partials + ((size_t)frame * G + wg) * 3 * kMRec;
  for (int k = tid; k < 3 * kMRec; k += kFThreads) out[k] = s_S[k];
}

}  // namespace g1s

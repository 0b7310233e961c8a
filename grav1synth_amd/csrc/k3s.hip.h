// k3s.hip.h -- the fused accumulation pass, second generation (G1S_K3=stream, the default).
//
// Same job as k3f.hip.h (source / denoised planes of the flat blocks' tiles -> residual tiles in LDS -> exact int8 SYRK on
// the matrix cores -> one partial system per workgroup and plane; block statistics, L plane and out-of-int8 deferrals on
// the way), same lists (k3m_units), same finisher (k3m_finish), same records.  What differs is how a unit moves through
// the workgroup:
//
//  * 16x16x64 MFMAs on operand PAIRS.  The 32 matrix rows (neighbour cx columns right, a rows up; av1-grain diff/solver.rs
//    add_block_observations) split into two 16-row operands P and Q that hold two values of `a` each, and the symmetric
//    32x32 product into three 16x16 products P P^T, P Q^T, Q Q^T (the fourth is the transpose of the second): 3 x 16
//    cycles of matrix pipe per 64 samples instead of 2 x 32.  A step is 64 samples: one row of both blocks of the unit
//    (blocks 32 wide; P = {a = 0, 2}, Q = {a = 1, 3}) or two rows (blocks 16 wide; P = {0, 1}, Q = {2, 3}).  Either way
//    the Q operand of a step IS the P operand of the step before -- the same 16 bytes of the same tile rows in the same
//    lanes -- so a step reads ONE operand from LDS (one conflict-free ds_read_b128) and renames the other: half the LDS
//    operand traffic of the 32x32 scheme, which read every tile row four times.
//  * Two tile buffers, ONE workgroup barrier per unit: the copies of unit k + 1 are written while unit k is multiplied.
//  * Both halo words of a row come from the neighbours' registers when the workgroup's run of units is contiguous (the
//    usual case): a row of a luma unit costs the memory pipe its own 128-byte line and nothing else.  The residuals of
//    unit k + 2 exist before the copies of unit k + 1 are written (which need their first dword).
//  * Rows 64 bytes apart in the copies (no halo columns: a copy is already shifted), so a luma tile set is 16 KB.
//
// Bit-exact against k3f.hip.h and the oracle (tests/test_gpu_parity.py::test_accumulation_modes_agree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "k0.hip.h"
#include "k3f.hip.h"
#include "k3m.hip.h"
#include "kernels.hip.h"

namespace g1s {

// ---- matrix rows: lane l of an operand holds 16 bytes of row i = l & 15 for the k-group l >> 4 --------------------------
// i -> (u, s): u = which of the operand's two `a` values, s = 0..6 the copy (cx = s - 3), s = 7 the chroma regressor L (u = 0
// of the operand that holds a = 0) or a spare row.  u = 0 sits on the lanes {0-3, 12-15}, u = 1 on {4-11}: with the copies
// 2 (mod 16) 16-byte slots apart, the 16 lanes a ds_read_b128 is served in ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}) then
// read 16 different slots (mod 16) -- the u = 0 rows of one k-group land on the even slots, the u = 1 rows of the next one
// on the odd ones -- whatever the row pitch (two-row steps: an even number of slots).
__device__ __forceinline__ void s_row(int i, int &u, int &s) {
  if (i < 4) { u = 0; s = i; }
  else if (i < 12) { u = 1; s = i - 4; }
  else { u = 0; s = i - 8; }
}
// operand (0 = P, 1 = Q), row i -> the row's `a`
template <bool TWO_ROW>
__device__ __forceinline__ int s_row_a(int op, int u) { return TWO_ROW ? 2 * op + u : 2 * u + op; }
// index in the record's (nc + 1)-vector (as m_rec_index): 0..n-1 neighbours, n = L (chroma), nc = the sample; -1 = no part of it
template <bool TWO_ROW>
__device__ __forceinline__ int s_rec_index(int op, int i, int lag, int n, bool chroma) {
  int u, s;
  s_row(i, u, s);
  const int a = s_row_a<TWO_ROW>(op, u);
  if (s == 7) return (a == 0 && chroma) ? n : -1;
  const int cx = s - 3;
  if (a == 0 && cx == 0) return n + (chroma ? 1 : 0);
  if (a == 0 && cx > 0) return -1;  // (not causal: the row exists because it is a = 1's or a = 2's row one step later)
  if (a > lag || cx < -lag || cx > lag) return -1;
  return (lag - a) * (2 * lag + 1) + (cx + lag);
}

// ---- tile geometry of a plane kind: block BW x BH, unit of two blocks -----------------------------------------------
__host__ __device__ constexpr int s_pitch(int BW) { return kMUnitBlocks * BW; }  // bytes of a copy's row: the unit's samples
__host__ __device__ constexpr int s_copy_stride(int BW, int BH) {
  int slots = ((BH + 3) * s_pitch(BW) + 15) / 16;  // rows -3 .. BH - 1
  while ((slots & 15) != 2) ++slots;
  return slots * 16;
}
// one buffer: luma launch 7 copies; chroma launch [Cb: 7 copies][L: an eighth "copy" of the Cb tile][Cr: 7 copies]
__host__ __device__ constexpr int s_buf_bytes(int CBW, int CBH, int PL) {
  return PL == 0 ? kMCopies * s_copy_stride(32, kBlock) : 15 * s_copy_stride(CBW, CBH);
}
__host__ __device__ constexpr int s_lds_bytes(int CBW, int CBH, int PL) { return 2 * s_buf_bytes(CBW, CBH, PL); }

typedef int v4i32s __attribute__((ext_vector_type(4)));

// NSTEP steps of RS rows from lane address a0 (the P operand of the first step), pitch P.  MASKED: `rm` bit j * RS says
// whether the lane's sample row of step j lies inside its block's window rows (a sample outside contributes nothing: both
// operands of the step are zeroed for the lane's k-group -- copies of them, the tile rows stay what they are for the next step)
template <int NSTEP, int RS, int P, bool MASKED>
__device__ __forceinline__ void s_multiply(v4i32s &aPP, v4i32s &aPQ, v4i32s &aQQ, const uint8_t *smem, int a0, uint32_t rm) {
  v4i32s q = m_lds16(smem, a0 - RS * P);
  constexpr int H = NSTEP > 4 ? 4 : NSTEP;  // operand reads in flight
#pragma unroll
  for (int j0 = 0; j0 < NSTEP; j0 += H) {
    v4i32s p[H];
#pragma unroll
    for (int j = 0; j < H; ++j) p[j] = m_lds16(smem, a0 + (j0 + j) * RS * P);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      if constexpr (MASKED) {
        const int m = __builtin_amdgcn_sbfe((int)rm, (j0 + j) * RS, 1);  // 0 or -1
        const v4i32s pm = p[j] & m, qm = q & m;
        aPP = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, pm, aPP, 0, 0, 0);
        aPQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, q, aPQ, 0, 0, 0);
        aQQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(qm, qm, aQQ, 0, 0, 0);
      } else {
        aPP = __builtin_amdgcn_mfma_i32_16x16x64_i8(p[j], p[j], aPP, 0, 0, 0);
        aPQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(p[j], q, aPQ, 0, 0, 0);
        aQQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(q, q, aQQ, 0, 0, 0);
      }
      q = p[j];
    }
  }
}

template <int CBW, int CBH>
struct SShape {
  static constexpr bool CH = CBW != 0;
  static constexpr int CW_ = CH ? CBW : 16, CH_ = CH ? CBH : 16;
  // luma: words -1 .. 8 of a row (8 samples each; the first and the last are the halo words), row pairs
  static constexpr int WY = 10, PAIRS = (kBlock + 4) / 2, PPJ = 64 / WY;
  static constexpr int PY = s_pitch(32), CSY = s_copy_stride(32, kBlock);
  static_assert(PPJ * (kFWaves - 1) >= PAIRS, "luma row pairs: three staging waves");
  // chroma: words -1 .. 2 CW / 8 of a row, one row a lane and round
  static constexpr int WC = kMUnitBlocks * CW_ / 8 + 2, RC = CH_ + 3, RPW = 64 / WC;
  static constexpr int PC = s_pitch(CW_), CSC = s_copy_stride(CW_, CH_);
  static constexpr int CROUNDS = CH ? (RC + 2 * RPW - 1) / (2 * RPW) : 0;
  static constexpr int NL = CH ? CH_ * kMUnitBlocks * CW_ / 8 : 0;  // 8-byte words of the unit's L tile
  static constexpr bool TWO_ROW_C = CW_ == 16;                      // chroma steps: two rows of 32 samples
};

// ---------------------------------------------------------------------------------
// k3s_fused<CBW, CBH, BPS, PL>: as k3f_fused (chroma block 32 >> xdec by 32 >> ydec, 0 0: luma only; PL = 0 the luma plane
// and L, PL = 1 the chroma planes).  grid = frames x workgroups per frame (1-D), block = 256, dynamic LDS = s_lds_bytes.
// ---------------------------------------------------------------------------------
template <int CBW, int CBH, int BPS, int PL>
__global__ __launch_bounds__(kFThreads, G1S_F_OCC) void k3s_fused(Geom g, FParams fpar) {
  extern __shared__ __attribute__((aligned(16))) uint8_t m_smem[];
  using SH = SShape<CBW, CBH>;
  constexpr bool CH = SH::CH;
  constexpr bool LUMA = PL == 0, CHROMA = PL == 1;
  static_assert(LUMA || CH, "the chroma launch needs chroma planes");
  constexpr int CW_ = SH::CW_, CH_ = SH::CH_, CROUNDS = CHROMA ? SH::CROUNDS : 0, NCR = CROUNDS > 0 ? CROUNDS : 1;
  constexpr int BUF = s_buf_bytes(CBW, CBH, PL);
  constexpr int OFF_CB = 0, OFF_L = 7 * SH::CSC, OFF_CR = 8 * SH::CSC;
  // per-unit side data, slot = unit & 3: written when the unit's residuals are formed (two iterations before it is
  // multiplied), read when it is multiplied, zeroed an iteration later
  //   block statistics, ONE 64-bit LDS atomic a lane: sum d^2 << 37 | sum src8 << 19 | sum (d + bias)
  __shared__ unsigned long long s_sum[4][3][kMUnitBlocks];
  __shared__ int s_bad[4][2][kMUnitBlocks];  // [slot][kind][block]: a residual (kind 1: or L) outside int8 in the block's tile
  __shared__ int s_ring[4][kMStatInts];      // statistics records on their way out (wave 3)
  __shared__ uint2 s_L[4][LUMA && SH::NL > 0 ? SH::NL : 1];  // luma launch: the L tile of a unit on its way to the L plane (wave 3)
  __shared__ uint4 s_ent[kMMaxUnits];

  const int G = fpar.wgs, frame = g.frame0 + (int)blockIdx.x % fpar.frames, wg = (int)blockIdx.x / fpar.frames;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nx = G, jx = wg;
  const uint32_t cnt_g = fpar.unit_count[2 * frame], cnt_p = fpar.unit_count[2 * frame + 1];
  const uint32_t ustride = fpar.deal ? 1u : (uint32_t)nx;
  auto share = [&](uint32_t cnt, uint32_t &first, int &n) {
    if (fpar.deal) {
      first = (uint32_t)((unsigned long long)cnt * (uint32_t)jx / (uint32_t)nx);
      n = (int)((uint32_t)((unsigned long long)cnt * (uint32_t)(jx + 1) / (uint32_t)nx) - first);
    } else {
      first = (uint32_t)jx;
      n = cnt > first ? (int)((cnt - first + (uint32_t)nx - 1) / (uint32_t)nx) : 0;
    }
  };
  uint32_t first_p, first_g;
  int n_p, n_g;
  share(cnt_p, first_p, n_p);
  share(cnt_g, first_g, n_g);
  auto upos = [&](int k) {
    return k < n_p ? (uint32_t)fpar.nunits - 1u - (first_p + (uint32_t)k * ustride) : first_g + (uint32_t)(k - n_p) * ustride;
  };
  const uint32_t *units = fpar.units + (size_t)frame * fpar.nunits * kMUnitDwords;
  int32_t *ustats = fpar.ustats + (size_t)frame * fpar.nunits * kMStatInts;
  uint8_t *lplane = fpar.lplane + (size_t)frame * fpar.lframe_bytes;
  const FramePlanes fp = fpar.ft.f[frame];
  constexpr int sx = CH && CBW == 16 ? 1 : 0, sy = CH && CBH == 16 ? 1 : 0;
  const int cpw = g.W >> sx, cph = g.H >> sy;
  const int sbps = f_bps<BPS>(g.src_bps), dbps = f_bps<BPS>(g.den_bps);
  const bool vec_all = (g.vec_mask & (LUMA ? 0x09 : 0x36)) == (LUMA ? 0x09 : 0x36);
  // halo words from the neighbours' registers: only where every word is a vector load (planes 16-byte aligned, no word
  // straddling the right plane edge)
  const bool reuse = LUMA && fpar.reuse && vec_all && (g.W & 7) == 0;

  // ---- this lane's operand address inside a buffer ----
  const int mi = lane & 15, mg = lane >> 4;
  int mu, ms;
  s_row(mi, mu, ms);
  // luma / chroma blocks 32 wide: one row a step, P = {a = 0, 2}; chroma blocks 16 wide: two rows a step, P = {a = 0, 1}
  constexpr bool TWO_ROW = CHROMA && SH::TWO_ROW_C;
  constexpr int MP = LUMA ? SH::PY : SH::PC, MCS = LUMA ? SH::CSY : SH::CSC, MBH = LUMA ? kBlock : CH_;
  const int m_rho = TWO_ROW ? (mg >> 1) : 0;                 // the lane's sample row inside a step
  const int m_blk = TWO_ROW ? (mg & 1) : (mg >> 1);          // the block its 16 samples belong to
  const int m_xo = TWO_ROW ? 16 * (mg & 1) : 16 * mg;
  const int m_ro = TWO_ROW ? m_rho - mu : -2 * mu;           // tile row of the lane's bytes, relative to the step's first sample row
  // this wave's share of the unit's steps
  constexpr int WPP = LUMA ? kFWaves : kFWaves / 2;          // waves per plane
  constexpr int RS = TWO_ROW ? 2 : 1;
  constexpr int NSTEP = MBH / RS / WPP;
  const int m_plane = LUMA ? 0 : 1 + (wave >> 1);            // plane this wave multiplies
  const int m_y0 = (LUMA ? wave : (wave & 1)) * NSTEP * RS;  // its first sample row
  int m_addr;
  {
    const int s_eff = (LUMA && ms == 7) ? 6 : ms;  // luma launch: the spare rows read what row s = 6 reads (a broadcast)
    int base = (CHROMA && m_plane == 2) ? OFF_CR : OFF_CB;
    int so = s_eff * MCS;
    if (CHROMA && ms == 7) { base = 0; so = OFF_L; }  // L: an eighth copy of the Cb tile (both planes' waves)
    m_addr = base + so + (m_y0 + 3 + m_ro) * MP + m_xo;
  }

  // ---- this lane's staging work ----
  // luma: pair ypair = tile rows 2 ypair - 1, 2 ypair (the two rows under a 4:2:0 chroma row); odd pairs take their two rows in
  // the opposite order, so that the 16 lanes a ds_write_b64 is served in write rows an odd number of rows apart (rows are 64
  // bytes: the two 64-byte halves of the 32 banks)
  const int ypl = lane / SH::WY, ywd = lane - ypl * SH::WY;
  const int ypair = wave * SH::PPJ + ypl;
  const bool y_wave = wave * SH::PPJ < SH::PAIRS;
  const bool yon = LUMA && ypl < SH::PPJ && ypair < SH::PAIRS;
  const int ytr0 = yon ? 2 * ypair - 1 : -9;
  const int yswap = ypair & 1;
  // chroma: waves 0, 1 stage Cb, waves 2, 3 Cr; round k, tile row (2 k + (wave & 1)) * RPW + lane / WC
  const int cwd = lane % SH::WC;
  const int cplane = 1 + (wave >> 1);
  const uint8_t *c_src = cplane == 2 ? fp.src[2] : fp.src[1], *c_den = cplane == 2 ? fp.den[2] : fp.den[1];
  const uint32_t c_sst = cplane == 2 ? fp.src_stride[2] : fp.src_stride[1], c_dst = cplane == 2 ? fp.den_stride[2] : fp.den_stride[1];
  int cpl[NCR], ctr[NCR];
#pragma unroll
  for (int k = 0; k < CROUNDS; ++k) {
    const int rr = (2 * k + (wave & 1)) * SH::RPW + lane / SH::WC;
    const bool on = lane / SH::WC < SH::RPW && rr < SH::RC;
    cpl[k] = on ? cplane : 0;
    ctr[k] = on ? rr : 0;
  }
  uint32_t cso[NCR], cdo[NCR];
#pragma unroll
  for (int k = 0; k < CROUNDS; ++k) {
    cso[k] = (uint32_t)ctr[k] * c_sst + (uint32_t)(8 * cwd * sbps);
    cdo[k] = (uint32_t)ctr[k] * c_dst + (uint32_t)(8 * cwd * dbps);
  }

  v4i32s aPP = {0, 0, 0, 0}, aPQ = {0, 0, 0, 0}, aQQ = {0, 0, 0, 0};

  // ---- this workgroup's units: their entries parked in LDS ----
  const int nmine = n_p + n_g;
  if (tid < nmine) {
    uint4 e = *reinterpret_cast<const uint4 *>(units + (size_t)upos(tid) * kMUnitDwords);
    // bit 31 of .x: the unit before this one in the workgroup's sequence is its left neighbour in the block row
    if (reuse && tid > 0) {
      const uint32_t a = units[(size_t)upos(tid - 1) * kMUnitDwords] & 0xffffffu, here = e.x & 0xffffffu;
      if ((a & 0xfff000u) == (here & 0xfff000u) && (a & 0xfffu) + 1u == (here & 0xfffu)) e.x |= 1u << 31;
    }
    e.w = CHROMA ? (uint32_t)ustats[(size_t)upos(tid) * kMStatInts + 14] : 0u;  // the luma launch's deferral bits
    s_ent[tid] = e;
  }
  if (tid < 4 * 3 * kMUnitBlocks) (&s_sum[0][0][0])[tid] = 0ull;
  if (tid < 4 * 2 * kMUnitBlocks) (&s_bad[0][0][0])[tid] = 0;
  __syncthreads();
  // is unit k's left / right neighbour the unit before / after it in the sequence
  auto adj_left = [&](int k) __attribute__((always_inline)) {
    return k >= 0 && k < nmine && (__builtin_amdgcn_readfirstlane(s_ent[k].x) >> 31) != 0;
  };

  // ---- registers of the pipeline ----
  u32x4 ys_[2], yd_[2];      // luma raw words in flight: two rows, source and denoised
  u32x4 cs_[NCR], cd_[NCR];  // chroma raw words in flight
  uint2 lraw = make_uint2(0u, 0u);  // chroma launch: this thread's word of the L tile, in flight
  uint32_t Dn[2][2] = {{0u, 0u}, {0u, 0u}}, Dc1[2][2] = {{0u, 0u}, {0u, 0u}}, DlastY[2] = {0u, 0u};  // luma residual words: of the unit just formed (k + 2), of unit k + 1; last dwords of unit k
  uint32_t Cn[NCR][2] = {}, Cc1[NCR][2] = {};                  // chroma residual words: unit k + 2, unit k + 1
  uint2 Ln = make_uint2(0u, 0u), L1 = make_uint2(0u, 0u);
  bool carry_y = false;
  const bool l_on = CHROMA && tid < SH::NL;
  constexpr int LWR = kMUnitBlocks * CW_ / 8;  // 8-byte words of an L tile row
  const int l_row = tid / LWR, l_wd = tid - l_row * LWR;

  const int dbg = fpar.dbg;
  auto request = [&](int k) __attribute__((always_inline)) {
    if (dbg & 1) return;
    const uint32_t ex = __builtin_amdgcn_readfirstlane(s_ent[k].x);
    const int bx0 = kMUnitBlocks * (int)(ex & 0xfffu), by = (int)((ex >> 12) & 0xfffu);
    const int X0y = bx0 * 32 - 8, Y0y = by * kBlock - 3, X0c = bx0 * CW_ - 8, Y0c = by * CH_ - 3;
    const bool aL = (ex >> 31) != 0, aR = adj_left(k + 1);  // the halo word is a neighbour's own word: not read
    if constexpr (CHROMA) {
      if (l_on) lraw = *reinterpret_cast<const uint2 *>(lplane + (size_t)(by * CH_ + l_row) * fpar.lpitch + bx0 * CW_ + 8 * l_wd);
    }
    const bool slow = !vec_all || (LUMA ? ((g.W & 7) != 0 && X0y + 8 * SH::WY > g.W) : ((cpw & 7) != 0 && X0c + 8 * SH::WC > cpw));
    if (__builtin_expect(slow, 0)) {
      if constexpr (LUMA) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int tr = ytr0 + (r ^ yswap);
          ys_[r] = f_load_slow(fp.src[0], fp.src_stride[0], sbps, X0y + 8 * ywd, yon && tr >= 0 ? Y0y + tr : -1, g.W, g.H);
          yd_[r] = f_load_slow(fp.den[0], fp.den_stride[0], dbps, X0y + 8 * ywd, yon && tr >= 0 ? Y0y + tr : -1, g.W, g.H);
        }
      }
#pragma unroll
      for (int q = 0; q < CROUNDS; ++q) {
        const int c = cpl[q];
        cs_[q] = f_load_slow(c_src, c_sst, sbps, X0c + 8 * cwd, c ? Y0c + ctr[q] : -1, cpw, cph);
        cd_[q] = f_load_slow(c_den, c_dst, dbps, X0c + 8 * cwd, c ? Y0c + ctr[q] : -1, cpw, cph);
      }
      return;
    }
    if (LUMA && y_wave) {
      const uint8_t *sb = fp.src[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.src_stride[0] + (ptrdiff_t)X0y * sbps);
      const uint8_t *db = fp.den[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.den_stride[0] + (ptrdiff_t)X0y * dbps);
      const bool inside = X0y >= 0 && X0y + 8 * SH::WY <= g.W && Y0y >= 0 && Y0y + kBlock + 3 <= g.H;
      const bool skip = (aL && ywd == 0) || (aR && ywd == SH::WY - 1);
      bool xok = inside || (X0y + 8 * ywd >= 0 && X0y + 8 * ywd + 8 <= g.W);
      xok = xok && !skip;
      int l_tr = ytr0, l_w = ywd, l_sw = yswap;
      asm volatile("" : "+v"(l_tr), "+v"(l_w), "+v"(l_sw));
      if (inside) {
        // every lane loads, no predicate; a halo lane whose word comes from a neighbour re-reads the own word next to it
        // (the same 128-byte line: no halo line is touched); tile row -1 and the lanes past the last pair read row 0
        if (aL && ywd == 0) l_w = 1;
        if (aR && ywd == SH::WY - 1) l_w = SH::WY - 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          ys_[r] = f_load<BPS>(sb, (uint32_t)max(l_tr + (r ^ l_sw), 0) * fp.src_stride[0] + (uint32_t)(8 * l_w * sbps), g.src_bps, true);
          yd_[r] = f_load<BPS>(db, (uint32_t)max(l_tr + (r ^ l_sw), 0) * fp.den_stride[0] + (uint32_t)(8 * l_w * dbps), g.den_bps, true);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int tr = ytr0 + (r ^ yswap), Y = Y0y + tr;
          const bool ok = xok && tr >= 0 && Y >= 0 && Y < g.H;
          ys_[r] = f_load<BPS>(sb, (uint32_t)max(l_tr + (r ^ l_sw), 0) * fp.src_stride[0] + (uint32_t)(8 * l_w * sbps), g.src_bps, ok);
          yd_[r] = f_load<BPS>(db, (uint32_t)max(l_tr + (r ^ l_sw), 0) * fp.den_stride[0] + (uint32_t)(8 * l_w * dbps), g.den_bps, ok);
        }
      }
    }
    if constexpr (CHROMA) {
      const uint8_t *sb = c_src + ((ptrdiff_t)Y0c * (ptrdiff_t)c_sst + (ptrdiff_t)X0c * sbps);
      const uint8_t *db = c_den + ((ptrdiff_t)Y0c * (ptrdiff_t)c_dst + (ptrdiff_t)X0c * dbps);
      const bool inside = X0c >= 0 && X0c + 8 * SH::WC <= cpw && Y0c >= 0 && Y0c + CH_ + 3 <= cph;
      const bool xok = inside || (X0c + 8 * cwd >= 0 && X0c + 8 * cwd + 8 <= cpw);
      if (inside) {
#pragma unroll
        for (int q = 0; q < CROUNDS; ++q) {
          cs_[q] = f_load<BPS>(sb, cso[q], g.src_bps, true);
          cd_[q] = f_load<BPS>(db, cdo[q], g.den_bps, true);
        }
      } else {
#pragma unroll
        for (int q = 0; q < CROUNDS; ++q) {
          const int Y = Y0c + ctr[q];
          const bool ok = xok && cpl[q] != 0 && Y >= 0 && Y < cph;
          cs_[q] = f_load<BPS>(sb, cso[q], g.src_bps, ok);
          cd_[q] = f_load<BPS>(db, cdo[q], g.den_bps, ok);
        }
      }
    }
  };

  const bool y_interior = ywd >= 1 && ywd <= SH::WY - 2, c_interior = cwd >= 1 && cwd <= SH::WC - 2;
  const int y_xw = 8 * (ywd - 1), y_bq = (y_xw >> 5) & 1;
  const int c_xw = 8 * (cwd - 1), c_bq = (c_xw / CW_) & 1;

  // raw words of unit k -> residual words (Dn / Cn / Ln), block statistics and out-of-int8 flags (slot k & 3), L -> its LDS tile
  auto form = [&](int k) __attribute__((always_inline)) {
    const int slot = k & 3;
    if (dbg & 2) {
      if (LUMA && y_wave) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          Dn[r][0] = ys_[r].x ^ yd_[r].x;
          Dn[r][1] = ys_[r].y ^ yd_[r].y;
        }
      }
#pragma unroll
      for (int q = 0; q < CROUNDS; ++q) {
        Cn[q][0] = cs_[q].x ^ cd_[q].x;
        Cn[q][1] = cs_[q].y ^ cd_[q].y;
      }
      return;
    }
    if (LUMA && y_wave) {
      uint32_t mx = 0, mn = 0, lmx = 0, lmn = 0, keep16[4] = {0, 0, 0, 0};
      int sd = 0, sd2 = 0, ls = 0;
      const uint32_t ex = __builtin_amdgcn_readfirstlane(s_ent[k].x);
      uint8_t *ltile = reinterpret_cast<uint8_t *>(&s_L[slot][0]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int tr = ytr0 + (r ^ yswap);  // (the row this register set holds)
        uint32_t hs[4], hv[4], d16[4];
        f_narrow<BPS>(ys_[r], g.src_bps, g.src_shift, hs);
        f_narrow<BPS>(yd_[r], g.den_bps, g.den_shift, hv);
        f_residual(hs, hv, d16, mx, mn);
        Dn[r][0] = pk_bytes(d16[0], d16[1]);
        Dn[r][1] = pk_bytes(d16[2], d16[3]);
        if (tr >= 3 && y_interior) {
          sd = __builtin_amdgcn_sdot4((int)Dn[r][0], 0x01010101, sd, false);
          sd = __builtin_amdgcn_sdot4((int)Dn[r][1], 0x01010101, sd, false);
          sd2 = __builtin_amdgcn_sdot4((int)Dn[r][0], (int)Dn[r][0], sd2, false);
          sd2 = __builtin_amdgcn_sdot4((int)Dn[r][1], (int)Dn[r][1], sd2, false);
          ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[0], hs[1]), 0u, (uint32_t)ls);
          ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[2], hs[3]), 0u, (uint32_t)ls);
        }
        if constexpr (CH) {
          // ---- the chroma regressor L (chroma resolution) -> the unit's L tile in LDS (wave 3 stores it) ----
          uint32_t v[4] = {0, 0, 0, 0};
          bool have = false;
          int cy = 0;
          if (sy) {  // the pair's two rows are one chroma row (either order)
            if (r == 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) keep16[q] = d16[q];
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = pk_add(keep16[q], d16[q]);
              have = ytr0 + 1 >= 4;
              cy = (ytr0 + 1 - 4) >> 1;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = d16[q];
            have = tr >= 3;
            cy = tr - 3;
          }
          if (have && y_interior) {
            uint8_t *lp = ltile + cy * (kMUnitBlocks * CW_) + (y_xw >> sx);
            if (sx) {
              const uint32_t p0 = ((uint32_t)pk_dot(v[0], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[1], 0x00010001u, 0) << 16);
              const uint32_t p1 = ((uint32_t)pk_dot(v[2], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[3], 0x00010001u, 0) << 16);
              lmx = pk_max(lmx, pk_max(p0, p1));
              lmn = pk_min(lmn, pk_min(p0, p1));
              *reinterpret_cast<uint32_t *>(lp) = pk_bytes(p0, p1);
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                lmx = pk_max(lmx, v[q]);
                lmn = pk_min(lmn, v[q]);
              }
              *reinterpret_cast<uint2 *>(lp) = make_uint2(pk_bytes(v[0], v[1]), pk_bytes(v[2], v[3]));
            }
          }
        }
      }
      if (y_interior && ytr0 >= 3 && !(dbg & 4))
        atomicAdd(&s_sum[slot][0][y_bq],
                  ((unsigned long long)(uint32_t)sd2 << 37) | ((unsigned long long)(uint32_t)ls << 19) | (unsigned long long)(uint32_t)(sd + kFBiasY));
      // a residual outside int8 flags the blocks whose tile holds it.  A halo word that is not read is a neighbour's own
      // word: the unit before carries the flag of its last word to this unit's first block, and this unit's first word
      // flags the second block of the unit before (whose slot is still open: it is multiplied an iteration after this)
      const bool aL = (ex >> 31) != 0;
      const bool badw = yon && range_bad(mx, mn);
      if (badw) f_flag_blocks(&s_bad[slot][0][0], ywd, 4);
      if (aL && carry_y) s_bad[slot][0][0] = 1;
      if (aL && badw && ywd == 1) s_bad[(k - 1) & 3][0][kMUnitBlocks - 1] = 1;
      carry_y = badw && ywd == SH::WY - 2;
      if (CH && y_interior && range_bad(lmx, lmn)) s_bad[slot][1][y_bq] = 1;
    }
    if constexpr (CHROMA) Ln = lraw;
#pragma unroll
    for (int q = 0; q < CROUNDS; ++q) {
      const int c = cpl[q];
      uint32_t hs[4], hv[4], d16[4], mx = 0, mn = 0;
      f_narrow<BPS>(cs_[q], g.src_bps, g.src_shift, hs);
      f_narrow<BPS>(cd_[q], g.den_bps, g.den_shift, hv);
      f_residual(hs, hv, d16, mx, mn);
      Cn[q][0] = pk_bytes(d16[0], d16[1]);
      Cn[q][1] = pk_bytes(d16[2], d16[3]);
      if (c && ctr[q] >= 3 && c_interior && !(dbg & 4)) {
        int sd = __builtin_amdgcn_sdot4((int)Cn[q][0], 0x01010101, 0, false);
        sd = __builtin_amdgcn_sdot4((int)Cn[q][1], 0x01010101, sd, false);
        int sd2 = __builtin_amdgcn_sdot4((int)Cn[q][0], (int)Cn[q][0], 0, false);
        sd2 = __builtin_amdgcn_sdot4((int)Cn[q][1], (int)Cn[q][1], sd2, false);
        atomicAdd(&s_sum[slot][c][c_bq], ((unsigned long long)(uint32_t)sd2 << 37) | (unsigned long long)(uint32_t)(sd + kFBiasC));
      }
      if (c && range_bad(mx, mn)) f_flag_blocks(&s_bad[slot][1][0], cwd, CW_ / 8);
    }
  };
  // the L tile of unit k (slot k & 3) -> the L plane (luma launch, wave 3)
  auto flush_l = [&](int k) __attribute__((always_inline)) {
    if constexpr (LUMA && CH) {
      if (wave == kFWaves - 1) {
        const uint32_t ex = __builtin_amdgcn_readfirstlane(s_ent[k].x);
        const int bx0 = kMUnitBlocks * (int)(ex & 0xfffu), by = (int)((ex >> 12) & 0xfffu);
        constexpr int LW = kMUnitBlocks * CW_ / 8;
#pragma unroll
        for (int w0 = 0; w0 < SH::NL; w0 += 64) {
          const int w = w0 + lane, row = w / LW, wd = w - row * LW;
          if (w < SH::NL)
            *reinterpret_cast<uint2 *>(lplane + (size_t)(by * CH_ + row) * fpar.lpitch + bx0 * CW_ + 8 * wd) = s_L[k & 3][w];
        }
      }
    }
  };
  // the copies of unit k1 = the unit in Dc1 / Cc1 / L1 -> buffer k1 & 1.  Its left neighbour's last dwords are DlastY when that
  // neighbour is the unit before it, its right neighbour's first dwords are in Dn when it is the unit after it.
  auto write_copies = [&](int k1) __attribute__((always_inline)) {
    if (dbg & 8) return;
    const uint4 e0 = s_ent[k1];
    const uint32_t ex0 = __builtin_amdgcn_readfirstlane(e0.x), ey = __builtin_amdgcn_readfirstlane(e0.y),
                   ez = __builtin_amdgcn_readfirstlane(e0.z);
    const bool plain = k1 < n_p;
    const uint32_t wins[4] = {ey & 0xffffu, ey >> 16, ez & 0xffffu, ez >> 16};
    uint8_t *buf = m_smem + (k1 & 1) * BUF;
    if (LUMA && y_wave) {
      const bool aL = (ex0 >> 31) != 0, aR = adj_left(k1 + 1);
      uint2 cm = make_uint2(0u, 0u);
      if (y_interior) cm = plain ? make_uint2(~0u, ~0u) : m_colmask8(m_unpack(y_bq ? wins[1] : wins[0], g.lag), y_xw - 32 * y_bq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int tr = ytr0 + (r ^ yswap);
        uint32_t d1h = Dc1[r][1], d0h = Dc1[r][0];
        if (aL) {  // the left halo lane's last dword: the last dword of the unit before, same row pair, word 8
          const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + SH::WY - 2), (int)DlastY[r]);
          if (ywd == 0) d1h = v;
        }
        if (aR) {  // the right halo lane's first dword: the first dword of the unit after, same row pair, word 1
          const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane - (SH::WY - 2)), (int)Dn[r][0]);
          if (ywd == SH::WY - 1) d0h = v;
        }
        const uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)d1h, 0x138, 0xf, 0xf, true);  // wave_shr:1
        const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)d0h, 0x130, 0xf, 0xf, true);  // wave_shl:1
        if (tr >= 0 && y_interior) {
          if (plain) m_write_copies<false>(buf + tr * SH::PY + y_xw, SH::CSY, prev1, Dc1[r][0], Dc1[r][1], next0, cm);
          else m_write_copies<true>(buf + tr * SH::PY + y_xw, SH::CSY, prev1, Dc1[r][0], Dc1[r][1], next0, cm);
        }
      }
    }
    if constexpr (CHROMA) {
      if (l_on) {  // the unit's L tile: this thread's word, under the window columns of its chroma block
        const int lb = (8 * l_wd / CW_) & 1;
        const uint2 lm = plain ? make_uint2(~0u, ~0u) : m_colmask8(m_unpack(lb ? wins[3] : wins[2], g.lag), 8 * l_wd - CW_ * lb);
        *reinterpret_cast<uint2 *>(buf + OFF_L + (l_row + 3) * SH::PC + 8 * l_wd) = make_uint2(L1.x & lm.x, L1.y & lm.y);
      }
    }
#pragma unroll
    for (int q = 0; q < CROUNDS; ++q) {
      const int c = cpl[q];
      uint2 cm = make_uint2(0u, 0u);
      if (c_interior && c) cm = plain ? make_uint2(~0u, ~0u) : m_colmask8(m_unpack(c_bq ? wins[3] : wins[2], g.lag), c_xw - CW_ * c_bq);
      const uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Cc1[q][1], 0x138, 0xf, 0xf, true);  // wave_shr:1
      const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Cc1[q][0], 0x130, 0xf, 0xf, true);  // wave_shl:1
      if (c_interior && c) {
        uint8_t *dst = buf + (c == 2 ? OFF_CR : OFF_CB) + ctr[q] * SH::PC + c_xw;
        if (plain) m_write_copies<false>(dst, SH::CSC, prev1, Cc1[q][0], Cc1[q][1], next0, cm);
        else m_write_copies<true>(dst, SH::CSC, prev1, Cc1[q][0], Cc1[q][1], next0, cm);
      }
    }
  };
  // unit k + 1 <- unit k + 2 (after the copies of k + 1 are out)
  auto advance = [&]() __attribute__((always_inline)) {
    if (LUMA && y_wave) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        DlastY[r] = Dc1[r][1];
        Dc1[r][0] = Dn[r][0];
        Dc1[r][1] = Dn[r][1];
      }
    }
#pragma unroll
    for (int q = 0; q < CROUNDS; ++q) {
      Cc1[q][0] = Cn[q][0];
      Cc1[q][1] = Cn[q][1];
    }
    if constexpr (CHROMA) L1 = Ln;
  };

  // ---- prologue: units 0 and 1 formed, the copies of unit 0 written, the words of unit 2 requested ----
  if (nmine > 0) {
    request(0);
    form(0);
    advance();  // (unit 0 -> the k + 1 registers)
    if (nmine > 1) {
      request(1);
      form(1);
      if (nmine > 2) request(2);
    }
    write_copies(0);
    advance();
  }
  __syncthreads();
  if (nmine > 0) flush_l(0);

  // units [k0, k1) of this workgroup's sequence; two calls (plain units, then the others) are ONE pipeline
  auto run = [&](auto plain_tag, int k0, int k1) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    for (int k = k0; k < k1; ++k) {
      const int slot = k & 3;
      const uint4 e0 = s_ent[k];
      const uint32_t ey = __builtin_amdgcn_readfirstlane(e0.y), ez = __builtin_amdgcn_readfirstlane(e0.z);
      const uint32_t ex0 = __builtin_amdgcn_readfirstlane(e0.x);
      const uint32_t fbits = PLAIN ? (1u << kMUnitBlocks) - 1u : (ex0 >> 24) & ((1u << kMUnitBlocks) - 1u);
      const uint32_t lbad = CHROMA ? __builtin_amdgcn_readfirstlane(e0.w) >> kMUnitBlocks : 0u;  // L outside int8 (luma launch)
      const uint32_t wins[4] = {ey & 0xffffu, ey >> 16, ez & 0xffffu, ez >> 16};
      // ---- the unit after next: its words have had an iteration to land ----
      if (k + 2 < nmine) {
        form(k + 2);
        if (k + 3 < nmine) request(k + 3);
      }
      // ---- the next unit's copies -> the other buffer (free since the barrier: unit k - 1 has been multiplied) ----
      if (k + 1 < nmine) {
        write_copies(k + 1);
        advance();
      }
      // ------------------------------- multiply unit k -------------------------------
      uint32_t defer = 0;
      {
        // A step spans both blocks of the unit: a residual outside int8 in either tile sends all of the unit's flat blocks to
        // the exact kernel (which redoes their statistics too)
        bool bad = false;
#pragma unroll
        for (int b = 0; b < kMUnitBlocks; ++b) {
          if (!((fbits >> b) & 1u)) continue;
          if (LUMA) {
            if (CH && __builtin_amdgcn_readfirstlane(s_bad[slot][1][b])) defer |= 1u << (kMUnitBlocks + b);  // L: the chroma launch's business
            if (__builtin_amdgcn_readfirstlane(s_bad[slot][0][b])) bad = true;
          } else {
            if (__builtin_amdgcn_readfirstlane(s_bad[slot][1][b]) || ((lbad >> b) & 1u)) bad = true;
          }
        }
        if (bad) {
          defer |= fbits << (LUMA ? 0 : kMUnitBlocks);
        } else if (!(dbg & 16)) {
          const uint8_t *buf = m_smem + (k & 1) * BUF;
          if constexpr (PLAIN) {
            s_multiply<NSTEP, RS, MP, false>(aPP, aPQ, aQQ, buf, m_addr, ~0u);
          } else {
            const MWin w0 = m_unpack(wins[LUMA ? 0 : kMUnitBlocks], g.lag), w1 = m_unpack(wins[LUMA ? 1 : kMUnitBlocks + 1], g.lag);
            if (w0.go || w1.go) {
              const uint32_t r0 = w0.go ? m_rowmask(w0.ys, w0.ye) : 0u, r1 = w1.go ? m_rowmask(w1.ys, w1.ye) : 0u;
              const uint32_t rm = (m_blk ? r1 : r0) >> (m_y0 + m_rho);
              s_multiply<NSTEP, RS, MP, true>(aPP, aPQ, aQQ, buf, m_addr, rm);
            }
          }
        }
      }
      // ---- wave 3: the unit's statistics record (a four-unit ring, stored four at a time), the next unit's L tile ----
      if (wave == kFWaves - 1 && !(dbg & 32)) {
        auto mine_entry = [](int t) {
          const int b = t >= 7 ? 1 : 0, e = t - 7 * b, c = e < 3 ? 0 : (e < 5 ? 1 : 2);
          return t < 14 ? (LUMA ? c == 0 : c != 0) : t == 14 + PL;
        };
        if (lane < kMStatInts && mine_entry(lane)) {
          const int b = lane >= 7 ? 1 : 0, e = lane - 7 * b, c = e < 3 ? 0 : (e < 5 ? 1 : 2), f = e < 3 ? e : (e - 3) & 1;
          int val = (int)defer;
          if (lane < 14) {
            const unsigned long long pk = s_sum[slot][c][b];
            const int bias = c == 0 ? kFBiasY * 16 * 4 : kFBiasC * CH_ * (CW_ / 8);
            if (f == 1) val = (int)(pk >> 37);
            else if (f == 2) val = (int)((pk >> 19) & 0x3ffffu);
            else val = (int)(c == 0 ? (pk & 0x7ffffu) : (pk & 0x1fffffffffull)) - bias;
          }
          s_ring[k & 3][lane] = val;
        }
        if ((k & 3) == 3 || k == nmine - 1) {
          const int first = k & ~3, u = lane >> 4, e = lane & 15;
          if (first + u <= k && mine_entry(e)) ustats[(size_t)upos(first + u) * kMStatInts + e] = s_ring[u][e];
        }
        if (k + 1 < nmine) flush_l(k + 1);
      }
      // (the sums and flags of the unit before this one: consumed an iteration ago, written again two iterations on)
      if (tid >= 64 && tid < 64 + 3 * kMUnitBlocks) (&s_sum[(k + 3) & 3][0][0])[tid - 64] = 0ull;
      else if (tid >= 128 && tid < 128 + 2 * kMUnitBlocks) (&s_bad[(k + 3) & 3][0][0])[tid - 128] = 0;
      __syncthreads();
    }
  };
  run(std::true_type{}, 0, n_p);
  run(std::false_type{}, n_p, nmine);

  // ---- the workgroup's partial systems: waves add into LDS (int64), one plain store per entry ----
  constexpr int NPL = LUMA ? 1 : 2, PL0 = LUMA ? 0 : 1;
  long long *s_S = reinterpret_cast<long long *>(m_smem);
  __syncthreads();
  for (int k = tid; k < NPL * kMRec; k += kFThreads) s_S[k] = 0;
  __syncthreads();
  {
    const bool ch = CHROMA;
    const int nc = g.n + (ch ? 1 : 0);
    long long *dst = s_S + (m_plane - PL0) * kMRec;
    auto add = [&](int er, int ec, int v, bool cross) {
      if (er < 0 || ec < 0 || v == 0) return;
      if (cross && er == nc) {  // (the sample itself sits in P: as a row of P Q^T it is the `b` entry of the Q row)
        const int t = er;
        er = ec;
        ec = t;
      }
      if (er == nc) return;
      int idx = -1;
      if (ec == nc) idx = nc * nc + er;
      else if (cross) idx = min(er, ec) * nc + max(er, ec);
      else if (er <= ec) idx = er * nc + ec;
      if (idx >= 0) atomicAdd(reinterpret_cast<unsigned long long *>(&dst[idx]), (unsigned long long)(long long)v);
    };
    const int cP = s_rec_index<TWO_ROW>(0, mi, g.lag, g.n, ch), cQ = s_rec_index<TWO_ROW>(1, mi, g.lag, g.n, ch);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * mg + r;
      const int rP = s_rec_index<TWO_ROW>(0, row, g.lag, g.n, ch), rQ = s_rec_index<TWO_ROW>(1, row, g.lag, g.n, ch);
      add(rP, cP, aPP[r], false);
      add(rP, cQ, aPQ[r], true);
      add(rQ, cQ, aQQ[r], false);
    }
  }
  __syncthreads();
  long long *out = fpar.partials + (((size_t)frame * G + wg) * 3 + PL0) * kMRec;
  for (int k = tid; k < NPL * kMRec; k += kFThreads) out[k] = s_S[k];
}

}  // namespace g1s

// k3s.hip.h -- the fused accumulation pass, second generation (G1S_K3=stream).
//
// Same job as k3s_params.hip.h (source / denoised planes of the flat blocks' tiles -> residual tiles in LDS -> exact int8 SYRK on
// the matrix cores -> one partial system per workgroup and plane; block statistics, L plane and out-of-int8 deferrals on
// the way), same lists (k3m_units), same finisher (k3m_finish), same records.  What differs is how a unit moves through
// the workgroup -- and how many instructions that takes: the SQ counters of k3f and of this kernel's first form
// (profiles/r03a_sq_counters_stream_vs_fused.txt) say the pass is bound by instruction issue (1 800 - 2 200 wave-instructions
// a unit, a third of them scalar: exec-mask bookkeeping of per-lane conditions, entry decoding, 64-bit address arithmetic),
// not by LDS, the matrix pipe or HBM.
//
//  * 16x16x64 MFMAs on operand PAIRS.  The 32 matrix rows (neighbour cx columns right, a rows up; av1-grain diff/solver.rs
//    add_block_observations) split into two 16-row operands P and Q that hold two values of `a` each, and the symmetric
//    32x32 product into three 16x16 products P P^T, P Q^T, Q Q^T (the fourth is the transpose of the second).  A step is
//    64 samples: one row of both blocks of the unit (blocks 32 wide; P = {a = 0, 2}, Q = {a = 1, 3}) or two rows (blocks
//    16 wide; P = {0, 1}, Q = {2, 3}).  Either way the Q operand of a step IS the P operand of the step before -- the same
//    16 bytes of the same tile rows in the same lanes -- so a step reads ONE operand from LDS (one conflict-free
//    ds_read_b128) and renames the other.
//  * Two tile buffers, ONE workgroup barrier per unit: the copies of unit k + 1 are written while unit k is multiplied.
//  * A FAST PATH for the usual unit -- inside the plane, its left and right neighbours the units before and after it in the
//    workgroup's run: four unpredicated loads at per-lane offsets that never change, both halo dwords of a row from the
//    neighbours' registers by DPP row rotations (a row of 16 lanes is two row pairs of 8 own words), residual arithmetic,
//    statistics and copies as straight-line code; every per-lane condition is a constant of the lane (dummy LDS targets
//    instead of exec masks).  Everything else -- plane borders, run breaks, unaligned planes, residuals outside int8 --
//    takes the general path (predicated loads, halo lanes, flags), a wave-uniform branch away.
//  * One LDS read per iteration for everything uniform a unit needs (a control word built when the entries are parked).
//
// Bit-exact against k3s_params.hip.h and the oracle (tests/test_gpu_parity.py::test_accumulation_modes_agree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pixel_helpers.hip.h"
#include "k3s_params.hip.h"
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
// tile row t = block row + 4: rows 1 .. BH + 3 are the block's rows -3 .. BH - 1, row 0 is a dummy (lanes with nothing to
// write write there instead of being switched off)
__host__ __device__ constexpr int s_pitch(int BW) { return kMUnitBlocks * BW; }  // bytes of a copy's row: the unit's samples
__host__ __device__ constexpr int s_copy_stride(int BW, int BH) {
  int slots = ((BH + 4) * s_pitch(BW) + 15) / 16;
  while ((slots & 15) != 2) ++slots;
  return slots * 16;
}
// one buffer: luma launch 7 copies; chroma launch [Cb: 7 copies][L: an eighth "copy" of the Cb tile][Cr: 7 copies]
// (PL = 2 / 3: the Cb / the Cr plane alone: [plane: 7 copies][L])
__host__ __device__ constexpr int s_buf_bytes(int CBW, int CBH, int PL) {
  return PL == 0 ? kMCopies * s_copy_stride(32, kBlock) : (PL == 1 ? 15 : 8) * s_copy_stride(CBW, CBH);
}
__host__ __device__ constexpr int s_lds_bytes(int CBW, int CBH, int PL) { return 2 * s_buf_bytes(CBW, CBH, PL); }

typedef int v4i32s __attribute__((ext_vector_type(4)));

// timing experiments (FParams::dbg, G1S_S_DBG): parts of the kernel left out -- only in builds with -DG1S_S_DBG_BUILD (a
// wave-uniform test costs the hot loop three instructions a time)
#ifdef G1S_S_DBG_BUILD
#define G1S_S_DBGBIT(bit) ((dbg & (bit)) != 0)
#else
#define G1S_S_DBGBIT(bit) false
#endif

// a raw 8-sample word -> packed 16-bit pairs 0x00vv00vv of the narrowed samples (f_narrow with plain 32-bit shifts: what the
// shift drags from the upper sample into the lower one is masked away)
template <int BPS>
__device__ __forceinline__ void s_narrow(const u32x4 &v, int rbps, int shift, uint32_t (&h)[4]) {
  if (f_bps<BPS>(rbps) == 2) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = (w[k] >> shift) & 0x00ff00ffu;
  } else {
    f_narrow<BPS>(v, rbps, shift, h);
  }
}
// residuals of a word; acc |= (d + 128) of every residual: some d outside -128 .. 127 <=> (acc & 0xff00ff00) != 0
__device__ __forceinline__ void s_residual(const uint32_t (&hs)[4], const uint32_t (&hv)[4], uint32_t (&d16)[4], uint32_t &acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) d16[q] = pk_sub(hs[q], hv[q]);
  acc = acc | pk_add(d16[0], 0x00800080u) | pk_add(d16[1], 0x00800080u);
  acc = acc | pk_add(d16[2], 0x00800080u) | pk_add(d16[3], 0x00800080u);
}
// NSTEP steps of RS rows from lane address a0 (the P operand of the first step), pitch P.
// The products of a step: P P^T, P Q^T, Q Q^T with Q = the P of the step before -- so Q Q^T of a step IS P P^T of the step
// before.  Where no step is masked (PLAIN: every row of both blocks inside its window) the wave therefore multiplies
//   aS += P P^T of steps 0 .. NSTEP - 2   (counts for P P^T and, as the Q Q^T of steps 1 .. NSTEP - 1, for Q Q^T)
//   aP += P P^T of the last step, aQ += Q Q^T of the first step (the operand read for the renaming), aX += P Q^T:
// 2 NSTEP + 1 products instead of 3 NSTEP; P P^T = aS + aP and Q Q^T = aS + aQ when the accumulators are written out.
// MASKED: `rm` bit j * RS says whether the lane's sample row of step j lies inside its block's window rows (a sample outside
// contributes nothing: both operands of the step are zeroed for the lane's k-group -- copies of them, the tile rows stay what
// they are for the next step); all three products of every step, into aP, aX, aQ.
template <int NSTEP, int RS, int P, bool MASKED, int READS>
__device__ __forceinline__ void s_multiply(v4i32s &aS, v4i32s &aP, v4i32s &aX, v4i32s &aQ, const uint8_t *smem, int a0, uint32_t rm) {
  v4i32s q = m_lds16(smem, a0 - RS * P);
  constexpr int H = NSTEP > READS ? READS : NSTEP;  // operand reads in flight
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
        aP = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, pm, aP, 0, 0, 0);
        aX = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, q, aX, 0, 0, 0);
        aQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(qm, qm, aQ, 0, 0, 0);
      } else {
        if (j0 + j == 0) aQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(q, q, aQ, 0, 0, 0);
        if (j0 + j == NSTEP - 1) aP = __builtin_amdgcn_mfma_i32_16x16x64_i8(p[j], p[j], aP, 0, 0, 0);
        else aS = __builtin_amdgcn_mfma_i32_16x16x64_i8(p[j], p[j], aS, 0, 0, 0);
        aX = __builtin_amdgcn_mfma_i32_16x16x64_i8(p[j], q, aX, 0, 0, 0);
      }
      q = p[j];
    }
  }
}

#ifndef G1S_S_READS
#define G1S_S_READS 2  // operand reads in flight in the multiplies (4: the 4:2:0 luma launch spills)
#endif
template <int CBW, int CBH>
struct SShape {
  static constexpr bool CH = CBW != 0;
  static constexpr int CW_ = CH ? CBW : 16, CH_ = CH ? CBH : 16;
  // luma: words 0 .. 9 of a row (8 samples each; 0 and 9 are the halo words), row pairs: pair p = tile rows 2 p, 2 p + 1
  static constexpr int WY = 10, PAIRS = (kBlock + 4) / 2, PPJ = 6;
  static constexpr int PY = s_pitch(32), CSY = s_copy_stride(32, kBlock);
  static_assert(PPJ * (kFWaves - 1) == PAIRS, "luma row pairs: three staging waves of six");
  // chroma: words 0 .. 2 CW / 8 + 1 of a row, one row a lane and round
  static constexpr int WC = kMUnitBlocks * CW_ / 8 + 2, RC = CH_ + 3, RPW = 64 / WC;
  static constexpr int PC = s_pitch(CW_), CSC = s_copy_stride(CW_, CH_);
  static constexpr int CROUNDS = CH ? (RC + 2 * RPW - 1) / (2 * RPW) : 0;
  static constexpr int NL = CH ? CH_ * kMUnitBlocks * CW_ / 8 : 0;  // 8-byte words of the unit's L tile
  static constexpr bool TWO_ROW_C = CW_ == 16;                      // chroma steps: two rows of 32 samples
};

// bits (block) of the blocks whose tile holds word wd of a row (WB words to a block, word 0 = the left halo word)
__device__ __forceinline__ uint32_t s_flag_bits(int wd, int WB) {
  uint32_t r = 0;
  const int b = wd / WB;
  if (b < kMUnitBlocks) r |= 1u << b;
  if (wd - b * WB <= 1 && b >= 1) r |= 1u << (b - 1);
  return r;
}

// control word of iteration k (one LDS read for everything uniform the iteration needs):
//   .x  unit k + 3: chunk | block row << 12 | fast << 29 | right neighbour follows << 30 | left neighbour precedes << 31
//   .y  unit k + 1: its two windows of this launch's plane kind
//   .z  unit k:     its two windows of this launch's plane kind
//   .w  unit k: flat bits | L-out-of-int8 bits of the luma launch << 2;  unit k + 1: left << 4 | right << 5
// ---------------------------------------------------------------------------------
// k3s_fused<CBW, CBH, BPS, PL>: as k3f_fused (chroma block 32 >> xdec by 32 >> ydec, 0 0: luma only; PL = 0 the luma plane
// and L, PL = 1 the chroma planes).  grid = frames x workgroups per frame (1-D), block = 256, dynamic LDS = s_lds_bytes.
// ---------------------------------------------------------------------------------
#ifndef G1S_S_OCC_C
#define G1S_S_OCC_C 4  // waves per SIMD the 4:2:0 chroma launch is compiled for (5 = 96 registers: measured 9 % slower, profiles/r03b)
#endif
// workgroups a CU holds (= waves a SIMD holds) for an instantiation: what its LDS leaves room for, at most G1S_F_OCC.  The
// kernel is compiled for that many -- 4:4:4 planes (70 KB of tiles a chroma workgroup: two to a CU; 46 KB a luma workgroup:
// three) get 256 / 168 registers instead of 128 and no longer spill.
__host__ __device__ constexpr int s_occupancy(int CBW, int CBH, int PL) {
  const int nl = (PL == 0 && CBW != 0) ? CBH * kMUnitBlocks * CBW / 8 : 0;      // words of the luma launch's L tile
  const int fixed = 4400 + (nl ? 8 * (nl + 1) * 4 : 0);                          // static LDS (control words, rings, L tiles)
  const int fit = (160 * 1024) / (s_lds_bytes(CBW, CBH, PL) + fixed);
  const int want = (PL == 1 && CBW == 16 && CBH == 16) ? G1S_S_OCC_C : G1S_F_OCC;
  return fit < 1 ? 1 : (fit < want ? fit : want);
}
template <int CBW, int CBH, int BPS, int PL>
__global__ __launch_bounds__(kFThreads, s_occupancy(CBW, CBH, PL)) void k3s_fused(Geom g, FParams fpar) {
  extern __shared__ __attribute__((aligned(16))) uint8_t m_smem[];
  using SH = SShape<CBW, CBH>;
  constexpr bool CH = SH::CH;
  // PL: 0 the luma plane (and L); 1 both chroma planes (Cb on waves 0-1, Cr on waves 2-3); 2 / 3 the Cb / the Cr plane alone on
  // all four waves -- for 4:4:4, where two planes' tiles (70 KB) leave room for two workgroups on a CU and one plane's for four
  constexpr bool LUMA = PL == 0, CHROMA = PL >= 1, SINGLE = PL >= 2;
  static_assert(LUMA || CH, "the chroma launch needs chroma planes");
  constexpr int CPW = SINGLE ? kFWaves : kFWaves / 2;  // waves that stage and multiply a chroma plane
  constexpr int CW_ = SH::CW_, CH_ = SH::CH_, CROUNDS = CHROMA ? (SH::RC + CPW * SH::RPW - 1) / (CPW * SH::RPW) : 0, NCR = CROUNDS > 0 ? CROUNDS : 1;
  constexpr int BUF = s_buf_bytes(CBW, CBH, PL);
  constexpr int OFF_CB = 0, OFF_L = 7 * SH::CSC, OFF_CR = 8 * SH::CSC;
  // per-unit side data, slot = unit & 3: written when the unit's residuals are formed (two iterations before it is
  // multiplied), read when it is multiplied, zeroed an iteration later
  //   block statistics, ONE 64-bit LDS atomic a lane: sum d^2 << 37 | sum src8 << 19 | sum (d + bias); [..][3] a dummy target
  __shared__ unsigned long long s_sum[4][4][kMUnitBlocks];
  __shared__ uint32_t s_badbits[4];      // bit kind * 2 + block: a residual (kind 1: or L) outside int8 in the block's tile
  __shared__ int s_ring[4][kMStatInts];  // statistics records on their way out (wave 3)
  __shared__ uint2 s_L[4][LUMA && SH::NL > 0 ? SH::NL + 1 : 1];  // luma launch: the L tile of a unit on its way to the L plane (wave 3); [NL] a dummy target
  __shared__ uint4 s_ctl[kMMaxUnits];
  // the entries (+ 5 empty ones behind the last): needed until the control words exist and the prologue has read its own --
  // they live where the tiles will
  uint4 *s_ent = reinterpret_cast<uint4 *>(m_smem);
  static_assert((kMMaxUnits + 5) * 16 <= s_lds_bytes(CBW, CBH, PL), "the parked entries fit the tile buffers");

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
  // the fast path and the halo words from the neighbours' registers: only where every word is a vector load (planes 16-byte
  // aligned, no word straddling the right plane edge)
  const bool reuse = LUMA && fpar.reuse && vec_all && (g.W & 7) == 0;
  const int dbg = fpar.dbg;
  (void)dbg;

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
  constexpr int WPP = LUMA ? kFWaves : CPW;                  // waves per plane
  constexpr int RS = TWO_ROW ? 2 : 1;
  constexpr int NSTEP = MBH / RS / WPP;
  const int m_plane = LUMA ? 0 : (SINGLE ? PL - 1 : 1 + (wave >> 1));  // plane this wave multiplies
  const int m_y0 = ((LUMA || SINGLE) ? wave : (wave & 1)) * NSTEP * RS;  // its first sample row
  int m_addr;
  {
    const int s_eff = (LUMA && ms == 7) ? 6 : ms;  // luma launch: the spare rows read what row s = 6 reads (a broadcast)
    int base = (CHROMA && !SINGLE && m_plane == 2) ? OFF_CR : OFF_CB;
    int so = s_eff * MCS;
    if (CHROMA && ms == 7) { base = 0; so = OFF_L; }  // L: an eighth copy of the Cb tile (both planes' waves)
    m_addr = base + so + (m_y0 + 4 + m_ro) * MP + m_xo;
  }

  // ---- this lane's staging work ----
  // luma: lanes 0-47 = 6 row pairs x the unit's 8 own words, lanes 48-59 the pairs' halo words (general path only), 60-63
  // idle.  Pair p = tile rows 2 p, 2 p + 1 (the two rows under a 4:2:0 chroma row); odd pairs take their two rows in the
  // opposite order, so that the 16 lanes a ds_write_b64 is served in write rows an odd number of rows apart (rows are 64
  // bytes: the two 64-byte halves of the 32 banks).
  const bool y_own = lane < 48, y_halo = lane >= 48 && lane < 60;
  const int ypl = y_own ? (lane >> 3) : (y_halo ? ((lane - 48) >> 1) : 0);
  const int ywd = y_own ? 1 + (lane & 7) : (y_halo ? ((lane & 1) ? SH::WY - 1 : 0) : 1);
  const bool y_wave = LUMA && wave < kFWaves - 1;
  const int ypair = wave * SH::PPJ + ypl;
  const int yswap = ypair & 1;
  const int yt[2] = {2 * ypair + yswap, 2 * ypair + (1 ^ yswap)};  // tile rows of the lane's two register sets
  const bool y_stat = y_own && ypair >= 2;                         // an own word of the block's own rows: statistics, L
  const int y_xw = 8 * (ywd - 1), y_bq = (y_xw >> 5) & 1;
  // constants of the lane: fast-path load offsets from the unit's origin (tile row 0, word 0), LDS targets
  uint32_t y_los[2], y_lod[2];
  int y_wa[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int wl = y_own ? ywd : 1, tl = max(yt[r], 1);  // (lanes with nothing to load read something that is there)
    y_los[r] = (uint32_t)tl * fp.src_stride[0] + (uint32_t)(8 * wl * sbps);
    y_lod[r] = (uint32_t)tl * fp.den_stride[0] + (uint32_t)(8 * wl * dbps);
    y_wa[r] = y_own ? yt[r] * SH::PY + y_xw : 8 * (lane & 7);  // (the others: the dummy row)
  }
  const uint32_t y_flagbits = s_flag_bits(ywd, 4);
  const int y_lrow = ypair - 2;  // chroma row of the pair (vertically subsampled chroma)
  // chroma: waves 0, 1 stage Cb, waves 2, 3 Cr; round k, row (2 k + (wave & 1)) * RPW + lane / WC of the rows -3 .. CH - 1
  const int cwd = lane % SH::WC;
  const int cplane = SINGLE ? PL - 1 : 1 + (wave >> 1);
  const uint8_t *c_src = cplane == 2 ? fp.src[2] : fp.src[1], *c_den = cplane == 2 ? fp.den[2] : fp.den[1];
  const uint32_t c_sst = cplane == 2 ? fp.src_stride[2] : fp.src_stride[1], c_dst = cplane == 2 ? fp.den_stride[2] : fp.den_stride[1];
  const bool c_interior = cwd >= 1 && cwd <= SH::WC - 2;
  const int c_xw = 8 * (cwd - 1), c_bq = (c_xw / CW_) & 1;
  int cpl[NCR], ctr[NCR];  // plane (0: the lane is idle in this round), row 0 .. RC - 1 (= block row + 3)
  uint32_t cso[NCR], cdo[NCR];
  int c_wa[NCR];
  bool c_stat[NCR];
#pragma unroll
  for (int k = 0; k < CROUNDS; ++k) {
    const int rr = (CPW * k + (SINGLE ? wave : (wave & 1))) * SH::RPW + lane / SH::WC;
    const bool on = lane / SH::WC < SH::RPW && rr < SH::RC;
    cpl[k] = on ? cplane : 0;
    ctr[k] = on ? rr : 0;
    cso[k] = (uint32_t)ctr[k] * c_sst + (uint32_t)(8 * cwd * sbps);
    cdo[k] = (uint32_t)ctr[k] * c_dst + (uint32_t)(8 * cwd * dbps);
    c_wa[k] = (on && c_interior) ? ((cplane == 2 && !SINGLE) ? OFF_CR : OFF_CB) + (ctr[k] + 1) * SH::PC + c_xw : 8 * (lane & (SH::PC / 8 - 1));
    c_stat[k] = on && c_interior && rr >= 3;
  }
  const uint32_t c_flagbits = s_flag_bits(cwd, CW_ / 8) << kMUnitBlocks;

  v4i32s aSS = {0, 0, 0, 0}, aPP = {0, 0, 0, 0}, aPQ = {0, 0, 0, 0}, aQQ = {0, 0, 0, 0};  // (P P^T = aSS + aPP, Q Q^T = aSS + aQQ: s_multiply)

  // ---- this workgroup's units: their entries, then the iterations' control words, parked in LDS ----
  const int nmine = n_p + n_g;
  if (tid < nmine + 5) {
    uint4 e = make_uint4(0u, 0u, 0u, 0u);
    if (tid < nmine) {
      e = *reinterpret_cast<const uint4 *>(units + (size_t)upos(tid) * kMUnitDwords);
      // bit 31 of .x: the unit before this one in the workgroup's sequence is its left neighbour in the block row
      if (reuse && tid > 0) {
        const uint32_t a = units[(size_t)upos(tid - 1) * kMUnitDwords] & 0xffffffu, here = e.x & 0xffffffu;
        if ((a & 0xfff000u) == (here & 0xfff000u) && (a & 0xfffu) + 1u == (here & 0xfffu)) e.x |= 1u << 31;
      }
      e.w = CHROMA ? (uint32_t)ustats[(size_t)upos(tid) * kMStatInts + 14] : 0u;  // the luma launch's deferral bits
      if (PL == 3) e.y = (uint32_t)ustats[(size_t)upos(tid) * kMStatInts + 15];   // the Cb launch's (a chroma launch has no use for the luma windows)
    }
    s_ent[tid] = e;
  }
  if (tid < 4 * 4 * kMUnitBlocks) (&s_sum[0][0][0])[tid] = 0ull;
  if (tid < 4) s_badbits[tid] = 0u;
  __syncthreads();
  // .x of the control word for unit k (k may run past the last unit: an empty entry).  fast: every row and own word of the
  // unit's tile inside the plane, both neighbours in the sequence, vector loads
  auto unit_x = [&](int k) {
    const uint32_t ex = s_ent[k].x, nxt = s_ent[k + 1].x;
    const int bx0 = kMUnitBlocks * (int)(ex & 0xfffu), by = (int)((ex >> 12) & 0xfffu);
    const bool aL = (ex >> 31) != 0, aR = (nxt >> 31) != 0;
    const bool fast = reuse && aL && aR && by >= 1 && by * kBlock + kBlock <= g.H && bx0 * 32 + 64 <= g.W;
    return (ex & 0xffffffu) | (fast ? 1u << 29 : 0u) | (aR ? 1u << 30 : 0u) | (aL ? 1u << 31 : 0u);
  };
  if (tid < nmine) {
    const uint4 e0 = s_ent[tid], e1 = s_ent[tid + 1];
    uint4 c;
    c.x = unit_x(tid + 3);
    c.y = LUMA ? e1.y : e1.z;
    c.z = LUMA ? e0.y : e0.z;
    c.w = ((e0.x >> 24) & 3u) | (((e0.w >> kMUnitBlocks) & 3u) << 2) | ((e1.x >> 31) << 4) | ((s_ent[tid + 2].x >> 31) << 5);
    if (PL == 3) c.w |= (e0.y & 0xffu) << 8;  // what the Cb launch wrote into the unit's deferral entry: kept
    s_ctl[tid] = c;
  }
  // what the prologue needs of the entries (the tiles take their place)
  const uint32_t u0 = __builtin_amdgcn_readfirstlane(unit_x(0)), u1 = __builtin_amdgcn_readfirstlane(unit_x(1)),
                 u2 = __builtin_amdgcn_readfirstlane(unit_x(2));
  const uint32_t w0 = __builtin_amdgcn_readfirstlane(LUMA ? s_ent[0].y : s_ent[0].z);
  __syncthreads();

  // ---- registers of the pipeline ----
  u32x4 ys_[2], yd_[2];      // luma raw words in flight: two rows, source and denoised
  u32x4 cs_[NCR], cd_[NCR];  // chroma raw words in flight
  uint2 lraw = make_uint2(0u, 0u);  // chroma launch: this thread's word of the L tile, in flight
  uint32_t Dn[2][2] = {{0u, 0u}, {0u, 0u}}, Dc1[2][2] = {{0u, 0u}, {0u, 0u}}, DlastY[2] = {0u, 0u};  // luma residual words: unit k + 2, unit k + 1; last dwords of unit k
  uint32_t Cn[NCR][2] = {}, Cc1[NCR][2] = {};                                                        // chroma residual words: unit k + 2, unit k + 1
  uint2 Ln = make_uint2(0u, 0u), L1 = make_uint2(0u, 0u);
  bool carry_y = false;   // (general path) the last own word of the unit before held a residual outside int8
  bool carry_u = false;   // ... in some lane of the wave
  const bool l_on = CHROMA && tid < SH::NL;
  constexpr int LWR = kMUnitBlocks * CW_ / 8;  // 8-byte words of an L tile row
  const int l_row = tid / LWR, l_wd = tid - l_row * LWR;
  const uint32_t l_off = (uint32_t)l_row * fpar.lpitch + (uint32_t)(8 * l_wd);  // this thread's word of a unit's L tile in the L plane

  // ux: the unit's control .x (chunk, block row, fast, right, left)
  auto request = [&](uint32_t ux) __attribute__((always_inline)) {
    if (G1S_S_DBGBIT(1)) return;
    const int bx0 = kMUnitBlocks * (int)(ux & 0xfffu), by = (int)((ux >> 12) & 0xfffu);
    const int X0y = bx0 * 32 - 8, Y0y = by * kBlock - 4, X0c = bx0 * CW_ - 8, Y0c = by * CH_ - 3;
    if constexpr (CHROMA) {
      if (l_on) {  // (scalar base + the lane's constant offset)
        const uint8_t *lb = lplane + ((uint32_t)(by * CH_) * fpar.lpitch + (uint32_t)(bx0 * CW_));
        uint32_t lo = l_off;
        asm volatile("" : "+v"(lo));
        const u32x2 t = *(gptr_u2)(as_global(lb) + lo);
        lraw = make_uint2(t.x, t.y);
      }
    }
    if (LUMA && y_wave) {
      if ((ux >> 29) & 1u) {  // the fast path: every lane loads at its constant offset (the origin lies inside the plane)
        const uint8_t *sb = fp.src[0] + ((uint32_t)Y0y * fp.src_stride[0] + (uint32_t)(X0y * sbps));
        const uint8_t *db = fp.den[0] + ((uint32_t)Y0y * fp.den_stride[0] + (uint32_t)(X0y * dbps));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          ys_[r] = f_load<BPS>(sb, y_los[r], g.src_bps, true);  // (f_load hides the offset from the optimiser: it would turn
          yd_[r] = f_load<BPS>(db, y_lod[r], g.den_bps, true);  //  base + offset into 64-bit lane addresses, and spill)
        }
        return;
      }
      const uint8_t *sb = fp.src[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.src_stride[0] + (ptrdiff_t)X0y * sbps);
      const uint8_t *db = fp.den[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.den_stride[0] + (ptrdiff_t)X0y * dbps);
      // the general path: halo lanes read their words unless a neighbour holds them, everything inside the plane
      const bool aL = (ux >> 31) != 0, aR = ((ux >> 30) & 1u) != 0;
      const bool slow = !vec_all || ((g.W & 7) != 0 && X0y + 8 * SH::WY > g.W);
      const bool skip = !(y_own || y_halo) || (aL && ywd == 0) || (aR && ywd == SH::WY - 1);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int Y = Y0y + yt[r], X = X0y + 8 * ywd;
        if (__builtin_expect(slow, 0)) {
          ys_[r] = f_load_slow(fp.src[0], fp.src_stride[0], sbps, X, !skip && yt[r] >= 1 ? Y : -1, g.W, g.H);
          yd_[r] = f_load_slow(fp.den[0], fp.den_stride[0], dbps, X, !skip && yt[r] >= 1 ? Y : -1, g.W, g.H);
        } else {
          const bool ok = !skip && yt[r] >= 1 && Y >= 0 && Y < g.H && X >= 0 && X + 8 <= g.W;
          ys_[r] = f_load<BPS>(sb, (uint32_t)yt[r] * fp.src_stride[0] + (uint32_t)(8 * ywd * sbps), g.src_bps, ok);
          yd_[r] = f_load<BPS>(db, (uint32_t)yt[r] * fp.den_stride[0] + (uint32_t)(8 * ywd * dbps), g.den_bps, ok);
        }
      }
    }
    if constexpr (CHROMA) {
      const bool slow = !vec_all || ((cpw & 7) != 0 && X0c + 8 * SH::WC > cpw);
      if (__builtin_expect(slow, 0)) {
#pragma unroll
        for (int q = 0; q < CROUNDS; ++q) {
          const int c = cpl[q];
          cs_[q] = f_load_slow(c_src, c_sst, sbps, X0c + 8 * cwd, c ? Y0c + ctr[q] : -1, cpw, cph);
          cd_[q] = f_load_slow(c_den, c_dst, dbps, X0c + 8 * cwd, c ? Y0c + ctr[q] : -1, cpw, cph);
        }
        return;
      }
      const bool inside = X0c >= 0 && X0c + 8 * SH::WC <= cpw && Y0c >= 0 && Y0c + CH_ + 3 <= cph;
      if (inside) {  // (every lane loads: the idle lanes of a round read row 0, unused)
        const uint8_t *sb = c_src + ((uint32_t)Y0c * c_sst + (uint32_t)(X0c * sbps));
        const uint8_t *db = c_den + ((uint32_t)Y0c * c_dst + (uint32_t)(X0c * dbps));
#pragma unroll
        for (int q = 0; q < CROUNDS; ++q) {
          cs_[q] = f_load<BPS>(sb, cso[q], g.src_bps, true);
          cd_[q] = f_load<BPS>(db, cdo[q], g.den_bps, true);
        }
      } else {
        const uint8_t *sb = c_src + ((ptrdiff_t)Y0c * (ptrdiff_t)c_sst + (ptrdiff_t)X0c * sbps);
        const uint8_t *db = c_den + ((ptrdiff_t)Y0c * (ptrdiff_t)c_dst + (ptrdiff_t)X0c * dbps);
        const bool xok = X0c + 8 * cwd >= 0 && X0c + 8 * cwd + 8 <= cpw;
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

  // raw words of unit k -> residual words (Dn / Cn / Ln), block statistics and out-of-int8 flags (slot k & 3), L -> its LDS tile.
  // aL / fast: of unit k.
  auto form = [&](int k, bool aL, bool fast) __attribute__((always_inline)) {
    const int slot = k & 3;
    if (G1S_S_DBGBIT(2)) {
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
      uint32_t racc = 0, lacc = 0, d16[2][4];
      int sd = 0, sd2 = 0, ls = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        uint32_t hs[4], hv[4];
        s_narrow<BPS>(ys_[r], g.src_bps, g.src_shift, hs);
        s_narrow<BPS>(yd_[r], g.den_bps, g.den_shift, hv);
        s_residual(hs, hv, d16[r], racc);
        Dn[r][0] = pk_bytes(d16[r][0], d16[r][1]);
        Dn[r][1] = pk_bytes(d16[r][2], d16[r][3]);
        // (every lane sums; the lanes outside the block's own rows and words drop theirs into a dummy below)
        sd = __builtin_amdgcn_sdot4((int)Dn[r][0], 0x01010101, sd, false);
        sd = __builtin_amdgcn_sdot4((int)Dn[r][1], 0x01010101, sd, false);
        sd2 = __builtin_amdgcn_sdot4((int)Dn[r][0], (int)Dn[r][0], sd2, false);
        sd2 = __builtin_amdgcn_sdot4((int)Dn[r][1], (int)Dn[r][1], sd2, false);
        ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[0], hs[1]), 0u, (uint32_t)ls);
        ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[2], hs[3]), 0u, (uint32_t)ls);
      }
      if (!G1S_S_DBGBIT(4)) {
        // int8 arithmetic: a block that holds a residual outside int8 is redone by the exact kernel, statistics included
        unsigned long long *tgt = &s_sum[slot][y_stat ? 0 : 3][y_bq];
        atomicAdd(tgt, ((unsigned long long)(uint32_t)sd2 << 37) | ((unsigned long long)(uint32_t)ls << 19) | (unsigned long long)(uint32_t)(sd + kFBiasY));
      }
      if constexpr (CH) {
        // ---- the chroma regressor L (chroma resolution) -> the unit's L tile in LDS (wave 3 stores it) ----
        uint8_t *ltile = reinterpret_cast<uint8_t *>(&s_L[slot][0]);
        constexpr int LROW = kMUnitBlocks * CW_;
#pragma unroll
        for (int r = 0; r < (sy ? 1 : 2); ++r) {
          uint32_t v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = sy ? pk_add(d16[0][q], d16[1][q]) : d16[r][q];
          const int cy = sy ? y_lrow : yt[r] - 4;  // the row of the L tile: the pair's chroma row, or the row's own
          uint8_t *lp = y_stat ? ltile + cy * LROW + (y_xw >> sx) : ltile + 8 * SH::NL;
          if (sx) {
            const uint32_t p0 = ((uint32_t)pk_dot(v[0], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[1], 0x00010001u, 0) << 16);
            const uint32_t p1 = ((uint32_t)pk_dot(v[2], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[3], 0x00010001u, 0) << 16);
            lacc = lacc | pk_add(p0, 0x00800080u) | pk_add(p1, 0x00800080u);
            *reinterpret_cast<uint32_t *>(lp) = pk_bytes(p0, p1);
          } else {
            lacc = lacc | pk_add(v[0], 0x00800080u) | pk_add(v[1], 0x00800080u);
            lacc = lacc | pk_add(v[2], 0x00800080u) | pk_add(v[3], 0x00800080u);
            *reinterpret_cast<uint2 *>(lp) = make_uint2(pk_bytes(v[0], v[1]), pk_bytes(v[2], v[3]));
          }
        }
      }
      // ---- residuals (or L) outside int8 (-128 .. 127): rare; ONE wave-uniform test on the usual way (every lane takes part:
      //      the lanes without a word of their own hold copies of real words, or zeros) ----
      const uint32_t out = (racc | (y_stat ? lacc : 0u)) & 0xff00ff00u;
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(out != 0) != 0 || carry_u, 0)) {
        // A residual outside int8 flags the blocks whose tile holds it.  A halo word that is not read is a neighbour's own
        // word: the unit before carries the flag of its last word to this unit's first block, and this unit's first word
        // flags the second block of the unit before (whose slot is still open: it is multiplied an iteration after this)
        const bool lane_on = fast ? y_own : (y_own || y_halo);  // (fast path: the other lanes hold copies of a word that is not theirs)
        const bool badw = lane_on && (racc & 0xff00ff00u) != 0;
        if (badw) atomicOr(&s_badbits[slot], y_flagbits);
        if (aL && carry_y) atomicOr(&s_badbits[slot], 1u);
        if (aL && badw && ywd == 1) atomicOr(&s_badbits[(k - 1) & 3], 1u << (kMUnitBlocks - 1));
        carry_y = badw && ywd == SH::WY - 2;
        carry_u = __builtin_amdgcn_ballot_w64(carry_y) != 0;
        if (CH && y_stat && (lacc & 0xff00ff00u) != 0) atomicOr(&s_badbits[slot], 1u << (kMUnitBlocks + y_bq));
      }
    }
    if constexpr (CHROMA) {
      Ln = lraw;
      uint32_t rall = 0;
#pragma unroll
      for (int q = 0; q < CROUNDS; ++q) {
        uint32_t hs[4], hv[4], d16[4], racc = 0;
        s_narrow<BPS>(cs_[q], g.src_bps, g.src_shift, hs);
        s_narrow<BPS>(cd_[q], g.den_bps, g.den_shift, hv);
        s_residual(hs, hv, d16, racc);
        Cn[q][0] = pk_bytes(d16[0], d16[1]);
        Cn[q][1] = pk_bytes(d16[2], d16[3]);
        if (!G1S_S_DBGBIT(4)) {
          int sd = __builtin_amdgcn_sdot4((int)Cn[q][0], 0x01010101, 0, false);
          sd = __builtin_amdgcn_sdot4((int)Cn[q][1], 0x01010101, sd, false);
          int sd2 = __builtin_amdgcn_sdot4((int)Cn[q][0], (int)Cn[q][0], 0, false);
          sd2 = __builtin_amdgcn_sdot4((int)Cn[q][1], (int)Cn[q][1], sd2, false);
          unsigned long long *tgt = &s_sum[slot][c_stat[q] ? cplane : 3][c_bq];
          atomicAdd(tgt, ((unsigned long long)(uint32_t)sd2 << 37) | (unsigned long long)(uint32_t)(sd + kFBiasC));
        }
        rall |= cpl[q] ? racc : 0u;
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64((rall & 0xff00ff00u) != 0) != 0, 0)) {
        if ((rall & 0xff00ff00u) != 0) atomicOr(&s_badbits[slot], c_flagbits);
      }
    }
  };
  // the L tile (slot) of the unit whose control .x is ux -> the L plane (luma launch, wave 3)
  auto flush_l = [&](uint32_t ux, int slot) __attribute__((always_inline)) {
    if constexpr (LUMA && CH) {
      const int bx0 = kMUnitBlocks * (int)(ux & 0xfffu), by = (int)((ux >> 12) & 0xfffu);
      constexpr int LW = kMUnitBlocks * CW_ / 8;
      uint8_t *lb = lplane + ((uint32_t)(by * CH_) * fpar.lpitch + (uint32_t)(bx0 * CW_));
#pragma unroll
      for (int w0 = 0; w0 < SH::NL; w0 += 64) {
        const int w = w0 + lane, row = w / LW, wd = w - row * LW;
        uint32_t lo = (uint32_t)row * fpar.lpitch + (uint32_t)(8 * wd);
        asm volatile("" : "+v"(lo));
        if (w < SH::NL) *reinterpret_cast<uint2 *>(lb + lo) = s_L[slot][w];
      }
    }
  };
  // the copies of unit k1 = the unit in Dc1 / Cc1 / L1 -> buffer k1 & 1.  Its left neighbour's last dwords are DlastY when that
  // neighbour is the unit before it (aL), its right neighbour's first dwords are in Dn when it is the unit after it (aR);
  // otherwise they are in the halo lanes of Dc1 (general path).  wy: the unit's two windows of this launch's plane kind.
  auto write_copies = [&](int k1, bool aL, bool aR, uint32_t wy) __attribute__((always_inline)) {
    if (G1S_S_DBGBIT(8)) return;
    const bool plain = k1 < n_p;
    const uint32_t wins[2] = {wy & 0xffffu, wy >> 16};
    uint8_t *buf = m_smem + (k1 & 1) * BUF;
    if (LUMA && y_wave) {
      uint2 cm = make_uint2(~0u, ~0u);
      if (!plain) cm = m_colmask8(m_unpack(y_bq ? wins[1] : wins[0], g.lag), y_xw - 32 * y_bq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        // the dword left of word 1: the last dword of the unit before (same row pair, word 8: 7 lanes up, inside the row of
        // 16 lanes the two pairs share) / the dword right of word 8: the first dword of the unit after (7 lanes down)
        uint32_t left = DlastY[r], right = Dn[r][0];
        if (__builtin_expect(!aL, 0)) left = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (48 + 2 * ypl), (int)Dc1[r][1]);  // (the pair's halo lane)
        if (__builtin_expect(!aR, 0)) right = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (49 + 2 * ypl), (int)Dc1[r][0]);
        uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Dc1[r][1], 0x138, 0xf, 0xf, true);  // wave_shr:1
        uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Dc1[r][0], 0x130, 0xf, 0xf, true);  // wave_shl:1
        const uint32_t lrot = (uint32_t)__builtin_amdgcn_mov_dpp((int)left, 0x129, 0xf, 0xf, true);   // row_ror:9: lane i <- lane i + 7 (mod 16)
        const uint32_t rrot = (uint32_t)__builtin_amdgcn_mov_dpp((int)right, 0x127, 0xf, 0xf, true);  // row_ror:7: lane i <- lane i - 7 (mod 16)
        if (ywd == 1) prev1 = lrot;
        if (ywd == SH::WY - 2) next0 = rrot;
        uint8_t *dst = buf + y_wa[r];
        if (plain) m_write_copies<false>(dst, SH::CSY, prev1, Dc1[r][0], Dc1[r][1], next0, cm);
        else m_write_copies<true>(dst, SH::CSY, prev1, Dc1[r][0], Dc1[r][1], next0, cm);
      }
    }
    if constexpr (CHROMA) {
      if (l_on) {  // the unit's L tile: this thread's word, under the window columns of its chroma block
        const int lb = (8 * l_wd / CW_) & 1;
        const uint2 lm = plain ? make_uint2(~0u, ~0u) : m_colmask8(m_unpack(lb ? wins[1] : wins[0], g.lag), 8 * l_wd - CW_ * lb);
        *reinterpret_cast<uint2 *>(buf + OFF_L + (l_row + 4) * SH::PC + 8 * l_wd) = make_uint2(L1.x & lm.x, L1.y & lm.y);
      }
      uint2 cm = make_uint2(~0u, ~0u);
      if (!plain) cm = m_colmask8(m_unpack(c_bq ? wins[1] : wins[0], g.lag), c_xw - CW_ * c_bq);
#pragma unroll
      for (int q = 0; q < CROUNDS; ++q) {
        const uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Cc1[q][1], 0x138, 0xf, 0xf, true);  // wave_shr:1
        const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Cc1[q][0], 0x130, 0xf, 0xf, true);  // wave_shl:1
        uint8_t *dst = buf + c_wa[q];
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
  uint32_t ux1 = u1, ux2 = u2;  // control .x of units k + 1, k + 2 (k + 3 comes with the iteration's control word)
  if (nmine > 0) {
    request(u0);
    form(0, false, false);
    advance();  // (unit 0 -> the k + 1 registers)
    if (nmine > 1) {
      request(u1);
      form(1, (u1 >> 31) != 0, ((u1 >> 29) & 1u) != 0);
      if (nmine > 2) request(u2);
    }
    write_copies(0, false, ((u0 >> 30) & 1u) != 0, w0);
    advance();
  }
  __syncthreads();
  if (nmine > 0 && wave == kFWaves - 1) flush_l(u0, 0);

  // units [k0, k1) of this workgroup's sequence; two calls (plain units, then the others) are ONE pipeline
  auto run = [&](auto plain_tag, int k0, int k1) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    for (int k = k0; k < k1; ++k) {
      const int slot = k & 3;
      const uint4 c0 = s_ctl[k];
      const uint32_t bb = s_badbits[slot];
      const uint32_t ux3 = __builtin_amdgcn_readfirstlane(c0.x), wy1 = __builtin_amdgcn_readfirstlane(c0.y),
                     wy0 = __builtin_amdgcn_readfirstlane(c0.z), cw = __builtin_amdgcn_readfirstlane(c0.w);
      const uint32_t badbits = __builtin_amdgcn_readfirstlane(bb);
      const uint32_t fbits = PLAIN ? (1u << kMUnitBlocks) - 1u : cw & 3u;
      // ---- the unit after next: its words have had an iteration to land ----
      if (k + 2 < nmine) {
        form(k + 2, (ux2 >> 31) != 0, ((ux2 >> 29) & 1u) != 0);
        if (k + 3 < nmine) request(ux3);
      }
      // ---- the next unit's copies -> the other buffer (free since the barrier: unit k - 1 has been multiplied) ----
      if (k + 1 < nmine) {
        write_copies(k + 1, ((cw >> 4) & 1u) != 0, ((cw >> 5) & 1u) != 0, wy1);
        advance();
      }
      // ------------------------------- multiply unit k -------------------------------
      uint32_t defer = 0;
      {
        // A step spans both blocks of the unit: a residual outside int8 in either tile sends all of the unit's flat blocks to
        // the exact kernel (which redoes their statistics too)
        const uint32_t lbad = (cw >> 2) & 3u;
        const uint32_t mine = LUMA ? badbits & 3u : ((badbits >> kMUnitBlocks) | lbad) & 3u;
        if (LUMA && CH) defer |= ((badbits >> kMUnitBlocks) & fbits) << kMUnitBlocks;  // L: the chroma launch's business
        if (__builtin_expect((mine & fbits) != 0, 0)) {
          // (chroma: bits 2, 3 Cb, bits 4, 5 Cr: k3m_finish reads them per plane)
          defer |= LUMA ? fbits : (PL == 1 ? (fbits << kMUnitBlocks) | (fbits << (2 * kMUnitBlocks)) : fbits << ((PL - 1) * kMUnitBlocks));
        } else if (!G1S_S_DBGBIT(16)) {
          const uint8_t *buf = m_smem + (k & 1) * BUF;
          if constexpr (PLAIN) {
            s_multiply<NSTEP, RS, MP, false, G1S_S_READS>(aSS, aPP, aPQ, aQQ, buf, m_addr, ~0u);
          } else {
            const MWin w0m = m_unpack(wy0 & 0xffffu, g.lag), w1m = m_unpack(wy0 >> 16, g.lag);
            if (w0m.go || w1m.go) {
              const uint32_t r0 = w0m.go ? m_rowmask(w0m.ys, w0m.ye) : 0u, r1 = w1m.go ? m_rowmask(w1m.ys, w1m.ye) : 0u;
              const uint32_t rm = (m_blk ? r1 : r0) >> (m_y0 + m_rho);
              s_multiply<NSTEP, RS, MP, true, G1S_S_READS>(aSS, aPP, aPQ, aQQ, buf, m_addr, rm);
            }
          }
        }
      }
      // ---- wave 3: the unit's statistics record (a four-unit ring, stored four at a time), the next unit's L tile ----
      if (wave == kFWaves - 1 && !G1S_S_DBGBIT(32)) {
        auto mine_entry = [](int t) {
          const int b = t >= 7 ? 1 : 0, e = t - 7 * b, c = e < 3 ? 0 : (e < 5 ? 1 : 2);
          return t < 14 ? (LUMA ? c == 0 : (SINGLE ? c == PL - 1 : c != 0)) : t == (LUMA ? 14 : 15);
        };
        if (lane < kMStatInts && mine_entry(lane)) {
          const int b = lane >= 7 ? 1 : 0, e = lane - 7 * b, c = e < 3 ? 0 : (e < 5 ? 1 : 2), f = e < 3 ? e : (e - 3) & 1;
          int val = (int)(PL == 3 ? defer | ((cw >> 8) & 0xffu) : defer);
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
        if (k + 1 < nmine) flush_l(ux1, (k + 1) & 3);
      }
      // (the sums and flags of the unit before this one: consumed an iteration ago, written again two iterations on)
      if (wave == kFWaves - 1) {
        if (lane < 4 * kMUnitBlocks) (&s_sum[(k + 3) & 3][0][0])[lane] = 0ull;
        else if (lane == 32) s_badbits[(k + 3) & 3] = 0u;
      }
      ux1 = ux2;
      ux2 = ux3;
      if (!G1S_S_DBGBIT(64)) __syncthreads();
    }
  };
  run(std::true_type{}, 0, n_p);
  run(std::false_type{}, n_p, nmine);

  // ---- the workgroup's partial systems: waves add into LDS (int64), one plain store per entry ----
  constexpr int NPL = (LUMA || SINGLE) ? 1 : 2, PL0 = LUMA ? 0 : (SINGLE ? PL - 1 : 1);
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
      add(rP, cP, aSS[r] + aPP[r], false);
      add(rP, cQ, aPQ[r], true);
      add(rQ, cQ, aSS[r] + aQQ[r], false);
    }
  }
  __syncthreads();
  long long *out = fpar.partials + (((size_t)frame * fpar.wg_cap + wg) * 3 + PL0) * kMRec;
  for (int k = tid; k < NPL * kMRec; k += kFThreads) out[k] = s_S[k];
}

}  // namespace g1s

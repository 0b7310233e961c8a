// k3w.hip.h -- the accumulation pass, third generation (G1S_K3=wide): WIDE units, one tile buffer, windows at multiply time.
//
// Same job as k3s.hip.h: source / denoised planes of the flat blocks' tiles -> int8 residual tiles in LDS (7 shifted copies)
// -> exact int8 SYRK on the matrix cores (add_block_observations of av1-grain diff/solver.rs == libaom noise_model.c) -> one
// partial system per workgroup and plane; block statistics, the chroma regressor L and out-of-int8 deferrals on the way.
// What changed, and the measurement behind each change (profiles/r04_issue_probe.txt, profiles/r03_sq_counters_final.txt):
//
//  * A unit is 128 samples wide (4 luma blocks; 8 chroma blocks of 16): a row of a unit is ONE row of 16 lanes (one DPP row),
//    every staging lane holds a word of its own (k3s: 48 of 64), the per-unit costs (entry, branches, barriers, halo
//    exchange) are paid once per 4 096 samples instead of once per 2 048, and a row of a 10-bit unit is two whole 128-byte
//    lines (a chroma row of k3s's units was half a line).
//  * The unit entries (8 dwords) of a workgroup's slice are parked in LDS once and read back a unit at a time
//    (ds_read + v_readfirstlane into SGPRs).  Scalar loads straight from the list were tried first and cost more than
//    half of the luma launch: an s_load in the unit loop is a 2 - 3 us round trip that nothing hides
//    (profiles/r04b_elim.txt).
//  * Observation windows are applied when the tile is MULTIPLIED, as a byte mask on the A operand only
//    (S = sum_p m(p) v(p) v(p)^T = (M V) V^T): a k-group's 16 samples sit in fixed columns, so the column window of a block
//    is one 16-byte lane constant per unit and a row outside the window rows zeroes it for that step.  The staging code
//    therefore writes the copies unmasked, always (k3s: 14 v_and per word and a second code path).
//  * Residual arithmetic in plain 32-bit SWAR on 16-bit lanes (v_and / v_sub / v_add / v_lshrrev issue in ~2.6 cycles a
//    wave, packed-16 / perm / DPP / VOP3 forms in ~4.6): t = ((s >> sh) & 0xff00ff) + 0x800080 - ((v >> sh) & 0xff00ff) holds
//    d + 128 in each half; "some d outside int8" is ONE OR-accumulated test of the high bytes (a borrow between the halves
//    happens only when the low half is out of range, i.e. when the unit is deferred anyway); the bytes are packed with
//    one v_perm per four samples and flipped to two's complement with one v_xor.
//  * No halo words are ever loaded: the left / right halo dwords of a unit are the neighbouring unit's own dwords, taken
//    from the registers of the unit before / after it in the workgroup's run (a DPP row rotation); the list is in raster
//    order (k2w_select_units compacts it deterministically), a neighbour that exists is therefore adjacent in the list, and the
//    one in front of / behind the workgroup's slice is formed as a GHOST (loads and residuals only).
//  * ONE tile buffer (two barriers a unit): 36 KB a luma workgroup, four to a CU.
//
// Exactness: int8 x int8 -> int32 products, int32 accumulators (a workgroup's slice is bounded so that they cannot
// overflow), int64 partial systems: every sum is the reference's own sum of integers, in another order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k3m.hip.h"
#include "kernels.hip.h"

namespace g1s {

constexpr int kWThreads = 256, kWWaves = 4;
constexpr int kWUnitW = 128;       // samples a unit row
constexpr int kWEntry = 8;         // dwords a list entry
constexpr int kWMaxUnits = 100;    // units a workgroup (luma launch): 100 units * 16 steps * 64 samples * 128^2 < 2^31
constexpr int kWMaxUnitsC = 32;    // ... chroma launch: what fits beside four workgroups' tiles in a CU's LDS (32-row planes: 32 steps a unit, 50 at most)


// entry: .x = c | by << 10 | aL << 22 | aR << 23 | plain << 24 | interior << 25 | top << 26;  .y = flat bits;  .z = grid index;
//        [4 .. 7] = the windows of the unit's blocks, 16 bits each (m_unpack's format)
struct WParams {
  FrameTable ft;
  const uint32_t *units;    // [batch][ncell][kWEntry]  this launch's list (k2w_select_units), raster order
  const uint32_t *count;    // [batch]
  uint8_t *records;         // [batch] x g.rec_size: the workgroups scatter the block statistics themselves
  long long *partials;      // [batch][wg_cap][3][kMRec]  one partial system per workgroup and plane (k3w_tail sums them)
  uint8_t *only;            // [batch][3][nblocks]  flat blocks left to the exact kernel (zeroed per batch)
  uint32_t *only_any;       // [batch]
  uint8_t *lbad;            // [batch][ncell_y]  luma unit whose L left int8 (zeroed per batch); the chroma launch reads it
  uint8_t *lplane;          // [batch][lrows][lpitch]  L at chroma resolution, int8
  uint32_t lpitch, lframe_bytes;
  int ncell, ncell_y;       // grid cells a frame of this launch's kind / of the luma kind
  int gx_y;                 // luma cells a block row
  int frames, wgs, wg_cap;
  int rev;                  // 1: the launch walks the batch's frames last to first (what the kernel before it touched last is read first)
  int dbg;                  // timing experiments (G1S_W_DBG, builds with -DG1S_W_DBG_BUILD only): 1 no global loads, 2 no residual arithmetic, 4 no statistics / L, 8 no copy writes, 16 no multiplies, 32 no barriers in the loop, 64 no statistics stores, 128 no L loads; wrong results
};
#ifdef G1S_W_DBG_BUILD
#define G1S_W_DBGBIT(bit) ((wp.dbg & (bit)) != 0)
#else
#define G1S_W_DBGBIT(bit) false
#endif

// ---- matrix rows (as k3s.hip.h): lane l of an operand holds 16 bytes of row i = l & 15 for the k-group l >> 4.
// i -> (u, s): u = which of the operand's two `a`, s = 0..6 the copy (cx = s - 3), s = 7 the chroma regressor L (u = 0 of P)
// or a spare.  P = {a = 0, 2}, Q = {a = 1, 3}; the Q operand of a step is the P operand of the step before.
__device__ __forceinline__ void w_row(int i, int &u, int &s) {
  if (i < 4) { u = 0; s = i; }
  else if (i < 12) { u = 1; s = i - 4; }
  else { u = 0; s = i - 8; }
}
__device__ __forceinline__ int w_rec_index(int op, int i, int lag, int n, bool chroma) {
  int u, s;
  w_row(i, u, s);
  const int a = 2 * u + op;
  if (s == 7) return (a == 0 && chroma) ? n : -1;
  const int cx = s - 3;
  if (a == 0 && cx == 0) return n + (chroma ? 1 : 0);
  if (a == 0 && cx > 0) return -1;
  if (a > lag || cx < -lag || cx > lag) return -1;
  return (lag - a) * (2 * lag + 1) + (cx + lag);
}

// ---- tile geometry: rows t = 0 .. BH + 3 (t = block row + 4; rows 1 .. 3 the halo rows, row 0 unused), 128 bytes a row ----
__host__ __device__ constexpr int w_copy_stride(int BH) {
  int slots = (BH + 4) * (kWUnitW / 16);
  while ((slots & 15) != 2) ++slots;  // copies 2 (mod 16) 16-byte slots apart: conflict-free operand reads
  return slots * 16;
}
// luma: 7 copies; chroma: [Cb: 7 copies][L: an eighth copy][Cr: 7 copies]
__host__ __device__ constexpr int w_lds_bytes(int KIND, int BH) { return (KIND == 0 ? 7 : 15) * w_copy_stride(BH); }

typedef int w_v4 __attribute__((ext_vector_type(4)));
typedef uint32_t w_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t w_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t w_store2 __attribute__((ext_vector_type(2)));  // (the operand type of the 64-bit buffer store builtin)

__device__ __forceinline__ uint32_t w_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }  // v_bfi_b32

// ---------------------------------------------------------------------------------
// the raster-ordered unit list of a frame and plane kind (built by k2w_select_units behind the frame's threshold)
// ---------------------------------------------------------------------------------
struct WUnitParams {
  uint32_t *units[2];   // [batch][ncell[k]][kWEntry]
  uint32_t *count;      // [batch][2]
  int ncell[2], gx[2], ub[2];  // cells a frame, cells a block row, blocks a unit
};
// (the body: the list of plane kind `kind` of the frame, by one workgroup of 1024 threads)
// BITS: `maskp` is the mask as a bitmap in LDS (k2_flat_select_sized: bit 32 + i = block i is flat, zeros around it): a cell's own
// blocks, its neighbour cells and the blocks above come out of two windows of it.  Otherwise (frames of more than 32 768 blocks)
// `maskp` is the frame's mask bytes in the record: a cell asks for ~25 of them, one dependent load each.
__device__ __forceinline__ uint32_t w_bits_at(const uint32_t *bm, int p, int n) {  // n <= 24 bits from block p on (p >= -32)
  const int q = p + 32;
  const uint32_t lo = bm[q >> 5], hi = bm[(q >> 5) + 1];
  return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (q & 31)) & ((1u << n) - 1u);
}
template <bool BITS>
__device__ __forceinline__ void w_build_units(const Geom &g, const void *maskp, const WUnitParams &up, int kind, int frame) {
  const int UB = up.ub[kind], gx = up.gx[kind], ncell = up.ncell[kind];
  const int bw = kind ? (kBlock >> g.xdec) : kBlock, bh = kind ? (kBlock >> g.ydec) : kBlock;
  const int pw = kind ? (g.W >> g.xdec) : g.W, ph = kind ? (g.H >> g.ydec) : g.H;
  uint32_t *out = up.units[kind] + (size_t)frame * ncell * kWEntry;
  const uint8_t *mask = reinterpret_cast<const uint8_t *>(maskp);
  const uint32_t *bm = reinterpret_cast<const uint32_t *>(maskp);
  // (one barrier a chunk of 1024 cells: the waves' counts go into one of two rows by the chunk's parity, every thread adds the
  //  sixteen up for itself and keeps the running total in a register)
  __shared__ __attribute__((aligned(16))) uint32_t s_wave[2][16];
  uint32_t base_count = 0;
  auto at = [&](int x, int y) { return (x >= 0 && x < g.nbw && y >= 0 && y < g.nbh) ? (int)mask[y * g.nbw + x] : 0; };
  auto cell_bits = [&](int c, int by) {
    uint32_t b = 0;
    if (c < 0 || c >= gx) return b;
    for (int k = 0; k < UB; ++k) b |= at(c * UB + k, by) ? 1u << k : 0u;
    return b;
  };
  const uint32_t all = (1u << UB) - 1u;
  for (int base = 0; base < ncell; base += 1024) {
    const int idx = base + (int)threadIdx.x;
    uint32_t bits = 0, e[kWEntry] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (idx < ncell) {
      const int by = idx / gx, c = idx - by * gx;
      // cur: the flat bits of the blocks c UB - UB .. c UB + 2 UB - 1 of block row by (left cell | this cell | right cell), upb: of the
      // blocks above this cell's; blocks outside the row read as not flat
      uint32_t cur = 0, upb = 0;
      if (BITS) {
        const int b0 = c * UB - UB;  // first block of the window
        const int lo = max(-b0, 0), hi = min(max(g.nbw - b0, 0), 3 * UB);
        cur = w_bits_at(bm, by * g.nbw + b0, 3 * UB) & ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        if (by > 0) upb = w_bits_at(bm, (by - 1) * g.nbw + c * UB, UB) & ((1u << min(max(g.nbw - c * UB, 0), UB)) - 1u);
        bits = (cur >> UB) & all;
      } else {
        bits = cell_bits(c, by);
      }
      if (bits) {
        const bool aL = BITS ? (cur & all) != 0 : cell_bits(c - 1, by) != 0, aR = BITS ? ((cur >> (2 * UB)) & all) != 0 : cell_bits(c + 1, by) != 0;
        bool plain = bits == all, top = false;
        for (int k = 0; k < UB; ++k) {
          if (!((bits >> k) & 1u)) continue;
          const int bx = c * UB + k;
          const int left = BITS ? (int)((cur >> (UB + k - 1)) & 1u) : at(bx - 1, by), right = BITS ? (int)((cur >> (UB + k + 1)) & 1u) : at(bx + 1, by),
                    upm = BITS ? (int)((upb >> k) & 1u) : at(bx, by - 1);
          const int ys = upm ? 0 : g.lag, xs = left ? 0 : g.lag;
          const int ye = min(ph - by * bh, bh), xe = min(pw - bx * bw - g.lag, right ? bw : (bw - g.lag));
          const bool go = xe > xs && ye > ys;
          if (!go || xs != 0 || ys != 0 || xe != bw || ye != bh) plain = false;
          if (!go) continue;
          if (ys == 0) top = true;
          const uint32_t wc = (uint32_t)xe | ((uint32_t)ye << 6) | (ys ? 1u << 13 : 0u) | (xs ? 1u << 14 : 0u) | (1u << 15);
          e[4 + (k >> 1)] |= wc << (16 * (k & 1));
        }
        const bool interior = by >= 1 && (by + 1) * bh <= ph && (c + 1) * kWUnitW <= pw;
        e[0] = (uint32_t)c | ((uint32_t)by << 10) | (aL ? 1u << 22 : 0u) | (aR ? 1u << 23 : 0u) | (plain ? 1u << 24 : 0u) |
               (interior ? 1u << 25 : 0u) | (top ? 1u << 26 : 0u);
        e[1] = bits;
        e[2] = (uint32_t)idx;
      }
    }
    // ordered compaction: ballot inside the wave, prefix over the 16 waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, par = (base >> 10) & 1;
    const unsigned long long vote = __ballot(bits != 0);
    if (lane == 0) s_wave[par][wv] = (uint32_t)__popcll(vote);
    __syncthreads();
    // (lane k < 16 takes wave k's count; a DPP scan; the totals in front of this wave and of all sixteen read back as scalars)
    const uint32_t mine = lane < 16 ? s_wave[par][lane] : 0u;
    const uint32_t incl = wave_scan_incl(mine);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
    const uint32_t before = base_count + (wv > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (wv - 1) & 15) : 0u);
    if (bits) {
      const uint32_t pos = before + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull));
      uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)pos * kWEntry);
      dst[0] = make_uint4(e[0], e[1], e[2], e[3]);
      dst[1] = make_uint4(e[4], e[5], e[6], e[7]);
    }
    base_count += total;
  }
  if (threadIdx.x == 0) up.count[2 * frame + kind] = base_count;
}
// k2w_select_units: k2_flat_select (the frame's threshold score, its mask bytes) and, behind it, the frame's unit lists: one launch
// instead of two in the finder's chain.  grid = (batch, kinds of planes: 1 or 2), block = 1024 (= kK2Threads).
static_assert(kK2Threads == 1024, "k2w_select_units: the list builder's workgroup");
__global__ __launch_bounds__(1024) void k2w_select_units(Geom g, uint8_t *__restrict__ records, const uint8_t *__restrict__ flags, WUnitParams up) {
  const int frame = blockIdx.x;
  // the mask also goes into LDS as a bitmap (up to an 8K frame's 32 400 blocks: what the select keeps in registers): the list
  // builder reads it there
  __shared__ uint32_t s_bits[kK2BitWords];
  const bool in_lds = g.nblocks <= kK2Threads * 32;
  k2_flat_select_body(g, records, flags, frame, in_lds ? s_bits : nullptr);
  // (larger frames: the bytes were written by this workgroup -- a workgroup-scope fence and a barrier make them visible to its own loads)
  __threadfence_block();
  __syncthreads();
  // (grid.y = the list's kind: the two workgroups of a frame both find the threshold and write the same mask bytes, then each
  //  builds one list -- the lists were a third of this kernel's time one behind the other, and a frame's workgroup is alone on
  //  its CU either way)
  const int kind = (int)blockIdx.y;
  if (in_lds) w_build_units<true>(g, s_bits, up, kind, frame);
  else w_build_units<false>(g, records + (size_t)frame * g.rec_size + g.off_mask, up, kind, frame);
}

// ---------------------------------------------------------------------------------
// residual arithmetic of one row word (8 samples): raw source / denoised words -> T[4], 16-bit halves holding d + 128
//   BPS 2: T[q] = samples (2 q, 2 q + 1);  BPS 1: T[0] = (0, 2), T[1] = (1, 3), T[2] = (4, 6), T[3] = (5, 7).
// acc |= every T (some d outside int8 <=> (acc & 0xff00ff00) != 0).  (The sum of the narrowed source samples -- the record's
// luma_sum -- is no longer formed here: k1_certify writes it from the finder's moments.)
// ---------------------------------------------------------------------------------
// BPS 2: both inputs are narrowed by the same shift sh <= 4 (the engine sends other pairs of depths down the stream chain), so
// the eight bits are masked where they lie and the difference is shifted once: km = 0xff << sh in both halves, bm = 128 << sh.
template <int BPS>
__device__ __forceinline__ void w_residual(const w_u4 &s, const w_u4 &v, int sh, uint32_t km, uint32_t bm, uint32_t (&T)[4], uint32_t &acc) {
  constexpr uint32_t K = 0x00ff00ffu, B = 0x00800080u;
  if (BPS == 2) {
    const uint32_t ws[4] = {s.x, s.y, s.z, s.w}, wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t a = ws[q] & km, b = wv[q] & km;
      T[q] = ((a - b) + bm) >> sh;
      acc |= T[q];
    }
  } else {
    const uint32_t ws[2] = {s.x, s.y}, wv[2] = {v.x, v.y};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t a0 = ws[q] & K, b0 = wv[q] & K, a1 = (ws[q] >> 8) & K, b1 = (wv[q] >> 8) & K;
      T[2 * q] = (a0 + B) - b0;
      T[2 * q + 1] = (a1 + B) - b1;
      acc |= T[2 * q];
      acc |= T[2 * q + 1];
    }
  }
}
// The general form (GEN: the inputs differ in sample size or in narrowing shift): each input narrowed on its own into 16-bit
// halves in the BPS 2 order -- 16-bit samples shifted and masked, bytes spread by one v_perm per two samples -- then the same
// difference.
template <int BS, int BD>
__device__ __forceinline__ void w_residual_gen(const w_u4 &s, const w_u4 &v, int sh_s, int sh_d, uint32_t (&T)[4], uint32_t &acc) {
  constexpr uint32_t K = 0x00ff00ffu, B = 0x00800080u;
  uint32_t a[4], b[4];
  if (BS == 2) {
    const uint32_t ws[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = (ws[q] >> sh_s) & K;
  } else {
    a[0] = __builtin_amdgcn_perm(0u, s.x, 0x0c010c00u), a[1] = __builtin_amdgcn_perm(0u, s.x, 0x0c030c02u);
    a[2] = __builtin_amdgcn_perm(0u, s.y, 0x0c010c00u), a[3] = __builtin_amdgcn_perm(0u, s.y, 0x0c030c02u);
  }
  if (BD == 2) {
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = (wv[q] >> sh_d) & K;
  } else {
    b[0] = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u), b[1] = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u);
    b[2] = __builtin_amdgcn_perm(0u, v.y, 0x0c010c00u), b[3] = __builtin_amdgcn_perm(0u, v.y, 0x0c030c02u);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    T[q] = (a[q] + B) - b[q];
    acc |= T[q];
  }
}
// T -> the word's 8 residual bytes (two's complement)
template <int BPS>
__device__ __forceinline__ void w_pack(const uint32_t (&T)[4], uint32_t &d0, uint32_t &d1) {
  constexpr uint32_t sel = BPS == 2 ? 0x06040200u : 0x06020400u;
  d0 = __builtin_amdgcn_perm(T[1], T[0], sel) ^ 0x80808080u;
  d1 = __builtin_amdgcn_perm(T[3], T[2], sel) ^ 0x80808080u;
}

// 7 shifted copies of a row word -> LDS (copy s at dst + s * CS): 9 alignbytes, 7 ds_write_b64 with immediate offsets
template <int CS>
__device__ __forceinline__ void w_write_copies(uint8_t *dst, uint32_t prev1, uint32_t d0, uint32_t d1, uint32_t next0) {
  const uint32_t a1 = __builtin_amdgcn_alignbyte(d1, d0, 1), a2 = __builtin_amdgcn_alignbyte(d1, d0, 2), a3 = __builtin_amdgcn_alignbyte(d1, d0, 3);
  const uint32_t b1 = __builtin_amdgcn_alignbyte(d0, prev1, 1), b2 = __builtin_amdgcn_alignbyte(d0, prev1, 2), b3 = __builtin_amdgcn_alignbyte(d0, prev1, 3);
  const uint32_t c1 = __builtin_amdgcn_alignbyte(next0, d1, 1), c2 = __builtin_amdgcn_alignbyte(next0, d1, 2), c3 = __builtin_amdgcn_alignbyte(next0, d1, 3);
  *reinterpret_cast<uint2 *>(dst + 0 * CS) = make_uint2(b1, a1);  // cx = -3
  *reinterpret_cast<uint2 *>(dst + 1 * CS) = make_uint2(b2, a2);
  *reinterpret_cast<uint2 *>(dst + 2 * CS) = make_uint2(b3, a3);
  *reinterpret_cast<uint2 *>(dst + 3 * CS) = make_uint2(d0, d1);
  *reinterpret_cast<uint2 *>(dst + 4 * CS) = make_uint2(a1, c1);  // cx = +1
  *reinterpret_cast<uint2 *>(dst + 5 * CS) = make_uint2(a2, c2);
  *reinterpret_cast<uint2 *>(dst + 6 * CS) = make_uint2(a3, c3);
}

// ---- the multiplies of a chain of NSTEP one-row steps from lane address a0 (the P operand of the first step), pitch 128 ----
// MODE 0 (plain): 2 NSTEP + 1 products (Q Q^T of a step is P P^T of the step before: aS counts for both; k3s.hip.h s_multiply).
// MODE 1 (column windows only: every sample row of the chain inside its window rows): the same 2 NSTEP + 1 products with the
//   A operand under the lane's 16-byte column mask cm -- the mask does not change from step to step, so the renaming still holds.
// MODE 2 (general): rm bit j = the lane's sample row of step j lies inside its window rows: all three products of every step,
//   the A operand under cm and the row bit, the B operand as it is.
template <int NSTEP, int MODE>
__device__ __forceinline__ void w_multiply(w_v4 &aS, w_v4 &aP, w_v4 &aX, w_v4 &aQ, const uint8_t *smem, int a0, w_v4 cm, uint32_t rm) {
  constexpr int P = kWUnitW;
  w_v4 q = *reinterpret_cast<const w_v4 *>(smem + a0 - P);
  if constexpr (MODE == 1) aQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(q & cm, q, aQ, 0, 0, 0);
  // (measured and dropped, profiles/r06f_halo_rotation.txt: the plain chain with FOUR operand reads in flight, pinned with scheduling
  //  barriers -- same registers, same time: the SIMD has three other waves to issue from while one waits for its operands)
  constexpr int H = MODE == 2 ? 1 : 2;  // operand reads in flight (the general form holds three masked copies besides)
#pragma unroll
  for (int j0 = 0; j0 < NSTEP; j0 += H) {
    w_v4 p[H];
#pragma unroll
    for (int j = 0; j < H; ++j) p[j] = *reinterpret_cast<const w_v4 *>(smem + a0 + (j0 + j) * P);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      if constexpr (MODE == 2) {
        const int m = __builtin_amdgcn_sbfe((int)rm, j0 + j, 1);  // 0 or -1
        const w_v4 cmj = cm & m;
        const w_v4 pm = p[j] & cmj, qm = q & cmj;
        aP = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, p[j], aP, 0, 0, 0);
        aX = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, q, aX, 0, 0, 0);
        aQ = __builtin_amdgcn_mfma_i32_16x16x64_i8(qm, q, aQ, 0, 0, 0);
      } else if constexpr (MODE == 1) {
        const w_v4 pm = p[j] & cm;
        if (j0 + j == NSTEP - 1) aP = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, p[j], aP, 0, 0, 0);
        else aS = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, p[j], aS, 0, 0, 0);
        aX = __builtin_amdgcn_mfma_i32_16x16x64_i8(pm, q, aX, 0, 0, 0);
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

// ---------------------------------------------------------------------------------
// k3w_pass<KIND, BPS, SX, SY>
//   KIND 0: the luma plane (blocks 32 x 32, 4 to a unit); leaves L behind when the frame has chroma planes
//           (SX, SY = the chroma subsampling; -1, -1: no chroma planes);
//   KIND 1: both chroma planes (blocks 32 >> SX by 32 >> SY, 128 / width to a unit), Cb on waves 0-1, Cr on waves 2-3, the L
//           tile from the luma launch's L plane.
// grid = frames x workgroups per frame (1-D, frame = blockIdx % frames), block = 256, dynamic LDS = w_lds_bytes.
// ---------------------------------------------------------------------------------
template <int KIND, int SX, int SY>
struct WShape {
  static constexpr bool CHR = KIND == 1;
  static constexpr int BW = CHR ? (32 >> SX) : 32, BH = CHR ? (32 >> SY) : 32;
  static constexpr int UB = kWUnitW / BW;            // blocks a unit
  static constexpr int WPB = BW / 8;                 // words a block row
  static constexpr int CS = w_copy_stride(BH);
  static constexpr int NPL = CHR ? 2 : 1;
  static constexpr int NOWN = NPL * BH / 32;         // own iterations (8 rows) a wave: 1, or 2 for 32-row chroma planes
  static constexpr int NSTEP = NPL * BH * 2 / kWWaves;  // steps of a wave's chain: 16, or 32
  static constexpr int OFF_L = 7 * CS, OFF_P1 = 8 * CS;
  static constexpr bool LOUT = KIND == 0 && SX >= 0;
  static constexpr int LBW = LOUT ? (32 >> SX) : 32, LBH = LOUT ? (32 >> SY) : 32;
};

#ifndef G1S_W_OCC_C
#define G1S_W_OCC_C 4  // workgroups a CU the 4:2:0 / 4:4:0 chroma launch is compiled for (a variant build's switch)
#endif
#ifndef G1S_W_OCC_L
#define G1S_W_OCC_L 4
#endif
// BPD, GEN: the denoised planes' sample size, and the general residual form (inputs of different sample sizes or narrowing
// shifts: w_residual_gen); the layouts behind the residual words are the BPS 2 ones then
template <int KIND, int BPS, int SX, int SY, int BPD = BPS, bool GEN = false>
__global__ __launch_bounds__(kWThreads, KIND == 1 ? (SY == 0 ? 2 : G1S_W_OCC_C) : G1S_W_OCC_L) void k3w_pass(Geom g, WParams wp) {
  constexpr int LAY = GEN ? 2 : BPS;  // the order of the samples in the residual words T
  extern __shared__ __attribute__((aligned(16))) uint8_t w_smem[];
  using SH = WShape<KIND, SX, SY>;
  constexpr bool CHR = SH::CHR, LOUT = SH::LOUT;
  constexpr int BW = SH::BW, BH = SH::BH, UB = SH::UB, CS = SH::CS, NPL = SH::NPL, NOWN = SH::NOWN, NSTEP = SH::NSTEP;
  // per-unit side data, slot = (sequence position + 1) & 3
  __shared__ unsigned long long s_sum[4][NPL][UB];  // block statistics, one 64-bit LDS atomic a lane and iteration
  __shared__ uint32_t s_bad[4];                     // bit 0: a residual outside int8 somewhere in the unit's tile rows (plane 0); 1: in word 0; 2: in word 15; 3: L (luma launch)
                                                    // chroma: bits 4 .. 6 the same for plane 1
  // this workgroup's entries (the ghost's in front, three behind), parked once: a scalar load per iteration costs the launch more
  // than everything else in the loop (an entry is a cache miss far away; profiles/r04b_elim.txt)
  constexpr int MAXU = CHR ? kWMaxUnitsC : kWMaxUnits;
  __shared__ uint4 s_ent[2 * (MAXU + 4)];

  // (blockIdx -> frame = blockIdx % frames: every frame of the batch live at once, on one XCD when the batch is a multiple of 8.
  //  Measured and dropped, profiles/r05_ab_knobs.txt: whole frames dealt to the XCDs one after the other)
  const int fi = (int)blockIdx.x % wp.frames;
  const int G = wp.wgs, frame = g.frame0 + (wp.rev ? wp.frames - 1 - fi : fi), wg = (int)blockIdx.x / wp.frames;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t cnt = wp.count[2 * frame];  // ([batch][2 kinds]: the pointer is this kind's)
  const uint32_t first = (uint32_t)((unsigned long long)cnt * (uint32_t)wg / (uint32_t)G);
  const int nmine = (int)((uint32_t)((unsigned long long)cnt * (uint32_t)(wg + 1) / (uint32_t)G) - first);
  uint8_t *rec = wp.records + (size_t)frame * g.rec_size;
  uint32_t nobs_acc = 0;  // (statistics threads) observations of the blocks this workgroup multiplied (< 2^32: a slice holds at most kWMaxUnits units)
  const FramePlanes fp = wp.ft.f[frame];
  constexpr int sxc = CHR ? SX : 0, syc = CHR ? SY : 0;
  const int pw = g.W >> sxc, ph = g.H >> syc;
  const int ssh = g.src_shift;  // (== g.den_shift, <= 4: wide_ok)
  const uint32_t r_km = (0xffu << ssh) * 0x00010001u, r_bm = (128u << ssh) * 0x00010001u;

  // ---- this lane's staging work: pair p (two tile rows), word w ----
  const int p = lane >> 4, w = lane & 15;
  const int s_plane = CHR ? (wave >> 1) : 0;  // plane of this wave's own rows and of its chain
  const uint8_t *psrc = CHR ? (s_plane ? fp.src[2] : fp.src[1]) : fp.src[0];
  const uint8_t *pden = CHR ? (s_plane ? fp.den[2] : fp.den[1]) : fp.den[0];
  const uint32_t sst = CHR ? (s_plane ? fp.src_stride[2] : fp.src_stride[1]) : fp.src_stride[0];
  const uint32_t dst_ = CHR ? (s_plane ? fp.den_stride[2] : fp.den_stride[1]) : fp.den_stride[0];
  // own iteration i: tile rows t = 4 + own_row0 + 8 i + 2 p + r
  // (measured and dropped, twice: the halo wave rotating with the workgroup so that the four workgroups of a CU would not all
  //  load the same SIMD with it -- round 4, pseudo-randomly: luma 399 - 429 -> 430 - 441 us; round 6, by the bits of the workgroup
  //  index that differ between the workgroups dealt to one CU: 363 - 367 -> 367 - 374, chroma 217 -> 216 - 220, all-flat 537 -> 543:
  //  nothing.  The halo wave's extra row is not what the workgroup's barriers wait for.  profiles/r06f_halo_rotation.txt)
  const int swave = wave;
  const int own_row0 = CHR ? (swave & 1) * (BH / 2) : 8 * swave;
  // the halo rows (tile rows 0 .. 3) of a plane: one more QUARTER iteration on the plane's last wave -- its 64 lanes are the
  // 4 rows x 16 words, ONE row a lane (lane = row p, word w: the same rows of 16 lanes, so the neighbour exchange is the own
  // rows').  (Round 4 gave the lower 32 lanes two rows each: a whole iteration's instructions on the wave every other wave
  // of the workgroup then waits for at the barrier.)
  const bool h_wave = CHR ? (swave & 1) == 1 : swave == kWWaves - 1;
  // constants of the lane: ONE load offset per input (tile row 4 + own_row0 + 2 p, word w, from the unit's origin = tile row 0,
  // word 0); the second row of the pair and the further own iterations move the SCALAR base instead; the halo row has an
  // offset of its own (tile row p, word w).
  const uint32_t lo_s = (uint32_t)(4 + own_row0 + 2 * p) * sst + (uint32_t)(8 * w * BPS);
  const uint32_t lo_v = (uint32_t)(4 + own_row0 + 2 * p) * dst_ + (uint32_t)(8 * w * BPD);
  // (tile row 0 is never read by a multiply: its lanes ask for row 1's words again -- the same lines as the lanes of row 1, no
  //  bytes of their own from memory: 1 / 36 of the luma launch's tile bytes, 1 / 20 of the 4:2:0 chroma launch's)
  const int ph_ = p > 0 ? p : 1;
  const uint32_t lo_hs = (uint32_t)ph_ * sst + (uint32_t)(8 * w * BPS), lo_hv = (uint32_t)ph_ * dst_ + (uint32_t)(8 * w * BPD);

  // the L plane of the frame; this thread's word(s) of a unit's L tile (chroma launch) / this lane's L bytes (luma launch)
  uint8_t *lframe = wp.lplane + (size_t)frame * wp.lframe_bytes;
  // (its descriptor: the L stores of the luma launch and the L-tile loads of the chroma launch are buffer instructions like the
  //  loads of the planes -- the unit's 32-bit offset in an SGPR, the lane's constant offset in a VGPR)
  const __amdgpu_buffer_rsrc_t bL = __builtin_amdgcn_make_buffer_rsrc(lframe, 0, 0x7fffffff, 0x00020000);
  uint32_t l_off[CHR ? (BH / 16) : NOWN];
  if (CHR) {
#pragma unroll
    for (int q = 0; q < BH / 16; ++q) l_off[q] = (uint32_t)(16 * q + (tid >> 4)) * wp.lpitch + (uint32_t)(8 * (tid & 15));
  } else if (LOUT) {
#pragma unroll
    for (int i = 0; i < NOWN; ++i) l_off[i] = (uint32_t)((own_row0 + 8 * i + 2 * p) >> (SY > 0 ? 1 : 0)) * wp.lpitch + (uint32_t)(8 * w >> (SX > 0 ? 1 : 0));
  }
  (void)l_off;

  // ---- this lane's operand address: row i = lane & 15 -> (u, s); k-group g4 = lane >> 4 ----
  const int mi = lane & 15, mg = lane >> 4;
  int mu, ms;
  w_row(mi, mu, ms);
  const int m_strip = wave & 1;
  const int m_row0 = CHR ? 0 : 16 * (wave >> 1);  // first sample row of this wave's chain
  int m_addr;
  {
    const int s_eff = (!CHR && ms == 7) ? 6 : ms;  // luma: the spare rows read what row s = 6 reads
    int base = CHR ? (s_plane ? SH::OFF_P1 : 0) : 0;
    int so = s_eff * CS;
    if (CHR && ms == 7) { base = 0; so = SH::OFF_L; }
    m_addr = base + so + (m_row0 + 4 - 2 * mu) * kWUnitW + 64 * m_strip + 16 * mg;
  }
  const int m_blk = (64 * m_strip + 16 * mg) / BW, m_xo = (64 * m_strip + 16 * mg) % BW;

  w_v4 aSS = {0, 0, 0, 0}, aPP = {0, 0, 0, 0}, aPQ = {0, 0, 0, 0}, aQQ = {0, 0, 0, 0};

  {
    const uint4 *src = reinterpret_cast<const uint4 *>(wp.units + ((size_t)frame * wp.ncell + first) * kWEntry);
    for (int i = tid; i < 2 * (nmine + 4); i += kWThreads) s_ent[i] = src[i - 2];  // (entry -1: the ghost in front; the engine keeps a pad in front of the first list)
  }
  if constexpr (CHR) {
    // "L left int8 in a luma unit under this unit" (the luma launch's flags), looked up once: a load in the loop would make the
    // loop wait for everything in flight.  Parked in the entry's spare word.
    __syncthreads();
    if (tid < nmine) {
      const uint32_t ex = s_ent[2 * (tid + 1)].x;
      const int c = (int)(ex & 0x3ffu), by = (int)((ex >> 10) & 0xfffu);
      constexpr int LPU = UB / 4;  // luma units under a chroma unit
      uint32_t lb = 0;
#pragma unroll
      for (int q = 0; q < LPU; ++q) {
        const int cy = c * LPU + q;
        if (cy < wp.gx_y) lb |= wp.lbad[(size_t)frame * wp.ncell_y + (size_t)by * wp.gx_y + cy];
      }
      s_ent[2 * (tid + 1)].w = lb;
    }
  }
  if (tid < 4 * NPL * UB) (&s_sum[0][0][0])[tid] = 0ull;
  if (tid < 4) s_bad[tid] = 0u;
  __syncthreads();

  // ---- pipeline registers ----
  w_u4 rs[NOWN][2], rv[NOWN][2];                      // raw words in flight: own iterations
  w_u4 hs = {}, hv = {};                              // ... the halo row
  uint32_t Dc[NOWN][2][2] = {}, Dn[NOWN][2][2] = {}, Dl[NOWN][2] = {};   // residual words: unit k, unit k + 1; last dwords of unit k - 1
  uint32_t Hc[2] = {}, Hn[2] = {}, Hl = 0;

  auto entry_x = [&](int j) -> uint32_t { return __builtin_amdgcn_readfirstlane(s_ent[2 * (j + 1)].x); };

  // the raw words of the unit with entry word ex, into rs / rv (and hs / hv)
  // (16-bit planes are read with the non-temporal hint: a launch walks gigabytes it touches once, and read that way the lines the
  //  finder's moments kernel left in the caches -- the luma source of the batch's last frames -- survive until this launch asks
  //  for them.  4K 10-bit: luma launch 391 -> 377 us, chroma 220 -> 212 with its L tiles read the same way; all-flat 578 -> 560,
  //  324 -> 305.  A 1080p 8-bit batch is about the size of the Infinity Cache: plain loads there (178 vs 184 us).
  //  profiles/r05e_nontemporal.txt)
  auto load8 = [&](const uint8_t *base, uint32_t off, bool wide) __attribute__((always_inline)) -> w_u4 {
    w_u4 r = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(off));  // (opaque: scalar base + 32-bit lane offset, not a 64-bit lane address)
    if (wide) {
      r = __builtin_nontemporal_load((gptr_u4)(as_global(base) + off));
    } else {
      const u32x2 a = *(gptr_u2)(as_global(base) + off);
      r.x = a.x, r.y = a.y;
    }
    return r;
  };
  // (buffer form of the same loads for the units that lie inside the plane: descriptor of the plane in SGPRs, the lane's constant
  //  offset as the instruction's VGPR offset, the row's 32-bit offset from the plane's origin as its SGPR offset -- no 64-bit
  //  address arithmetic a row and no copy of the lane offset into the destination registers)
  const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(psrc), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t bden = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(pden), 0, 0x7fffffff, 0x00020000);
  auto bload8 = [&](__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, bool wide) __attribute__((always_inline)) -> w_u4 {
    w_u4 r = {0u, 0u, 0u, 0u};
    if (wide) {
      r = __builtin_bit_cast(w_u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 2));  // (aux 2: non-temporal)
    } else {
      const w_u2 a = __builtin_bit_cast(w_u2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)soff, 0));
      r.x = a.x, r.y = a.y;
    }
    return r;
  };
  auto request = [&](uint32_t ex) __attribute__((always_inline)) {
    if (G1S_W_DBGBIT(1)) return;
    const int c = (int)(ex & 0x3ffu), by = (int)((ex >> 10) & 0xfffu);
    const int X0 = c * kWUnitW, Y0 = by * BH - 4;
    if ((ex >> 25) & 1u) {  // every row and word of the tile inside the plane (Y0 >= 0)
      const uint32_t so_s = (uint32_t)Y0 * sst + (uint32_t)(X0 * BPS), so_v = (uint32_t)Y0 * dst_ + (uint32_t)(X0 * BPD);
#pragma unroll
      for (int i = 0; i < NOWN; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          rs[i][r] = bload8(bsrc, lo_s, so_s + (uint32_t)(8 * i + r) * sst, BPS == 2);
          rv[i][r] = bload8(bden, lo_v, so_v + (uint32_t)(8 * i + r) * dst_, BPD == 2);
        }
      if (h_wave) {
        hs = bload8(bsrc, lo_hs, so_s, BPS == 2);
        hv = bload8(bden, lo_hv, so_v, BPD == 2);
      }
      return;
    }
    // (scalar origin + the lane's constant offset)
    const uint8_t *sb = psrc + ((ptrdiff_t)Y0 * (ptrdiff_t)sst + (ptrdiff_t)(X0 * BPS));
    const uint8_t *vb = pden + ((ptrdiff_t)Y0 * (ptrdiff_t)dst_ + (ptrdiff_t)(X0 * BPD));
    {
      const bool xok = X0 + 8 * w + 8 <= pw;
#pragma unroll
      for (int i = 0; i < NOWN; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int t = 4 + own_row0 + 8 * i + 2 * p + r;
          rs[i][r] = w_u4{0u, 0u, 0u, 0u};
          rv[i][r] = w_u4{0u, 0u, 0u, 0u};
          if (xok && Y0 + t < ph) {
            rs[i][r] = load8(sb + (size_t)(8 * i + r) * sst, lo_s, BPS == 2);
            rv[i][r] = load8(vb + (size_t)(8 * i + r) * dst_, lo_v, BPD == 2);
          }
        }
      if (h_wave) {
        hs = w_u4{0u, 0u, 0u, 0u};
        hv = w_u4{0u, 0u, 0u, 0u};
        if (xok && Y0 + ph_ >= 0 && Y0 + ph_ < ph) {
          hs = load8(sb, lo_hs, BPS == 2);
          hv = load8(vb, lo_hv, BPD == 2);
        }
      }
    }
  };
  // chroma launch: this thread's words of the unit's L tile (BH rows of 128 bytes, one 8-byte word a thread and 16 rows)
  w_u2 Lc[CHR ? (BH / 16) : 1];
  (void)Lc;
  auto load_L = [&](uint32_t ex) __attribute__((always_inline)) {
    if (G1S_W_DBGBIT(128)) return;
    if constexpr (CHR) {
      const int c = (int)(ex & 0x3ffu), by = (int)((ex >> 10) & 0xfffu);
      const uint32_t so = (uint32_t)(by * BH) * wp.lpitch + (uint32_t)(c * kWUnitW);
#pragma unroll
      for (int q = 0; q < BH / 16; ++q)
        Lc[q] = __builtin_bit_cast(w_u2, __builtin_amdgcn_raw_buffer_load_b64(bL, (int)l_off[q], (int)so, BPS == 2 ? 2 : 0));  // (16-bit jobs: non-temporal)
    }
  };

  // raw words -> residual words of the unit at sequence position j (D = Dn, H = Hn); REAL: statistics, L, flags of a unit
  // of this workgroup's own (a ghost leaves nothing behind but its words and its edge flags)
  auto form = [&](int j, uint32_t ex, bool real) __attribute__((always_inline)) {
    const int slot = (j + 1) & 3;
    uint32_t racc = 0, lacc = 0;
    if (G1S_W_DBGBIT(2)) {
#pragma unroll
      for (int i = 0; i < NOWN; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) Dn[i][r][0] = rs[i][r].x ^ rv[i][r].x, Dn[i][r][1] = rs[i][r].y ^ rv[i][r].y;
      Hn[0] = hs.x ^ hv.x, Hn[1] = hs.y ^ hv.y;
      return;
    }
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
      uint32_t T[2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if constexpr (GEN) w_residual_gen<BPS, BPD>(rs[i][r], rv[i][r], ssh, g.den_shift, T[r], racc);
        else w_residual<BPS>(rs[i][r], rv[i][r], ssh, r_km, r_bm, T[r], racc);
        w_pack<LAY>(T[r], Dn[i][r][0], Dn[i][r][1]);
        __builtin_amdgcn_sched_barrier(0);  // (row by row: the scheduler would otherwise keep both rows' temporaries alive)
      }
      if (real && !G1S_W_DBGBIT(4)) {
        int sd = 0, sd2 = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          sd = __builtin_amdgcn_sdot4((int)Dn[i][r][0], 0x01010101, sd, false);
          sd = __builtin_amdgcn_sdot4((int)Dn[i][r][1], 0x01010101, sd, false);
          sd2 = __builtin_amdgcn_sdot4((int)Dn[i][r][0], (int)Dn[i][r][0], sd2, false);
          sd2 = __builtin_amdgcn_sdot4((int)Dn[i][r][1], (int)Dn[i][r][1], sd2, false);
        }
        const unsigned long long pk = ((unsigned long long)(uint32_t)sd2 << 32) | (unsigned long long)(uint32_t)(sd + 16 * 128);
        atomicAdd(&s_sum[slot][s_plane][w / SH::WPB], pk);
        if constexpr (LOUT) {
          // ---- the chroma regressor L of this lane's samples -> the L plane ----
          const int c = (int)(ex & 0x3ffu), by = (int)((ex >> 10) & 0xfffu);
          const uint32_t lso = (uint32_t)(by * SH::LBH) * wp.lpitch + (uint32_t)(c * (kWUnitW >> (SX > 0 ? 1 : 0)));
          if (SX == 0 && SY == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(w_store2, w_u2{Dn[i][r][0], Dn[i][r][1]}), bL, (int)l_off[i], (int)(lso + (uint32_t)r * wp.lpitch), 0);
            lacc = racc;  // (L is the residual itself: outside int8 exactly where the residual is)
          } else {
#pragma unroll
            for (int r = 0; r < (SY ? 1 : 2); ++r) {
              uint32_t V[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) V[q] = SY ? T[0][q] + T[1][q] : T[r][q];
              constexpr uint32_t bias1 = SY ? 256u : 128u;
              if (SX) {
                uint32_t x01, x23;
                if (LAY == 2) {
                  uint32_t h[4];
#pragma unroll
                  for (int q = 0; q < 4; ++q) h[q] = V[q] + (V[q] >> 16);
                  x01 = __builtin_amdgcn_perm(h[1], h[0], 0x05040100u);
                  x23 = __builtin_amdgcn_perm(h[3], h[2], 0x05040100u);
                } else {
                  x01 = V[0] + V[1];
                  x23 = V[2] + V[3];
                }
                constexpr uint32_t add = (640u - 2u * bias1) * 0x00010001u;
                const uint32_t y01 = x01 + add, y23 = x23 + add;  // halves: L + 640; inside int8 <=> high byte 2
                lacc |= (y01 ^ 0x02000200u) | (y23 ^ 0x02000200u);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(y23, y01, 0x06040200u) ^ 0x80808080u, bL, (int)l_off[i], (int)(lso + (uint32_t)r * wp.lpitch), 0);
              } else {
                // (SY = 1, SX = 0: eight values a row pair, laid out like T)
                constexpr uint32_t add = (640u - bias1) * 0x00010001u;
                uint32_t y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  y[q] = V[q] + add;
                  lacc |= y[q] ^ 0x02000200u;
                }
                constexpr uint32_t sel = LAY == 2 ? 0x06040200u : 0x06020400u;
                __builtin_amdgcn_raw_buffer_store_b64(
                    __builtin_bit_cast(w_store2, w_u2{__builtin_amdgcn_perm(y[1], y[0], sel) ^ 0x80808080u, __builtin_amdgcn_perm(y[3], y[2], sel) ^ 0x80808080u}), bL,
                    (int)l_off[i], (int)(lso + (uint32_t)r * wp.lpitch), 0);
              }
            }
          }
        }
      }
    }
    uint32_t hacc = 0;
    if (h_wave) {
      uint32_t T[4];
      if constexpr (GEN) w_residual_gen<BPS, BPD>(hs, hv, ssh, g.den_shift, T, hacc);
      else w_residual<BPS>(hs, hv, ssh, r_km, r_bm, T, hacc);
      w_pack<LAY>(T, Hn[0], Hn[1]);
    }
    // ---- residuals (or L) outside int8: rare; one wave-uniform test on the usual way ----
    const uint32_t out = (racc | hacc | lacc) & 0xff00ff00u;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(out != 0) != 0, 0)) {
      const int sh0 = CHR ? 4 * s_plane : 0;
      uint32_t bits = 0;
      if ((racc & 0xff00ff00u) != 0) bits |= (1u | (w == 0 ? 2u : 0u) | (w == 15 ? 4u : 0u)) << sh0;
      if ((lacc & 0xff00ff00u) != 0) bits |= 8u;
      if ((hacc & 0xff00ff00u) != 0) bits |= (1u | (w == 0 ? 2u : 0u) | (w == 15 ? 4u : 0u)) << sh0;
      if (!real) bits &= ~(1u | 16u | 8u);  // a ghost: only what its edge words mean to the neighbour
      if (bits) atomicOr(&s_bad[slot], bits);
    }
  };

  // the copies of unit k (Dc / Hc; left halo dwords Dl / Hl, right halo dwords in Dn / Hn) -> the tile buffer.
  // The dword left of word w is word w - 1's second dword (row_shr:1; lane 0 of the row keeps what the first move put there: the
  // unit before's last dword, rotated in from lane 15), the dword right of it word w + 1's first (row_shl:1; lane 15 likewise).
  // (What lane 0 / lane 15 get when the unit has NO neighbour on that side -- words of some other unit, or zeros -- is never
  //  multiplied into a sum: a block without a flat neighbour keeps `lag` columns from that edge out of its window, the window
  //  masks the A operand, and a sample inside the window reads at most `lag` columns to its side: words of its own unit.  The
  //  plain products need every block's whole window, i.e. real neighbours.  Round 4 zeroed them: two v_and a row.)
  auto neighbours = [&](uint32_t dl, uint32_t d0, uint32_t d1, uint32_t dn0, uint32_t &prev1, uint32_t &next0)
                        __attribute__((always_inline)) {
    const int hl = __builtin_amdgcn_mov_dpp((int)dl, 0x121, 0xf, 0xf, true);   // row_ror:1:  lane 0 <- lane 15
    const int hr = __builtin_amdgcn_mov_dpp((int)dn0, 0x12f, 0xf, 0xf, true);  // row_ror:15: lane 15 <- lane 0
    prev1 = (uint32_t)__builtin_amdgcn_update_dpp(hl, (int)d1, 0x111, 0xf, 0xf, false);  // row_shr:1
    next0 = (uint32_t)__builtin_amdgcn_update_dpp(hr, (int)d0, 0x101, 0xf, 0xf, false);  // row_shl:1
  };
  auto write_copies = [&](uint32_t ex) __attribute__((always_inline)) {
    if (G1S_W_DBGBIT(8)) return;
    uint8_t *base = w_smem + (CHR && s_plane ? SH::OFF_P1 : 0) + 2 * p * kWUnitW + 8 * w;
#pragma unroll
    for (int i = 0; i < NOWN; ++i)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        uint32_t prev1, next0;
        neighbours(Dl[i][r], Dc[i][r][0], Dc[i][r][1], Dn[i][r][0], prev1, next0);
        w_write_copies<CS>(base + (4 + own_row0 + 8 * i + r) * kWUnitW, prev1, Dc[i][r][0], Dc[i][r][1], next0);
      }
    if (h_wave) {  // (tile row p, word w)
      uint32_t prev1, next0;
      neighbours(Hl, Hc[0], Hc[1], Hn[0], prev1, next0);
      w_write_copies<CS>(base - p * kWUnitW, prev1, Hc[0], Hc[1], next0);
    }
  };
  // unit k <- unit k + 1
  auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NOWN; ++i)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        Dl[i][r] = Dc[i][r][1];
        Dc[i][r][0] = Dn[i][r][0];
        Dc[i][r][1] = Dn[i][r][1];
      }
    Hl = Hc[1];
    Hc[0] = Hn[0];
    Hc[1] = Hn[1];
  };

  // the block statistics of unit j (sequence position) -> the frame's record (flat blocks); its deferred blocks -> the exact
  // kernel's list; the observation count of the blocks that were multiplied.  Called right behind the loads of the iteration after
  // the unit's own: global stores share the loads' counter, and a store issued late in an iteration makes the next wait for the
  // raw words a wait for the store (measured: +130 us on the luma launch).
  uint32_t defer_prev = 0;
  auto stats_out = [&](int j, uint32_t dfr) __attribute__((always_inline)) {
    if (G1S_W_DBGBIT(64)) return;
    if (tid < NPL * UB) {
      const int slot = (j + 1) & 3;
      const int pl = tid / UB, b = tid - pl * UB;
      const unsigned long long pk = s_sum[slot][pl][b];
      s_sum[slot][pl][b] = 0ull;
      const uint4 ta = s_ent[2 * (j + 1)];
      if ((ta.y >> b) & 1u) {
        const int c = (int)(ta.x & 0x3ffu), by = (int)((ta.x >> 10) & 0xfffu);
        const int blk = by * g.nbw + c * UB + b, plane = CHR ? 1 + pl : 0;
        const int samples = BW * BH;
        if (!CHR) {
          reinterpret_cast<int32_t *>(rec + g.off_sum_d[0])[blk] = (int)(uint32_t)(pk & 0xffffffffu) - samples * 128;
          reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[0])[blk] = (uint32_t)(pk >> 32);
        } else {
          // (arithmetic, not an indexed read of the kernel arguments: that is a global load and a wait in the loop)
          const uint32_t od = g.off_sum_d[1] + (uint32_t)pl * (g.off_sum_d[2] - g.off_sum_d[1]), od2 = g.off_sum_d2[1] + (uint32_t)pl * (g.off_sum_d2[2] - g.off_sum_d2[1]);
          reinterpret_cast<int32_t *>(rec + od)[blk] = (int)(uint32_t)(pk & 0xffffffffu) - samples * 128;
          reinterpret_cast<uint32_t *>(rec + od2)[blk] = (uint32_t)(pk >> 32);
        }
        if ((dfr >> pl) & 1u) {
          wp.only[((size_t)frame * 3 + plane) * g.nblocks + blk] = 1;  // (rare)
          wp.only_any[frame] = 1u;
        } else {
          const uint32_t wd = reinterpret_cast<const uint32_t *>(s_ent)[8 * (j + 1) + 4 + (b >> 1)];
          const MWin mw = m_unpack((wd >> (16 * (b & 1))) & 0xffffu, g.lag);
          if (mw.go) nobs_acc += (uint32_t)((mw.xe - mw.xs) * (mw.ye - mw.ys));
        }
      }
    }
  };

  // ---- prologue: the ghost in front of the slice (when the first unit has a left neighbour), unit 0 ----
  if (nmine > 0) {
    const uint32_t e0 = entry_x(0);
    if ((e0 >> 22) & 1u) {
      const uint32_t eg = entry_x(-1);
      request(eg);
      form(-1, eg, false);
      advance();  // (the ghost -> the k registers)
    }
    request(e0);
    form(0, e0, true);
    advance();  // ghost (or zeros) -> the k - 1 registers, unit 0 -> the k registers
    load_L(e0);
    // the words of unit 1 (or of the ghost behind a one-unit slice)
    if (nmine > 1 || ((e0 >> 23) & 1u)) request(entry_x(1));
  }
  __syncthreads();

  for (int k = 0; k < nmine; ++k) {
    // the entry of unit k and the first words of the next two: LDS -> SGPRs
    w_u4 ea, eb;
    {
      const uint4 ta = s_ent[2 * (k + 1)], tb = s_ent[2 * (k + 1) + 1];
      ea.x = __builtin_amdgcn_readfirstlane(ta.x), ea.y = __builtin_amdgcn_readfirstlane(ta.y), ea.z = __builtin_amdgcn_readfirstlane(ta.z), ea.w = CHR ? __builtin_amdgcn_readfirstlane(ta.w) : 0u;
      eb.x = __builtin_amdgcn_readfirstlane(tb.x), eb.y = __builtin_amdgcn_readfirstlane(tb.y);
      eb.z = UB > 4 ? __builtin_amdgcn_readfirstlane(tb.z) : 0u, eb.w = UB > 4 ? __builtin_amdgcn_readfirstlane(tb.w) : 0u;
    }
    const uint32_t x1 = entry_x(k + 1), x2 = entry_x(k + 2);
    // (no explicit wait for the loads here: measured, profiles/r04d -- it costs the chroma launch 24 us: it also waits for the
    //  block-statistics stores of the iteration before, which nothing needs)
    const uint32_t ex = ea.x;
    const bool last = k + 1 == nmine;
    // ---- the next unit's residual words (its raw words have had an iteration to land); the unit after it is requested ----
    if (!last || ((ex >> 23) & 1u)) {
      form(k + 1, x1, !last);
    } else {
#pragma unroll
      for (int i = 0; i < NOWN; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) Dn[i][r][0] = Dn[i][r][1] = 0u;
      Hn[0] = Hn[1] = 0u;
    }
    // (every wait for memory is behind us: what follows only issues -- the L tile of this unit into the buffer, the stores of the
    //  unit before, the loads of the units ahead -- and nothing in the rest of the iteration waits for a load or a store)
    if constexpr (CHR) {
#pragma unroll
      for (int q = 0; q < BH / 16; ++q) {
        const int row = 16 * q + (tid >> 4);
        *reinterpret_cast<uint2 *>(w_smem + SH::OFF_L + (row + 4) * kWUnitW + 8 * (tid & 15)) = make_uint2(Lc[q].x, Lc[q].y);
      }
    }
    if (!last) load_L(x1);  // (the next unit's L tile, an iteration ahead)
    if (!last && (k + 2 < nmine || ((x1 >> 23) & 1u))) request(x2);
    // (the stores BEHIND the loads: the compiler guards the loads' destination registers with a wait that would take the stores
    //  with it; the next wait for memory, form's in the next iteration, is a whole iteration away)
    if (k > 0) stats_out(k - 1, defer_prev);
    // ---- this unit's copies -> the tile buffer (free since the barrier at the end of the iteration before) ----
    write_copies(ex);
    if (!G1S_W_DBGBIT(32)) __syncthreads();
    // ------------------------------- multiply unit k -------------------------------
    const int slot = (k + 1) & 3;
    const uint32_t b0 = __builtin_amdgcn_readfirstlane(s_bad[slot]), bl = __builtin_amdgcn_readfirstlane(s_bad[k & 3]),
                   br = __builtin_amdgcn_readfirstlane(s_bad[(k + 2) & 3]);
    uint32_t defer;  // bit pl: plane pl of this launch is left to the exact kernel; luma bit 3: L left int8
    {
      const uint32_t aL = (ex >> 22) & 1u, aR = (ex >> 23) & 1u;
      const uint32_t d0 = (b0 & 1u) | (aL & (bl >> 2)) | (aR & (br >> 1));
      defer = d0 & 1u;
      if (CHR) defer |= (((b0 >> 4) & 1u) | (aL & (bl >> 6)) | (aR & (br >> 5))) << 1;
      if (LOUT) defer |= b0 & 8u;
    }
    if (CHR && ea.w) defer |= 3u;  // L outside int8 in a luma unit under this unit: both planes
    const bool mine_deferred = ((defer >> (CHR ? s_plane : 0)) & 1u) != 0;
    // (round 6: a wave that multiplies outranks, at its SIMD's arbiter, the three waves of other workgroups that stage beside it -- the
    //  matrix pipe is the scarcer issue slot, and a product issued late is a barrier reached late by four waves.  luma 362 - 365 ->
    //  354 - 358 us, chroma 212 - 219 -> 209 - 214, the chain - 5 to - 20 by the box; priorities 1 / 2 / 3, the halo wave raised while it
    //  stages, or the staging phase raised instead: all within 3 us of each other.  profiles/r06h_setprio.txt)
    __builtin_amdgcn_s_setprio(1);
    if (!mine_deferred && !G1S_W_DBGBIT(16)) {
      if ((ex >> 24) & 1u) {
        w_multiply<NSTEP, 0>(aSS, aPP, aPQ, aQQ, w_smem, m_addr, w_v4{0, 0, 0, 0}, 0u);
      } else if ([&]() {
                   // the windows of this wave's strip (64 columns: 64 / BW blocks), in SGPRs: every one of them the whole block
                   // -> the plain products, without building a mask
                   constexpr int DPS = (64 / BW + 1) / 2;  // dwords of window codes a strip
                   constexpr uint32_t whole1 = (uint32_t)BW | ((uint32_t)BH << 6) | (1u << 15), whole2 = whole1 | (whole1 << 16);
                   const uint32_t wsel[4] = {eb.x, eb.y, eb.z, eb.w};
                   bool all = true;
#pragma unroll
                   for (int q = 0; q < DPS; ++q) all = all && wsel[m_strip * DPS + q] == whole2;
                   return all;
                 }()) {
        w_multiply<NSTEP, 0>(aSS, aPP, aPQ, aQQ, w_smem, m_addr, w_v4{0, 0, 0, 0}, 0u);
      } else {
        // this lane's window: block m_blk of the unit
        const uint32_t wsel[4] = {eb.x, eb.y, eb.z, eb.w};
        uint32_t wd = wsel[0];
#pragma unroll
        for (int q = 1; q < (UB + 1) / 2; ++q) wd = (m_blk >> 1) == q ? wsel[q] : wd;
        const MWin mw = m_unpack((wd >> (16 * (m_blk & 1))) & 0xffffu, g.lag);
        const int lo = mw.go ? min(max(mw.xs - m_xo, 0), 16) : 0, hi = mw.go ? min(max(mw.xe - m_xo, 0), 16) : 0;
        // bytes [lo, hi) of the lane's 16
        w_v4 cm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int hj = min(max(hi - 4 * j, 0), 4), lj = min(max(lo - 4 * j, 0), 4);
          cm[j] = (int)((uint32_t)((1ull << (8 * hj)) - 1ull) & ~(uint32_t)((1ull << (8 * lj)) - 1ull));
        }
        constexpr uint32_t full = NSTEP >= 32 ? ~0u : (1u << NSTEP) - 1u;
        const uint32_t rm = mw.go ? (m_rowmask(mw.ys, mw.ye) >> m_row0) & full : 0u;
        // what the chain needs, wave-uniform: nothing (no window sample in its strip), the plain products (every lane's window
        // covers its 16 samples and the chain's rows), column masks only, or the general form
        const bool empty = hi <= lo || rm == 0u;
        const bool whole = lo == 0 && hi == 16;
        const bool rows_in = rm == full || empty;
        if (__builtin_amdgcn_ballot_w64(!empty) != 0) {
          if (__builtin_amdgcn_ballot_w64(!rows_in) != 0) w_multiply<NSTEP, 2>(aSS, aPP, aPQ, aQQ, w_smem, m_addr, cm, rm);
          else if (__builtin_amdgcn_ballot_w64(!(whole && !empty)) != 0) w_multiply<NSTEP, 1>(aSS, aPP, aPQ, aQQ, w_smem, m_addr, empty ? w_v4{0, 0, 0, 0} : cm, 0u);
          else w_multiply<NSTEP, 0>(aSS, aPP, aPQ, aQQ, w_smem, m_addr, w_v4{0, 0, 0, 0}, 0u);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (tid == 64) {
      if (LOUT && (defer & 8u)) wp.lbad[(size_t)frame * wp.ncell_y + ea.z] = 1;
      s_bad[(k + 3) & 3] = 0u;  // (the slot of unit k - 2 = of unit k + 2: dead since the iteration before, written again in the next)
    }
    advance();
    defer_prev = defer;
    if (!G1S_W_DBGBIT(32)) __syncthreads();
  }
  if (nmine > 0) stats_out(nmine - 1, defer_prev);

  // ---- the workgroup's partial systems: waves add into LDS (int64), one plain store per entry ----
  long long *s_S = reinterpret_cast<long long *>(w_smem);
  for (int k = tid; k < NPL * kMRec; k += kWThreads) s_S[k] = 0;
  __syncthreads();
  {
    const bool ch = CHR;
    const int nc = g.n + (ch ? 1 : 0);
    long long *dst = s_S + (CHR ? s_plane : 0) * kMRec;
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
    const int cP = w_rec_index(0, mi, g.lag, g.n, ch), cQ = w_rec_index(1, mi, g.lag, g.n, ch);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * mg + r;
      const int rP = w_rec_index(0, row, g.lag, g.n, ch), rQ = w_rec_index(1, row, g.lag, g.n, ch);
      add(rP, cP, aSS[r] + aPP[r], false);
      add(rP, cQ, aPQ[r], true);
      add(rQ, cQ, aSS[r] + aQQ[r], false);
    }
  }
  __syncthreads();
  // the partial systems: plain stores (k3w_tail sums a frame's; atomics into the record were measured: 32 workgroups adding to
  // the same 41 lines cost the launch 70 us); the observation counts: one atomic per statistics thread
  {
    long long *outp = wp.partials + (((size_t)frame * wp.wg_cap + wg) * 3 + (CHR ? 1 : 0)) * kMRec;
    for (int k = tid; k < NPL * kMRec; k += kWThreads) outp[k] = s_S[k];
    const int nc0 = g.n + (CHR ? 1 : 0), ne = nc0 * nc0 + nc0;
    if (tid < NPL * UB && nobs_acc != 0)
      atomicAdd(reinterpret_cast<unsigned long long *>(rec + g.off_ar[CHR ? 1 + tid / UB : 0]) + ne, (unsigned long long)nobs_acc);
  }
}

// ---------------------------------------------------------------------------------
// k3w_tail: the launch behind the accumulation launches.  x < kWTailParts: the G partial systems of plane y of the frame, summed
// into its record -- a thread an entry, eight of the workgroups' systems in flight at once (the kernel is as long as its longest
// chain of dependent loads); x >= kWTailParts: the exact kernel (k3_ar_generic's body) on the blocks the launches deferred --
// on most frames none: it returns at once.  The grid is kept SMALL: at (11 + 16) x 3 workgroups a frame the launch was as
// long as the dispatch of its 5 000 - 10 000 workgroups (48 us at 4K, 62 at 1080p with 128-frame batches), whatever they did.
// grid = (kWTailParts + chunks, nplanes, batch), block = 256.
// ---------------------------------------------------------------------------------
constexpr int kWTailParts = 3, kWTailChunks = 4;
static_assert(kK3Threads == 256 && kWTailParts * 256 >= 26 * 26 + 26, "k3w_tail: a thread per entry");
__global__ __launch_bounds__(kK3Threads) void k3w_tail(const FrameTable ft, Geom g, uint8_t *__restrict__ records, const uint8_t *__restrict__ only,
                                                       const uint32_t *__restrict__ only_any, const long long *__restrict__ partials, int wg_cap, int G_luma,
                                                       int G_chroma) {
  const int c = blockIdx.y, frame = g.frame0 + (int)blockIdx.z;
  if ((int)blockIdx.x < kWTailParts) {
    const int nc = g.n + (c > 0), k = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (k >= nc * nc + nc) return;
    const int G = c == 0 ? G_luma : G_chroma;
    const long long *p = partials + (size_t)frame * wg_cap * 3 * kMRec + (size_t)c * kMRec + k;
    constexpr size_t kStep = (size_t)3 * kMRec;
    long long s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int w = 0;
    for (; w + 8 <= G; w += 8) {  // (independent loads in flight)
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += p[(size_t)(w + u) * kStep];
    }
    for (; w < G; ++w) s[0] += p[(size_t)w * kStep];
    const long long tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    // (an atomic: the exact kernel's workgroups of this launch add to the same entries)
    if (tot != 0) {
      long long *ar = reinterpret_cast<long long *>(records + (size_t)frame * g.rec_size + g.off_ar[c]);
      atomicAdd(reinterpret_cast<unsigned long long *>(ar) + k, (unsigned long long)tot);
    }
    return;
  }
  k3_ar_generic_body(ft, g, records, only, only_any, (int)blockIdx.x - kWTailParts, (int)gridDim.x - kWTailParts, c, (int)blockIdx.z);
}

}  // namespace g1s

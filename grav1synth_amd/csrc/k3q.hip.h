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
// All kernels read the planes K0 left behind (k0.hip.h): int8 d8, L8 and the window bits w1.
//   k3_classify          per (frame, kind, area): class + compacted INT / MIX / GENERIC lists
//   k3_lag<KIND, false>  INT areas: 46 (+26 chroma) v_dot4c_i32_i8 per group
//   k3_lag<KIND, true>   MIX areas: the same for their FULL groups (+ L terms under the window); the
//                        coordinates of their PARTIAL groups go to one dense list per (frame, kind)
//   k3_partial_dense     the 324 masked products of every listed group, one lane per group
//   k3q_generic          areas that touch a block with |d| > 127 (or |L| > 127), plain int32
//   k3q_reduce           lag sums / masked products -> record int64 S / Sb / nobs
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "k0.hip.h"
#include "kernels.hip.h"

namespace g1s {

constexpr int kQN = 24;
constexpr int kNumLags = 46;            // distinct c_j - c_i (incl. 0) and -c_i
constexpr int kNumLTerms = 24 + 2;      // (i,L) for the 24 neighbours, L*L, L*y  (L as ONE int8)
enum : uint8_t { kClsExt = 0, kClsInt = 1, kClsMix = 2 };

// lag index: dy = 0: dx 0..6 -> 0..6 ; dy = 1..3: dx -6..6 -> 7 + (dy-1)*13 + (dx+6)
__host__ __device__ constexpr int lag_index(int dx, int dy) { return dy == 0 ? dx : 7 + (dy - 1) * 13 + (dx + 6); }
__host__ __device__ constexpr int coord_x(int k) { return k % 7 - 3; }
__host__ __device__ constexpr int coord_y(int k) { return k / 7 - 3; }

// lag-kernel sums per (frame, plane): [46 lag sums][26 L terms][nobs]
constexpr int kQPart = kNumLags + kNumLTerms + 1;
// masked products: anchor i owns (24 - i) + 1 of the 324; anchors i and 23 - i together 27.
// The 12 such pairs are dealt round-robin to kPParts register-sized parts.
constexpr int kPParts = 4;
constexpr int kPPart = 324;
constexpr int kPSub = kPPart / kPParts;
__host__ __device__ constexpr bool p_in_part(int part, int i) { return ((i < 12 ? i : 23 - i) % kPParts) == part; }

// ar3: lag 1 / 2 generators: the lag-3 systems [frame][3][kAr3] the kernels fill (the record holds the lag's own,
// smaller system: k3q_compact picks it out); nullptr for lag 3 (the kernels write the record directly)
constexpr int kAr3 = (kQN + 1) * (kQN + 1) + (kQN + 1) + 1 + 3;  // S 25x25, Sb 25, nobs (+ pad)
struct QParams {
  long long *ar3;
  int mixed_fast;    // 1: MIX areas by k3_lag<true> + k3_partial_dense; 0: by k3q_generic (debug)
  long long *lagacc;   // [batch][3][kQPart]  int64 sums of all lag-kernel workgroups (zeroed per batch)
  long long *paracc;   // [batch][3][kPPart]  int64 sums of all k3_partial_dense workgroups (zeroed per batch)
  uint8_t *cls;        // [batch][2][nblocks]  area class per kind (luma, chroma)
  uint8_t *bad;        // [batch][2][nblocks]  K0: block holds a residual (or L) outside int8 (zeroed per batch)
  uint32_t *lists;     // [batch][2 kinds][3 (INT, MIX, GENERIC)][nblocks] compacted areas
  uint32_t *counts;    // [batch][2][3] list lengths (zeroed before k3_classify)
  uint32_t *pglist;    // [batch][2][pg_cap] partial groups: x/4 + kPadX/4 .. | y << 16 (plane coordinates)
  uint32_t *pgcount;   // [batch][2] (zeroed per batch)
  uint32_t pg_cap;     // nblocks * 256: every group of every area, the list cannot overflow
  uint8_t *planes;     // K0 planes, [batch] x ps.frame_bytes
  PlaneSet ps;
};
// INT / MIX list entry: bx | by << 16.  GENERIC list entry: the block index.
constexpr uint32_t kEntryNone = 0xffffffffu;

__device__ __forceinline__ int sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, int sh) {
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
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
// N wave sums at once: every DPP stage runs over all N values before the next stage, so no
// instruction waits on its predecessor (a lone wave_sum() is a chain of dependent DPP adds
// with wait states between them).  Totals in lane 63.
template <int N>
__device__ __forceinline__ void wave_sums_dpp(int (&v)[N]) {
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
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += __builtin_amdgcn_update_dpp(0, v[i], 0x143, 0xc, 0xf, false);
}
typedef const G1S_GLOBAL uint32_t *gptr_u1;
// dword-aligned wide global loads (global_load_dwordx3 / x4 need no more than that)
typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// ---------------------------------------------------------------------------------
// k3_classify: one thread per block area, both plane kinds.  The area of block (bx, by)
// needs w on rows by*bh .. by*bh+bh+2, cols bx*bw-3 .. bx*bw+bw+2: the windows of the six
// blocks (bx-1..bx+1, by..by+1), which are functions of the flat mask on (bx-2..bx+2,
// by-1..by+1).  That neighbourhood is read once into registers and serves both kinds.
// grid = (ceil(nblocks/kClsThreads), 1, batch), block = kClsThreads; each of the six lists takes ONE
// global atomic per workgroup (they all hit the same six counters: a per-wave atomic serialises).
// ---------------------------------------------------------------------------------
constexpr int kClsThreads = 256;  // small workgroups: they must find room next to the pixel pass of the next batch
__global__ __launch_bounds__(kClsThreads) void k3_classify(Geom g, const uint8_t *__restrict__ records, QParams qp) {
  __shared__ uint32_t s_cnt[6][kClsThreads / 64];  // [kind * 3 + which][wave] -> count, then list position
  const int blk = blockIdx.x * kClsThreads + threadIdx.x;
  const int frame = g.frame0 + (int)blockIdx.z;
  const uint8_t *mask = records + (size_t)frame * g.rec_size + g.off_mask;
  const bool valid = blk < g.nblocks;
  const int bx = valid ? blk % g.nbw : 0, by = valid ? blk / g.nbw : 0;
  uint8_t m[3][5];  // m[1 + dy][2 + dx] = mask(bx + dx, by + dy), 0 outside the grid
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int X = bx + dx, Y = by + dy;
      m[1 + dy][2 + dx] = (valid && X >= 0 && Y >= 0 && X < g.nbw && Y < g.nbh) ? mask[Y * g.nbw + X] : (uint8_t)0;
    }
  }
  const int kinds = g.nplanes == 3 ? 2 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int which_of[2] = {-1, -1};  // list of this area per kind: 0 INT, 1 MIX, 2 GENERIC, -1 none
#pragma unroll
  for (int kind = 0; kind < 2; ++kind) {
    if (kind >= kinds) break;
    const uint8_t *bad = qp.bad + ((size_t)frame * 2 + kind) * g.nblocks;
    uint8_t c = kClsExt;
    bool gen = false;
    if (valid) {
      const int sx = kind ? g.xdec : 0, sy = kind ? g.ydec : 0;
      const int bw = kBlock >> sx, bh = kBlock >> sy, pw = g.W >> sx, ph = g.H >> sy;
      const int AX0 = bx * bw - kQLag, AX1 = bx * bw + bw + kQLag, AY0 = by * bh, AY1 = by * bh + bh + kQLag;
      bool all1 = !(AX0 < 0 || AX1 > pw || AY1 > ph);
      bool any1 = false, anybad = false;
#pragma unroll
      for (int dby = -1; dby <= 1; ++dby) {
#pragma unroll
        for (int dbx = -1; dbx <= 1; ++dbx) {
          const int Bx = bx + dbx, By = by + dby;
          if (Bx < 0 || By < 0 || Bx >= g.nbw || By >= g.nbh) continue;
          // the tiles of this area reach into these blocks (chroma: also the row above, for the L terms)
          if ((dby >= 0 || kind) && bad[By * g.nbw + Bx]) anybad = true;
          if (dby < 0) continue;
          // window of block (Bx, By), as block_window() computes it
          Win w{0, 0, 0, 0, 0};
          if (m[1 + dby][2 + dbx]) {
            w.flat = 1;
            w.ys = m[dby][2 + dbx] ? 0 : g.lag;
            w.xs = m[1 + dby][1 + dbx] ? 0 : g.lag;
            w.ye = min(ph - By * bh, bh);
            w.xe = min(pw - Bx * bw - g.lag, m[1 + dby][3 + dbx] ? bw : (bw - g.lag));
            if (w.xe <= w.xs || w.ye <= w.ys) w.flat = 0;
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
      qp.cls[((size_t)frame * 2 + kind) * g.nblocks + blk] = c;
      gen = c != kClsExt && (anybad || (c == kClsMix && !qp.mixed_fast));
    }
    if (valid) which_of[kind] = gen ? 2 : (c == kClsInt ? 0 : (c == kClsMix ? 1 : -1));
  }
  // compacted lists (any order: the sums are exact integers): wave counts -> LDS, one global
  // atomic per workgroup and list, then every lane writes its entry
  unsigned long long bal[2][3];
#pragma unroll
  for (int kind = 0; kind < 2; ++kind) {
#pragma unroll
    for (int which = 0; which < 3; ++which) {
      bal[kind][which] = __ballot(which_of[kind] == which);
      if (lane == 0) s_cnt[kind * 3 + which][wave] = (uint32_t)__popcll(bal[kind][which]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int l = threadIdx.x;
    uint32_t total = 0;
    for (int w = 0; w < kClsThreads / 64; ++w) total += s_cnt[l][w];
    uint32_t base = 0;
    if (total) base = atomicAdd(&qp.counts[(size_t)frame * 6 + l], total);
    for (int w = 0; w < kClsThreads / 64; ++w) {
      const uint32_t n = s_cnt[l][w];
      s_cnt[l][w] = base;
      base += n;
    }
  }
  __syncthreads();
#pragma unroll
  for (int kind = 0; kind < 2; ++kind) {
    const int which = which_of[kind];
    if (which < 0) continue;
    const unsigned long long b = which == 0 ? bal[kind][0] : (which == 1 ? bal[kind][1] : bal[kind][2]);
    const size_t lo = ((size_t)frame * 2 + kind) * 3 + which;
    qp.lists[lo * g.nblocks + s_cnt[kind * 3 + which][wave] + __popcll(b & ((1ull << lane) - 1ull))] =
        which == 2 ? (uint32_t)blk : ((uint32_t)bx | ((uint32_t)by << 16));
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
  // LDS row pitch: 3*G dwords (>= the G+4 payload dwords, a multiple of 16 bytes).  A half wave reads
  // 32/G rows x G groups; row r starts at bank 3*G*r mod 32 = 24r (G=8) or 12r (G=4): disjoint runs of G banks.
  static constexpr int PITCH_DW = 3 * G;
  static constexpr int PITCH = PITCH_DW * 4;
  static constexpr int UP = kChroma ? kQLag : 0;  // rows above the area (chroma L terms are p-centric)
  static constexpr int TH = BH + kQLag + UP;      // tile rows: -UP .. BH+2
  static constexpr int SEG = (BW + 16) / 16;      // 16-byte segments per tile row: x = -8 .. BW+7
  static constexpr int LSEG = BW / 16;
  static constexpr int TILE_BYTES = TH * PITCH;
  static constexpr int LTILE_BYTES = BH * PITCH;
  static constexpr int DATA_BYTES = NPL * TILE_BYTES + (kChroma ? LTILE_BYTES : 0);
  static constexpr int WTILE_BYTES = (BH + kQLag) * PITCH;  // rows 0..BH+2
  static constexpr int NACC = kNumLags + (kChroma ? kNumLTerms : 0);
};

// ---------------------------------------------------------------------------------
// TileRegs: the tiles ONE WAVE needs of one area, HBM/L2 -> registers (prefetch) -> LDS,
// 16 bytes a piece: the d tile of the wave's plane, (chroma) the L tile, (MIXED) the window
// tile.  LDS layout of a wave's region: d tile, sample (x, y) (x in -8..BW+7, y in -UP..BH+2)
// at byte (y + UP) * PITCH + 8 + x, so group g (x = 4g) is dword g + 2; then the L tile
// (block proper, byte y*PITCH + x); then the window-indicator tile, rows 0..BH+2, same
// column layout, bytes 0xFF / 0x00.
// ---------------------------------------------------------------------------------
template <int KIND, bool MIXED>
struct TileRegs {
  using S = QShape<KIND>;
  static constexpr int ND = S::TH * S::SEG;
  static constexpr int NLI = S::kChroma ? S::BH * S::LSEG : 0;
  static constexpr int NWI = MIXED ? (S::BH + kQLag) * S::SEG : 0;
  static constexpr int NITEMS = ND + NLI + NWI;
  static constexpr int MAXIT = (NITEMS + 63) / 64;
  static constexpr int L_OFF = S::TILE_BYTES, W_OFF = S::TILE_BYTES + (S::kChroma ? S::LTILE_BYTES : 0);
  static constexpr int BYTES = W_OFF + (MIXED ? S::WTILE_BYTES : 0);
  u32x4 regs[MAXIT];

  __device__ __forceinline__ void fetch(const uint8_t *fbase, const PlaneSet &ps, int lane, int pl, int bx, int by) {
    constexpr bool CHROMA = S::kChroma;
    const uint32_t pitch = CHROMA ? ps.pitch[1] : ps.pitch[0];
    const uint32_t off_w = CHROMA ? ps.off_w[1] : ps.off_w[0];
    const uint32_t wpitch = CHROMA ? ps.wpitch[1] : ps.wpitch[0];
    const uint32_t off_d = CHROMA ? (pl ? ps.off_d[2] : ps.off_d[1]) : ps.off_d[0];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
      const int it = lane + k * 64;
      // one straight-line path, parameters chosen by selects
      const bool isd = it < ND, isl = !isd && it < ND + NLI;
      const int r = isd ? it : (isl ? it - ND : it - ND - NLI);
      const int segs = isl ? S::LSEG : S::SEG;
      const int y = r / segs, sg = r - y * segs;
      const uint32_t off = isd ? off_d : (isl ? ps.off_l : off_w);
      const uint32_t pt = isl ? ps.lpitch : (isd ? pitch : wpitch);
      const int row = by * S::BH + y + (isd ? kPadY - S::UP : (isl ? 0 : kPadY));
      // d / L items: 16 bytes of samples; window items: the 16 bits of the same 16 samples
      const int col = isd || isl ? bx * S::BW + 16 * sg : (bx * S::BW + 16 * sg) >> 3;
      gptr_u8 p = as_global(fbase) + off + (size_t)row * pt + (size_t)col;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (it < ND + NLI) v = *(gptr_u4)p;
      else if (it < NITEMS) v.x = *(const G1S_GLOBAL uint16_t *)p;
      regs[k] = v;
    }
  }
  __device__ __forceinline__ void store(uint8_t *lds, int lane) const {
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
      const int it = lane + k * 64;
      const bool isd = it < ND, isl = !isd && it < ND + NLI;
      const int r = isd ? it : (isl ? it - ND : it - ND - NLI);
      const int segs = isl ? S::LSEG : S::SEG;
      const int y = r / segs, sg = r - y * segs;
      const int base = isd ? 0 : (isl ? L_OFF : W_OFF);
      u32x4 v = regs[k];
      if (!isd && !isl) {  // window bits -> bytes
        const uint32_t b16 = v.x;
        v.x = expand_bits4(b16 & 15u);
        v.y = expand_bits4((b16 >> 4) & 15u);
        v.z = expand_bits4((b16 >> 8) & 15u);
        v.w = expand_bits4((b16 >> 12) & 15u);
      }
      if (it < NITEMS) *reinterpret_cast<u32x4 *>(lds + base + y * S::PITCH + 16 * sg) = v;
    }
  }
};

// ---------------------------------------------------------------------------------
// TileT: the luma d tile ROW-INTERLEAVED.  Dword (rq, x) of the LDS tile = the samples of column x
// in the four rows 4 rq .. 4 rq + 3, one per byte (x = -8 .. 39 at dword index 0 .. 47, rq = 0 .. 8:
// rows 0 .. 35, row 35 never used).  A lane then owns 4 columns x 4 rows: the operands of a horizontal
// lag are simply other registers, the operands of a vertical lag dy are ONE v_alignbyte per column over
// this row quad and the next -- 48 alignbytes per 184 dot4 where the row-major tile needs 144, and 2.5
// LDS dwords per sample instead of 4.5.  The transposition (8 v_perm per 4 x 4 bytes) is done once per
// area by the 27 lanes that stage the tile: lane = (row quad, 16-byte segment), four rows of it.
// ---------------------------------------------------------------------------------
struct TileT {
  static constexpr int NQ = 9, SEG = 3, UNITS = NQ * SEG, ROW_DW = 48, BYTES = NQ * ROW_DW * 4;
  u32x4 r[4];
  __device__ __forceinline__ void fetch(const uint8_t *fbase, const PlaneSet &ps, int lane, int bx, int by) {
    const int rq = lane / SEG, sg = lane - rq * SEG;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 4 * rq + k;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (lane < UNITS && y <= kBlock + kQLag - 1) {
        gptr_u8 p = as_global(fbase) + ps.off_d[0] + (size_t)(by * kBlock + y + kPadY) * ps.pitch[0] + (size_t)(bx * kBlock + 16 * sg);
        v = *(gptr_u4)p;
      }
      r[k] = v;
    }
  }
  __device__ __forceinline__ void store(uint8_t *lds, int lane) const {
    if (lane >= UNITS) return;
    const int rq = lane / SEG, sg = lane - rq * SEG;
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds + (rq * ROW_DW + 16 * sg) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t a = r[0][j], b = r[1][j], c = r[2][j], d = r[3][j];
      const uint32_t ab_lo = __builtin_amdgcn_perm(b, a, 0x05010400u), ab_hi = __builtin_amdgcn_perm(b, a, 0x07030602u);
      const uint32_t cd_lo = __builtin_amdgcn_perm(d, c, 0x05010400u), cd_hi = __builtin_amdgcn_perm(d, c, 0x07030602u);
      u32x4 o;
      o.x = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u);  // column 4 j    : rows 0 .. 3
      o.y = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);  // column 4 j + 1
      o.z = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u);
      o.w = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
      dst[j] = o;
    }
  }
};

// The luma window tile next to a TileT: row-major bytes 0xFF / 0x00, rows 0 .. BH+2, the column layout
// of TileRegs (sample x at byte 8 + x, pitch 3 G dwords); 105 items of 16 samples, loaded as 16 bits.
struct WTileT {
  static constexpr int ROWS = kBlock + kQLag, SEG = 3, ITEMS = ROWS * SEG, PITCH = 3 * (kBlock / 4) * 4;
  static constexpr int BYTES = ROWS * PITCH;
  uint32_t bits[2];
  __device__ __forceinline__ void fetch(const uint8_t *fbase, const PlaneSet &ps, int lane, int bx, int by) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int it = lane + 64 * k;
      const int y = it / SEG, sg = it - y * SEG;
      uint32_t v = 0;
      if (it < ITEMS) {
        gptr_u8 p = as_global(fbase) + ps.off_w[0] + (size_t)(by * kBlock + y + kPadY) * ps.wpitch[0] +
                    (size_t)((bx * kBlock + 16 * sg) >> 3);
        v = *(const G1S_GLOBAL uint16_t *)p;
      }
      bits[k] = v;
    }
  }
  __device__ __forceinline__ void store(uint8_t *lds, int lane) const {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int it = lane + 64 * k;
      const int y = it / SEG, sg = it - y * SEG;
      const uint32_t b16 = bits[k];
      u32x4 v;
      v.x = expand_bits4(b16 & 15u);
      v.y = expand_bits4((b16 >> 4) & 15u);
      v.z = expand_bits4((b16 >> 8) & 15u);
      v.w = expand_bits4((b16 >> 12) & 15u);
      if (it < ITEMS) *reinterpret_cast<u32x4 *>(lds + y * PITCH + 16 * sg) = v;
    }
  }
};

// Group classification from window bytes: w32 points at the dword of x = 4g - 8 in row 0 of
// the group, rows are `pitch_dw` dwords apart; the group's own samples are dword +2.
// FULL / EMPTY are decided on the bytes x-3 .. x+6 of rows 0..3 (a superset of what the 24
// masks read; every kernel uses this same predicate, so the partition into full / partial /
// empty groups is consistent).
template <typename P>
__device__ __forceinline__ void group_state(P w32, int pitch_dw, bool &full, bool &empty) {
  uint32_t all_and = 0xffffffffu, all_or = 0;
#pragma unroll
  for (int dy = 0; dy <= 3; ++dy) {
    P rp = w32 + dy * pitch_dw;
    const uint32_t q1 = rp[1], q2 = rp[2], q3 = rp[3];
    all_and &= (q1 | 0x000000ffu) & q2 & (q3 | 0xff000000u);
    all_or |= (q1 & 0xffffff00u) | q2 | (q3 & 0x00ffffffu);
  }
  full = all_and == 0xffffffffu;
  empty = all_or == 0;
}

// ---------------------------------------------------------------------------------
// k3_lag<KIND, MIXED>: 46 lag sums (+26 chroma L terms) per group.
//   MIXED = false: the INT list (every group full, own window = whole block)
//   MIXED = true : the MIX list; only FULL groups enter the lag sums; L terms and nobs use the
//                  block's own window; the PARTIAL groups are listed for k3_partial_dense.
// grid = (chunks, 1, batch), block = 256 = four AUTONOMOUS waves: a wave walks its own slice of
// the list (chroma: and owns one of the two planes), stages its tiles alone in its own LDS
// region and multiplies them; no workgroup barrier until the final reduction.  LDS operations
// of one wave execute in order, so a single tile buffer per wave is enough: the loads of area
// k+1 fly (in registers) during the products of area k.
// int32 safety: per step |sum| <= 4*127^2; the launch keeps <= 128 area steps per wave; x64 lanes < 2^31.
// ---------------------------------------------------------------------------------
constexpr int kLagWaves = 4;
template <int KIND, bool MIXED>
__global__ __launch_bounds__(64 * kLagWaves) void k3_lag(Geom g, QParams qp) {
  using S = QShape<KIND>;
  constexpr bool CHROMA = S::kChroma;
  constexpr int NACC = S::NACC;
  using Tile = TileRegs<KIND, MIXED>;
  // partial-group coordinates of this wave; normally flushed ONCE, by the whole workgroup at the
  // end (every flush is a returning atomic on the one counter of the (frame, kind) list)
  constexpr int PGBUF = MIXED ? (S::NG < 256 ? 256 : 2 * S::NG) : 1;
  constexpr bool TRANSPOSED = KIND == 0;  // luma: row-interleaved tile (TileT)
  __shared__ __attribute__((aligned(16))) uint8_t lds_all[kLagWaves][TRANSPOSED ? TileT::BYTES + (MIXED ? WTileT::BYTES : 0) : Tile::BYTES];
  __shared__ uint32_t s_pg_all[kLagWaves][PGBUF];
  __shared__ int red[kLagWaves][kQPart + 1];
  __shared__ uint32_t s_pgn[kLagWaves];

  const int frame = g.frame0 + (int)blockIdx.z;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // global wave index -> (list slice, plane).  The slice count is a multiple of 8 (list_slice).
  const int gw = (int)blockIdx.x * kLagWaves + wave;
  const int pl = CHROMA ? (gw & 1) : 0;
  const int slice = CHROMA ? (gw >> 1) : gw;
  const int nslices = (int)gridDim.x * kLagWaves / S::NPL;
  const size_t lsel = ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * 3 + (MIXED ? 1 : 0);
  gptr_u1 list = (gptr_u1)as_global(reinterpret_cast<const uint8_t *>(qp.lists + lsel * g.nblocks));
  const int nlist = (int)qp.counts[lsel];
  const uint8_t *fbase = qp.planes + (size_t)frame * qp.ps.frame_bytes;
  uint8_t *lds = lds_all[wave];
  uint32_t *s_pg = s_pg_all[wave];
  const int lg = lane % S::G, lr = lane / S::G;

  int acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  int nobs = 0;  // INT: number of areas (x BW*BH in the reducer); MIX: window samples (plane 0 waves)
  uint32_t pgn = 0;  // wave-uniform: entries in s_pg
  uint32_t *pg_out = qp.pglist + ((size_t)frame * 2 + (CHROMA ? 1 : 0)) * qp.pg_cap;
  uint32_t *pg_cnt = qp.pgcount + (size_t)frame * 2 + (CHROMA ? 1 : 0);
  auto pg_flush = [&]() {  // one global atomic per flush (a per-step atomic would serialise on the counter)
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(pg_cnt, pgn);
    base = __shfl(base, 0, 64);
    for (uint32_t i = lane; i < pgn; i += 64) pg_out[base + i] = s_pg[i];
    pgn = 0;
  };

  int li, li_end;
  list_slice(slice, nslices, nlist, li, li_end);
  auto entry_at = [&](int pos) -> uint32_t {
    return pos < li_end ? (uint32_t)__builtin_amdgcn_readfirstlane((int)list[pos]) : kEntryNone;
  };
  if constexpr (TRANSPOSED) {
    TileT tt;
    WTileT wt;
    uint32_t e_cur = entry_at(li), e_nxt = entry_at(li + 1);
    if (e_cur != kEntryNone) {
      tt.fetch(fbase, qp.ps, lane, (int)(e_cur & 0xffffu), (int)(e_cur >> 16));
      if (MIXED) wt.fetch(fbase, qp.ps, lane, (int)(e_cur & 0xffffu), (int)(e_cur >> 16));
    }
    const int rq = lane >> 3, cg = lane & 7;
    for (; e_cur != kEntryNone; e_cur = e_nxt, e_nxt = entry_at(li + 1)) {
      __builtin_amdgcn_wave_barrier();
      tt.store(lds, lane);  // in order after the reads of the previous area
      if (MIXED) wt.store(lds + TileT::BYTES, lane);
      __builtin_amdgcn_wave_barrier();
      const int bx = (int)(e_cur & 0xffffu), by = (int)(e_cur >> 16);
      ++li;
      if (e_nxt != kEntryNone) {
        tt.fetch(fbase, qp.ps, lane, (int)(e_nxt & 0xffffu), (int)(e_nxt >> 16));
        if (MIXED) wt.fetch(fbase, qp.ps, lane, (int)(e_nxt & 0xffffu), (int)(e_nxt >> 16));
      }
      if (!MIXED) ++nobs;
      if (MIXED && pgn + S::NG > (uint32_t)PGBUF) pg_flush();
      // the lane's four groups: rows 4 rq + r of group column cg.  FULL groups enter the lag sums (their
      // byte of the row-interleaved operand stays), PARTIAL ones are listed, nobs counts window samples.
      uint32_t M = 0xffffffffu;
      if (MIXED) {
        constexpr int WP = WTileT::PITCH / 4;
        const uint32_t *wb = reinterpret_cast<const uint32_t *>(lds + TileT::BYTES) + (4 * rq) * WP + cg;
        uint32_t ra[7], ro[7];
#pragma unroll
        for (int y = 0; y < 7; ++y) {
          const uint32_t q1 = wb[y * WP + 1], q2 = wb[y * WP + 2], q3 = wb[y * WP + 3];
          ra[y] = (q1 | 0x000000ffu) & q2 & (q3 | 0xff000000u);
          ro[y] = (q1 & 0xffffff00u) | q2 | (q3 & 0x00ffffffu);
          if (y < 4) nobs = sdot4((int)(q2 & 0x01010101u), 0x01010101, nobs);
        }
        M = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool full = (ra[r] & ra[r + 1] & ra[r + 2] & ra[r + 3]) == 0xffffffffu;
          const bool empty = (ro[r] | ro[r + 1] | ro[r + 2] | ro[r + 3]) == 0u;
          if (full) M |= 0xffu << (8 * r);
          const bool part = !full && !empty;
          const unsigned long long bal = __ballot(part);
          if (part)
            s_pg[pgn + __popcll(bal & ((1ull << lane) - 1ull))] =
                (uint32_t)(bx * S::G + cg) | ((uint32_t)(by * S::BH + 4 * rq + r) << 16);
          pgn += (uint32_t)__popcll(bal);
        }
      }
      // columns x0 - 8 .. x0 + 11 (x0 = 4 cg) of this row quad (T) and of the next (N)
      const u32x4 *tq = reinterpret_cast<const u32x4 *>(lds + (rq * TileT::ROW_DW + 4 * cg) * 4);
      const u32x4 *nq = tq + TileT::ROW_DW / 4;
      uint32_t T[20], N[20];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const u32x4 a = tq[k], b = nq[k];
        T[4 * k] = a.x, T[4 * k + 1] = a.y, T[4 * k + 2] = a.z, T[4 * k + 3] = a.w;
        N[4 * k] = b.x, N[4 * k + 1] = b.y, N[4 * k + 2] = b.z, N[4 * k + 3] = b.w;
      }
      // own columns: T[8 + c], c = 0 .. 3
      uint32_t D[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) D[c] = T[8 + c] & M;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int dx = 0; dx <= 6; ++dx) acc[dx] = sdot4((int)D[c], (int)T[8 + c + dx], acc[dx]);
      }
#pragma unroll
      for (int dy = 1; dy <= 3; ++dy) {
        uint32_t Sv[16];  // rows + dy of columns x0 - 6 .. x0 + 9
#pragma unroll
        for (int i = 0; i < 16; ++i) Sv[i] = alignbyte(N[2 + i], T[2 + i], dy);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int dx = -6; dx <= 6; ++dx) {
            const int a = 7 + (dy - 1) * 13 + dx + 6;
            acc[a] = sdot4((int)D[c], (int)Sv[6 + c + dx], acc[a]);
          }
        }
      }
    }
  } else {
  Tile tr;
  uint32_t e_cur = entry_at(li), e_nxt = entry_at(li + 1);
  if (e_cur != kEntryNone) tr.fetch(fbase, qp.ps, lane, pl, (int)(e_cur & 0xffffu), (int)(e_cur >> 16));

  for (; e_cur != kEntryNone; e_cur = e_nxt, e_nxt = entry_at(li + 1)) {
    __builtin_amdgcn_wave_barrier();
    tr.store(lds, lane);  // in order after the reads of the previous area
    __builtin_amdgcn_wave_barrier();
    const int bx = (int)(e_cur & 0xffffu), by = (int)(e_cur >> 16);
    ++li;
    if (e_nxt != kEntryNone) tr.fetch(fbase, qp.ps, lane, pl, (int)(e_nxt & 0xffffu), (int)(e_nxt >> 16));
    if (!MIXED && pl == 0) ++nobs;
    if (MIXED && pl == 0 && pgn + S::NG > (uint32_t)PGBUF) pg_flush();

#pragma unroll 1
    for (int step = 0; step < S::NS; ++step) {
      const int row = step * S::ROWS_PER_STEP + lr;  // sample row of the area
      const uint32_t *t32 = reinterpret_cast<const uint32_t *>(lds) + (row + S::UP) * S::PITCH_DW + lg;
      const uint32_t c0 = t32[2], c1 = t32[3], c2 = t32[4];
      uint32_t D0 = c0, Wc = 0xffffffffu;
      if (MIXED) {
        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(lds + Tile::W_OFF) + row * S::PITCH_DW + lg;
        bool full, empty;
        group_state(w32, S::PITCH_DW, full, empty);
        Wc = w32[2];
        if (!full) D0 = 0;  // partial groups belong to k3_partial_dense, empty ones to nobody
        if (pl == 0) {      // both chroma planes share the window: counted and listed once
          nobs = sdot4((int)(Wc & 0x01010101u), 0x01010101, nobs);
          const bool part = !full && !empty;
          const unsigned long long bal = __ballot(part);
          if (part)
            s_pg[pgn + __popcll(bal & ((1ull << lane) - 1ull))] =
                (uint32_t)(bx * S::G + lg) | ((uint32_t)(by * S::BH + row) << 16);
          pgn += (uint32_t)__popcll(bal);
        }
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
      if (CHROMA) {
        // p-centric L terms under the block's own window (Wc; all ones for INT areas)
        const uint32_t *ta = reinterpret_cast<const uint32_t *>(lds + Tile::L_OFF);
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
  }
  }
  if (MIXED && lane == 0) s_pgn[wave] = pgn;

  // ---- wave reduction (batched DPP: no dependent-instruction bubbles), then the waves of the same
  //      plane add up: one atomic set per workgroup ----
  {
    constexpr int CH = 23;
#pragma unroll
    for (int b0 = 0; b0 < NACC; b0 += CH) {
      constexpr int dummy = 0;
      (void)dummy;
      int tmp[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) tmp[i] = (b0 + i < NACC) ? acc[b0 + i] : 0;
      wave_sums_dpp<CH>(tmp);
      if (lane == 63) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
          if (b0 + i < NACC) red[wave][b0 + i] = tmp[i];
      }
    }
  }
  if (MIXED) nobs = wave_sum(nobs);
  if (lane == 0) red[wave][kQPart] = nobs;
  __syncthreads();
  if (MIXED) {  // the partial groups of the four waves: one atomic
    if (tid == 0) {
      uint32_t total = 0;
      for (int w = 0; w < kLagWaves; ++w) total += s_pgn[w];
      uint32_t base = total ? atomicAdd(pg_cnt, total) : 0u;
      for (int w = 0; w < kLagWaves; ++w) {
        const uint32_t n = s_pgn[w];
        s_pgn[w] = base;
        base += n;
      }
    }
    __syncthreads();
    const uint32_t base = s_pgn[wave];
    for (uint32_t i = lane; i < pgn; i += 64) pg_out[base + i] = s_pg[i];
  }
  for (int p = 0; p < S::NPL; ++p) {
    unsigned long long *out =
        reinterpret_cast<unsigned long long *>(qp.lagacc) + ((size_t)frame * 3 + (CHROMA ? 1 + p : 0)) * kQPart;
    // waves of plane p: CHROMA: wave & 1 == p (blockIdx.x * 4 is even); luma: all
    for (int i = tid; i < NACC; i += 64 * kLagWaves) {
      int v = 0;
      for (int w = 0; w < kLagWaves; ++w)
        if (!CHROMA || (w & 1) == p) v += red[w][i];
      if (v != 0) atomicAdd(&out[i], (unsigned long long)(long long)v);
    }
    if (tid == 0) {
      long long v = 0;
      for (int w = 0; w < kLagWaves; ++w)
        if (!CHROMA || (w & 1) == 0) v += red[w][kQPart];  // counted by the plane-0 waves
      if (!MIXED) v *= S::BW * S::BH;
      if (v != 0) atomicAdd(&out[kQPart - 1], (unsigned long long)v);
    }
  }
}

// ---------------------------------------------------------------------------------
// k3_partial_dense: the 324 masked products of the listed PARTIAL groups,
//     acc(i, j) += dot4(D(q) & W(q - c_i), D(q + c_j - c_i)),   acc(i, y) likewise,
// barrier-free: a lane owns a group per step, gathers its 18 + 12 operand dwords from the
// d8 / w8 planes (neighbouring groups share cache lines), rebuilds the 46 shifted operands
// and multiplies.  wave = which part of the anchors (kPSub accumulators a lane),
// blockIdx.y = frame * nplanes + component.  grid = (chunks, batch * nplanes).
// int32 safety: <= 64516 per step and accumulator; the launch keeps steps per lane < 520.
// ---------------------------------------------------------------------------------
template <int PART>
__device__ __forceinline__ void partial_products(int (&acc)[kPSub], const uint32_t (&D)[kNumLags],
                                                 const uint32_t (&wb)[4]) {
  // wb[dy]: window bits of samples x-4 .. x+7 of row dy (bit 0 = x-4)
  int idx = 0;
#pragma unroll
  for (int i = 0; i < kQN; ++i) {
    if (!p_in_part(PART, i)) continue;
    const int dx = -coord_x(i), dy = -coord_y(i);  // W(q - c_i): samples x+dx .. x+dx+3 of row dy
    const uint32_t wi = expand_bits4((wb[dy] >> (4 + dx)) & 15u);
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

// Asynchronous gathers straight into LDS (gfx950 global_load_lds_dword / _dwordx4): lane l's data
// lands at dst + l * size, no VGPRs are held while the load is in flight and vmcnt counts it.
// Issued through inline asm on purpose: the compiler would otherwise wait for vmcnt(0) before
// every LDS read that may alias the destination; the waits are placed by hand below.
typedef __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
__device__ __forceinline__ uint32_t lds_address(const void *p) {
  return (uint32_t)(uintptr_t)(lds_u32_ptr)(uint32_t *)p;
}
__device__ __forceinline__ void dma16(gptr_u8 src, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(gptr_u8 src, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(lds_dst) : "memory");
}
// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

constexpr int kDStages = 5;     // ring of step buffers: the gathers of four later steps are in flight
constexpr int kDEnt = 2048;     // list entries staged per block (32 steps)
struct DenseStep {              // operand dwords of the 64 groups of a step
  uint32_t A[4][64][4];         // A[0]: d row 0, samples x .. x+15; A[r]: d row r, samples x-8 .. x+7
  uint32_t B[3][64];            // d row r+1, samples x+8 .. x+11
  uint32_t W[9][64];            // window bit words: row dy -> W[2 dy], W[2 dy + 1]; W[8] is a dummy slot
};
__global__ __launch_bounds__(256, 3) void k3_partial_dense(Geom g, QParams qp) {
  __shared__ __attribute__((aligned(16))) DenseStep ring[kDStages];
  __shared__ uint32_t s_ent[kDEnt];
  __shared__ long long red[4][kPSub];
  static_assert(kPParts == 4, "a wave per anchor part");
  const int fz = (int)blockIdx.y / g.nplanes, c = (int)blockIdx.y - fz * g.nplanes;
  const int frame = g.frame0 + fz;
  const int kind = c > 0 ? 1 : 0;
  const uint32_t n = min(qp.pgcount[(size_t)frame * 2 + kind], qp.pg_cap);
  // contiguous share of the list for this workgroup, in whole steps
  const uint32_t per = (((n + gridDim.x - 1) / gridDim.x) + 63u) & ~63u;
  const uint32_t lo = min(n, (uint32_t)blockIdx.x * per), hi = min(n, lo + per);
  if (lo >= hi) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  gptr_u1 list = (gptr_u1)as_global(reinterpret_cast<const uint8_t *>(qp.pglist + ((size_t)frame * 2 + kind) * qp.pg_cap));
  gptr_u8 fb = as_global(qp.planes) + (size_t)frame * qp.ps.frame_bytes;
  gptr_u8 dplane = fb + (c == 0 ? qp.ps.off_d[0] : (c == 1 ? qp.ps.off_d[1] : qp.ps.off_d[2]));
  gptr_u8 wplane = fb + (kind ? qp.ps.off_w[1] : qp.ps.off_w[0]);
  const uint32_t wpitch = kind ? qp.ps.wpitch[1] : qp.ps.wpitch[0];
  const uint32_t pitch = kind ? qp.ps.pitch[1] : qp.ps.pitch[0];
  int acc[kPSub];
#pragma unroll
  for (int i = 0; i < kPSub; ++i) acc[i] = 0;

  // The four waves take the four anchor parts of the SAME 64 groups; each wave gathers a
  // quarter of the operands of a step (one 16-byte and three 4-byte pieces per lane, exactly
  // four loads per wave and step so that the in-order vmcnt can be counted):
  //   wave 0: A0 W0 W1 W2 | wave 1: A1 B0 W3 W4 | wave 2: A2 B1 W5 W6 | wave 3: A3 B2 W7 (W7 again -> dummy)
  auto issue = [&](uint32_t ent, int stage) {
    const uint32_t gx = ent & 0xffffu, gy = ent >> 16;
    gptr_u8 drow = dplane + (size_t)(gy + kPadY + wave) * pitch + 4u * gx;  // row `wave`, sample x-8
    // window bits of samples x-4 .. x+7: bit kPadX + x - 4 = 4 * gx + 4 of the bit row
    gptr_u8 wbase = wplane + (size_t)(gy + kPadY) * wpitch + (((gx * 4u + 4u) >> 5) << 2);
    const uint32_t st = lds_address(&ring[stage]);
    const uint32_t oA = st + (uint32_t)wave * 1024u, oB = st + 4096u, oW = st + 4096u + 768u;
    dma16(drow + (wave == 0 ? 8 : 0), oA);
    const int k1 = wave == 0 ? 0 : -1, k2 = 2 * wave + 1, k3 = wave == 3 ? 7 : 2 * wave + 2;
    // piece 1: wave 0: W0; others: B[wave-1]
    dma4(wave == 0 ? wbase : drow + 16, wave == 0 ? oW : oB + (uint32_t)(wave - 1) * 256u);
    (void)k1;
    dma4(wbase + (size_t)(k2 >> 1) * wpitch + 4u * (k2 & 1), oW + (uint32_t)k2 * 256u);
    dma4(wbase + (size_t)(k3 >> 1) * wpitch + 4u * (k3 & 1), oW + (uint32_t)(wave == 3 ? 8 : k3) * 256u);
  };

  // the whole step loop is instantiated per anchor part (a dispatch inside the loop would merge
  // the 81 accumulators of the four variants after every step: 81 register copies per step)
  auto run = [&](auto part_tag) {
  for (uint32_t blo = lo; blo < hi; blo += kDEnt) {
    const uint32_t cnt = min((uint32_t)kDEnt, hi - blo);
    const int S = (int)((cnt + 63u) >> 6);
    __syncthreads();  // everybody is done with the previous block's entries (and all its loads have landed)
    for (uint32_t i = tid; i < cnt; i += 256) s_ent[i] = list[blo + i];
    __syncthreads();
    // lanes past the end of the list gather the last entry (valid memory) and skip the products
    auto entry = [&](int st) -> uint32_t { return s_ent[min((uint32_t)(st * 64 + lane), cnt - 1u)]; };
    for (int st = 0; st < kDStages - 1 && st < S; ++st) issue(entry(st), st % kDStages);
    for (int st = 0; st < S; ++st) {
      // this wave's loads of step st have landed when at most 4 * (steps issued after it) are outstanding
      const int ahead = min(S - 1, st + kDStages - 2) - st;
      if (ahead >= 4) wait_vmcnt<16>();
      else if (ahead == 3) wait_vmcnt<12>();
      else if (ahead == 2) wait_vmcnt<8>();
      else if (ahead == 1) wait_vmcnt<4>();
      else wait_vmcnt<0>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();  // ... and so have the other waves'; the buffer of step st-1 is free
      asm volatile("" ::: "memory");  // (a bare s_barrier: no fence, hence no vmcnt(0); keep LDS reads below it)
      if (st + kDStages - 1 < S) issue(entry(st + kDStages - 1), (st + kDStages - 1) % kDStages);
      const DenseStep &ops = ring[st % kDStages];
      if ((uint32_t)(st * 64 + lane) < cnt) {
        const uint32_t wshift = ((s_ent[st * 64 + lane] & 0xffffu) * 4u + 4u) & 31u;
        uint32_t D[kNumLags];
        {
          const uint32_t c0 = ops.A[0][lane][0], c1 = ops.A[0][lane][1], c2 = ops.A[0][lane][2];
          D[0] = c0;
          D[1] = alignbyte(c1, c0, 1);
          D[2] = alignbyte(c1, c0, 2);
          D[3] = alignbyte(c1, c0, 3);
          D[4] = c1;
          D[5] = alignbyte(c2, c1, 1);
          D[6] = alignbyte(c2, c1, 2);
        }
#pragma unroll
        for (int dy = 1; dy <= 3; ++dy) {
          const uint32_t e0 = ops.A[dy][lane][0], e1 = ops.A[dy][lane][1], e2 = ops.A[dy][lane][2],
                         e3 = ops.A[dy][lane][3], e4 = ops.B[dy - 1][lane];
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
        uint32_t wb[4];
#pragma unroll
        for (int dy = 0; dy <= 3; ++dy)
          wb[dy] = __builtin_amdgcn_alignbit(ops.W[2 * dy + 1][lane], ops.W[2 * dy][lane], wshift) & 0xfffu;
        partial_products<decltype(part_tag)::value>(acc, D, wb);
      }
    }
  }
  };
  if (wave == 0) run(std::integral_constant<int, 0>{});
  else if (wave == 1) run(std::integral_constant<int, 1>{});
  else if (wave == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});
  // cross-lane sums on 16-bit halves (a lane may exceed 2^31 / 64), 27 values a round
  {
    constexpr int CH = 27;
    static_assert(kPSub % CH == 0, "");
#pragma unroll
    for (int b0 = 0; b0 < kPSub; b0 += CH) {
      int lo16[CH], hi16[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        lo16[i] = acc[b0 + i] & 0xffff;
        hi16[i] = acc[b0 + i] >> 16;
      }
      wave_sums_dpp<CH>(lo16);
      wave_sums_dpp<CH>(hi16);
      if (lane == 63) {
#pragma unroll
        for (int i = 0; i < CH; ++i) red[wave][b0 + i] = ((long long)hi16[i] << 16) + lo16[i];
      }
    }
  }
  __syncthreads();
  unsigned long long *out = reinterpret_cast<unsigned long long *>(qp.paracc) + ((size_t)frame * 3 + c) * kPPart;
  for (int i = tid; i < kPPart; i += 256) {
    const long long v = red[i / kPSub][i % kPSub];
    if (v != 0) atomicAdd(&out[i], (unsigned long long)v);
  }
}

// ---------------------------------------------------------------------------------
// k3q_generic: the GENERIC list (areas touching a block outside int8, or MIX when the fast
// mixed path is off), plain int32 from the original planes.  Thread t owns up to two of the 350
// products:
//   (i, j):  sum_q W(q - c_i) d(q) d(q + c_j - c_i)        (i, y): ... d(q - c_i)
//   (i, L):  sum_p w(p) L(p) d(p + c_i);  (L,L), (L,y) likewise      [chroma]
// Adds straight into the record with int64 atomics (few areas), writes the block stats.
// grid = (chunks, nplanes, batch), block = 256.
// ---------------------------------------------------------------------------------
constexpr int kGW = kBlock + 12, kGH = kBlock + 6;  // d tile: cols -6..37, rows -3..34
__global__ __launch_bounds__(256) void k3q_generic(const FrameTable ft, Geom g, QParams qp,
                                                   uint8_t *__restrict__ records) {
  __shared__ int dt[kGH * kGW];        // d(q), x in -6..bw+5, y in -3..bh+2
  __shared__ uint8_t wt[kGH * kGW];    // w at the same positions
  __shared__ int lt[kBlock * kBlock];  // L(p) on the block proper
  __shared__ int red[4];
  const int c = blockIdx.y, frame = g.frame0 + (int)blockIdx.z;
  const int kind = c > 0 ? 1 : 0;
  const size_t lsel = ((size_t)frame * 2 + kind) * 3 + 2;
  const int nlist = (int)qp.counts[lsel];
  if ((int)blockIdx.x >= nlist) return;  // the common case: nothing deferred
  const uint32_t *list = qp.lists + lsel * g.nblocks;
  const FramePlanes fp = ft.f[frame];
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
  unsigned long long *ar = qp.ar3 ? reinterpret_cast<unsigned long long *>(qp.ar3) + ((size_t)frame * 3 + c) * kAr3
                                  : reinterpret_cast<unsigned long long *>(rec + g.off_ar[c]);
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
        wv = window_at(mask, g.nbw, g.nbh, bw, bh, pw, ph, X, Y, g.lag);
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
//   + the 324 masked products of k3_partial_dense.
// grid = (nplanes, batch), block = 256: 4 lanes-groups of 64 split the chunk range.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3q_reduce(Geom g, QParams qp, uint8_t *__restrict__ records) {
  const int c = blockIdx.x, frame = g.frame0 + (int)blockIdx.y;
  const bool chroma = c > 0;
  const int nc = kQN + (chroma ? 1 : 0);
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  long long *ar = qp.ar3 ? qp.ar3 + ((size_t)frame * 3 + c) * kAr3 : reinterpret_cast<long long *>(rec + g.off_ar[c]);
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
    const int h = (i < 12 ? i : 23 - i) % kPParts;
    int idx = 0;
    for (int a = 0; a < i; ++a)
      if (p_in_part(h, a)) idx += (kQN - a) + 1;
    const long long *t = par + h * kPSub + idx;
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

// ---------------------------------------------------------------------------------
// k3q_compact (lag 1 / 2): the lag's own system out of the lag-3 one.  The lag-L neighbourhood is a subset
// of the lag-3 neighbourhood and, with the windows built for lag L (block_window), entry (a, b) of the lag-L
// normal equations IS entry (i3(a), i3(b)) of the lag-3 ones: same samples, same products.
// grid = (nplanes, batch), block = 256.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int lag3_index(int a, int lag, int n) {  // coefficient a of lag `lag` (n of them) -> lag-3 index
  if (a >= n) return kQN;                                            // the chroma luma regressor
  const int row = 2 * lag + 1;
  const int y = a / row - lag, x = a % row - lag;                    // raster over (-lag..0, -lag..lag), causal part
  return (y + kQLag) * 7 + (x + kQLag);
}
__global__ __launch_bounds__(256) void k3q_compact(Geom g, QParams qp, uint8_t *__restrict__ records) {
  const int c = blockIdx.x, frame = g.frame0 + (int)blockIdx.y;
  const int nc = g.n + (c > 0), nc3 = kQN + (c > 0);
  long long *dst = reinterpret_cast<long long *>(records + (size_t)frame * g.rec_size + g.off_ar[c]);
  const long long *src = qp.ar3 + ((size_t)frame * 3 + c) * kAr3;
  for (int p = threadIdx.x; p < nc * nc; p += 256) {
    const int a = p / nc, b = p % nc;
    if (b < a) continue;  // (the host mirrors the upper triangle)
    dst[a * nc + b] = src[lag3_index(a, g.lag, g.n) * nc3 + lag3_index(b, g.lag, g.n)];
  }
  if (threadIdx.x < nc) dst[nc * nc + threadIdx.x] = src[nc3 * nc3 + lag3_index(threadIdx.x, g.lag, g.n)];
  if (threadIdx.x == 64) dst[nc * nc + nc] = src[nc3 * nc3 + nc3];
}

}  // namespace g1s

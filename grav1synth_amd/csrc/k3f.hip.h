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
// The pass is two launches of one kernel template, PL = 0 (luma) then PL = 1 (the two chroma planes): each
// keeps few enough registers (one resp. two accumulators, one plane kind's words in flight) for four waves
// to a SIMD, and a 16-20 KB tile set.  The luma launch leaves L behind as an int8 plane at chroma
// resolution (2 MB a 4K frame) for the chroma launch.
// Workgroup = 4 waves, unit = 2 adjacent blocks of a block row (k3m_units).  Per unit:
//   staging   luma: waves 0-2 take 6 row pairs each, a lane one 8-sample word of both rows (the two rows under a
//             4:2:0 chroma row: L needs no cross-lane traffic), source and denoised: four 16-byte loads.
//             Chroma: waves 0, 1 take the rows of Cb, waves 2, 3 those of Cr, a lane one word of one row.
//             A residual (or L) outside int8 flags the blocks whose tile holds it: they are left to the
//             exact int32 kernel (k3_ar_generic, `only` list).
//   multiply  wave w takes rows 8w .. 8w+7 of every luma block / its share of the chroma steps.
// Software pipeline: iteration k writes the tile copies of unit k, multiplies them, then turns the words of
// unit k+1 (requested a whole iteration earlier) into residual bytes behind its own MFMAs, and requests
// the words of unit k+2.  Two barriers per unit; accumulators stay in registers for the whole slice of the
// frame's unit lists the workgroup walks; one partial system per workgroup and plane (k3m_finish).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "k0.hip.h"
#include "k3m.hip.h"
#include "kernels.hip.h"

namespace g1s {

constexpr int kFWaves = 4, kFThreads = 64 * kFWaves;
#ifndef G1S_F_OCC
#define G1S_F_OCC 4  // waves per SIMD the kernel is compiled for (4 workgroups to a CU)
#endif

struct FParams {
  FrameTable ft;
  const uint32_t *units;  // [batch][nunits][kMUnitDwords]  (k3m_units)
  const uint32_t *unit_count;
  long long *partials;    // [batch][G][3][kMRec]
  int32_t *ustats;        // [batch][nunits][kMStatInts]  per-unit block statistics + deferral bits (k3m_finish)
  int nunits;
  uint8_t *lplane;        // [batch][lrows][lpitch]  L (sum of the co-located luma residuals) at chroma resolution, int8
  uint32_t lpitch, lframe_bytes;
  int frames, wgs;        // the launch: frames x workgroups per frame, as a 1-D grid (see the kernel)
  int wg_cap;             // workgroups per frame the partial systems are laid out for (>= wgs: the luma and the chroma launch may differ)
  int deal;               // units to workgroups: 0 round-robin, 1 contiguous runs
  int reuse;              // 1: (luma launch) the left halo word of a unit whose left neighbour was the unit before it in the run is not read
  const uint8_t *planes;  // SRC = 1: the int8 planes of the pixel pass K0 (k0.hip.h), [batch] x ps.frame_bytes
  PlaneSet ps;
  int dbg;                  // timing experiments (G1S_S_DBG, k3s.hip.h): bit 0 no global loads, 1 no residual arithmetic, 2 no statistics atomics, 3 no copy writes, 4 no multiplies, 5 no statistics / L stores, 6 no barrier in the loop; wrong results
  long long *phase_cycles;  // profiling aid (built with -DG1S_F_PHASES, run with G1S_F_PHASES=1): [workgroup][wave][8] cycles: tile copies, barrier 2, multiply, barrier 1, wait for the words, residuals, requests, stores; or null
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
  // (opaque to the optimiser: it would otherwise keep a 64-bit copy of the offset across the unit loop and lose the
  //  scalar-base + 32-bit-offset form of the load)
  asm volatile("" : "+v"(off));
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

constexpr int kFBiasY = 16 * 255, kFBiasC = 8 * 255;  // bias of a lane's sum of residuals: two rows / one row of 8 samples

// blocks whose tile holds word wd of a row (WB words to a block; the tile reaches one word into its neighbours)
__device__ __forceinline__ void f_flag_blocks(int *flags, int wd, int WB) {
  const int b = wd / WB;
  if (b < kMUnitBlocks) flags[b] = 1;
  if (wd - b * WB <= 1 && b >= 1) flags[b - 1] = 1;
}

// ---------------------------------------------------------------------------------
// k3f_fused<CBW, CBH, BPS, PL>: chroma block 32 >> xdec by 32 >> ydec (0, 0: luma only); PL = 0: the luma plane (and L),
// PL = 1: the chroma planes.  grid = (G, 1, batch), block = 256, dynamic LDS = f_lds_bytes(CBW, CBH, PL).
// ---------------------------------------------------------------------------------
template <int CBW, int CBH>
struct FShape {
  static constexpr bool CH = CBW != 0;
  static constexpr int CW_ = CH ? CBW : 16, CH_ = CH ? CBH : 16;
  // luma tile: rows -3 .. 31, samples -8 .. 71 of the chunk
  static constexpr int PY = m_pitch(32), WY = PY / 8, CSY = m_copy_stride(32, kBlock);
  static constexpr int PAIRS = (kBlock + 4) / 2;                                     // row pairs of the tile
  // ... per wave: as many as fit its 64 lanes.  The kernels are bound by the number of vector instructions the SIMDs issue, not
  // by a wave's latency: three full waves (and an idle one) issue a quarter less than four waves of five pairs
  static constexpr int PPJ = 64 / WY;
  static_assert(PPJ * kFWaves >= PAIRS, "luma row pairs: one job per wave");
  // chroma tiles: rows -3 .. CBH-1
  static constexpr int PC = m_pitch(CW_), WC = PC / 8, CSC = m_copy_stride(CW_, CH_);
  static constexpr int RC = CH_ + 3, RPW = 64 / WC;             // tile rows per plane / rows per wave and round
  // waves 0, 1 take Cb, waves 2, 3 Cr (the plane is uniform in a wave: scalar base addresses)
  static constexpr int CROUNDS = CH ? (RC + 2 * RPW - 1) / (2 * RPW) : 0;
  static constexpr int NL = CH ? CH_ * kMUnitBlocks * CW_ / 8 : 0;  // 8-byte words of the unit's L tile (<= 256)
};
// LDS map: PL = 0: [luma tile][zero block]; PL = 1: [Cb tile][Cr tile][pad][L tile][zero block]
__host__ __device__ constexpr int f_lds_tiles(int CBW, int CBH, int PL) {
  return PL == 0 ? m_tile_bytes(32, kBlock) : 2 * m_tile_bytes(CBW, CBH) + m_l_pad(CBW, CBH) + CBH * m_pitch(CBW);
}
__host__ __device__ constexpr int f_lds_bytes(int CBW, int CBH, int PL) { return f_lds_tiles(CBW, CBH, PL) + 16; }

__device__ __forceinline__ void f_residual(const uint32_t (&hs)[4], const uint32_t (&hv)[4], uint32_t (&d16)[4], uint32_t &mx, uint32_t &mn) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    d16[q] = pk_sub(hs[q], hv[q]);
    mx = pk_max(mx, d16[q]);
    mn = pk_min(mn, d16[q]);
  }
}

// SRC = 0: the words come from the source / denoised planes (the fused pass); SRC = 1: from the int8 residual and L planes
// the pixel pass K0 left behind (k0.hip.h), which also took the statistics and flagged the residuals outside int8 (k3m_units
// then lists no window for those blocks): staging is a copy
template <int CBW, int CBH, int BPS, int PL, int SRC = 0>
__global__ __launch_bounds__(kFThreads, G1S_F_OCC) void k3f_fused(Geom g, FParams fpar) {
  extern __shared__ __attribute__((aligned(16))) uint8_t m_smem[];
  using SH = FShape<CBW, CBH>;
  constexpr bool CH = SH::CH;
  constexpr bool LUMA = PL == 0, CHROMA = PL == 1, RAW = SRC == 0;
  static_assert(LUMA || CH, "the chroma launch needs chroma planes");
  constexpr int CW_ = SH::CW_, CH_ = SH::CH_, CROUNDS = CHROMA ? SH::CROUNDS : 0, NCR = CROUNDS > 0 ? CROUNDS : 1;
  constexpr int ZOFF = f_lds_tiles(CBW, CBH, PL);
  constexpr int OFF_CB = 0, OFF_CR = m_tile_bytes(CW_, CH_), OFF_L = 2 * m_tile_bytes(CW_, CH_) + m_l_pad(CW_, CH_);
  // block statistics, ONE 64-bit LDS atomic a lane (they all hit the same few words): [unit parity][plane][block]
  //   sum d^2 << 37 | sum src8 << 19 | sum (d + bias): every contributing lane adds its bias, their number is fixed
  __shared__ unsigned long long s_sum[2][3][kMUnitBlocks];
  __shared__ int s_bad[2][2][kMUnitBlocks];  // [unit parity][kind][block]
  __shared__ int s_ring[4][kMStatInts];      // statistics records on their way out (wave 3)
  __shared__ uint2 s_L[2][PL == 0 && SH::NL > 0 ? SH::NL : 1];  // luma launch: the L tile of a unit on its way to the L plane (wave 3)

  // Workgroup b of the 1-D grid runs on XCD b % 8, and workgroups b, b + 256, ... share a CU (observed; speed only).  With
  // frame = b % frames (frames a multiple of 8, or few), the workgroups on a CU work on ONE frame -- few distinct pages under
  // the CU's address translation cache: issuing a load costs hundreds of cycles when it misses there -- and a frame's
  // workgroups share an XCD, i.e. the L2 the 128-byte lines under their units' halo columns are read through.
  const int G = fpar.wgs, frame = g.frame0 + (int)blockIdx.x % fpar.frames, wg = (int)blockIdx.x / fpar.frames;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the frame's two unit lists are dealt round-robin to its workgroups: adjacent units at about the same time
  const int nx = G, jx = wg;
  const uint32_t cnt_g = fpar.unit_count[2 * frame], cnt_p = fpar.unit_count[2 * frame + 1];
  const uint32_t ustride = fpar.deal ? 1u : (uint32_t)nx;
  auto share = [&](uint32_t cnt, uint32_t &first, int &n) {  // positions first, first + ustride, ... (n of them) of a list of cnt
    if (fpar.deal) {  // a contiguous run: the line under a unit's right halo is the next unit's own
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
  // list position of this workgroup's k-th unit: its plain units first (that list grows from the back of the array)
  auto upos = [&](int k) {
    return k < n_p ? (uint32_t)fpar.nunits - 1u - (first_p + (uint32_t)k * ustride) : first_g + (uint32_t)(k - n_p) * ustride;
  };
  const uint32_t *units = fpar.units + (size_t)frame * fpar.nunits * kMUnitDwords;
  int32_t *ustats = fpar.ustats + (size_t)frame * fpar.nunits * kMStatInts;
  uint8_t *lplane = fpar.lplane + (size_t)frame * fpar.lframe_bytes;
  const FramePlanes fp = fpar.ft.f[frame];
  // (the chroma block shape IS the subsampling: known at compile time -- the registers and branches of the other formats go)
  constexpr int sx = CH && CBW == 16 ? 1 : 0, sy = CH && CBH == 16 ? 1 : 0;
  const int cpw = g.W >> sx, cph = g.H >> sy;
  const int sbps = f_bps<BPS>(g.src_bps), dbps = f_bps<BPS>(g.den_bps);

  // ---- this lane's operand address inside a tile (k3m.hip.h) ----
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
  // luma: pair ypair = tile rows 2 ypair - 1, 2 ypair (= block rows 2 ypair - 4, 2 ypair - 3)
  const int ypl = lane / SH::WY, ywd = lane - ypl * SH::WY;
  const int ypair = wave * SH::PPJ + ypl;
  const bool y_wave = wave * SH::PPJ < SH::PAIRS;  // this wave has luma row pairs
  const bool yon = LUMA && ypl < SH::PPJ && ypair < SH::PAIRS;
  const int ytr0 = yon ? 2 * ypair - 1 : -9;
  // chroma: waves 0, 1 stage Cb, waves 2, 3 Cr; round k, tile row (2 k + (wave & 1)) * RPW + lane / WC
  const int cwd = lane % SH::WC;
  const int cplane = 1 + (wave >> 1);
  const uint8_t *c_src = cplane == 2 ? fp.src[2] : fp.src[1], *c_den = cplane == 2 ? fp.den[2] : fp.den[1];
  const uint32_t c_sst = cplane == 2 ? fp.src_stride[2] : fp.src_stride[1], c_dst = cplane == 2 ? fp.den_stride[2] : fp.den_stride[1];
  int cpl[NCR], ctr[NCR];  // plane (1, 2; 0: the lane is idle in this round), tile row
#pragma unroll
  for (int k = 0; k < CROUNDS; ++k) {
    const int rr = (2 * k + (wave & 1)) * SH::RPW + lane / SH::WC;
    const bool on = lane / SH::WC < SH::RPW && rr < SH::RC;
    cpl[k] = on ? cplane : 0;
    ctr[k] = on ? rr : 0;
  }
  // the lane's words: byte offsets from the unit's origin (tile row 0, sample -8 of the chunk), kept across the units
  // (chroma launch; the luma launch has no registers to spare and works them out at each request)
  uint32_t cso[NCR], cdo[NCR];
#pragma unroll
  for (int k = 0; k < CROUNDS; ++k) {
    cso[k] = (uint32_t)ctr[k] * c_sst + (uint32_t)(8 * cwd * sbps);
    cdo[k] = (uint32_t)ctr[k] * c_dst + (uint32_t)(8 * cwd * dbps);
  }
  // planes whose rows are 16-byte aligned take the vector loads; a chunk that reaches over the right plane edge
  // inside a word (W % 8 != 0) and unaligned planes go sample by sample
  const bool vec_all = (g.vec_mask & (LUMA ? 0x09 : 0x36)) == (LUMA ? 0x09 : 0x36);

  v16i32 accA, accB;  // luma launch: accA; chroma launch: Cb, Cr
#pragma unroll
  for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0;

  // ---- this workgroup's units: their entries parked in LDS (.w: the luma launch's deferral bits, for the chroma launch) ----
  __shared__ uint4 s_ent[kMMaxUnits];
  const int nmine = n_p + n_g;  // (<= kMMaxUnits: the host sizes G for it)
  if (tid < nmine) {
    uint4 e = *reinterpret_cast<const uint4 *>(units + (size_t)upos(tid) * kMUnitDwords);
    // bit 31 of .x: the unit before this one in the workgroup's sequence is its left neighbour in the block row (the usual
    // case in a contiguous run of the lists).  The three samples left of the unit's columns are then the last samples of
    // that unit's words -- still in this wave's registers, WY - 3 lanes away -- and the left halo word of every row is not
    // read from memory: a row of a luma unit is two 128-byte lines instead of three (the memory pipe of a CU takes a few
    // cycles per LINE a load touches, however little of it is used: profiles/r02_k3f_counters.txt).
    if (RAW && LUMA && fpar.reuse && tid > 0) {
      const uint32_t a = units[(size_t)upos(tid - 1) * kMUnitDwords] & 0xffffffu, here = e.x & 0xffffffu;
      if ((a & 0xfff000u) == (here & 0xfff000u) && (a & 0xfffu) + 1u == (here & 0xfffu)) e.x |= 1u << 31;
    }
    if (CHROMA && RAW) e.w = (uint32_t)ustats[(size_t)upos(tid) * kMStatInts + 14];
    if (!RAW) e.w = 0u;
    s_ent[tid] = e;
  }
  if (tid < 4) reinterpret_cast<uint32_t *>(m_smem + ZOFF)[tid] = 0u;
  if (tid < 2 * 3 * kMUnitBlocks) (&s_sum[0][0][0])[tid] = 0ull;
  if (tid < 2 * 2 * kMUnitBlocks) (&s_bad[0][0][0])[tid] = 0;
  __syncthreads();

  // ---- software pipeline over the units k = 0 .. nmine - 1 ----
  //   request(k)  the plane words of unit k -> raw registers (global loads, no wait)
  //   phase A(k)  raw words -> residual words, (luma launch) L -> its plane, block statistics and out-of-int8 flags (LDS, parity k & 1)
  //   phase B(k)  residual words -> the 7 shifted tile copies in LDS (needs the tiles free: after barrier 1)
  //   multiply(k) after barrier 2
  // Iteration k runs B(k), multiply(k), A(k + 1), request(k + 2): A's arithmetic issues behind the wave's own MFMAs,
  // and a request has a whole iteration to land.
  u32x4 ys_[2], yd_[2];      // luma raw words: two rows, source and denoised
  u32x4 cs_[NCR], cd_[NCR];  // chroma raw words: one row a round
  uint2 lraw = make_uint2(0u, 0u), Lk = make_uint2(0u, 0u);  // chroma launch: this thread's word of the L tile (requested / of the unit being staged)
  uint32_t Dy[2][2], Dc[NCR][2];  // residual bytes of the luma rows / of the chroma rows
  uint32_t DlastY[2] = {0u, 0u};  // luma launch: the last dword of the words of the unit before
  bool carry_y = false;           // ... its last word held a residual outside int8
  uint32_t Lw00 = 0, Lw01 = 0, Lw10 = 0, Lw11 = 0;  // luma launch: the L bytes under the lane's rows (scalars: an array the lambdas share goes to scratch)
  const bool l_on = CHROMA && tid < SH::NL;
  constexpr int LWR = kMUnitBlocks * CW_ / 8;  // 8-byte words of an L tile row
  const int l_row = tid / LWR, l_wd = tid - l_row * LWR;
  auto request = [&](int k) __attribute__((always_inline)) {
    const uint32_t ex = __builtin_amdgcn_readfirstlane(s_ent[k].x);
    const int bx0 = kMUnitBlocks * (int)(ex & 0xfffu), by = (int)((ex >> 12) & 0xfffu);
    const int X0y = bx0 * 32 - 8, Y0y = by * kBlock - 3, X0c = bx0 * CW_ - 8, Y0c = by * CH_ - 3;
    const bool aL = (ex >> 31) != 0;  // the left halo word is the left neighbour's own last word: not read
    if constexpr (!RAW) {
      // K0's planes: sample (x, y) of a residual plane at byte (y + 3) * pitch + 8 + x (zero padding around the plane); L without padding
      const uint8_t *fpl = fpar.planes + (size_t)frame * fpar.ps.frame_bytes;
      if constexpr (LUMA) {
        if (y_wave) {
          const uint32_t pitch = fpar.ps.pitch[0];
          const uint8_t *b = fpl + fpar.ps.off_d[0] + (size_t)(by * kBlock) * pitch + bx0 * 32;
          const bool cok = (uint32_t)(bx0 * 32 + 8 * ywd + 8) <= pitch;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            uint2 w = make_uint2(0u, 0u);
            if (cok && ytr0 + r >= 0) w = *reinterpret_cast<const uint2 *>(b + (uint32_t)(ytr0 + r) * pitch + 8u * (uint32_t)ywd);
            ys_[r].x = w.x;
            ys_[r].y = w.y;
          }
        }
      } else {
        const uint32_t pitch = fpar.ps.pitch[1];
        const uint8_t *b = fpl + fpar.ps.off_d[cplane] + (size_t)(by * CH_) * pitch + bx0 * CW_;
        const bool cok = (uint32_t)(bx0 * CW_ + 8 * cwd + 8) <= pitch;
#pragma unroll
        for (int q = 0; q < CROUNDS; ++q) {
          uint2 w = make_uint2(0u, 0u);
          if (cok && cpl[q]) w = *reinterpret_cast<const uint2 *>(b + (uint32_t)ctr[q] * pitch + 8u * (uint32_t)cwd);
          cs_[q].x = w.x;
          cs_[q].y = w.y;
        }
        lraw = make_uint2(0u, 0u);
        if (l_on && (uint32_t)(bx0 * CW_ + 8 * l_wd + 8) <= fpar.ps.lpitch)
          lraw = *reinterpret_cast<const uint2 *>(fpl + fpar.ps.off_l + (size_t)(by * CH_ + l_row) * fpar.ps.lpitch + bx0 * CW_ + 8 * l_wd);
      }
      return;
    }
    if constexpr (CHROMA) {
      if (l_on) lraw = *reinterpret_cast<const uint2 *>(lplane + (size_t)(by * CH_ + l_row) * fpar.lpitch + bx0 * CW_ + 8 * l_wd);
    }
    const bool slow = !vec_all || (LUMA ? ((g.W & 7) != 0 && X0y + SH::PY > g.W) : ((cpw & 7) != 0 && X0c + SH::PC > cpw));
    if (__builtin_expect(slow, 0)) {
      if constexpr (LUMA) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          ys_[r] = f_load_slow(fp.src[0], fp.src_stride[0], sbps, X0y + 8 * ywd, ytr0 + r >= 0 ? Y0y + ytr0 + r : -1, g.W, g.H);
          yd_[r] = f_load_slow(fp.den[0], fp.den_stride[0], dbps, X0y + 8 * ywd, ytr0 + r >= 0 ? Y0y + ytr0 + r : -1, g.W, g.H);
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
      // (pointers to the unit's origin: not dereferenced where the origin lies outside the plane)
      const uint8_t *sb = fp.src[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.src_stride[0] + (ptrdiff_t)X0y * sbps);
      const uint8_t *db = fp.den[0] + ((ptrdiff_t)Y0y * (ptrdiff_t)fp.den_stride[0] + (ptrdiff_t)X0y * dbps);
      // a unit whose tile lies inside the plane (all but the frame's border units) needs no per-lane bounds
      const bool inside = X0y >= 0 && X0y + SH::PY <= g.W && Y0y >= 0 && Y0y + kBlock + 3 <= g.H;
      bool xok = inside || (X0y + 8 * ywd >= 0 && X0y + 8 * ywd + 8 <= g.W);
      xok = xok && !(aL && ywd == 0);
      // (the lane's offsets from the unit's origin are worked out here from values the optimiser cannot see through:
      //  hoisted out of the unit loop they cost registers -- and a spill, whose reload from scratch waits for every
      //  load in flight)
      int l_tr = ytr0, l_w = ywd;
      asm volatile("" : "+v"(l_tr), "+v"(l_w));
      if (inside) {
        // Every lane loads, no predicate (a wave is as fast as its instruction count: a predicated load is a compare, an
        // exec save, a branch, the load, an exec restore and the zeroes of the other arm).  Tile row -1 and the lanes past
        // the last row pair read row 0 -- nothing uses what they get; the left halo lane of a unit whose neighbour holds
        // its samples re-reads word 1 (the same 128-byte line: no halo line is touched).
        if (aL && ywd == 0) l_w = 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          ys_[r] = f_load<BPS>(sb, (uint32_t)max(l_tr + r, 0) * fp.src_stride[0] + (uint32_t)(8 * l_w * sbps), g.src_bps, true);
          yd_[r] = f_load<BPS>(db, (uint32_t)max(l_tr + r, 0) * fp.den_stride[0] + (uint32_t)(8 * l_w * dbps), g.den_bps, true);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int Y = Y0y + ytr0 + r;
          const bool ok = xok && ytr0 + r >= 0 && Y >= 0 && Y < g.H;
          ys_[r] = f_load<BPS>(sb, (uint32_t)max(l_tr + r, 0) * fp.src_stride[0] + (uint32_t)(8 * l_w * sbps), g.src_bps, ok);
          yd_[r] = f_load<BPS>(db, (uint32_t)max(l_tr + r, 0) * fp.den_stride[0] + (uint32_t)(8 * l_w * dbps), g.den_bps, ok);
        }
      }
    }
    if constexpr (CHROMA) {
      const uint8_t *sb = c_src + ((ptrdiff_t)Y0c * (ptrdiff_t)c_sst + (ptrdiff_t)X0c * sbps);
      const uint8_t *db = c_den + ((ptrdiff_t)Y0c * (ptrdiff_t)c_dst + (ptrdiff_t)X0c * dbps);
      const bool inside = X0c >= 0 && X0c + SH::PC <= cpw && Y0c >= 0 && Y0c + CH_ + 3 <= cph;
      bool xok = inside || (X0c + 8 * cwd >= 0 && X0c + 8 * cwd + 8 <= cpw);
      if (inside) {  // (every lane loads: the idle lanes of a round read row 0, unused)
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
  const int y_xw = 8 * (ywd - 1), y_bq = (y_xw >> 5) & 1;     // luma word: first sample of the chunk, block
  const int c_xw = 8 * (cwd - 1), c_bq = (c_xw / CW_) & 1;   // chroma word
  auto phase_a = [&](int k) __attribute__((always_inline)) {
    const int par = k & 1;
    if constexpr (!RAW) {  // the words ARE the residual bytes
      if (LUMA && y_wave) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          Dy[r][0] = ys_[r].x;
          Dy[r][1] = ys_[r].y;
        }
      }
      if constexpr (CHROMA) Lk = lraw;
#pragma unroll
      for (int q = 0; q < CROUNDS; ++q) {
        Dc[q][0] = cs_[q].x;
        Dc[q][1] = cs_[q].y;
      }
      (void)par;
      return;
    }
    if (LUMA && y_wave) {
      // ---- luma: residuals of the two rows, their statistics, L ----
      uint32_t mx = 0, mn = 0, lmx = 0, lmn = 0, keep16[4] = {0, 0, 0, 0};
      int sd = 0, sd2 = 0, ls = 0;
      const uint32_t ex = __builtin_amdgcn_readfirstlane(s_ent[k].x);
      const int bx0 = kMUnitBlocks * (int)(ex & 0xfffu), by = (int)((ex >> 12) & 0xfffu);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int tr = ytr0 + r;
        uint32_t hs[4], hv[4], d16[4];
        f_narrow<BPS>(ys_[r], g.src_bps, g.src_shift, hs);
        f_narrow<BPS>(yd_[r], g.den_bps, g.den_shift, hv);
        f_residual(hs, hv, d16, mx, mn);
        Dy[r][0] = pk_bytes(d16[0], d16[1]);
        Dy[r][1] = pk_bytes(d16[2], d16[3]);
        if (tr >= 3 && y_interior) {  // the block proper: its statistics (int8 arithmetic: a block that holds a residual outside
                                      // int8 is redone by the exact kernel, statistics included)
          sd = __builtin_amdgcn_sdot4((int)Dy[r][0], 0x01010101, sd, false);
          sd = __builtin_amdgcn_sdot4((int)Dy[r][1], 0x01010101, sd, false);
          sd2 = __builtin_amdgcn_sdot4((int)Dy[r][0], (int)Dy[r][0], sd2, false);
          sd2 = __builtin_amdgcn_sdot4((int)Dy[r][1], (int)Dy[r][1], sd2, false);
          ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[0], hs[1]), 0u, (uint32_t)ls);
          ls = (int)__builtin_amdgcn_sad_u8(pk_bytes(hs[2], hs[3]), 0u, (uint32_t)ls);
        }
        if constexpr (CH) {
          // ---- the chroma regressor L (chroma resolution) -> its plane, for the chroma launch ----
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
          if (have && y_interior) {
            if (sx) {
              const uint32_t p0 = ((uint32_t)pk_dot(v[0], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[1], 0x00010001u, 0) << 16);
              const uint32_t p1 = ((uint32_t)pk_dot(v[2], 0x00010001u, 0) & 0xffffu) | ((uint32_t)pk_dot(v[3], 0x00010001u, 0) << 16);
              lmx = pk_max(lmx, pk_max(p0, p1));
              lmn = pk_min(lmn, pk_min(p0, p1));
              const uint32_t lw = pk_bytes(p0, p1);
              if (r == 0) Lw00 = lw;
              else Lw10 = lw;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                lmx = pk_max(lmx, v[q]);
                lmn = pk_min(lmn, v[q]);
              }
              const uint32_t la = pk_bytes(v[0], v[1]), lb = pk_bytes(v[2], v[3]);
              if (r == 0) {
                Lw00 = la;
                Lw01 = lb;
              } else {
                Lw10 = la;
                Lw11 = lb;
              }
            }
          }
        }
      }
      // (every interior lane of the 16 row pairs inside the block rows adds, zeros included: the bias total is a constant)
      if (y_interior && ytr0 >= 3)
        atomicAdd(&s_sum[par][0][y_bq],
                  ((unsigned long long)(uint32_t)sd2 << 37) | ((unsigned long long)(uint32_t)ls << 19) | (unsigned long long)(uint32_t)(sd + kFBiasY));
      // (a residual outside int8 flags the blocks whose tile holds it; with the left halo word unread, the last word of the
      //  unit before -- carried -- flags this unit's first block)
      const bool badw = yon && range_bad(mx, mn);
      if (badw) f_flag_blocks(&s_bad[par][0][0], ywd, 4);
      if ((ex >> 31) && carry_y) s_bad[par][0][0] = 1;
      carry_y = badw && ywd == SH::WY - 2;
      if (CH && y_interior && range_bad(lmx, lmn)) s_bad[par][1][y_bq] = 1;
    }
    // ---- chroma ----
    if constexpr (CHROMA) Lk = lraw;
#pragma unroll
    for (int q = 0; q < CROUNDS; ++q) {
      const int c = cpl[q];
      uint32_t hs[4], hv[4], d16[4], mx = 0, mn = 0;
      f_narrow<BPS>(cs_[q], g.src_bps, g.src_shift, hs);
      f_narrow<BPS>(cd_[q], g.den_bps, g.den_shift, hv);
      f_residual(hs, hv, d16, mx, mn);
      Dc[q][0] = pk_bytes(d16[0], d16[1]);
      Dc[q][1] = pk_bytes(d16[2], d16[3]);
      if (c && ctr[q] >= 3 && c_interior) {
        int sd = __builtin_amdgcn_sdot4((int)Dc[q][0], 0x01010101, 0, false);
        sd = __builtin_amdgcn_sdot4((int)Dc[q][1], 0x01010101, sd, false);
        int sd2 = __builtin_amdgcn_sdot4((int)Dc[q][0], (int)Dc[q][0], 0, false);
        sd2 = __builtin_amdgcn_sdot4((int)Dc[q][1], (int)Dc[q][1], sd2, false);
        atomicAdd(&s_sum[par][c][c_bq], ((unsigned long long)(uint32_t)sd2 << 37) | (unsigned long long)(uint32_t)(sd + kFBiasC));
      }
      if (c && range_bad(mx, mn)) f_flag_blocks(&s_bad[par][1][0], cwd, CW_ / 8);
    }
  };
  // the L words of unit k -> the L plane (luma launch).  A global store costs the staging waves ~500 cycles a unit at the
  // memory pipe's door (profiles/r02_k3f_counters.txt): they leave the words in an LDS tile (parity k & 1) and wave 3,
  // which stages no rows, stores the tile -- one 8-byte word a lane -- behind barrier 1 of the unit's own iteration.
  auto export_l = [&](int k) __attribute__((always_inline)) {
    if constexpr (LUMA && CH && RAW) {
      if (y_wave && y_interior) {
        uint8_t *tile = reinterpret_cast<uint8_t *>(&s_L[k & 1][0]);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int tr = ytr0 + r;
          const bool have = sy ? (r == 1 && tr >= 4) : tr >= 3;
          const int cy = sy ? (tr - 4) >> 1 : tr - 3;
          if (have) {
            uint8_t *lp = tile + cy * (kMUnitBlocks * CW_) + (y_xw >> sx);
            const uint32_t la = r == 0 ? Lw00 : Lw10, lb = r == 0 ? Lw01 : Lw11;
            if (sx) *reinterpret_cast<uint32_t *>(lp) = la;
            else *reinterpret_cast<uint2 *>(lp) = make_uint2(la, lb);
          }
        }
      }
    }
  };
  auto flush_l = [&](int k) __attribute__((always_inline)) {
    if constexpr (LUMA && CH && RAW) {
      if (wave == kFWaves - 1) {
        const uint32_t ex = __builtin_amdgcn_readfirstlane(s_ent[k].x);
        const int bx0 = kMUnitBlocks * (int)(ex & 0xfffu), by = (int)((ex >> 12) & 0xfffu);
        constexpr int LW = kMUnitBlocks * CW_ / 8;  // 8-byte words of an L tile row
#pragma unroll
        for (int w0 = 0; w0 < SH::NL; w0 += 64) {
          const int w = w0 + lane, row = w / LW, wd = w - row * LW;
          if (w < SH::NL)
            *reinterpret_cast<uint2 *>(lplane + (size_t)(by * CH_ + row) * fpar.lpitch + bx0 * CW_ + 8 * wd) = s_L[k & 1][w];
        }
      }
    }
  };
  // PLAIN: every window of the unit is its whole block (k3m_units): no column masks, every word is written
  // aL: the unit before is the left neighbour -- its last dword is the dword left of this unit's words (WY - 3 lanes away:
  // the same row pair, word 8)
  auto phase_b = [&](auto plain_tag, const uint32_t (&wins)[4], bool aL) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    if (LUMA && y_wave) {
      // every word of a block that is multiplied is written, the fully masked ones (columns past a window that ends at the
      // plane's right edge) as zeros: the multiplies read all 32 positions of the block's rows
      uint2 cm = make_uint2(0u, 0u);
      const uint32_t wsel = y_bq ? wins[1] : wins[0];
      const bool wr = y_interior && (PLAIN || ((wsel >> 15) & 1u) != 0);
      if (y_interior) cm = PLAIN ? make_uint2(~0u, ~0u) : m_colmask8(m_unpack(wsel, g.lag), y_xw - 32 * y_bq);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int tr = ytr0 + r;
        uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Dy[r][1], 0x138, 0xf, 0xf, true);  // wave_shr:1
        const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Dy[r][0], 0x130, 0xf, 0xf, true);  // wave_shl:1
        if (aL) {
          const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + SH::WY - 3), (int)DlastY[r]);
          if (ywd == 1) prev1 = v;
        }
        DlastY[r] = Dy[r][1];
        if (tr >= 0 && wr) m_write_copies<!PLAIN>(m_smem + tr * SH::PY + y_xw, SH::CSY, prev1, Dy[r][0], Dy[r][1], next0, cm);
      }
    }
    if constexpr (CHROMA) {
      if (l_on) {  // the unit's L tile: this thread's word, under the window columns of its chroma block
        const int lb = (8 * l_wd / CW_) & 1;
        const uint2 lm = PLAIN ? make_uint2(~0u, ~0u) : m_colmask8(m_unpack(lb ? wins[3] : wins[2], g.lag), 8 * l_wd - CW_ * lb);
        *reinterpret_cast<uint2 *>(m_smem + OFF_L + l_row * SH::PC + 8 * l_wd) = make_uint2(Lk.x & lm.x, Lk.y & lm.y);
      }
    }
#pragma unroll
    for (int q = 0; q < CROUNDS; ++q) {
      const int c = cpl[q];
      uint2 cm = make_uint2(0u, 0u);
      const uint32_t wsel = c_bq ? wins[3] : wins[2];
      const bool wr = c_interior && c && (PLAIN || ((wsel >> 15) & 1u) != 0);
      if (c_interior && c) cm = PLAIN ? make_uint2(~0u, ~0u) : m_colmask8(m_unpack(wsel, g.lag), c_xw - CW_ * c_bq);
      const uint32_t prev1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Dc[q][1], 0x138, 0xf, 0xf, true);  // wave_shr:1
      const uint32_t next0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)Dc[q][0], 0x130, 0xf, 0xf, true);  // wave_shl:1
      if (wr)
        m_write_copies<!PLAIN>(m_smem + (c == 2 ? OFF_CR : OFF_CB) + ctr[q] * SH::PC + c_xw, SH::CSC, prev1, Dc[q][0], Dc[q][1], next0, cm);
    }
  };

#ifdef G1S_F_PHASES
  long long t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = fpar.phase_cycles ? clock64() : 0;
  auto stamp = [&](int ph) {
    if (fpar.phase_cycles) {
      const long long t = clock64();
      t_ph[ph] += t - t_last;
      t_last = t;
    }
  };
#else
  auto stamp = [](int) {};
#endif
  if (nmine > 0) {
    request(0);
    phase_a(0);
    if (nmine > 1) request(1);
    export_l(0);
  }
  // units [k0, k1) of this workgroup's sequence; two calls (plain units, then the others) are ONE pipeline: the residuals and
  // requests a unit of the first loop prepares belong to units of the second
  auto run = [&](auto plain_tag, int k0, int k1) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    for (int k = k0; k < k1; ++k) {
      const int par = k & 1;
      const uint4 e0 = s_ent[k];
      const uint32_t ey = __builtin_amdgcn_readfirstlane(e0.y), ez = __builtin_amdgcn_readfirstlane(e0.z);
      const uint32_t ex0 = __builtin_amdgcn_readfirstlane(e0.x);
      const uint32_t fbits = PLAIN ? (1u << kMUnitBlocks) - 1u : ex0 >> 24;
      const uint32_t lbad = CHROMA ? __builtin_amdgcn_readfirstlane(e0.w) >> kMUnitBlocks : 0u;  // L outside int8 (luma launch)
      const uint32_t wins[4] = {ey & 0xffffu, ey >> 16, ez & 0xffffu, ez >> 16};  // luma block 0, 1; chroma block 0, 1
      __syncthreads();  // the previous unit's tiles are no longer read
      stamp(3);
      phase_b(plain_tag, wins, (ex0 >> 31) != 0);
      flush_l(k);
      // (the sums and flags of the unit before this one: read in its multiply phase, written again two units on)
      if (tid >= 64 && tid < 64 + 3 * kMUnitBlocks) (&s_sum[par ^ 1][0][0])[tid - 64] = 0ull;
      else if (tid >= 128 && tid < 128 + 2 * kMUnitBlocks) (&s_bad[par ^ 1][0][0])[tid - 128] = 0;
      stamp(0);
      __syncthreads();
      stamp(1);
#ifdef G1S_DBG_SALU
      {  // issue-rate probe: G1S_DBG_SALU harmless scalar instructions per unit
        int x_ = __builtin_amdgcn_readfirstlane(lane & 0);
#pragma unroll
        for (int q_ = 0; q_ < G1S_DBG_SALU; ++q_) asm volatile("s_add_i32 %0, %0, 1" : "+s"(x_));
        asm volatile("" ::"s"(x_));
      }
#endif
      // ------------------------------- multiply -------------------------------
      uint32_t defer = 0;
#pragma unroll
      for (int b = 0; b < kMUnitBlocks; ++b) {
        const bool flat_b = ((fbits >> b) & 1u) != 0;
        if constexpr (LUMA) {
          const MWin wy = m_unpack(wins[b], g.lag);
          constexpr int RPY = kBlock / kFWaves;
          if (CH && flat_b && __builtin_amdgcn_readfirstlane(s_bad[par][1][b])) defer |= 1u << (kMUnitBlocks + b);  // L
          if (flat_b && __builtin_amdgcn_readfirstlane(s_bad[par][0][b])) {
            defer |= 1u << b;  // (any flat block: the exact kernel redoes its statistics too)
          } else if (PLAIN || wy.go) {
            m_rows_one<RPY, SH::PY>(accA, m_smem, base_luma + 32 * b, PLAIN ? ~0u : m_rowmask(wy.ys, wy.ye) >> (wave * RPY), ZOFF);
          }
        } else {
          const MWin wc = m_unpack(wins[kMUnitBlocks + b], g.lag);
          constexpr int RPC = CH_ / kFWaves;
          if (flat_b && (__builtin_amdgcn_readfirstlane(s_bad[par][1][b]) || ((lbad >> b) & 1u))) {
            defer |= (1u << (kMUnitBlocks + b)) | (1u << (2 * kMUnitBlocks + b));  // (both chroma planes: k3m_finish reads them per plane)
          } else if (PLAIN || wc.go) {
            const uint32_t rm = PLAIN ? ~0u : m_rowmask(wc.ys, wc.ye) >> (wave * RPC);
            if constexpr (CW_ == 32) m_rows_two<RPC, SH::PC>(accA, accB, m_smem, addr_cb + CW_ * b, addr_cr + CW_ * b, rm, ZOFF);
            else m_steps_two<RPC / 2, SH::PC>(accA, accB, m_smem, addr_cb + CW_ * b, addr_cr + CW_ * b, PLAIN ? ~0u : rm >> h, ZOFF);
            (void)rm;
          }
        }
      }
      stamp(2);
      // ---- the next unit's words have had this whole iteration to land: their arithmetic runs behind the multiplies ----
      if (k + 1 < nmine) {
#ifdef G1S_F_PHASES
        if (fpar.phase_cycles) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          stamp(4);
        }
#endif
        phase_a(k + 1);
#ifdef G1S_F_PHASES
        if (fpar.phase_cycles) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          stamp(5);
        }
#endif
        if (k + 2 < nmine) request(k + 2);
        stamp(6);
        export_l(k + 1);
      }
      // ---- the unit's statistics record (k3m_finish scatters it): this launch's entries.  A global store costs the wave
      // that issues it a few hundred cycles at the memory pipe's door, and the slowest wave sets the workgroup's pace: the
      // records go through a four-unit ring in LDS and leave four at a time, from wave 3 (which stages no luma rows) ----
      if (RAW && wave == kFWaves - 1) {
        // entry 7 b + {0: luma sum d, 1: sum d^2, 2: sum src8, 3 / 4: Cb sum d / sum d^2, 5 / 6: Cr}; 14 / 15: the deferral
        // bits of the luma / chroma launch
        auto mine_entry = [](int t) {
          const int b = t >= 7 ? 1 : 0, e = t - 7 * b, c = e < 3 ? 0 : (e < 5 ? 1 : 2);
          return t < 14 ? (LUMA ? c == 0 : c != 0) : t == 14 + PL;
        };
        if (lane < kMStatInts && mine_entry(lane)) {
          const int b = lane >= 7 ? 1 : 0, e = lane - 7 * b, c = e < 3 ? 0 : (e < 5 ? 1 : 2), f = e < 3 ? e : (e - 3) & 1;
          int val = (int)defer;
          if (lane < 14) {
            const unsigned long long pk = s_sum[par][c][b];
            // contributing lanes per block: luma 16 row pairs x 4 words, chroma CBH rows x CBW / 8 words
            const int bias = c == 0 ? kFBiasY * 16 * 4 : kFBiasC * CH_ * (CW_ / 8);
            if (f == 1) val = (int)(pk >> 37);
            else if (f == 2) val = (int)((pk >> 19) & 0x3ffffu);
            else val = (int)(c == 0 ? (pk & 0x7ffffu) : (pk & 0x1fffffffffull)) - bias;
          }
          s_ring[k & 3][lane] = val;
        }
        if ((k & 3) == 3 || k == nmine - 1) {  // (the wave's own LDS writes above are ordered before these reads)
          const int first = k & ~3, u = lane >> 4, e = lane & 15;
          if (first + u <= k && mine_entry(e)) ustats[(size_t)upos(first + u) * kMStatInts + e] = s_ring[u][e];
        }
      }
      stamp(7);  // (with G1S_F_PHASES: slot 6 = the requests, 7 = the stores, 3 = the wait at barrier 1)
    }
  };
  run(std::true_type{}, 0, n_p);
  run(std::false_type{}, n_p, nmine);
#ifdef G1S_F_PHASES
  if (fpar.phase_cycles && lane == 0) {
    long long *o = fpar.phase_cycles + ((size_t)blockIdx.x * kFWaves + wave) * 8;
    for (int k = 0; k < 8; ++k) o[k] = t_ph[k];
  }
#endif

  // ---- the workgroup's partial systems: waves add into LDS (int64), one plain store per entry ----
  constexpr int NPL = LUMA ? 1 : 2, PL0 = LUMA ? 0 : 1;  // planes of this launch
  long long *s_S = reinterpret_cast<long long *>(m_smem);
  __syncthreads();
  for (int k = tid; k < NPL * kMRec; k += kFThreads) s_S[k] = 0;
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
        atomicAdd(reinterpret_cast<unsigned long long *>(&s_S[(c - PL0) * kMRec + idx]), (unsigned long long)(long long)acc[r]);
    }
  };
  flush(accA, PL0);
  if (CHROMA) flush(accB, 2);
  __syncthreads();
  long long *out = fpar.partials + (((size_t)frame * fpar.wg_cap + wg) * 3 + PL0) * kMRec;
  for (int k = tid; k < NPL * kMRec; k += kFThreads) out[k] = s_S[k];
}

}  // namespace g1s

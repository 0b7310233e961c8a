// k3m.hip.h -- K3 on the matrix cores: add_block_observations as an exact int8 SYRK.
//
// add_block_observations (av1-grain diff/solver.rs == libaom noise_model.c) sums, over the window
// samples p of every flat block, the outer product v(p) v(p)^T of
//     v(p) = [d(p + c_0) .. d(p + c_{n-1}), (L(p)), d(p)],     d = src8 - den8 (int8, K0 planes).
// With V = the matrix whose column p is v(p) (zero outside the window), that is S = V V^T: a symmetric
// rank-k update with K = samples.  v_mfma_i32_32x32x32_i8 takes 32 samples a step; lane l supplies the 16
// bytes of matrix row i = l & 31 for the sample half l >> 5 -- and because A[i][k] = V[i][k] = B[k][i],
// the SAME registers serve as the A and the B operand.  All integers: exact, order-independent.
//
// The 16 bytes of row i = (cx, a) (neighbour cx columns right, a rows up) are 16 consecutive samples of
// tile row y - a shifted by cx bytes.  A misaligned ds_read_b128 costs 64 cycles on gfx950 (measured:
// tools/mfma_lds_probe.hip), so the tile is staged as 7 copies, copy cx' = cx + 3 shifted by cx bytes
// at the time it is written; every operand read is then one aligned, bank-conflict-free ds_read_b128:
//     slot (16-byte unit) of lane (cx', a) = cx' * CS/16 - a * P/16 + const,   CS/16 = 2, P/16 odd (mod 16)
// and the two 16-lane groups a b128 read is served in, {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31},
// hold the rows a in {0, 1} (+ L) and a in {2, 3}: 16 distinct slots each.
//
// Work unit = kMUnitBlocks horizontally adjacent blocks of one block row (a `chunk`) holding at least one
// flat block, listed per frame by k3m_units together with the observation windows of its blocks.  This
// file holds the scheme's shared parts (matrix row map, tile geometry, the copy writer, the multiply
// loops, the unit lists, the reducer); the kernel that stages the tiles straight from the source /
// denoised planes and multiplies them is k3s.hip.h (round 2: k3f_fused, removed).  Any lag 1..3 (the lag-L neighbourhood and window
// border; the other matrix rows are ignored).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pixel_helpers.hip.h"
#include "kernels.hip.h"

namespace g1s {

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

constexpr int kMCopies = 7;
constexpr int kMRec = 656;        // int64 entries of one partial system (>= 25 * 25 + 25 + 1)
constexpr int kMUnitBlocks = 2;   // blocks per unit
constexpr int kMUnitDwords = 4;   // list entry: [0] chunk | block row << 12 | flat bits << 24, [1..2] 4 x u16 windows (luma, chroma)
constexpr int kMMaxUnits = 240;   // units per workgroup: 240 * 2 blocks * 8 steps * 32 samples * 127^2 < 2^31

struct MParams {
  const uint8_t *bad;     // [batch][2][nblocks]  residual (kind 1: or L) outside int8, when a pixel pass has flagged them; or null
  uint8_t *only;          // [batch][2][nblocks]  flat blocks left to k3_ar_generic (zeroed per batch)
  uint32_t *only_any;     // [batch]
  uint32_t *units;        // [batch][nunits][kMUnitDwords]  (two lists, see k3m_units)
  uint32_t *unit_count;   // [batch][2]  general units (from the front of the frame's array), plain units (from its back); zeroed per batch
  long long *partials;    // [batch][G][3][kMRec]
  int nunits;             // chunks per frame = ceil(nbw / 4) * nbh
};

// ---- per-block window of a unit, 16 bits: xe | ye << 6 | (ys != 0) << 13 | (xs != 0) << 14 | go << 15 ----
struct MWin {
  int go, xs, xe, ys, ye;
};
__device__ __forceinline__ MWin m_unpack(uint32_t w, int lag) {
  MWin r;
  r.go = (int)((w >> 15) & 1u);
  r.xe = (int)(w & 63u);
  r.ye = (int)((w >> 6) & 63u);
  r.ys = (w >> 13) & 1u ? lag : 0;
  r.xs = (w >> 14) & 1u ? lag : 0;
  return r;
}

// ---------------------------------------------------------------------------------
// k3m_units: per frame, the chunks with a flat block, the windows of their blocks per plane kind, and
// which of them go to the exact kernel instead.  grid = (ceil(nunits / 256), batch), block = 256;
// one atomic per wave.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3m_units(Geom g, const uint8_t *__restrict__ records, MParams mp) {
  const int frame = g.frame0 + (int)blockIdx.y;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int gx = (g.nbw + kMUnitBlocks - 1) / kMUnitBlocks;
  const uint8_t *mask = records + (size_t)frame * g.rec_size + g.off_mask;
  // The six mask bytes a unit's windows depend on -- its two blocks, their left and right neighbours, the two blocks above --
  // are read up front, unconditionally and independently (outside the frame: 0): the kernel is as long as its longest chain
  // of dependent loads, and block_window per block and kind was five of them in a row, four times over (35 us a launch).
  uint32_t bits = 0;
  int by = 0, ci = 0;
  int m_l = 0, m_0 = 0, m_1 = 0, m_r = 0, u_0 = 0, u_1 = 0;
  if (idx < mp.nunits) {
    by = idx / gx;
    ci = idx - by * gx;
    const int bxa = kMUnitBlocks * ci;
    auto at = [&](int x, int y) { return (x >= 0 && x < g.nbw && y >= 0 && y < g.nbh) ? (int)mask[y * g.nbw + x] : 0; };
    m_l = at(bxa - 1, by);
    m_0 = at(bxa, by);
    m_1 = at(bxa + 1, by);
    m_r = at(bxa + 2, by);
    u_0 = at(bxa, by - 1);
    u_1 = at(bxa + 1, by - 1);
    bits = (m_0 ? 1u : 0u) | (m_1 ? 2u : 0u);
  }
  static_assert(kMUnitBlocks == 2, "k3m_units: two blocks a unit");
  if (__ballot(bits != 0) == 0) return;
  const bool chroma = g.nplanes == 3;
  uint32_t win[2 * kMUnitBlocks];
  // `plain`: every block of the chunk is flat and its window is the whole block, in both plane kinds -- the accumulation
  // kernel runs those units through a loop without window masks
  bool plain = bits == (1u << kMUnitBlocks) - 1u;
#pragma unroll
  for (int t = 0; t < 2 * kMUnitBlocks; ++t) {
    const int kind = t / kMUnitBlocks, b = t % kMUnitBlocks, bx = kMUnitBlocks * ci + b;
    win[t] = 0;
    if (!((bits >> b) & 1u) || (kind && !chroma)) continue;
    const int bw = kind ? (kBlock >> g.xdec) : kBlock, bh = kind ? (kBlock >> g.ydec) : kBlock;
    const int pw = kind ? (g.W >> g.xdec) : g.W, ph = kind ? (g.H >> g.ydec) : g.H;
    // block_window (pixel_helpers.hip.h) on the bytes read above
    const int left = b ? m_0 : m_l, right = b ? m_r : m_1, up = b ? u_1 : u_0;
    Win w{1, 0, 0, 0, 0};
    w.ys = up ? 0 : g.lag;
    w.xs = left ? 0 : g.lag;
    w.ye = min(ph - by * bh, bh);
    w.xe = min(pw - bx * bw - g.lag, right ? bw : (bw - g.lag));
    if (w.xe <= w.xs || w.ye <= w.ys) w.flat = 0;  // empty window
    if (!w.flat || w.xs != 0 || w.ys != 0 || w.xe != bw || w.ye != bh) plain = false;
    if (!w.flat) continue;
    // (a pixel pass may have flagged residuals outside int8: the tile reaches into the left / right / upper neighbours;
    //  the fused pass finds them itself)
    bool defer = false;
    if (mp.bad) {
      const uint8_t *bad = mp.bad + ((size_t)frame * 2 + kind) * g.nblocks;
      for (int dy = -1; dy <= 0; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int x = bx + dx, y = by + dy;
          if (x >= 0 && x < g.nbw && y >= 0 && y < g.nbh && bad[y * g.nbw + x]) defer = true;
        }
    }
    if (defer) {
      // (the exact kernel's lists are per plane: a chroma deferral of this chain concerns both chroma planes)
      mp.only[((size_t)frame * 3 + kind) * g.nblocks + by * g.nbw + bx] = 1;
      if (kind) mp.only[((size_t)frame * 3 + 2) * g.nblocks + by * g.nbw + bx] = 1;
      mp.only_any[frame] = 1u;
      plain = false;
    } else {
      win[t] = (uint32_t)w.xe | ((uint32_t)w.ye << 6) | (w.ys ? 1u << 13 : 0u) | (w.xs ? 1u << 14 : 0u) | (1u << 15);
    }
  }
  // two lists in the frame's array: the general units from the front, the plain ones from the back
  const int lane = threadIdx.x & 63;
  const unsigned long long vote_g = __ballot(bits != 0 && !plain), vote_p = __ballot(bits != 0 && plain);
  uint32_t base_g = 0, base_p = 0;
  if (lane == 0) {
    if (vote_g) base_g = atomicAdd(&mp.unit_count[2 * frame], (uint32_t)__popcll(vote_g));
    if (vote_p) base_p = atomicAdd(&mp.unit_count[2 * frame + 1], (uint32_t)__popcll(vote_p));
  }
  base_g = __shfl(base_g, 0, 64);
  base_p = __shfl(base_p, 0, 64);
  if (!bits) return;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t pos = plain ? (uint32_t)mp.nunits - 1u - (base_p + (uint32_t)__popcll(vote_p & below)) : base_g + (uint32_t)__popcll(vote_g & below);
  static_assert(kMUnitBlocks == 2 && kMUnitDwords == 4, "list entry layout");
  uint32_t *e = mp.units + ((size_t)frame * mp.nunits + pos) * kMUnitDwords;
  *reinterpret_cast<uint4 *>(e) = make_uint4((uint32_t)ci | ((uint32_t)by << 12) | (bits << 24), win[0] | (win[1] << 16),
                                             win[2] | (win[3] << 16), plain ? 1u : 0u);
}

__device__ __forceinline__ uint32_t m_bytemask(int k) { return k >= 4 ? 0xffffffffu : ((1u << (8 * k)) - 1u); }

// 7 shifted copies of one row word (8 samples) -> LDS.  d0, d1: the word; prev1: the dword before it,
// next0: the dword after it.
template <bool MASKED = true>
__device__ __forceinline__ void m_write_copies(uint8_t *dst, int CS, uint32_t prev1, uint32_t d0, uint32_t d1, uint32_t next0,
                                               uint2 cm) {
#pragma unroll
  for (int cxp = 0; cxp < kMCopies; ++cxp) {
    const int cx = cxp - 3;
    uint32_t w0, w1;
    if (cx < 0) {
      w0 = __builtin_amdgcn_alignbyte(d0, prev1, 4 + cx);
      w1 = __builtin_amdgcn_alignbyte(d1, d0, 4 + cx);
    } else if (cx == 0) {
      w0 = d0;
      w1 = d1;
    } else {
      w0 = __builtin_amdgcn_alignbyte(d1, d0, cx);
      w1 = __builtin_amdgcn_alignbyte(next0, d1, cx);
    }
    *reinterpret_cast<uint2 *>(dst + cxp * CS) = MASKED ? make_uint2(w0 & cm.x, w1 & cm.y) : make_uint2(w0, w1);
  }
}

// byte mask of the 8 samples xb .. xb + 7 of a block under its window columns [xs, xe)
__device__ __forceinline__ uint2 m_colmask8(const MWin &bi, int xb) {
  if (!bi.go) return make_uint2(0u, 0u);
  const int lo = min(max(bi.xs - xb, 0), 8), hi = min(max(bi.xe - xb, 0), 8);
  const unsigned long long mh = hi >= 8 ? ~0ull : ((1ull << (8 * hi)) - 1ull);
  const unsigned long long ml = lo >= 8 ? ~0ull : ((1ull << (8 * lo)) - 1ull);
  const unsigned long long m = mh & ~ml;
  return make_uint2((uint32_t)m, (uint32_t)(m >> 32));
}

__device__ __forceinline__ v4i32 m_lds16(const uint8_t *smem, int a) { return *reinterpret_cast<const v4i32 *>(smem + a); }

// rows [ys, ye) of a block as a bit mask (bit y = row y), ye <= 32
__device__ __forceinline__ uint32_t m_rowmask(int ys, int ye) {
  return (uint32_t)(((1ull << ye) - 1ull) & ~((1ull << ys) - 1ull));
}

// ---------------------------------------------------------------------------------
// k3m_finish: what the accumulation workgroups left behind -> the frame's record.
//   x < 3 * nplanes:  a third (256 entries) of the G partial systems of plane x / 3, summed;
//   x >= 3 * nplanes: the per-unit statistics records (kMStatInts ints a unit: per block sum d, sum d^2, sum src8 of
//                 luma, sum d, sum d^2 of Cb and Cr; then the deferral bits kind * 2 + block of the luma and of the
//                 chroma launch) -> block statistics of the flat blocks, `only` flags of the deferred ones; and nobs of
//                 the blocks that were multiplied (one atomic per workgroup and plane).
// grid = (3 * nplanes + kMFinishWgs, batch), block = 256.  Every thread has at most one entry / one unit: the kernel is a
// few dependent loads deep (it was 55 us as 7 workgroups a frame that looped).
// ---------------------------------------------------------------------------------
constexpr int kMStatInts = 16, kMFinishWgs = 12, kMFinishParts = 3;
static_assert(kMFinishParts * 256 >= 26 * 26 + 26, "k3m_finish: one entry per thread");
// ustats == nullptr: a pixel pass took the statistics and the deferrals (K0 + k3m_units): only the systems and nobs.
__global__ __launch_bounds__(256) void k3m_finish(Geom g, MParams mp, int G_luma, int G_chroma, int G_cap, const int32_t *__restrict__ ustats,
                                                  uint8_t *__restrict__ records) {
  const int frame = g.frame0 + (int)blockIdx.y;
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  if ((int)blockIdx.x < kMFinishParts * g.nplanes) {
    const int c = (int)blockIdx.x / kMFinishParts, nc = g.n + (c > 0);
    const int k = ((int)blockIdx.x - c * kMFinishParts) * 256 + (int)threadIdx.x;
    if (k >= nc * nc + nc) return;
    long long *ar = reinterpret_cast<long long *>(rec + g.off_ar[c]);
    const int G = c == 0 ? G_luma : G_chroma;  // (workgroups a frame of the launch that made this plane's partial systems)
    const long long *p = mp.partials + (size_t)frame * G_cap * 3 * kMRec + (size_t)c * kMRec + k;
    long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int w = 0;
    for (; w + 4 <= G; w += 4) {  // (independent loads in flight)
      s0 += p[(size_t)(w + 0) * 3 * kMRec];
      s1 += p[(size_t)(w + 1) * 3 * kMRec];
      s2 += p[(size_t)(w + 2) * 3 * kMRec];
      s3 += p[(size_t)(w + 3) * 3 * kMRec];
    }
    for (; w < G; ++w) s0 += p[(size_t)w * 3 * kMRec];
    ar[k] += (s0 + s1) + (s2 + s3);
    return;
  }
  const uint32_t cnt_g = mp.unit_count[2 * frame], cnt = cnt_g + mp.unit_count[2 * frame + 1];
  // list position of the v-th unit of the frame: the general ones from the front, the plain ones from the back
  auto upos = [&](uint32_t v) { return v < cnt_g ? v : (uint32_t)mp.nunits - 1u - (v - cnt_g); };
  const uint32_t *units = mp.units + (size_t)frame * mp.nunits * kMUnitDwords;
  const int32_t *us = ustats + (size_t)frame * mp.nunits * kMStatInts;
  const int part = (int)blockIdx.x - kMFinishParts * g.nplanes;
  const bool chroma = g.nplanes == 3;
  long long n_y = 0, n_cb = 0, n_cr = 0;  // observations: the windows of the blocks that were multiplied (go and not deferred)
  for (uint32_t v = part * 256 + threadIdx.x; v < cnt; v += kMFinishWgs * 256) {
    const uint32_t u = upos(v);
    const uint4 e = *reinterpret_cast<const uint4 *>(units + (size_t)u * kMUnitDwords);
    const uint32_t e0 = e.x;
    const int bx0 = kMUnitBlocks * (int)(e0 & 0xfffu), by = (int)((e0 >> 12) & 0xfffu);
    const int32_t *r = us + (size_t)u * kMStatInts;
    int32_t rv[kMStatInts];
    if (ustats) {
#pragma unroll
      for (int q = 0; q < kMStatInts / 4; ++q) {
        const int4 t = *reinterpret_cast<const int4 *>(r + 4 * q);
        rv[4 * q] = t.x, rv[4 * q + 1] = t.y, rv[4 * q + 2] = t.z, rv[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < kMStatInts; ++q) rv[q] = 0;
    }
    // deferrals, per PLANE: entry 14 (the luma launch): bits 0, 1 the luma plane of block b, bits 2, 3 L outside int8 (both
    // chroma planes); entry 15 (the chroma launch(es)): bits 2, 3 Cb, bits 4, 5 Cr
    const uint32_t r14 = (uint32_t)rv[14], r15 = chroma ? (uint32_t)rv[15] : 0u;
    const uint32_t dpl[3] = {r14 & 3u, ((r14 | r15) >> kMUnitBlocks) & 3u, ((r14 >> kMUnitBlocks) | (r15 >> (2 * kMUnitBlocks))) & 3u};
#pragma unroll
    for (int b = 0; b < kMUnitBlocks; ++b) {
      const MWin wy = m_unpack((e.y >> (16 * b)) & 0xffffu, g.lag), wc = m_unpack((e.z >> (16 * b)) & 0xffffu, g.lag);
      if (wy.go && !((dpl[0] >> b) & 1u)) n_y += (long long)(wy.xe - wy.xs) * (wy.ye - wy.ys);
      if (wc.go && !((dpl[1] >> b) & 1u)) n_cb += (long long)(wc.xe - wc.xs) * (wc.ye - wc.ys);
      if (wc.go && !((dpl[2] >> b) & 1u)) n_cr += (long long)(wc.xe - wc.xs) * (wc.ye - wc.ys);
      if (!ustats || !((e0 >> (24 + b)) & 1u)) continue;
      const int blk = by * g.nbw + bx0 + b;
      reinterpret_cast<int32_t *>(rec + g.off_sum_d[0])[blk] = rv[7 * b + 0];
      reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[0])[blk] = (uint32_t)rv[7 * b + 1];
      reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)rv[7 * b + 2];
      if (chroma) {
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[1])[blk] = rv[7 * b + 3];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[1])[blk] = (uint32_t)rv[7 * b + 4];
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[2])[blk] = rv[7 * b + 5];
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[2])[blk] = (uint32_t)rv[7 * b + 6];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if ((dpl[c] >> b) & 1u) {
          mp.only[((size_t)frame * 3 + c) * g.nblocks + blk] = 1;
          mp.only_any[frame] = 1u;
        }
    }
  }
  __shared__ long long s_n[3][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n_y += __shfl_xor(n_y, o, 64);
    n_cb += __shfl_xor(n_cb, o, 64);
    n_cr += __shfl_xor(n_cr, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_n[0][threadIdx.x >> 6] = n_y;
    s_n[1][threadIdx.x >> 6] = n_cb;
    s_n[2][threadIdx.x >> 6] = n_cr;
  }
  __syncthreads();
  if ((int)threadIdx.x < g.nplanes) {
    const int c = threadIdx.x, nc = g.n + (c > 0);
    const long long n = s_n[c][0] + s_n[c][1] + s_n[c][2] + s_n[c][3];
    if (n) atomicAdd(reinterpret_cast<unsigned long long *>(rec + g.off_ar[c]) + (nc * nc + nc), (unsigned long long)n);
  }
}

}  // namespace g1s

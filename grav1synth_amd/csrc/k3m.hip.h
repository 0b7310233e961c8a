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
// Work unit = 4 horizontally adjacent blocks of one block row (a `chunk`) holding at least one flat
// block, compacted per frame by k3m_units.  A workgroup walks a contiguous slice of one frame's list,
// keeps the 32x32 int32 accumulators of the three planes in registers, and writes ONE partial system
// per plane at the end (k3m_reduce adds them into the record).  Blocks whose tile touches a residual
// outside int8 (K0 `bad` flags) are left to the exact int32 kernel (k3_ar_generic, `only` list).
// Any lag 1..3 (the lag-L neighbourhood and window border; rows of other neighbours are ignored).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k0.hip.h"
#include "kernels.hip.h"

namespace g1s {

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

constexpr int kMCopies = 7;
constexpr int kMRec = 656;        // int64 entries of one partial system (>= 25 * 25 + 25 + 1)
constexpr int kMUnitBlocks = 4;   // blocks per unit
constexpr int kMMaxUnits = 120;   // units per workgroup: 120 * 4 blocks * 8 steps * 32 samples * 127^2 < 2^31

struct MParams {
  const uint8_t *planes;  // K0 planes [batch] x ps.frame_bytes
  PlaneSet ps;
  const uint8_t *bad;     // [batch][2][nblocks]  K0: a residual (kind 1: or L) outside int8
  uint8_t *only;          // [batch][2][nblocks]  flat blocks left to k3_ar_generic (zeroed per batch)
  uint32_t *only_any;     // [batch]
  uint32_t *units;        // [batch][nunits]  chunk | block row << 12 | flat bits << 24
  uint32_t *unit_count;   // [batch]  (zeroed per batch)
  long long *partials;    // [batch][G][3][kMRec]
  int nunits;             // chunks per frame = ceil(nbw / 4) * nbh
};

// tile geometry of a plane kind: block BW x bh, chunk of 4 blocks
__host__ __device__ constexpr int m_pitch(int BW) { return 4 * BW + 16; }
__host__ __device__ constexpr int m_copy_stride(int BW, int bh) {
  // >= rows * pitch, in 16-byte slots == 2 (mod 16)
  int slots = ((bh + 3) * m_pitch(BW) + 15) / 16;
  while ((slots & 15) != 2) ++slots;
  return slots * 16;
}
__host__ __device__ constexpr int m_tile_bytes(int BW, int bh, bool with_l) {
  return kMCopies * m_copy_stride(BW, bh) + (with_l ? bh * m_pitch(BW) : 0);
}

// ---- matrix row i (0..31) -> what it holds ------------------------------------------
// order inside the two b128 lane groups
__device__ __forceinline__ int m_group_order(int i, int &grp) {
  if (i < 4) { grp = 0; return i; }
  if (i < 12) { grp = 1; return i - 4; }
  if (i < 16) { grp = 0; return 4 + (i - 12); }
  if (i < 20) { grp = 1; return 8 + (i - 16); }
  if (i < 28) { grp = 0; return 8 + (i - 20); }
  grp = 1;
  return 12 + (i - 28);
}
// special: 0 neighbour / the sample itself (a, cxp), 1 the luma regressor L, 2 spare
__device__ __forceinline__ void m_entry(int i, int &a, int &cxp, int &special) {
  int grp;
  const int k = m_group_order(i, grp);
  special = 0;
  a = 0;
  cxp = 3;
  if (grp == 0) {
    if (k < 4) { a = 0; cxp = k; }            // cx = -3 .. 0; k == 3: d(p) itself
    else if (k < 11) { a = 1; cxp = k - 4; }
    else if (k == 11) special = 1;
    else special = 2;
  } else {
    if (k < 7) { a = 2; cxp = k; }
    else if (k < 14) { a = 3; cxp = k - 7; }
    else special = 2;
  }
}
// index in the record's (nc+1)-vector: 0..n-1 neighbours, n = L (chroma), nc = the sample; -1 = not part of it
__device__ __forceinline__ int m_rec_index(int i, int lag, int n, bool chroma) {
  int a, cxp, sp;
  m_entry(i, a, cxp, sp);
  if (sp == 2) return -1;
  if (sp == 1) return chroma ? n : -1;
  const int cx = cxp - 3;
  if (a == 0 && cx == 0) return n + (chroma ? 1 : 0);
  if (a > lag || cx < -lag || cx > lag) return -1;
  return (lag - a) * (2 * lag + 1) + (cx + lag);
}

// ---------------------------------------------------------------------------------
// k3m_units: per frame, the chunks with a flat block.  grid = (ceil(nunits / 256), batch), block = 256;
// one atomic per wave.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3m_units(Geom g, const uint8_t *__restrict__ records, MParams mp) {
  const int frame = g.frame0 + (int)blockIdx.y;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int gx = (g.nbw + kMUnitBlocks - 1) / kMUnitBlocks;
  const uint8_t *mask = records + (size_t)frame * g.rec_size + g.off_mask;
  uint32_t bits = 0;
  int by = 0, ci = 0;
  if (idx < mp.nunits) {
    by = idx / gx;
    ci = idx - by * gx;
#pragma unroll
    for (int b = 0; b < kMUnitBlocks; ++b) {
      const int bx = kMUnitBlocks * ci + b;
      if (bx < g.nbw && mask[by * g.nbw + bx]) bits |= 1u << b;
    }
  }
  const unsigned long long vote = __ballot(bits != 0);
  if (vote == 0) return;
  const int lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&mp.unit_count[frame], (uint32_t)__popcll(vote));
  base = __shfl(base, 0, 64);
  if (bits) {
    const uint32_t pos = base + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull));
    mp.units[(size_t)frame * mp.nunits + pos] = (uint32_t)ci | ((uint32_t)by << 12) | (bits << 24);
  }
}

// ---------------------------------------------------------------------------------
// per-unit block info (one thread per (kind, block) writes it, everybody reads it)
// ---------------------------------------------------------------------------------
struct MBlockInfo {
  int go;  // flat, window not empty, not deferred
  int xs, xe, ys, ye;
};

__device__ __forceinline__ uint32_t m_bytemask(int k) { return k >= 4 ? 0xffffffffu : ((1u << (8 * k)) - 1u); }

// 7 shifted copies of one row word (8 samples) -> LDS.  d0, d1: the word; prev1: the dword before it,
// next0: the dword after it.
__device__ __forceinline__ void m_write_copies(uint8_t *dst, int CS, uint32_t prev1, uint32_t d0, uint32_t d1, uint32_t next0) {
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
    *reinterpret_cast<uint2 *>(dst + cxp * CS) = make_uint2(w0, w1);
  }
}

// stage the d8 tile of the unit (rows -3 .. bh-1, samples -8 .. 4 BW + 7 of the chunk) as 7 shifted copies
template <int BW>
__device__ __forceinline__ void m_stage_plane(uint8_t *tile, const uint8_t *__restrict__ plane, uint32_t pitch, int bx0, int by,
                                              int bh, int CS, int wave, int lane) {
  constexpr int P = m_pitch(BW), WPR = P / 8, RPP = 64 / WPR;
  const int rows = bh + 3;
  const int lr = lane / WPR, wd = lane - lr * WPR;
  for (int r0 = wave * RPP; r0 < rows; r0 += 4 * RPP) {
    const int row = r0 + lr;
    const bool active = lr < RPP && row < rows;
    const uint32_t col = (uint32_t)(bx0 * BW + 8 * wd);
    uint2 D = make_uint2(0u, 0u);
    if (active && col + 8u <= pitch) D = *reinterpret_cast<const uint2 *>(plane + (size_t)(by * bh + row) * pitch + col);
    const uint32_t prev1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)D.y, 0x138, 0xf, 0xf, false);  // wave_shr:1
    const uint32_t next0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)D.x, 0x130, 0xf, 0xf, false);  // wave_shl:1
    if (active && wd >= 1 && wd <= WPR - 2) m_write_copies(tile + row * P + 8 * (wd - 1), CS, prev1, D.x, D.y, next0);
  }
}

// the block's share of S: acc += V V^T over this wave's rows of the window
template <int BW>
__device__ __forceinline__ void m_block(v16i32 &acc, const uint8_t *tile, int addr, const MBlockInfo &bi, int bh, int wave, int h) {
  constexpr int P = m_pitch(BW);
  const int xh = BW == 32 ? 16 * h : 0;
  const int lo = min(max(bi.xs - xh, 0), 16), hi = min(max(bi.xe - xh, 0), 16);
  v4i32 m;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int l = min(max(lo - 4 * q, 0), 4), u = min(max(hi - 4 * q, 0), 4);
    m[q] = u > l ? (int)(m_bytemask(u) & ~m_bytemask(l)) : 0;
  }
  const bool full = bi.xs == 0 && bi.xe == BW;
  if constexpr (BW == 32) {
    const int rpw = bh >> 2;
    const int y0 = max(bi.ys, wave * rpw), y1 = min(bi.ye, (wave + 1) * rpw);
    if (full) {
      for (int y = y0; y < y1; ++y) {
        const v4i32 v = *reinterpret_cast<const v4i32 *>(tile + addr + y * P);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v, v, acc, 0, 0, 0);
      }
    } else {
      for (int y = y0; y < y1; ++y) {
        const v4i32 v = *reinterpret_cast<const v4i32 *>(tile + addr + y * P) & m;
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v, v, acc, 0, 0, 0);
      }
    }
  } else {
    const int spw = bh >> 3;  // steps (row pairs) per wave
    for (int s = wave * spw; s < (wave + 1) * spw; ++s) {
      const int ya = 2 * s;
      if (ya + 1 < bi.ys || ya >= bi.ye) continue;
      v4i32 v = *reinterpret_cast<const v4i32 *>(tile + addr + ya * P);
      if (ya < bi.ys || ya + 1 >= bi.ye) {
        const bool ok = ya + h >= bi.ys && ya + h < bi.ye;
        const v4i32 z = {0, 0, 0, 0};
        v = ok ? (v & m) : z;
      } else if (!full) {
        v &= m;
      }
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(v, v, acc, 0, 0, 0);
    }
  }
}

// one plane of one unit: stage, accumulate
template <int BW>
__device__ __forceinline__ void m_plane_pass(v16i32 &acc, long long &nobs, uint8_t *tile, const MBlockInfo *info, int c,
                                             const MParams &mp, const uint8_t *fplanes, int bx0, int by, int bh, bool with_l,
                                             int lag_base, int wave, int lane) {
  constexpr int P = m_pitch(BW);
  const int CS = m_copy_stride(BW, bh);
  const int kind = c > 0 ? 1 : 0;
  __syncthreads();  // the previous plane's reads are done
  m_stage_plane<BW>(tile, fplanes + mp.ps.off_d[c], mp.ps.pitch[kind], bx0, by, bh, CS, wave, lane);
  if (with_l && c == 1) {
    // L tile (no halo): rows 0 .. bh-1, samples 0 .. 4 BW - 1
    constexpr int WL = 4 * BW / 8;
    uint8_t *lt = tile + kMCopies * CS;
    for (int idx = threadIdx.x; idx < bh * WL; idx += 256) {
      const int row = idx / WL, wd = idx - row * WL;
      const uint32_t col = (uint32_t)(bx0 * BW + 8 * wd);
      uint2 D = make_uint2(0u, 0u);
      if (col + 8u <= mp.ps.lpitch) D = *reinterpret_cast<const uint2 *>(fplanes + mp.ps.off_l + (size_t)(by * bh + row) * mp.ps.lpitch + col);
      *reinterpret_cast<uint2 *>(lt + row * P + 8 * wd) = D;
    }
  }
  __syncthreads();
  const int h = lane >> 5;
#pragma unroll
  for (int b = 0; b < kMUnitBlocks; ++b) {
    const MBlockInfo bi = info[kind * kMUnitBlocks + b];
    if (!bi.go) continue;
    m_block<BW>(acc, tile, lag_base + BW * b, bi, bh, wave, h);
    if (threadIdx.x == 0) nobs += (long long)(bi.xe - bi.xs) * (bi.ye - bi.ys);
  }
}

// ---------------------------------------------------------------------------------
// k3m_accumulate<CBW>: CBW = chroma block width (32 >> xdec; 0 = luma only).
// grid = (G, 1, batch), block = 256, dynamic LDS = max tile bytes (see m_lds_bytes).
// ---------------------------------------------------------------------------------
template <int CBW>
__global__ __launch_bounds__(256) void k3m_accumulate(Geom g, MParams mp, const uint8_t *__restrict__ records) {
  extern __shared__ __attribute__((aligned(16))) uint8_t m_smem[];
  __shared__ MBlockInfo s_info[2 * kMUnitBlocks];
  uint8_t *tile = m_smem;
  const int frame = g.frame0 + (int)blockIdx.z;
  const int G = gridDim.x, wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t cnt = mp.unit_count[frame];
  const uint32_t u0 = (uint32_t)((unsigned long long)cnt * wg / G), u1 = (uint32_t)((unsigned long long)cnt * (wg + 1) / G);
  const uint32_t *units = mp.units + (size_t)frame * mp.nunits;
  const uint8_t *mask = records + (size_t)frame * g.rec_size + g.off_mask;
  const uint8_t *fplanes = mp.planes + (size_t)frame * mp.ps.frame_bytes;
  const bool chroma = CBW != 0 && g.nplanes == 3;
  const int cbh = kBlock >> g.ydec;

  // this lane's operand address inside a tile, per plane kind
  const int i = lane & 31, h = lane >> 5;
  int ea, ecxp, esp;
  m_entry(i, ea, ecxp, esp);
  int base_luma, base_chroma = 0;
  {
    constexpr int P = m_pitch(32);
    const int CS = m_copy_stride(32, kBlock);
    const bool plain = esp == 0;
    base_luma = (plain ? ecxp : 3) * CS + (3 - (plain ? ea : 0)) * P + 16 * h;
  }
  if constexpr (CBW != 0) {
    constexpr int P = m_pitch(CBW);
    const int CS = m_copy_stride(CBW, cbh);
    const bool plain = esp == 0;
    const int hoff = CBW == 32 ? 16 * h : h * P;
    base_chroma = esp == 1 ? kMCopies * CS + hoff : (plain ? ecxp : 3) * CS + (3 - (plain ? ea : 0)) * P + hoff;
  }

  v16i32 acc0, acc1, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = 0;
  long long nobs0 = 0, nobs1 = 0, nobs2 = 0;

  for (uint32_t u = u0; u < u1; ++u) {
    const uint32_t e = units[u];
    const int ci = (int)(e & 0xfffu), by = (int)((e >> 12) & 0xfffu);
    const int bx0 = kMUnitBlocks * ci;
    __syncthreads();  // s_info of the previous unit is no longer read
    if (tid < 2 * kMUnitBlocks) {
      const int kind = tid / kMUnitBlocks, b = tid % kMUnitBlocks;
      MBlockInfo bi{0, 0, 0, 0, 0};
      const int bx = bx0 + b;
      if ((kind == 0 || chroma) && ((e >> (24 + b)) & 1u)) {
        const int bw = kind ? (kBlock >> g.xdec) : kBlock, bh = kind ? cbh : kBlock;
        const int pw = kind ? (g.W >> g.xdec) : g.W, ph = kind ? (g.H >> g.ydec) : g.H;
        const Win w = block_window(mask, g.nbw, g.nbh, bx, by, bw, bh, pw, ph, g.lag);
        if (w.flat) {
          // the tile reaches into the left / right / upper neighbours
          const uint8_t *bad = mp.bad + ((size_t)frame * 2 + kind) * g.nblocks;
          bool defer = false;
          for (int dy = -1; dy <= 0; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int x = bx + dx, y = by + dy;
              if (x >= 0 && x < g.nbw && y >= 0 && bad[y * g.nbw + x]) defer = true;
            }
          if (defer) {
            mp.only[((size_t)frame * 2 + kind) * g.nblocks + by * g.nbw + bx] = 1;
            mp.only_any[frame] = 1u;
          } else {
            bi.go = 1;
            bi.xs = w.xs;
            bi.xe = w.xe;
            bi.ys = w.ys;
            bi.ye = w.ye;
          }
        }
      }
      s_info[tid] = bi;
    }
    // (m_plane_pass opens with a barrier)
    m_plane_pass<32>(acc0, nobs0, tile, s_info, 0, mp, fplanes, bx0, by, kBlock, false, base_luma, wave, lane);
    if constexpr (CBW != 0) {
      if (chroma) {
        m_plane_pass<CBW>(acc1, nobs1, tile, s_info, 1, mp, fplanes, bx0, by, cbh, true, base_chroma, wave, lane);
        m_plane_pass<CBW>(acc2, nobs2, tile, s_info, 2, mp, fplanes, bx0, by, cbh, true, base_chroma, wave, lane);
      }
    }
  }

  // ---- the workgroup's partial systems: waves add into LDS (int64), one plain store per entry ----
  long long *s_S = reinterpret_cast<long long *>(m_smem);
  __syncthreads();
  for (int k = tid; k < 3 * kMRec; k += 256) s_S[k] = 0;
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
  flush(acc0, 0, nobs0);
  if (chroma) {
    flush(acc1, 1, nobs1);
    flush(acc2, 2, nobs2);
  }
  __syncthreads();
  long long *out = mp.partials + ((size_t)frame * G + wg) * 3 * kMRec;
  for (int k = tid; k < 3 * kMRec; k += 256) out[k] = s_S[k];
}

inline size_t m_lds_bytes(int cbw, int cbh) {
  size_t b = (size_t)m_tile_bytes(32, kBlock, false);
  if (cbw == 32) b = std::max(b, (size_t)m_tile_bytes(32, cbh, true));
  if (cbw == 16) b = std::max(b, (size_t)m_tile_bytes(16, cbh, true));
  return std::max(b, sizeof(long long) * 3 * kMRec);
}

// ---------------------------------------------------------------------------------
// k3m_reduce: the G partial systems of a (frame, plane) -> record.  grid = (nplanes, batch), block = 256.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3m_reduce(Geom g, MParams mp, int G, uint8_t *__restrict__ records) {
  const int c = blockIdx.x, frame = g.frame0 + (int)blockIdx.y;
  const int nc = g.n + (c > 0);
  long long *ar = reinterpret_cast<long long *>(records + (size_t)frame * g.rec_size + g.off_ar[c]);
  const long long *p = mp.partials + (size_t)frame * G * 3 * kMRec + (size_t)c * kMRec;
  for (int k = threadIdx.x; k < nc * nc + nc + 1; k += 256) {
    long long s = 0;
    for (int w = 0; w < G; ++w) s += p[(size_t)w * 3 * kMRec + k];
    ar[k] += s;
  }
}

}  // namespace g1s

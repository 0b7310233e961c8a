// kernels.hip.h -- gfx950 device code of the `diff` estimator.
//
//   K1 flat_features : per 32x32 luma block, the plane-fit residual + gradient
//                      covariance features of FlatBlockFinder::run
//                      (av1-grain diff/solver.rs == libaom
//                      aom_flat_block_finder_run), one lane per block, f64 in
//                      the reference's summation order (decision-exact).
//   K2 flat_select   : per frame, the k-th order statistic of the f32 scores
//                      (index nblocks*90/100 of the ascending order) by radix
//                      select, then the 0/1/255 mask.
//   K3 ar_accumulate : per plane, exact integer AR normal-equation sums
//                      (add_block_observations) + per-block noise statistics
//                      (get_block_mean / get_noise_var) over the flat blocks.
//
// Reference call site of all of it: differ.diff_frame(...) src/main.rs:442.
// Compile with -ffp-contract=off: K1 must not contract mul+add into FMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "record.h"

namespace g1s {

struct FramePlanes {
  const uint8_t *src[3];
  const uint8_t *den[3];
  uint32_t src_stride[3];
  uint32_t den_stride[3];
};

// The frame table of a batch lives in device memory (72 bytes a frame pair); it is uploaded on a stream of
// its own when the batch is queued, long before the main stream gets to the batch: no copy in the launch chain.
constexpr int kMaxBatch = 256;
struct FrameTable {
  const FramePlanes *f;
};

struct Geom {
  int W, H, xdec, ydec, nplanes;
  int nbw, nbh, nblocks;
  int src_bps, den_bps, src_shift, den_shift;
  int lag, n;
  int frame0;     // first frame of the batch this launch covers (the K0 / K3 chain may run in sub-batches)
  int fast_rows;  // all luma source rows 16-byte aligned (base and stride)
  int vec_mask;   // bit c: src plane c rows 16-byte aligned in every frame of the batch; bit 3+c: den plane c
  // record layout (bytes from the start of a frame's record)
  uint32_t rec_size;
  uint32_t off_ar[3], off_luma_sum, off_sum_d[3], off_sum_d2[3], off_scores, off_mask;
};

struct FlatConsts {
  double ata_inv[9];  // (A^T A)^-1 of the 1024x3 plane-fit design matrix
};

// ----------------------------------------------------------------------------
// sample access: u8, or u16 narrowed by the truncating shift of
// av1-grain util.rs frame_into_u8 (`(v >> (bd - 8)) as u8`)
// ----------------------------------------------------------------------------
// Plane pointers come out of the frame table in memory, so the compiler would
// treat them as generic (flat) addresses; flat loads also tick lgkmcnt and would
// serialise against LDS traffic.  They are HBM pointers: say so.
#define G1S_GLOBAL __attribute__((address_space(1)))
typedef const G1S_GLOBAL uint8_t *gptr_u8;
typedef const G1S_GLOBAL uint16_t *gptr_u16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const G1S_GLOBAL u32x2 *gptr_u2;
typedef const G1S_GLOBAL u32x4 *gptr_u4;
__device__ __forceinline__ gptr_u8 as_global(const uint8_t *p) { return (gptr_u8)(uintptr_t)p; }
__device__ __forceinline__ uint4 gload4(gptr_u4 p) {
  const u32x4 v = *p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 gload2(gptr_u2 p) {
  const u32x2 v = *p;
  return make_uint2(v.x, v.y);
}

template <int BPS>
__device__ __forceinline__ int load_px(const uint8_t *base, uint32_t stride, int shift, int x, int y) {
  gptr_u8 row = as_global(base) + (size_t)y * stride;
  if (BPS == 1) return row[x];
  const uint16_t v = ((gptr_u16)row)[x];
  return (int)(uint8_t)(v >> shift);
}
__device__ __forceinline__ int load_px_rt(const uint8_t *base, uint32_t stride, int bps, int shift, int x, int y) {
  return bps == 1 ? load_px<1>(base, stride, shift, x, y) : load_px<2>(base, stride, shift, x, y);
}

// 32 consecutive samples of one row -> 32 bytes packed in 8 dwords.
template <int BPS>
__device__ __forceinline__ void load_row32(const uint8_t *base, uint32_t stride, int shift, int ox, int y,
                                           int W, bool fast, uint32_t (&pk)[8]) {
  if (fast) {
    if (BPS == 1) {
      gptr_u4 p = (gptr_u4)(as_global(base) + (size_t)y * stride + ox);
      const uint4 a = gload4(p), b = gload4(p + 1);
      pk[0] = a.x; pk[1] = a.y; pk[2] = a.z; pk[3] = a.w;
      pk[4] = b.x; pk[5] = b.y; pk[6] = b.z; pk[7] = b.w;
    } else {
      gptr_u4 p = (gptr_u4)(as_global(base) + (size_t)y * stride + 2 * (size_t)ox);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 a = gload4(p + q);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // `(v >> shift) as u8` of four samples: two packed 16-bit shifts and one byte permute (the low byte of each half);
          // written as shifts, masks and ors this was 12 instructions a dword -- 40 % of k1_moments, which is VALU bound
          // (a plain 32-bit shift: the bits that cross from the upper sample into the lower one's half land above its low byte,
          //  which is all the permute takes -- v_lshrrev_b32 issues in 2.6 cycles, the packed 16-bit shift in 4.6)
          const uint32_t lo = w[2 * h] >> shift, hi = w[2 * h + 1] >> shift;
          pk[2 * q + h] = __builtin_amdgcn_perm(hi, lo, 0x06040200u);
        }
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = min(ox + 4 * q + k, W - 1);
        v |= (uint32_t)load_px<BPS>(base, stride, shift, x, y) << (8 * k);
      }
      pk[q] = v;
    }
  }
}

// The 32 samples of a block row as loaded (fast path: 16-byte aligned rows inside the plane):
// kept raw in registers so that the loads of later rows are in flight while a row is
// processed (one lane per block and one wave per SIMD: nothing else hides the latency).
template <int BPS>
struct RowRaw {
  u32x4 v[BPS == 2 ? 4 : 2];
};
template <int BPS>
__device__ __forceinline__ void load_row_raw(const uint8_t *base, uint32_t stride, int ox, int y, RowRaw<BPS> &r) {
  gptr_u4 p = (gptr_u4)(as_global(base) + (size_t)y * stride + (size_t)ox * BPS);
#pragma unroll
  for (int q = 0; q < (BPS == 2 ? 4 : 2); ++q) r.v[q] = p[q];
}
template <int BPS>
__device__ __forceinline__ void narrow_row(const RowRaw<BPS> &r, int shift, uint32_t (&pk)[8]) {
  if (BPS == 1) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      pk[4 * q + 0] = r.v[q].x;
      pk[4 * q + 1] = r.v[q].y;
      pk[4 * q + 2] = r.v[q].z;
      pk[4 * q + 3] = r.v[q].w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t lo = r.v[q][2 * h] >> shift, hi = r.v[q][2 * h + 1] >> shift;  // (plain shifts: see load_row32)
        pk[2 * q + h] = __builtin_amdgcn_perm(hi, lo, 0x06040200u);
      }
    }
  }
}

// The finder's decision from the five interior sums (FlatBlockFinder::run after the loops): the four
// thresholds -> flag, the sigmoid -> f32 score.
__device__ __forceinline__ void flat_decide(double Gxx, double Gxy, double Gyy, double var, double mean, const Geom &g,
                                            uint8_t *__restrict__ records, uint8_t *__restrict__ flags, int frame, int blk) {
  const double nf = (double)((kBlock - 2) * (kBlock - 2));
  mean /= nf;
  Gxx /= nf;
  Gxy /= nf;
  Gyy /= nf;
  var = var / nf - mean * mean;
  const double trace = Gxx + Gyy;
  const double det = Gxx * Gyy - Gxy * Gxy;
  double disc = trace * trace - 4.0 * det;
  if (!(disc > 0.0)) disc = 0.0;
  const double sq = sqrt(disc);
  const double e1 = (trace + sq) / 2.0;
  const double e2 = (trace - sq) / 2.0;
  const double norm = e1;
  const double ratio = e1 / (e2 > 1e-6 ? e2 : 1e-6);
  const double kTrace = 0.15 / 1024.0, kRatio = 1.25, kNorm = 0.08 / 1024.0, kVar = 0.005 / 1024.0;
  const bool is_flat = (trace < kTrace) && (ratio < kRatio) && (norm < kNorm) && (var > kVar);
  double sw = -6682.0 * var + -0.2056 * ratio + 13087.0 * trace + -12434.0 * norm + 2.5694;
  sw = sw < -25.0 ? -25.0 : (sw > 100.0 ? 100.0 : sw);
  const float score = (float)(1.0 / (1.0 + exp(-sw)));
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  reinterpret_cast<float *>(rec + g.off_scores)[blk] = var > kVar ? score : 0.0f;
  flags[(size_t)frame * g.nblocks + blk] = is_flat ? 255 : 0;
}

// ----------------------------------------------------------------------------
// K1: flat-block features.  One lane per 32x32 block; every f64 sum runs in the
// reference's order (k = 0..1023 raster for the plane fit, (yi, xi) raster over
// the 30x30 interior for the gradient sums), so the four threshold decisions
// and the f32 score are those of the scalar CPU algorithm.
// grid = (ceil(nblocks/64), batch), block = 64.
// ----------------------------------------------------------------------------
// LISTED: the lane's block comes from a per-frame list (the blocks the certified fast path, k1f.hip.h,
// could not decide); otherwise lane = block.
template <int BPS, bool LISTED>
__global__ __launch_bounds__(64) void k1_flat_features(const FrameTable ft, Geom g,
                                                       FlatConsts fc, const double *__restrict__ lut_g,
                                                       uint8_t *__restrict__ records,
                                                       uint8_t *__restrict__ flags,
                                                       const uint32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ count) {
  __shared__ double lut[256];
  for (int i = threadIdx.x; i < 256; i += 64) lut[i] = lut_g[i];
  __syncthreads();
  const int frame = blockIdx.y;
  int blk = blockIdx.x * 64 + threadIdx.x;
  if (LISTED) {
    if (blk >= (int)count[frame]) return;
    blk = (int)list[(size_t)frame * g.nblocks + blk];
  }
  if (blk >= g.nblocks) return;
  const FramePlanes fp = ft.f[frame];
  const uint8_t *base = fp.src[0];
  const uint32_t stride = fp.src_stride[0];
  const int shift = g.src_shift;
  const int bx = blk % g.nbw, by = blk / g.nbw;
  const int ox = bx * kBlock, oy = by * kBlock;
  const bool fast = g.fast_rows && (ox + kBlock <= g.W);

  // ---- pass 1: t = block(1x1024) * A(1024x3), sequential sums from 0.0 ----
  double t0 = 0.0, t1 = 0.0, t2 = 0.0;
  auto fit_row = [&](int yi, const uint32_t (&pk)[8]) {
    const double yd = (double)(yi - 16) * 0.0625;
#pragma unroll
    for (int xi = 0; xi < kBlock; ++xi) {
      const double v = lut[(pk[xi >> 2] >> (8 * (xi & 3))) & 0xffu];
      const double xd = (double)(xi - 16) * 0.0625;
      t0 += v * yd;
      t1 += v * xd;
      t2 += v;  // v * 1.0
    }
  };
  if (fast) {
    // rows in groups of four: the next group's loads fly during this group's sums
    RowRaw<BPS> ra4[4], rb4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) load_row_raw<BPS>(base, stride, ox, min(oy + k, g.H - 1), ra4[k]);
    for (int y0 = 0; y0 < kBlock; y0 += 4) {
      if (y0 + 4 < kBlock) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load_row_raw<BPS>(base, stride, ox, min(oy + y0 + 4 + k, g.H - 1), rb4[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t pk[8];
        narrow_row<BPS>(ra4[k], shift, pk);
        fit_row(y0 + k, pk);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) ra4[k] = rb4[k];
    }
  } else {
    for (int yi = 0; yi < kBlock; ++yi) {
      uint32_t pk[8];
      load_row32<BPS>(base, stride, shift, ox, min(oy + yi, g.H - 1), g.W, false, pk);
      fit_row(yi, pk);
    }
  }
  // coef = AtA_inv(3x3) * t, each a sequential sum from 0.0
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
  c0 += fc.ata_inv[0] * t0; c0 += fc.ata_inv[1] * t1; c0 += fc.ata_inv[2] * t2;
  c1 += fc.ata_inv[3] * t0; c1 += fc.ata_inv[4] * t1; c1 += fc.ata_inv[5] * t2;
  c2 += fc.ata_inv[6] * t0; c2 += fc.ata_inv[7] * t1; c2 += fc.ata_inv[8] * t2;

  // ---- pass 2: residual rows + gradient covariance over the interior ----
  // residual(yi, xi) = block - (((0 + yd*c0) + xd*c1) + 1*c2)
  double ra[kBlock], rb[kBlock];
  // rows of pass 2 in order: 0, 1, then yi + 1 for yi = 1 .. 30, i.e. rows 0 .. 31; row r + 1 is
  // requested before row r is used
  RowRaw<BPS> nxt;
  if (fast) load_row_raw<BPS>(base, stride, ox, min(oy, g.H - 1), nxt);
  auto take_row = [&](int yi, uint32_t (&pk)[8]) {
    if (fast) {
      const RowRaw<BPS> cur = nxt;
      if (yi + 1 < kBlock) load_row_raw<BPS>(base, stride, ox, min(oy + yi + 1, g.H - 1), nxt);
      narrow_row<BPS>(cur, shift, pk);
    } else {
      load_row32<BPS>(base, stride, shift, ox, min(oy + yi, g.H - 1), g.W, false, pk);
    }
  };
  auto resid_row = [&](int yi, double (&out)[kBlock]) {
    uint32_t pk[8];
    take_row(yi, pk);
    const double yc = 0.0 + ((double)(yi - 16) * 0.0625) * c0;
#pragma unroll
    for (int xi = 0; xi < kBlock; ++xi) {
      const double xd = (double)(xi - 16) * 0.0625;
      const double fit = (yc + xd * c1) + c2;
      out[xi] = lut[(pk[xi >> 2] >> (8 * (xi & 3))) & 0xffu] - fit;
    }
  };
  double Gxx = 0.0, Gxy = 0.0, Gyy = 0.0, var = 0.0, mean = 0.0;
  // `up` holds row yi-1 and is overwritten in place by row yi+1 as we go
  auto grad_row = [&](int yi, double (&up)[kBlock], const double (&cur)[kBlock]) {
    uint32_t pk[8];
    take_row(yi + 1, pk);
    const double yc = 0.0 + ((double)(yi + 1 - 16) * 0.0625) * c0;
#pragma unroll
    for (int xi = 0; xi < kBlock; ++xi) {
      const double xd = (double)(xi - 16) * 0.0625;
      const double fit = (yc + xd * c1) + c2;
      const double nv = lut[(pk[xi >> 2] >> (8 * (xi & 3))) & 0xffu] - fit;
      if (xi >= 1 && xi <= kBlock - 2) {
        const double gx = (cur[xi + 1] - cur[xi - 1]) * 0.5;
        const double gy = (nv - up[xi]) * 0.5;
        Gxx += gx * gx;
        Gxy += gx * gy;
        Gyy += gy * gy;
        mean += cur[xi];
        var += cur[xi] * cur[xi];
      }
      up[xi] = nv;
    }
  };
  resid_row(0, ra);
  resid_row(1, rb);
  for (int yi = 1; yi < kBlock - 1; yi += 2) {
    grad_row(yi, ra, rb);      // ra: row yi-1 -> row yi+1
    grad_row(yi + 1, rb, ra);  // rb: row yi   -> row yi+2
  }

  flat_decide(Gxx, Gxy, Gyy, var, mean, g, records, flags, frame, blk);
}

// ----------------------------------------------------------------------------
// k1_flat_block<BPS>: the same literal evaluation, one WAVE per listed block (the few blocks the
// certified path leaves open: a lane-per-block launch of them is one long serial f64 chain per lane,
// ~0.1 ms whatever their number).  The products / residuals / gradient terms of the 1024 (900)
// pixels are worked out in parallel and laid down in LDS in raster order; every sum the reference
// takes sequentially is then one lane adding its array front to back from 0.0 (3 chains for the
// plane fit, 5 for the gradient sums), so each sum sees the same operands in the same order.
// grid = (kFbGrid), block = 64: the listed blocks of ALL frames of the launch, taken as one sequence, are dealt
// round-robin to one resident round of workgroups (3 to a CU by their 46 KB of LDS).  (A grid of 128 workgroups per
// frame was 8192 workgroups a 64-frame launch for ~420 listed blocks: dispatching the idle ones, 46 KB of LDS each,
// took as long as the work.)
// ----------------------------------------------------------------------------
constexpr int kFbGrid = 768;
constexpr int kFbInner = (kBlock - 2) * (kBlock - 2);
// GLOBAL: `list` is ONE sequence for the whole launch -- entry = frame * nblocks + block, count[0] its length (k1_certify's default
// mode appends to it with one atomic a wave): item p of workgroup w is entry w + p W, no per-frame counts to sum and search first
// (that prologue -- 64 counts from memory, their prefix, a walk over it -- was 5 of the kernel's 25 us).  Otherwise (the test aid
// "every block"): per-frame lists [batch][nblocks] + count[batch].
template <int BPS, bool GLOBAL>
__global__ __launch_bounds__(64) void k1_flat_block(const FrameTable ft, Geom g, FlatConsts fc,
                                                    const double *__restrict__ lut_g, uint8_t *__restrict__ records,
                                                    uint8_t *__restrict__ flags, const uint32_t *__restrict__ list,
                                                    const uint32_t *__restrict__ count, int nframes) {
  __shared__ double lut[256];
  __shared__ double s_v[kBlock * kBlock];  // pixel / 255, later the residual
  __shared__ double s_t[5 * kFbInner];     // [3][1024] fit products, then [5][900] gradient terms
  __shared__ uint32_t s_pre[GLOBAL ? 1 : kMaxBatch + 1];  // exclusive prefix of the frames' list lengths: the launch's sequence of blocks
  const int lane = threadIdx.x;
  const int W = (int)gridDim.x, w = (int)blockIdx.x;
  int total;
  if (GLOBAL) {
    total = (int)count[0];
    if (w >= total) return;
    // (the table is asked for while the entry is on its way)
    for (int i = lane; i < 256; i += 64) lut[i] = lut_g[i];
  } else {
    // (the wave sums the counts 64 frames at a time; a loop over the frames with a scalar load of each count in turn was most
    //  of the launch: 64 dependent round trips before the first block was touched)
    uint32_t carry = 0;
    if (lane == 0) s_pre[0] = 0;
    for (int base = 0; base < nframes; base += 64) {
      const uint32_t c = base + lane < nframes ? count[base + lane] : 0u;
      uint32_t incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += t;
      }
      if (base + lane < nframes) s_pre[base + lane + 1] = carry + incl;
      carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
    __syncthreads();
    total = (int)s_pre[nframes];
  }
  bool have_lut = GLOBAL;
  int frame = 0;
  for (int p = w; p < total; p += W) {
  int blk;
  if (GLOBAL) {
    const uint32_t e = list[p];
    frame = (int)(e / (uint32_t)g.nblocks);
    blk = (int)(e - (uint32_t)frame * (uint32_t)g.nblocks);
  } else {
    while ((int)s_pre[frame + 1] <= p) ++frame;  // (p grows: the search goes on from the last frame)
    blk = (int)list[(size_t)frame * g.nblocks + (p - (int)s_pre[frame])];
  }
  if (!have_lut) {
    for (int i = lane; i < 256; i += 64) lut[i] = lut_g[i];
    have_lut = true;
  }
  const FramePlanes fp = ft.f[frame];
  {
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int ox = bx * kBlock, oy = by * kBlock;
    __syncthreads();
    // ---- block(1x1024) and its products with the columns of A ----
    // (the lane's 16 pixels are requested at once: four at a time was four dependent round trips to memory)
    int pxs[kBlock * kBlock / 64];
#pragma unroll
    for (int k = 0; k < kBlock * kBlock / 64; ++k) {
      const int i = lane + 64 * k, yi = i >> 5, xi = i & 31;
      pxs[k] = load_px<BPS>(fp.src[0], fp.src_stride[0], g.src_shift, min(ox + xi, g.W - 1), min(oy + yi, g.H - 1));
    }
#pragma unroll
    for (int k = 0; k < kBlock * kBlock / 64; ++k) {
      const int i = lane + 64 * k, yi = i >> 5, xi = i & 31;
      const double v = lut[pxs[k]];
      const double yd = (double)(yi - 16) * 0.0625, xd = (double)(xi - 16) * 0.0625;
      s_v[i] = v;
      s_t[i] = v * yd;
      s_t[kBlock * kBlock + i] = v * xd;
      s_t[2 * kBlock * kBlock + i] = v;  // v * 1.0
    }
    __syncthreads();
    double sum = 0.0;
    if (lane < 3) {
      const double *src = s_t + lane * (kBlock * kBlock);
#pragma unroll 16
      for (int i = 0; i < kBlock * kBlock; ++i) sum += src[i];
    }
    const double t0 = __shfl(sum, 0, 64), t1 = __shfl(sum, 1, 64), t2 = __shfl(sum, 2, 64);
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    c0 += fc.ata_inv[0] * t0; c0 += fc.ata_inv[1] * t1; c0 += fc.ata_inv[2] * t2;
    c1 += fc.ata_inv[3] * t0; c1 += fc.ata_inv[4] * t1; c1 += fc.ata_inv[5] * t2;
    c2 += fc.ata_inv[6] * t0; c2 += fc.ata_inv[7] * t1; c2 += fc.ata_inv[8] * t2;
    __syncthreads();
    // ---- residual(yi, xi) = block - (((0 + yd*c0) + xd*c1) + 1*c2), in place ----
#pragma unroll 4
    for (int i = lane; i < kBlock * kBlock; i += 64) {
      const int yi = i >> 5, xi = i & 31;
      const double yc = 0.0 + ((double)(yi - 16) * 0.0625) * c0;
      const double xd = (double)(xi - 16) * 0.0625;
      const double fit = (yc + xd * c1) + c2;
      s_v[i] = s_v[i] - fit;
    }
    __syncthreads();
    // ---- gradient terms of the 30x30 interior, (yi, xi) raster ----
    for (int j = lane; j < kFbInner; j += 64) {
      const int yi = 1 + j / (kBlock - 2), xi = 1 + j % (kBlock - 2);
      const int o = yi * kBlock + xi;
      const double cur = s_v[o];
      const double gx = (s_v[o + 1] - s_v[o - 1]) * 0.5;
      const double gy = (s_v[o + kBlock] - s_v[o - kBlock]) * 0.5;
      s_t[j] = gx * gx;
      s_t[kFbInner + j] = gx * gy;
      s_t[2 * kFbInner + j] = gy * gy;
      s_t[3 * kFbInner + j] = cur;
      s_t[4 * kFbInner + j] = cur * cur;
    }
    __syncthreads();
    sum = 0.0;
    if (lane < 5) {
      const double *src = s_t + lane * kFbInner;
#pragma unroll 12
      for (int j = 0; j < kFbInner; ++j) sum += src[j];
    }
    const double Gxx = __shfl(sum, 0, 64), Gxy = __shfl(sum, 1, 64), Gyy = __shfl(sum, 2, 64);
    const double mean = __shfl(sum, 3, 64), var = __shfl(sum, 4, 64);
    if (lane == 0) flat_decide(Gxx, Gxy, Gyy, var, mean, g, records, flags, frame, blk);
  }
  }
}

// ----------------------------------------------------------------------------
// K2: per frame, threshold = scores_sorted_ascending[nblocks*90/100]; every
// block with score >= threshold gets `|= 1` (union with the 4-threshold flag).
// All scores are >= +0.0f, so their bit patterns order like the floats, and the
// k-th smallest pattern is the largest T with #{v < T} <= k: T is built bit by
// bit (32 counting rounds over register-resident scores, no atomics).
// grid = (batch), block = kK2Threads.
// ----------------------------------------------------------------------------
constexpr int kK2Threads = 1024;
constexpr int kK2PerThread = 8;  // scores kept in registers: up to 8192 blocks (a 4K frame has 8160); 32 for up to 32768 (8K: 32400)
// (the body: k2_flat_select below and the wide chain's k2w_select_units, k3w.hip.h, which builds the unit lists behind it)
// inclusive prefix sums over the 64 lanes of a wave, all in the VALU (DPP: the steps inside a row of 16, then the row totals)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast31 -> rows 2, 3
  return (uint32_t)x;
}

template <int PER>
__device__ __forceinline__ void k2_flat_select_sized(const Geom &g, uint8_t *__restrict__ records, const uint8_t *__restrict__ flags, int frame,
                                                     uint32_t *lds_bits) {
  // Three histograms in rotation and ONE barrier a pass: a pass counts into its own, which was zeroed a pass earlier; behind the
  // barrier EVERY wave finds the bin that holds the rank for itself (a lane takes 4 bins, a DPP scan over the wave: no
  // cross-wave step, no second and third barrier).  (Round 4: one histogram, four barriers a pass -- 16 waves meeting 17 times
  // were a third of the kernel.)
  __shared__ __attribute__((aligned(16))) uint32_t s_hist[3][256];
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  const uint32_t *sc = reinterpret_cast<const uint32_t *>(rec + g.off_scores);
  const int nb = g.nblocks;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool in_regs = nb <= kK2Threads * PER;
  const uint8_t *fl = flags + (size_t)frame * nb;
  uint32_t v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + k * kK2Threads;
    v[k] = i < nb ? sc[i] : 0xffffffffu;  // the filler sorts last
  }
  // (the finder's flag bytes are asked for now: they are needed when the threshold is known, a round trip to memory later)
  // (8 a thread; the 32 of an 8K frame would not fit the registers next to the scores: those are read when needed)
  constexpr bool kEarlyFlags = PER <= 8;
  uint8_t f[PER];
  if (kEarlyFlags && in_regs) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * kK2Threads;
      f[k] = i < nb ? fl[i] : 0;
    }
  }
  if (tid < 256) s_hist[0][tid] = 0;
  __syncthreads();
  // The k-th smallest pattern, a byte at a time from the top (radix select: 4 passes of histogram + scan; bit by bit it was
  // 32 rounds of count + barrier, 33 us a launch whatever the frame).  `rank` = 0-based rank among the patterns that share
  // the bytes decided so far.
  uint32_t rank = (uint32_t)(nb * 90 / 100);
  uint32_t thr = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const uint32_t himask = pass ? ~0u << (shift + 8) : 0u;  // the bytes decided so far
    uint32_t *hist = s_hist[pass % 3];
    if (tid < 256) s_hist[(pass + 1) % 3][tid] = 0;  // (the next pass's: its last readers are two barriers behind)
    // Scores cluster: the top byte of a score in [0, 1] takes two or three values, and many blocks share one score (0, or a
    // saturated sigmoid; the fillers of a small frame).  64 lanes adding to one LDS word are served one after the other, so in
    // the first pass the wave counts each distinct byte with a ballot and adds once (23 instead of 28 us a 4K launch); in the
    // later passes, whose bytes are spread, that only pays where a thread holds many scores (an 8K frame): two such rounds,
    // then the lanes left over add for themselves.
    if (in_regs && pass == 0) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t d = v[k] >> 24;
        unsigned long long todo = __ballot(1);
        while (todo) {
          const int first = __ffsll((long long)todo) - 1;
          const uint32_t dv = (uint32_t)__shfl((int)d, first, 64);
          const unsigned long long same = __ballot(d == dv) & todo;
          if (lane == first) atomicAdd(&hist[dv], (uint32_t)__popcll(same));
          todo &= ~same;
        }
      }
    } else if (in_regs && PER > 8) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const bool in = (v[k] & himask) == thr;
        const uint32_t d = (v[k] >> shift) & 0xffu;
        unsigned long long todo = __ballot(in);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (!todo) break;
          const int first = __ffsll((long long)todo) - 1;
          const uint32_t dv = (uint32_t)__shfl((int)d, first, 64);
          const unsigned long long same = __ballot(in && d == dv) & todo;
          if (lane == first) atomicAdd(&hist[dv], (uint32_t)__popcll(same));
          todo &= ~same;
        }
        if ((todo >> lane) & 1ull) atomicAdd(&hist[d], 1u);
      }
    } else if (in_regs) {
#pragma unroll
      for (int k = 0; k < PER; ++k)
        if ((v[k] & himask) == thr) atomicAdd(&hist[(v[k] >> shift) & 0xffu], 1u);
    } else {
      for (int i = tid; i < nb; i += kK2Threads) {
        const uint32_t x = sc[i];
        if ((x & himask) == thr) atomicAdd(&hist[(x >> shift) & 0xffu], 1u);
      }
    }
    __syncthreads();
    // the bin that holds the rank: every wave for itself (lane = bins 4 lane .. 4 lane + 3)
    {
      const uint4 c = *reinterpret_cast<const uint4 *>(&hist[4 * lane]);
      const uint32_t tot = (c.x + c.y) + (c.z + c.w);
      const uint32_t incl = wave_scan_incl(tot), excl = incl - tot;
      const bool mine = tot != 0 && excl <= rank && rank < incl;
      // (exactly one lane: the histogram holds more than `rank` entries)
      uint32_t bin = 4u * (uint32_t)lane, ex = excl;
      if (rank >= ex + c.x) {
        ex += c.x, ++bin;
        if (rank >= ex + c.y) {
          ex += c.y, ++bin;
          if (rank >= ex + c.z) ex += c.z, ++bin;
        }
      }
      const unsigned long long who = __ballot(mine);
      const int src = __ffsll((long long)who) - 1;
      thr |= (uint32_t)__builtin_amdgcn_readlane((int)bin, src) << shift;
      rank -= (uint32_t)__builtin_amdgcn_readlane((int)ex, src);
    }
  }
  // thr = bit pattern of the threshold score
  uint8_t *mask = rec + g.off_mask;

  // (the scores from the registers where they are.  `lds_bits`, when the caller builds the unit lists: the mask as a bitmap in
  //  raster order, bit 32 + i = block i is flat, a zero word in front and zeros behind -- a wave's 64 lanes hold 64 consecutive
  //  blocks, so a ballot is two words of it)
  if (in_regs) {
    if (!kEarlyFlags) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = tid + k * kK2Threads;
        f[k] = i < nb ? fl[i] : 0;
      }
    }
    if (lds_bits && tid < 3) lds_bits[tid == 0 ? 0 : kK2Threads * PER / 32 + tid] = 0u;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * kK2Threads;
      const uint8_t mb = i < nb ? (uint8_t)(f[k] | (v[k] >= thr ? 1 : 0)) : (uint8_t)0;
      if (i < nb) mask[i] = mb;
      if (lds_bits) {
        const unsigned long long bal = __ballot(mb != 0);
        if (lane == 0) {
          uint32_t *w = lds_bits + 1 + ((k * kK2Threads + (tid & ~63)) >> 5);
          w[0] = (uint32_t)bal;
          w[1] = (uint32_t)(bal >> 32);
        }
      }
    }
  } else {
    for (int i = tid; i < nb; i += kK2Threads) mask[i] = fl[i] | (sc[i] >= thr ? 1 : 0);
  }
}
// (the bitmap's size in words: the blocks the registers hold, a word in front, two behind)
constexpr int kK2BitWords = kK2Threads * 32 / 32 + 3;
__device__ __forceinline__ void k2_flat_select_body(const Geom &g, uint8_t *__restrict__ records, const uint8_t *__restrict__ flags, int frame,
                                                    uint32_t *lds_bits = nullptr) {
  if (g.nblocks <= kK2Threads * kK2PerThread) k2_flat_select_sized<kK2PerThread>(g, records, flags, frame, lds_bits);
  else k2_flat_select_sized<32>(g, records, flags, frame, lds_bits);  // (8K: 32 400 scores, still in registers; larger frames re-read them)
}
__global__ __launch_bounds__(kK2Threads) void k2_flat_select(Geom g, uint8_t *__restrict__ records,
                                                             const uint8_t *__restrict__ flags) {
  k2_flat_select_body(g, records, flags, (int)blockIdx.x);
}

// ----------------------------------------------------------------------------
// K3 (generic): exact integer AR sums + block statistics, any lag 1..3.
// One workgroup walks a strided subset of the frame's blocks; for each flat
// block it stages the residual tile d = src8 - den8 (+halo `lag` left/right/up)
// in LDS as int32, and every thread owns up to two (i, j) products of the
// (n+1)-vector [d(p+c_0) .. d(p+c_{n-1}), (luma residual sum for chroma), d(p)].
// Per-block partials are int32 (|product| <= 1020^2, <= 1024 samples), folded
// into int64 per thread, and added to the record with one atomic per product.
// grid = (chunks, nplanes, batch), block = 256.
// ----------------------------------------------------------------------------
constexpr int kK3Threads = 256;
constexpr int kMaxTile = (kBlock + 6) * (kBlock + 3);  // d tile, lag 3
constexpr int kMaxPairs = 350;                         // 25*26/2 + 25

__device__ __forceinline__ int block_reduce_sum(int v, int *scratch) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// `only` (optional): per-frame block lists [batch][3][nblocks] (one per plane); when given, only the
// blocks marked there are processed (the blocks the lag-3 fast kernel deferred
// because |d| > 127), and frames with only_any[frame] == 0 exit at once.
// (the body: chunk bx of nbx of plane c of the frame; k3_ar_generic below and the wide chain's tail kernel, k3w.hip.h, call it)
__device__ __forceinline__ void k3_ar_generic_body(const FrameTable ft, const Geom &g, uint8_t *__restrict__ records,
                                                   const uint8_t *__restrict__ only, const uint32_t *__restrict__ only_any,
                                                   int bx, int nbx, int c, int frame) {
  __shared__ int tile[kMaxTile + kBlock * kBlock];  // d tile, then luma-sum tile
  __shared__ int red[4];
  if (only_any && only_any[frame] == 0) return;
  const uint8_t *only_f = only ? only + ((size_t)frame * 3 + c) * g.nblocks : nullptr;
  const FramePlanes fp = ft.f[frame];
  uint8_t *rec = records + (size_t)frame * g.rec_size;
  const uint8_t *mask = rec + g.off_mask;
  const int lag = g.lag, n = g.n;
  const int nc = n + (c > 0);
  const int sx = c ? g.xdec : 0, sy = c ? g.ydec : 0;
  const int pw = g.W >> sx, ph = g.H >> sy;
  const int bw = kBlock >> sx, bh = kBlock >> sy;
  const int TW = bw + 2 * lag, TH = bh + lag;
  const int ltile0 = TW * TH;
  const int ntri = nc * (nc + 1) / 2;
  const int npairs = ntri + nc;

  // operand k of the (nc+1)-vector -> (base offset, row stride) in `tile`
  auto operand = [&](int k, int &base, int &stride) {
    if (k < n) {  // causal neighbour k: rows -lag..0, cols -lag..lag (row 0: -lag..-1)
      const int side = 2 * lag + 1;
      const int cy = k / side - lag, cx = k % side - lag;
      base = (lag + cy) * TW + (lag + cx);
      stride = TW;
    } else if (k < nc) {  // chroma: co-located luma residual sum
      base = ltile0;
      stride = bw;
    } else {  // the sample itself
      base = lag * TW + lag;
      stride = TW;
    }
  };
  int pa[2], sa[2], pb[2], sb[2], out_idx[2];
  bool have[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int p = threadIdx.x + s * kK3Threads;
    have[s] = p < npairs;
    int i = 0, j = 0;
    if (p < ntri) {
      int rem = p;
      i = 0;
      while (rem >= nc - i) {
        rem -= nc - i;
        ++i;
      }
      j = i + rem;
      out_idx[s] = i * nc + j;
    } else {
      i = p - ntri;
      j = nc;  // target
      out_idx[s] = nc * nc + i;
    }
    operand(have[s] ? i : 0, pa[s], sa[s]);
    operand(have[s] ? j : 0, pb[s], sb[s]);
  }
  long long acc64[2] = {0, 0};
  long long nobs = 0;

  const uint8_t *sp = fp.src[c], *dp = fp.den[c];
  const uint32_t sst = fp.src_stride[c], dst = fp.den_stride[c];

  for (int blk = bx; blk < g.nblocks; blk += nbx) {
    if (!mask[blk]) continue;
    if (only_f && !only_f[blk]) continue;
    const int bx = blk % g.nbw, by = blk / g.nbw;
    const int x_o = bx * bw, y_o = by * bh;
    // ---- stage the residual tile ----
    int s_d = 0, s_d2 = 0, s_l = 0;
    for (int idx = threadIdx.x; idx < TW * TH; idx += kK3Threads) {
      const int tx = idx % TW, ty = idx / TW;
      const int X = x_o - lag + tx, Y = y_o - lag + ty;
      int d = 0;
      if (X >= 0 && X < pw && Y >= 0 && Y < ph) {
        const int s = load_px_rt(sp, sst, g.src_bps, g.src_shift, X, Y);
        d = s - load_px_rt(dp, dst, g.den_bps, g.den_shift, X, Y);
        if (tx >= lag && tx < lag + bw && ty >= lag) {  // block proper (clipped by the plane)
          s_d += d;
          s_d2 += d * d;
          if (c == 0) s_l += s;
        }
      }
      tile[idx] = d;
    }
    if (c > 0) {
      for (int idx = threadIdx.x; idx < bw * bh; idx += kK3Threads) {
        const int x = idx % bw, y = idx / bw;
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
        tile[ltile0 + idx] = L;
      }
    }
    // ---- block statistics (get_block_mean / get_noise_var as exact sums) ----
    {
      const int td = block_reduce_sum(s_d, red);
      const int td2 = block_reduce_sum(s_d2, red);
      const int tl = c == 0 ? block_reduce_sum(s_l, red) : 0;
      if (threadIdx.x == 0) {
        reinterpret_cast<int32_t *>(rec + g.off_sum_d[c])[blk] = td;
        reinterpret_cast<uint32_t *>(rec + g.off_sum_d2[c])[blk] = (uint32_t)td2;
        if (c == 0) reinterpret_cast<uint32_t *>(rec + g.off_luma_sum)[blk] = (uint32_t)tl;
      }
    }
    __syncthreads();
    // ---- observation window of this block (add_block_observations) ----
    const int y_start = (by > 0 && mask[(by - 1) * g.nbw + bx]) ? 0 : lag;
    const int x_start = (bx > 0 && mask[by * g.nbw + bx - 1]) ? 0 : lag;
    const int y_end = min(ph - y_o, bh);
    const int x_end = min(pw - x_o - lag, (bx + 1 < g.nbw && mask[by * g.nbw + bx + 1]) ? bw : (bw - lag));
    if (x_end > x_start && y_end > y_start) {
      nobs += (long long)(x_end - x_start) * (y_end - y_start);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!have[s]) continue;
        int acc = 0;
        for (int y = y_start; y < y_end; ++y) {
          const int *ra = tile + pa[s] + y * sa[s];
          const int *rb = tile + pb[s] + y * sb[s];
          for (int x = x_start; x < x_end; ++x) acc += ra[x] * rb[x];
        }
        acc64[s] += acc;
      }
    }
    __syncthreads();
  }
  unsigned long long *ar = reinterpret_cast<unsigned long long *>(rec + g.off_ar[c]);
#pragma unroll
  for (int s = 0; s < 2; ++s)
    if (have[s] && acc64[s] != 0) atomicAdd(&ar[out_idx[s]], (unsigned long long)acc64[s]);
  if (threadIdx.x == 0 && nobs != 0) atomicAdd(&ar[nc * nc + nc], (unsigned long long)nobs);
}
__global__ __launch_bounds__(kK3Threads) void k3_ar_generic(const FrameTable ft, Geom g,
                                                            uint8_t *__restrict__ records,
                                                            const uint8_t *__restrict__ only,
                                                            const uint32_t *__restrict__ only_any) {
  k3_ar_generic_body(ft, g, records, only, only_any, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)blockIdx.z);
}

}  // namespace g1s

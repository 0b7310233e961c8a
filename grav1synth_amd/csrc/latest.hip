// latest.hip -- k4_latest: the per-frame half of the fold (fold.cpp: compute_latest) on the device.
//
// What the reference does per frame after the pixel work (av1-grain diff/solver.rs NoiseModel::update, the part before the
// combined model is touched: the latest state's AR solve, the noise-strength measurements of the flat blocks, the 20-bin
// strength solve) is f64 arithmetic in a fixed order on a few hundred KB of exact integers.  On the host it costs 72 - 85 us of
// a core per 4K frame and needs the whole 285 KB record over PCIe; here one workgroup per frame does it and the host receives
// the 27 KB latest-state blob of fold.h -- the same bytes compute_latest + latest_to_blob produce
// (tests/test_gpu_parity.py::test_device_latest_*):
//   * every f64 operation is the host's operation on the host's operands (-ffp-contract=off on both sides; IEEE divide and
//     square root; int64 -> f64 in one rounding);
//   * every f64 SUM runs in the host's order.
// Round 5 rebuilt the kernel around SHORT serial chains (round 3's took 1.1 ms a launch: barriers around every elimination step,
// chains of dependent L2 round trips):
//   * a linear system is solved by ONE wave with the rows in registers (lane = row, columns statically indexed): the
//     reference's "bubble the larger magnitude up" pivoting is a suffix arg-max over the column (the element carried past
//     position i is the first maximum of positions i .. n - 1, position i keeps the loser of its comparison), taken with five
//     shuffle steps; the pivot row travels by v_readlane; back substitution by every lane on its own row, the wanted lane's
//     result broadcast.  No barrier, no LDS.  The three planes' AR systems are solved by three waves at once;
//   * the strength system's entries are sums over the measured blocks in raster order.  Entry (k, k), (k + 1, k) = (k, k + 1)
//     and b[k] only ever meet blocks of bins k - 1 and k: the blocks are taken 512 at a time, their terms computed a thread a
//     block, partitioned by bin IN ORDER into lists in LDS (ballots and popcounts: a stable partition), and lane k adds its
//     list front to back -- one LDS read and one addition an element, no selection, no memory latency in the chain; `total`
//     is one sum over all blocks: a wave reads 64 terms at a time and adds them lane by lane (v_readlane).  Both chroma
//     planes go through one pass (same blocks, same bins).
#include "latest_dev.h"

#include "fold.h"

#include <stdio.h>

namespace g1s {
namespace {

constexpr int kT = 256, kWaves = kT / 64;
constexpr int kChunk = 512, kRounds = kChunk / kT;  // blocks a chunk; a thread's blocks of a chunk: base + kT * r + tid
constexpr double kTinyD = 1.0e-16;                  // TINY_NEAR_ZERO
constexpr double kNorm2D = 255.0 * 255.0;           // BLOCK_NORMALIZATION^2
constexpr int kNL = 2 * kNumBins;                   // lists of a chunk: D 0 .. 19 (bins k - 1 and k), L 20 .. 38 (bin k alone), T 39 (every block)
constexpr int kArenaD = 2 * kChunk + 8 * kNumBins + 8, kArenaL = kChunk + 8 * kNumBins + 8;
static_assert(kNumBins == 20 && kMaxN == 25, "latest.hip: the solvers' static sizes");

struct Shared {
  // a chunk's partitioned terms
  // (a list's slots start at a multiple of 8 and are zero-filled up to the next one: the chains add whole groups of eight --
  //  a +0.0 changes no sum that is >= +0 or NaN -- and read one group ahead)
  double D[kArenaD];      // matrix diagonal terms, list k at [beg[k], end[k])
  double Lw[kArenaL];     // off-diagonal terms, list 20 + k
  double Bv[2][kArenaD];  // b terms of the pass' planes, the D lists' slots
  double Tv[2][kChunk];   // noise stds in block order (list 39)
  uint16_t cnt[kRounds][kWaves][kNL];  // (16-bit: the workgroup's LDS stays under a quarter of a CU's, and with it the kernel at four waves a SIMD)
  uint16_t off[kRounds][kWaves][kNL];
  uint32_t beg[kNL], end[kNL];
  // results
  double arx[3][kMaxN + 1];
  double gain[3];
  long long nobs[3];
  int ar_ok[3];
  double diag[kNumBins], low[kNumBins], bsum[2][kNumBins], total[2];
  double sb[2][kNumBins], sx[2][kNumBins];
  int st_ok[2];
  double luma_x[kNumBins];
  uint32_t ne;
  uint32_t wsum[kWaves];
};

__device__ __forceinline__ double readlane_f64(double v, int j) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), j);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// gauss_solve of fold.cpp by one wave: lane r < n holds row r (a[j], j < n) and b of the system; x comes back in every lane
// (x must come in as what a failed solve leaves behind: the caller's cleared x).  The rows stay in their lanes; `rap` (lane p:
// the row at position p) is what the reference's row swaps move.
//   bubble pass over column k, bottom up:  before step i position i holds w(i), the first maximum of |column| over positions
//   i .. n - 1 (`!(up < carried)` moves on to the upper element on a tie), position i - 1 its own element; the larger moves
//   to i - 1.  So after the pass position k holds w(k) and position i > k holds (|c[i - 1]| < |c[w(i)]|) ? row(i - 1) : w(i).
//   w(.) is a suffix arg-max with the smaller position winning ties: five shuffle steps.  (No NaN can arise: the pivot is the
//   column's largest magnitude, every multiplier is <= 1 in magnitude.)
template <int NMAX>
__device__ __forceinline__ bool wave_gauss(const int n, double (&a)[NMAX], double b, double (&x)[NMAX], const int lane) {
  int rap = lane;
  bool done = false, ok = true;  // (ok: wave-uniform; a failed solve runs on without touching anything -- no early return out of the unrolled loops, which would send the rows to scratch memory)
#pragma unroll
  for (int k = 0; k < NMAX - 1; ++k) {
    if (k < n - 1 && ok) {
      // |column k| by position
      const double ck = fabs(a[k]);
      double cp = __shfl(ck, rap, 64);
      if (lane < k || lane >= n) cp = -1.0;
      double val = cp;
      int idx = lane;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const double v2 = __shfl_down(val, d, 64);
        const int i2 = __shfl_down(idx, d, 64);
        if (lane + d < 64 && v2 > val) {
          val = v2;
          idx = i2;
        }
      }
      const int rap_w = __shfl(rap, idx, 64);
      const int rap_up = __shfl_up(rap, 1, 64);
      const double cp_up = __shfl_up(cp, 1, 64);
      if (lane >= k && lane < n) rap = (lane == k) ? rap_w : ((cp_up < val) ? rap_up : rap_w);
      const int pr = uniform(__shfl(rap, k, 64));
      const double pk = readlane_f64(a[k], pr), bk = readlane_f64(b, pr);
      if (fabs(pk) < kTinyD) {
        ok = false;
      } else {
        if (lane == pr) done = true;
        const bool active = lane < n && !done;
        const double c = a[k] / pk;
#pragma unroll
        for (int j = k + 1; j < NMAX; ++j) {
          const double pj = readlane_f64(a[j], pr);
          const double t = c * pj;
          if (active && j < n) a[j] = a[j] - t;
        }
        const double t = c * bk;
        if (active) b = b - t;
      }
    }
  }
#pragma unroll
  for (int i = NMAX - 1; i >= 0; --i) {
    if (i < n && ok) {
      const int r = uniform(__shfl(rap, i, 64));
      const double piv = readlane_f64(a[i], r);
      if (fabs(piv) < kTinyD) {
        ok = false;
      } else {
        double c = 0;
#pragma unroll
        for (int j = i + 1; j < NMAX; ++j) {
          const double t = a[j] * x[j];  // (j >= n: a column of zeros times the cleared x: adds +0.0 to a sum that is not -0.0 ... kept out all the same)
          if (j < n) c += t;
        }
        const double xi = (b - c) / a[i];
        x[i] = __shfl(xi, r, 64);  // (a broadcast into vector registers: 25 solutions as scalar pairs do not fit the scalar file)
      }
    }
  }
  return ok;
}

__device__ inline double dev_clamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ inline double dev_bin_index(double value) { return (kNumBins - 1) * dev_clamp(value, 0.0, 255.0) / 255.0; }

// v / count: a division by a power of two (whole blocks' sample counts) is the scaling by it, bit for bit
__device__ __forceinline__ double div_count(double v, int count) {
  return (count & (count - 1)) == 0 ? __builtin_ldexp(v, -__builtin_ctz((unsigned)count)) : v / count;
}

__device__ inline void put_text(char *dst, const char *msg) {
  int i = 0;
  for (; msg[i] && i < 103; ++i) dst[i] = msg[i];
  dst[i] = 0;
}

#ifdef G1S_LATEST_TIMERS
#define K4_TICK(slot)                                 \
  do {                                                \
    __syncthreads();                                  \
    if (tid == 0 && frame == 0) {                     \
      const unsigned long long now_ = wall_clock64(); \
      k4_dbg[slot] += (double)(now_ - k4_t);          \
      k4_t = now_;                                    \
    }                                                 \
  } while (0)
#else
#define K4_TICK(slot) \
  do {                \
  } while (0)
#endif

struct FrameCtx {
  const uint8_t *rec;
  const uint8_t *mask;
  const uint32_t *luma_sum;
  int nb, nbw, W, H;
};

// One pass over the frame's blocks for the planes c0 .. c0 + NP - 1 (one geometry): the strength system's sums.
// Out: sh.diag / low (the matrix side: bins alone), sh.bsum[q], sh.total[q], sh.ne.
template <int NP>
__device__ __forceinline__ void strength_pass(Shared &sh, const LatestJob &job, const FrameCtx &fc, const int c0, const int tid, double *k4_dbg, const int frame) {
#ifdef G1S_LATEST_TIMERS
  unsigned long long k4_t = wall_clock64();
#endif
  const int lane = tid & 63, wave = tid >> 6;
  const bool is_chroma = c0 != 0;
  const int sx = is_chroma ? job.xdec : 0, sy = is_chroma ? job.ydec : 0;
  const int bw = kBlock >> sx, bh = kBlock >> sy;
  const int pw = job.W >> sx, ph = job.H >> sy;
  const int32_t *sum_d[NP];
  const uint32_t *sum_d2[NP];
  double noise_gain[NP], corr[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    sum_d[q] = reinterpret_cast<const int32_t *>(fc.rec + job.L.off_sum_d[c0 + q]);
    sum_d2[q] = reinterpret_cast<const uint32_t *>(fc.rec + job.L.off_sum_d2[c0 + q]);
    noise_gain[q] = sh.gain[c0 + q];
    corr[q] = is_chroma ? sh.arx[c0 + q][job.n] : 0;
  }
  const double luma_gain = sh.gain[0];
  // the chains' accumulators: wave 0 lane k < 20: (k, k) and b[k] of every plane; lanes 20 .. 38: (k + 1, k); wave 1: the totals
  // (one chain a lane -- wave 0: lanes 0 .. 19 the (k, k) sums, 20 .. 39 plane 0's b[k], 40 .. 59 plane 1's; wave 2 lanes 0 .. 18
  //  the (k + 1, k) sums; waves 1 and 3 the planes' totals: an addition waits for the one before it, the chains next to each other
  //  in a wave cost nothing)
  double acc = 0;
  uint32_t ne = 0;

  // raw words of a chunk (requested a chunk ahead)
  struct Raw {
    uint32_t m, ls;
    int32_t sd[NP];
    uint32_t sd2[NP];
  };
  Raw nxt[kRounds];
  auto fetch = [&](int base) {
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int bi = base + kT * r + tid;
      const bool in = bi < fc.nb;
      nxt[r].m = in ? fc.mask[bi] : 0u;
      nxt[r].ls = in ? fc.luma_sum[bi] : 0u;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        nxt[r].sd[q] = in ? sum_d[q][bi] : 0;
        nxt[r].sd2[q] = in ? sum_d2[q][bi] : 0u;
      }
    }
  };
  fetch(0);
  for (int base = 0; base < fc.nb; base += kChunk) {
    Raw cur[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) cur[r] = nxt[r];
    if (base + kChunk < fc.nb) fetch(base + kChunk);
    // ---- a thread a block: the measurement and its terms ----
    int key[kRounds];
    double tU[kRounds], tW[kRounds], tV[kRounds], tP[kRounds][NP], tQ[kRounds][NP], tS[kRounds][NP];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int bi = base + kT * r + tid;
      key[r] = -1;
      tU[r] = tW[r] = tV[r] = 0;
#pragma unroll
      for (int q = 0; q < NP; ++q) tP[r][q] = tQ[r][q] = tS[r][q] = 0;
      if (cur[r].m == 0) continue;
      const int by = bi / fc.nbw, bx = bi - by * fc.nbw;
      const int shh = min(ph - by * bh, bh), sw = min(pw - bx * bw, bw);
      if (!(sw * shh > kBlock)) continue;
      const int lw = min(fc.W - bx * kBlock, kBlock), lh = min(fc.H - by * kBlock, kBlock);
      const double block_mean = div_count((double)cur[r].ls, lw * lh);
      const double bin = dev_bin_index(block_mean);
      const int i0 = (int)bin;  // (bin >= 0: the floor)
      const int i1 = min(kNumBins - 1, i0 + 1);
      const double a = bin - i0;
      key[r] = i0;
      tU[r] = (1.0 - a) * (1.0 - a);
      tW[r] = a * (1.0 - a);
      tV[r] = a * a;
      const double luma_strength = is_chroma ? luma_gain * ((1.0 - a) * sh.luma_x[i0] + a * sh.luma_x[i1]) : 0;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        double noise_mean = (double)cur[r].sd[q];
        const double noise_sq = (double)cur[r].sd2[q];
        noise_mean = div_count(noise_mean, sw * shh);
        const double noise_var = div_count(noise_sq, sw * shh) - noise_mean * noise_mean;
        const double cl = corr[q] * luma_strength;
        const double t0 = noise_var / 16, t1 = noise_var - cl * cl;
        const double uncorr_std = sqrt(t0 > t1 ? t0 : t1);
        const double noise_std = uncorr_std / noise_gain[q];
        tS[r][q] = noise_std;
        tP[r][q] = (1.0 - a) * noise_std;
        tQ[r][q] = a * noise_std;
      }
    }
    K4_TICK(6);
    // ---- stable partition by bin: ranks inside the wave by ballots ----
    // list k (D): blocks of bin k ("own": the (k, k) term (1 - a)^2, the b term (1 - a) std) and of bin k - 1 ("prev": a^2, a std)
    // list 20 + k (L): blocks of bin k alone (a (1 - a)); list 39 (T): every measured block.
    // A block of the last bin is its own neighbour (i1 = i0 = 19) -- and sits exactly on it: bin = 19 means a = 0, so what the
    // reference adds besides (1 - a)^2 and (1 - a) std are zeros (a (1 - a), a^2, a std with a finite or NaN std), which change
    // no sum that is >= +0 or NaN: the block goes to list 19 as "own" only.
    uint32_t rk_own[kRounds], rk_prev[kRounds], rk_low[kRounds], rk_all[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      unsigned long long prevB = 0;
      uint32_t mycnt = 0;
      rk_own[r] = rk_prev[r] = rk_low[r] = rk_all[r] = 0;
      const int kk = key[r];
#pragma unroll
      for (int k = 0; k < kNumBins; ++k) {
        const unsigned long long Bk = __ballot(kk == k);
        const unsigned long long M = Bk | prevB;
        if (M == 0) continue;  // (a frame's blocks sit in a few bins: most lists of a chunk are empty)
        const uint32_t preM = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
        const uint32_t preB = __builtin_amdgcn_mbcnt_hi((uint32_t)(Bk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Bk, 0u));
        if (kk == k) rk_own[r] = preM, rk_low[r] = preB;
        if (kk == k - 1) rk_prev[r] = preM;
        if (lane == k) mycnt = (uint32_t)__popcll(M);
        if (lane == kNumBins + k) mycnt = (uint32_t)__popcll(Bk);
        prevB = Bk;
      }
      {
        const unsigned long long Ma = __ballot(kk >= 0);
        rk_all[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(Ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Ma, 0u));
        if (lane == kNL - 1) mycnt = (uint32_t)__popcll(Ma);
      }
      if (lane < kNL) sh.cnt[r][wave][lane] = (uint16_t)mycnt;
    }
    __syncthreads();
    K4_TICK(7);
    // ---- offsets: list l's slots of (round, wave), rounds first (block order) ----
    if (wave == 0) {
      uint32_t run = 0;
      if (lane < kNL) {
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
#pragma unroll
          for (int w = 0; w < kWaves; ++w) {
            sh.off[r][w][lane] = (uint16_t)run;
            run += sh.cnt[r][w][lane];
          }
      }
      // the lists' bases inside their arenas (each list's room a multiple of 8): D lists 0 .. 19 one behind the other, L lists
      // 20 .. 38 likewise, T at 0
      const uint32_t room = (run + 7u) & ~7u;
      uint32_t incl = room;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += o;
      }
      const uint32_t upto19 = (uint32_t)__shfl((int)incl, kNumBins - 1, 64);
      uint32_t bse = incl - room;
      if (lane >= kNumBins) bse -= upto19;
      if (lane == kNL - 1) bse = 0;
      if (lane < kNL) {
        sh.beg[lane] = bse;
        sh.end[lane] = bse + run;
        // the zeros behind the list
        if (lane < kNumBins) {
          for (uint32_t i = bse + run; i < bse + room; ++i) {
            sh.D[i] = 0.0;
#pragma unroll
            for (int q = 0; q < NP; ++q) sh.Bv[q][i] = 0.0;
          }
        } else if (lane < kNL - 1) {
          for (uint32_t i = bse + run; i < bse + room; ++i) sh.Lw[i] = 0.0;
        }
      }
    }
    __syncthreads();
    K4_TICK(8);
    // ---- the terms into their lists ----
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int kk = key[r];
      if (kk < 0) continue;
      const uint32_t so = sh.beg[kk] + sh.off[r][wave][kk] + rk_own[r];
      sh.D[so] = tU[r];
#pragma unroll
      for (int q = 0; q < NP; ++q) sh.Bv[q][so] = tP[r][q];
      if (kk < kNumBins - 1) {
        const uint32_t sp = sh.beg[kk + 1] + sh.off[r][wave][kk + 1] + rk_prev[r];
        sh.D[sp] = tV[r];
#pragma unroll
        for (int q = 0; q < NP; ++q) sh.Bv[q][sp] = tQ[r][q];
        const uint32_t sl = sh.beg[kNumBins + kk] + sh.off[r][wave][kNumBins + kk] + rk_low[r];
        sh.Lw[sl] = tW[r];
      }
      const uint32_t st = sh.off[r][wave][kNL - 1] + rk_all[r];
#pragma unroll
      for (int q = 0; q < NP; ++q) sh.Tv[q][st] = tS[r][q];
    }
    __syncthreads();
    K4_TICK(9);
    // ---- the chains: a list front to back ----
    if (wave == 0 || wave == 2) {
      // wave 0: lanes 0 .. 19 the diagonal terms of list k, 20 .. 39 / 40 .. 59 the planes' b terms; wave 2: lanes 0 .. 18 the
      // off-diagonal terms.  Groups of eight, the next group's reads in flight while this one is added.
      const int arr = wave == 0 ? lane / kNumBins : 3;
      const int li = wave == 2 ? kNumBins + lane : lane % kNumBins;
      const bool on = wave == 2 ? lane < kNumBins - 1 : arr <= NP;
      const double2 *src = reinterpret_cast<const double2 *>(arr == 0 ? sh.D : arr == 1 ? sh.Bv[0] : arr == 2 ? sh.Bv[NP - 1] : sh.Lw);
      if (on) {
        uint32_t i = sh.beg[li] >> 1;
        const uint32_t e = (sh.end[li] + 1) >> 1;
        double2 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) va[u] = src[i + u];
        while (i < e) {
#pragma unroll
          for (int u = 0; u < 4; ++u) vb[u] = src[i + 4 + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += va[u].x, acc += va[u].y;
          i += 4;
          if (i < e) {
#pragma unroll
            for (int u = 0; u < 4; ++u) va[u] = src[i + 4 + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += vb[u].x, acc += vb[u].y;
            i += 4;
          }
        }
      }
    } else if (wave == 1 || NP == 2) {
      // the totals: waves 1 and 3, a plane each; 64 stds at a time, added lane by lane
      const int q = wave == 1 ? 0 : NP - 1;
      const uint32_t e = sh.end[kNL - 1];
      for (uint32_t b0 = 0; b0 < e; b0 += 64) {
        const double v = b0 + lane < e ? sh.Tv[q][b0 + lane] : 0.0;
        const int cntc = (int)min(64u, e - b0);
        if (cntc == 64) {
#pragma unroll
          for (int j = 0; j < 64; ++j) acc += readlane_f64(v, j);
        } else {
          for (int j = 0; j < cntc; ++j) acc += readlane_f64(v, j);
        }
      }
      if (wave == 1) ne += e;
    }
    __syncthreads();  // (the lists are written again by the next chunk)
    K4_TICK(10);
  }
  if (wave == 0) {
    const int k = lane % kNumBins, arr = lane / kNumBins;
    if (arr == 0) sh.diag[k] = acc;
    else if (arr <= NP) sh.bsum[arr - 1][k] = acc;
  } else if (wave == 2) {
    if (lane < kNumBins - 1) sh.low[lane] = acc;
  } else if (lane == 0 && (wave == 1 || NP == 2)) {
    sh.total[wave == 1 ? 0 : NP - 1] = acc;
    if (wave == 1) sh.ne = ne;
  }
  __syncthreads();
}

// StrengthSolver::solve of plane q of the pass by the calling wave: b += mean / 8192 (kept: the reference does not undo it), the
// regularised copy of A, the elimination.  Out: sh.sb[q], sh.sx[q], sh.st_ok[q].
__device__ __forceinline__ void strength_solve(Shared &sh, const int q, const int lane) {
  const uint32_t ne = sh.ne;
  double r[kNumBins], x[kNumBins];
  const int i = lane < kNumBins ? lane : 0, nn = kNumBins;
  const double mean = sh.total[q] / (int)ne;
  const double sb = sh.bsum[q][i] + mean / 8192.;
  const double alpha = 2.0 * (double)(int)ne / nn;
  const int lo = max(0, i - 1), hi = min(nn - 1, i + 1);
#pragma unroll
  for (int j = 0; j < kNumBins; ++j) {
    double v = 0.0;
    if (i == j) v = sh.diag[i];
    else if (i == j + 1) v = sh.low[j];
    else if (j == i + 1) v = sh.low[i];
    r[j] = v;
    x[j] = 0.0;
  }
  // (the reference's three updates in its order; lo or hi may be i itself)
#pragma unroll
  for (int j = 0; j < kNumBins; ++j)
    if (j == lo) r[j] -= alpha;
#pragma unroll
  for (int j = 0; j < kNumBins; ++j)
    if (j == i) r[j] += 2 * alpha;
#pragma unroll
  for (int j = 0; j < kNumBins; ++j)
    if (j == hi) r[j] -= alpha;
#pragma unroll
  for (int j = 0; j < kNumBins; ++j)
    if (j == i) r[j] += 1.0 / 8192.;
  const bool ok = wave_gauss<kNumBins>(kNumBins, r, sb, x, lane);
  if (lane < kNumBins) {
    sh.sb[q][lane] = sb;
    double xv = 0.0;
#pragma unroll
    for (int j = 0; j < kNumBins; ++j)
      if (j == lane) xv = x[j];
    sh.sx[q][lane] = xv;
  }
  if (lane == 0) sh.st_ok[q] = ok ? 1 : 0;
}

__global__ __launch_bounds__(kT) void k4_latest(LatestJob job) {
  __shared__ Shared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frame = blockIdx.x;
  const uint8_t *rec = job.records + job.L.size * (size_t)frame;
  uint8_t *blob = job.blobs + job.blob_bytes * (size_t)frame;
  const int nb = (int)job.L.nblocks;
  const int n = job.n, ncm = n + 1;
  FrameCtx fc;
  fc.rec = rec;
  fc.mask = rec + job.L.off_mask;
  fc.luma_sum = reinterpret_cast<const uint32_t *>(rec + job.L.off_luma_sum);
  fc.nb = nb;
  fc.nbw = job.nbw;
  fc.W = job.W;
  fc.H = job.H;

  double *k4_dbg = reinterpret_cast<double *>(job.scratch);  // (a -DG1S_LATEST_TIMERS build: frame 0's phases, 100 MHz ticks = us x 100)
#ifdef G1S_LATEST_TIMERS
  unsigned long long k4_t = wall_clock64();
  if (tid < 16 && frame == 0) k4_dbg[tid] = 0.0;
  __syncthreads();
#endif
  // ---- the blob: zeros, the header, every plane "cleared" (ar_gain 1) ----
  for (size_t k = tid; k < job.blob_bytes / 8; k += kT) reinterpret_cast<unsigned long long *>(blob)[k] = 0ull;
  LatestHeader *hdr = reinterpret_cast<LatestHeader *>(blob);
  auto plane_head = [&](int c) { return reinterpret_cast<LatestPlaneHead *>(blob + sizeof(LatestHeader) + c * plane_blob_bytes(ncm)); };
  auto plane_doubles = [&](int c) { return reinterpret_cast<double *>(blob + sizeof(LatestHeader) + c * plane_blob_bytes(ncm) + sizeof(LatestPlaneHead)); };
  // flat blocks of the frame
  uint32_t nflat_t = 0;
  for (int b = tid; b < nb; b += kT) nflat_t += fc.mask[b] != 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) nflat_t += (uint32_t)__shfl_down((int)nflat_t, d, 64);
  if (lane == 0) sh.wsum[wave] = nflat_t;
  __syncthreads();  // (also: the zeros of the blob are behind every later store of this workgroup)
  const uint32_t num_flat = sh.wsum[0] + sh.wsum[1] + sh.wsum[2] + sh.wsum[3];
  if (tid == 0) {
    hdr->magic = kLatestMagic;
    hdr->lag = (uint32_t)job.lag;
    hdr->nplanes = (uint32_t)job.nplanes;
    hdr->status = G1S_OK;
    hdr->size_bytes = (uint32_t)job.blob_bytes;
    hdr->reserved = num_flat;  // (not part of the state: the caller's statistics)
    for (int c = 0; c < 3; ++c) plane_head(c)->ar_gain = 1.0;
  }
  if (num_flat <= 1) {
    if (tid == 0) {
      hdr->status = G1S_ERR_NOT_ENOUGH_FLAT;
      put_text(hdr->err, "Not enough flat blocks to update noise estimate");
    }
    return;
  }
  K4_TICK(0);

  // ---- the planes' AR systems: wave c solves plane c's (exact integer sums -> f64 normal equations, one rounding each) ----
  // den(i, j): 255^2, times ns for the row and for the column of the chroma regressor
  auto ar_entry = [&](int c, int i, int j) -> double {
    const bool is_chroma = c != 0;
    const int nc = n + (is_chroma ? 1 : 0);
    const int64_t *S = reinterpret_cast<const int64_t *>(rec + job.L.off_ar[c]);
    const double ns = is_chroma ? (double)((1 << job.xdec) * (1 << job.ydec)) : 1.0;
    double den = kNorm2D;
    if (is_chroma && i == nc - 1) den *= ns;
    if (is_chroma && j == nc - 1) den *= ns;
    const int64_t s = j == nc ? S[(size_t)nc * nc + i] : (i <= j ? S[i * nc + j] : S[j * nc + i]);  // (j == nc: the b side)
    return (double)s / den;
  };
  if (wave < job.nplanes) {
    const int c = wave;
    const bool is_chroma = c != 0;
    const int nc = n + (is_chroma ? 1 : 0);
    const int row = lane < nc ? lane : 0;
    double a[kMaxN], x[kMaxN];
#pragma unroll
    for (int j = 0; j < kMaxN; ++j) {
      a[j] = j < nc ? ar_entry(c, row, j) : 0.0;
      x[j] = 0.0;
    }
    const double b0 = ar_entry(c, row, nc);
    const double a_diag = ar_entry(c, row, row), a_last = ar_entry(c, row, nc - 1);
    const int64_t nobs = reinterpret_cast<const int64_t *>(rec + job.L.off_ar[c])[(size_t)nc * nc + nc];
    const bool ok = wave_gauss<kMaxN>(nc, a, b0, x, lane);
    // ar_solve's gain (fold.cpp), the sums in its order: lane i holds the i-th term
    double gain = 1.0;
    if (ok) {
      const int m = nc - (is_chroma ? 1 : 0);
      double x_mine = 0.0;
#pragma unroll
      for (int j = 0; j < kMaxN; ++j)
        if (j == lane) x_mine = x[j];
      const double tv = a_diag / nobs;
      double bi = b0, x_last = 0.0;
#pragma unroll
      for (int j = 0; j < kMaxN; ++j)
        if (j == nc - 1) x_last = x[j];
      if (is_chroma) bi -= a_last * x_last;
      const double tc = (bi * x_mine) / nobs;
      double var = 0, sum_covar = 0;
      for (int i = 0; i < m; ++i) var += readlane_f64(tv, i);
      var /= m;
      for (int i = 0; i < m; ++i) sum_covar += readlane_f64(tc, i);
      const double t = var - sum_covar;
      const double noise_var = t > 1e-6 ? t : 1e-6;
      const double q = var / noise_var;
      const double g = sqrt(q > 1e-6 ? q : 1e-6);
      gain = 1 > g ? 1 : g;
    } else if (is_chroma) {  // chroma_fallback: zero AR coefficients, keep only the luma correlation
      const int last = nc - 1;
      const double all = readlane_f64(a_diag, last), bl = readlane_f64(b0, last);
#pragma unroll
      for (int j = 0; j < kMaxN; ++j) x[j] = 0.0;
      if (fabs(all) > 1e-6) {
        const double xl = bl / all;
#pragma unroll
        for (int j = 0; j < kMaxN; ++j)
          if (j == last) x[j] = xl;
      }
    }
    if (lane < nc) {
      double xv = 0.0;
#pragma unroll
      for (int j = 0; j < kMaxN; ++j)
        if (j == lane) xv = x[j];
      sh.arx[c][lane] = xv;
    }
    if (lane == 0) {
      sh.gain[c] = gain;
      sh.nobs[c] = nobs;
      sh.ar_ok[c] = ok ? 1 : 0;
    }
  }
  __syncthreads();
  K4_TICK(1);
  // a plane's AR side into the blob (the host has loaded and solved a plane's system by the time it refuses the frame in that
  // plane's strength solve, and has not touched the planes behind it: each plane's is written when the host would have)
  auto write_ar = [&](int c) {
    const bool is_chroma = c != 0;
    const int nc = n + (is_chroma ? 1 : 0);
    double *pd = plane_doubles(c);
    for (int e = tid; e < nc * nc + nc; e += kT) {
      if (e < nc * nc) {
        const int i = e / nc, j = e - i * nc;
        pd[e] = ar_entry(c, i, j);
      } else {
        const int i = e - nc * nc;
        pd[ncm * ncm + i] = ar_entry(c, i, nc);
        pd[ncm * ncm + ncm + i] = sh.arx[c][i];
      }
    }
    if (tid == 0) {
      LatestPlaneHead *ph = plane_head(c);
      ph->num_observations = sh.nobs[c];
      ph->ar_gain = sh.gain[c];
    }
  };
  auto write_strength = [&](int c, int q) {
    double *qd = plane_doubles(c) + ncm * ncm + 2 * ncm;
    for (int e = tid; e < kNumBins * kNumBins; e += kT) {
      const int i = e / kNumBins, j = e - i * kNumBins;
      double v = 0.0;
      if (i == j) v = sh.diag[i];
      else if (i == j + 1) v = sh.low[j];
      else if (j == i + 1) v = sh.low[i];
      qd[e] = v;
    }
    if (tid < kNumBins) {
      qd[kNumBins * kNumBins + tid] = sh.sb[q][tid];
      qd[kNumBins * kNumBins + kNumBins + tid] = sh.sx[q][tid];
    }
    if (tid == 0) {
      LatestPlaneHead *ph = plane_head(c);
      ph->num_equations = (int32_t)sh.ne;
      ph->total = sh.total[q];
    }
  };
  write_ar(0);
  if (!sh.ar_ok[0]) {
    if (tid == 0) {
      hdr->status = G1S_ERR_SOLVE;
      put_text(hdr->err, "Solving latest noise equation system failed 0!");
    }
    return;
  }
  // ---- luma: noise strength vs. intensity ----
  strength_pass<1>(sh, job, fc, 0, tid, k4_dbg, frame);
  K4_TICK(2);
  if (wave == 0) strength_solve(sh, 0, lane);
  __syncthreads();
  write_strength(0, 0);
  if (tid < kNumBins) sh.luma_x[tid] = sh.sx[0][tid];
  if (!sh.st_ok[0]) {
    if (tid == 0) {
      hdr->status = G1S_ERR_SOLVE;
      put_text(hdr->err, "Solving latest noise strength failed!");
    }
    return;
  }
  __syncthreads();
  K4_TICK(3);
  if (job.nplanes < 3) return;
  // ---- both chroma planes: one pass (the same blocks, the same bins), two solves side by side ----
  strength_pass<2>(sh, job, fc, 1, tid, k4_dbg, frame);
  K4_TICK(4);
  if (wave < 2) strength_solve(sh, wave, lane);
  __syncthreads();
  write_ar(1);
  write_strength(1, 0);
  if (!sh.st_ok[0]) {
    if (tid == 0) {
      hdr->status = G1S_ERR_SOLVE;
      put_text(hdr->err, "Solving latest noise strength failed!");
    }
    return;
  }
  write_ar(2);
  write_strength(2, 1);
  if (!sh.st_ok[1]) {
    if (tid == 0) {
      hdr->status = G1S_ERR_SOLVE;
      put_text(hdr->err, "Solving latest noise strength failed!");
    }
  }
  K4_TICK(5);
}

}  // namespace

size_t latest_scratch_bytes(uint32_t) { return 256; }  // (the timers of a -DG1S_LATEST_TIMERS build; the kernel keeps everything else in LDS)
const char *latest_kernel_name() { return "k4_latest"; }

hipError_t launch_latest(const LatestJob &job, int frames, hipStream_t stream) {
  if (frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(k4_latest, dim3(frames), dim3(kT), 0, stream, job);
#ifdef G1S_LATEST_TIMERS
  {
    double t[16];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(t, job.scratch, sizeof(t), hipMemcpyDeviceToHost);
    fprintf(stderr, "k4_latest phases, us (frame 0): head %.1f  ar solves %.1f  luma pass %.1f  luma solve %.1f  chroma pass %.1f  chroma solves %.1f | inside both passes: "
                    "terms %.1f  ranks %.1f  offsets %.1f  scatter %.1f  chains %.1f\n",
            t[0] / 100, t[1] / 100, t[2] / 100, t[3] / 100, t[4] / 100, t[5] / 100, t[6] / 100, t[7] / 100, t[8] / 100, t[9] / 100, t[10] / 100);
  }
#endif
  return hipGetLastError();
}

}  // namespace g1s

// latest.hip -- k4_latest: the per-frame half of the fold (fold.cpp: compute_latest) on the device.
//
// What the reference does per frame after the pixel work (av1-grain diff/solver.rs NoiseModel::update, the part before the
// combined model is touched: the latest state's AR solve, the noise-strength measurements of the flat blocks, the 20-bin
// strength solve) is f64 arithmetic in a fixed order on a few hundred KB of exact integers.  On the host it costs 158 us of
// a core per 4K frame and needs the whole 285 KB record over PCIe; here one workgroup per frame does it next to the
// accumulation kernels of the following batch and the host receives the 27 KB latest-state blob of fold.h -- the same
// bytes compute_latest + latest_to_blob produce (tests/test_gpu_parity.py::test_device_latest_*):
//   * every f64 operation is the host's operation on the host's operands (-ffp-contract=off on both sides; IEEE divide and
//     square root; int64 -> f64 in one rounding);
//   * every f64 SUM runs in the host's order.  The elimination's row operations are independent per row (a thread a row),
//     pivot search and back substitution are serial (one thread); the strength system's entries are sums over the flat
//     blocks in raster order: entry (k, k), (k + 1, k) and b[k] only meet blocks of bins k - 1 and k, so the blocks are
//     partitioned by bin, in order (ballots), and lane k walks its own list; `total` is one sum over all blocks: a wave
//     loads 64 terms at a time and adds them lane by lane (v_readlane).
#include "latest_dev.h"

#include "fold.h"

#include <stdio.h>

namespace g1s {
namespace {

constexpr int kT = 256;
constexpr double kTinyD = 1.0e-16;         // TINY_NEAR_ZERO
constexpr double kNorm2D = 255.0 * 255.0;  // BLOCK_NORMALIZATION^2

struct Scratch {  // one frame's, in HBM (L2 resident while the frame is worked on)
  double *e_a, *e_std;  // the plane's measurements in raster order: interpolation weight a, noise std
  uint8_t *e_i0;        // ... their bin
  double2 *s_e;         // the same, partitioned: list k = the blocks of bins k - 1 and k, in raster order: (a, std), a's sign bit = bin k - 1
};
__host__ __device__ inline size_t scratch_bytes_for(uint32_t nblocks) {
  const size_t nb = (nblocks + 7) & ~size_t(7);
  return nb * (8 + 8 + 8 + 32) + 128;  // (+ the timers of a -DG1S_LATEST_TIMERS build)
}
__device__ inline Scratch carve(uint8_t *p, uint32_t nblocks) {
  const size_t nb = (nblocks + 7) & ~size_t(7);
  Scratch s;
  s.e_a = reinterpret_cast<double *>(p);
  s.e_std = s.e_a + nb;
  s.s_e = reinterpret_cast<double2 *>(s.e_std + nb);
  s.e_i0 = reinterpret_cast<uint8_t *>(s.s_e + 2 * nb);
  return s;
}

struct GaussShared {
  double colv[kMaxN];
  int R[kMaxN];
  int flag;
};
struct Shared {
  double A[kMaxN * kMaxN], At[kMaxN * kMaxN], b[kMaxN], bt[kMaxN], x[kMaxN];
  double SA[kNumBins * kNumBins], SAt[kNumBins * kNumBins], Sb[kNumBins], Sbt[kNumBins], Sx[kNumBins];
  double luma_x[kNumBins];
  double diag[kNumBins], low[kNumBins], bsum[kNumBins];
  double total, luma_gain, gain;
  GaussShared gs;
  uint32_t cnt[kNumBins], off2[kNumBins + 1];
  uint32_t wsum[4];
  uint32_t ne;
};

// gauss_solve of fold.cpp: the reference's elimination with its "bubble the larger magnitude up one row at a time" pivoting.
// The rows stay where they are in LDS (At, n x n; b by ROW in bt); what moves is the map position -> row (gs.R).  A step:
//   * the bubble pass over column k, from the bottom up, is `carried = |v[i - 1]| < |carried| ? carried : v[i - 1]` -- the
//     element carried past position i is the first maximum of positions i .. n - 1 (w(i)), and position i is left with the
//     loser of the comparison at step i: the thread of position i walks that chain itself (<= n comparisons on the column,
//     gathered by position) and names the row that lands there.  Same comparisons, same order, same NaN behaviour as the
//     serial pass;
//   * the thread of every position below k does `row[j] -= c * pivot[j]`, j > k, on its row.
// Back substitution: one thread, the reference's order.
// All threads call (barriers); At, bt are consumed; x must hold what a failed solve leaves behind (the caller's cleared x).
__device__ __forceinline__ bool dev_gauss(int n, double *__restrict__ At, double *__restrict__ bt, double *__restrict__ x, GaussShared &gs, int tid) {
  const bool mine = tid < n;
  if (mine) gs.R[tid] = tid;
  __syncthreads();
  for (int k = 0; k < n - 1; ++k) {
    if (mine && tid >= k) gs.colv[tid] = fabs(At[gs.R[tid] * n + k]);
    __syncthreads();
    int newrow = -1;
    if (mine && tid >= k) {  // this thread names the row of position i = tid after the pass
      const int i = tid;
      int w = n - 1;  // w(n - 1)
      double wv = gs.colv[n - 1];
      int w_i = w;  // w(i), on the way to w(i - 1)
      for (int q = n - 1; q >= (i == k ? k + 1 : i); --q) {  // after this iteration: w = w(q - 1)
        if (q == i) w_i = w;
        const double up = gs.colv[q - 1];
        if (!(up < wv)) {
          w = q - 1;
          wv = up;
        }
      }
      // i == k: w = w(k).  i > k: w = w(i - 1), w_i = w(i): the loser of the comparison at step i stays at position i
      if (i == k) newrow = gs.R[w];
      else newrow = (w == w_i) ? gs.R[i - 1] : gs.R[w_i];
    }
    __syncthreads();
    if (newrow >= 0) gs.R[tid] = newrow;
    __syncthreads();
    const int pr = gs.R[k];
    const double *pivot = At + pr * n;
    const double pk = pivot[k];
    if (fabs(pk) < kTinyD) return false;  // (every thread reads the same value)
    if (mine && tid > k) {
      const int r = gs.R[tid];
      double *__restrict__ row = At + r * n;
      const double *__restrict__ prow = pivot;  // (another row: r != pr)
      const double c = row[k] / pk;
#pragma unroll 4
      for (int j = k + 1; j < n; ++j) row[j] -= c * prow[j];
      bt[r] -= c * bt[pr];
    }
    __syncthreads();
  }
  if (tid == 0) {
    gs.flag = 1;
    for (int i = n - 1; i >= 0; --i) {
      const int r = gs.R[i];
      const double *row = At + r * n;
      if (fabs(row[i]) < kTinyD) {
        gs.flag = 0;
        break;
      }
      double c = 0;
      for (int j = i + 1; j <= n - 1; ++j) c += row[j] * x[j];
      x[i] = (bt[r] - c) / row[i];
    }
  }
  __syncthreads();
  return gs.flag != 0;
}

__device__ inline double dev_clamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ inline double dev_bin_index(double value) { return (kNumBins - 1) * dev_clamp(value, 0.0, 255.0) / 255.0; }
__device__ inline double dev_value_at(const double *sx, double x) {
  const double bin = dev_bin_index(x);
  const int i0 = (int)floor(bin);
  const int i1 = min(kNumBins - 1, i0 + 1);
  const double a = bin - i0;
  return (1.0 - a) * sx[i0] + a * sx[i1];
}

__device__ inline void put_text(char *dst, const char *msg) {
  int i = 0;
  for (; msg[i] && i < 103; ++i) dst[i] = msg[i];
  dst[i] = 0;
}

__device__ inline double readlane_f64(double v, int j) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), j);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

#ifdef G1S_LATEST_TIMERS
#define K4_TICK(slot)                                                       \
  do {                                                                      \
    if (tid == 0 && frame == 0) {                                           \
      const unsigned long long now_ = wall_clock64();                       \
      k4_dbg[slot] += (double)(now_ - k4_t);                                \
      k4_t = now_;                                                          \
    }                                                                       \
  } while (0)
#else
#define K4_TICK(slot) \
  do {                \
  } while (0)
#endif

__global__ __launch_bounds__(kT, 4) void k4_latest(LatestJob job) {
  __shared__ Shared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frame = blockIdx.x;
  const uint8_t *rec = job.records + job.L.size * (size_t)frame;
  uint8_t *blob = job.blobs + job.blob_bytes * (size_t)frame;
  const Scratch sc = carve(job.scratch + job.scratch_bytes * (size_t)frame, job.L.nblocks);
  const int nb = (int)job.L.nblocks, nbw = job.nbw;
  const int n = job.n, ncm = n + 1;
  const uint8_t *mask = rec + job.L.off_mask;
  const uint32_t *luma_sum = reinterpret_cast<const uint32_t *>(rec + job.L.off_luma_sum);

#ifdef G1S_LATEST_TIMERS
  double *k4_dbg = reinterpret_cast<double *>(job.scratch + job.scratch_bytes - 128);  // (frame 0's, 100 MHz ticks: us x 100)
  unsigned long long k4_t = wall_clock64();
  if (tid < 16 && frame == 0) k4_dbg[tid] = 0.0;
#endif
  // ---- the blob: zeros, the header, every plane "cleared" (ar_gain 1) ----
  for (size_t k = tid; k < job.blob_bytes / 8; k += kT) reinterpret_cast<unsigned long long *>(blob)[k] = 0ull;
  __syncthreads();
  LatestHeader *hdr = reinterpret_cast<LatestHeader *>(blob);
  auto plane_head = [&](int c) { return reinterpret_cast<LatestPlaneHead *>(blob + sizeof(LatestHeader) + c * plane_blob_bytes(ncm)); };
  auto plane_doubles = [&](int c) { return reinterpret_cast<double *>(blob + sizeof(LatestHeader) + c * plane_blob_bytes(ncm) + sizeof(LatestPlaneHead)); };
  // flat blocks of the frame
  const int E = (nb + kT - 1) / kT;
  const int b_lo = min(tid * E, nb), b_hi = min(b_lo + E, nb);
  auto block_sum = [&](uint32_t v, uint32_t &excl) {  // exclusive prefix over the threads + the total; two barriers
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
      if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) sh.wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) before += sh.wsum[w];
      total += sh.wsum[w];
    }
    excl = before + inc - v;
    return total;
  };
  uint32_t dummy;
  uint32_t nflat_t = 0;
  for (int b = b_lo; b < b_hi; ++b) nflat_t += mask[b] != 0;
  const uint32_t num_flat = block_sum(nflat_t, dummy);
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

  for (int c = 0; c < job.nplanes; ++c) {
    const bool is_chroma = c != 0;
    const int sx = is_chroma ? job.xdec : 0, sy = is_chroma ? job.ydec : 0;
    const int nc = n + (is_chroma ? 1 : 0);
    LatestPlaneHead *ph = plane_head(c);
    double *pd = plane_doubles(c);
    // ---- exact integer sums -> f64 normal equations (one rounding each) ----
    const int64_t *S = reinterpret_cast<const int64_t *>(rec + job.L.off_ar[c]);
    const int64_t *Sb = S + (size_t)nc * nc;
    const double ns = (double)((1 << sx) * (1 << sy));
    for (int e = tid; e < nc * nc + nc; e += kT) {
      if (e < nc * nc) {
        const int i = e / nc, j = e - i * nc;
        double den = kNorm2D;
        if (is_chroma && i == nc - 1) den *= ns;
        if (is_chroma && j == nc - 1) den *= ns;
        const int64_t s = i <= j ? S[i * nc + j] : S[j * nc + i];
        const double v = (double)s / den;
        sh.A[e] = v;
        sh.At[e] = v;
        pd[e] = v;
      } else {
        const int i = e - nc * nc;
        double den = kNorm2D;
        if (is_chroma && i == nc - 1) den *= ns;
        const double v = (double)Sb[i] / den;
        sh.b[i] = v;
        sh.bt[i] = v;
        sh.x[i] = 0.0;
        pd[ncm * ncm + i] = v;
      }
    }
    const int64_t nobs = Sb[nc];
    __syncthreads();
    K4_TICK(0);
    // ---- ar_solve ----
    const bool ar_ok = dev_gauss(nc, sh.At, sh.bt, sh.x, sh.gs, tid);
    if (tid == 0) {
      double gain = 1.0;
      if (ar_ok) {
        const int m = nc - (is_chroma ? 1 : 0);
        double var = 0;
        for (int i = 0; i < m; ++i) var += sh.A[i * nc + i] / nobs;
        var /= m;
        double sum_covar = 0;
        for (int i = 0; i < m; ++i) {
          double bi = sh.b[i];
          if (is_chroma) bi -= sh.A[i * nc + (nc - 1)] * sh.x[nc - 1];
          sum_covar += (bi * sh.x[i]) / nobs;
        }
        const double t = var - sum_covar;
        const double noise_var = t > 1e-6 ? t : 1e-6;
        const double q = var / noise_var;
        const double g = sqrt(q > 1e-6 ? q : 1e-6);
        gain = 1 > g ? 1 : g;
      } else if (is_chroma) {  // chroma_fallback: zero AR coefficients, keep only the luma correlation
        const int last = nc - 1;
        for (int i = 0; i < nc; ++i) sh.x[i] = 0.0;
        if (fabs(sh.A[last * nc + last]) > 1e-6) sh.x[last] = sh.b[last] / sh.A[last * nc + last];
      }
      sh.gain = gain;
      if (!is_chroma) sh.luma_gain = gain;
      ph->num_observations = nobs;
      ph->ar_gain = gain;
      for (int i = 0; i < nc; ++i) pd[ncm * ncm + ncm + i] = sh.x[i];
      if (!ar_ok && !is_chroma) {
        hdr->status = G1S_ERR_SOLVE;
        put_text(hdr->err, "Solving latest noise equation system failed 0!");
      }
    }
    __syncthreads();
    if (!ar_ok && !is_chroma) return;

    K4_TICK(1);
    // ---- noise strength vs. intensity: the measurements of the flat blocks, raster order ----
    const int bw = kBlock >> sx, bh = kBlock >> sy;
    const int32_t *sum_d = reinterpret_cast<const int32_t *>(rec + job.L.off_sum_d[c]);
    const uint32_t *sum_d2 = reinterpret_cast<const uint32_t *>(rec + job.L.off_sum_d2[c]);
    auto takes_part = [&](int bi, int &sw, int &shh) {
      if (!mask[bi]) return false;
      const int by = bi / nbw, bx = bi - by * nbw;
      shh = min((job.H >> sy) - by * bh, bh);
      sw = min((job.W >> sx) - bx * bw, bw);
      return sw * shh > kBlock;
    };
    uint32_t mine = 0;
    for (int b = b_lo; b < b_hi; ++b) {
      int sw, shh;
      mine += takes_part(b, sw, shh) ? 1u : 0u;
    }
    if (tid < kNumBins) sh.cnt[tid] = 0;  // (block_sum's barriers stand between this and the first count)
    uint32_t at;
    const uint32_t ne = block_sum(mine, at);
    {
      const double luma_gain = sh.luma_gain, noise_gain = sh.gain;
      const double corr = is_chroma ? sh.x[n] : 0;
      for (int b = b_lo; b < b_hi; ++b) {
        int sw, shh;
        if (!takes_part(b, sw, shh)) continue;
        const int by = b / nbw, bx = b - by * nbw;
        const int lw = min(job.W - bx * kBlock, kBlock), lh = min(job.H - by * kBlock, kBlock);
        const double block_mean = (double)luma_sum[b] / (lw * lh);
        double noise_mean = (double)sum_d[b];
        const double noise_sq = (double)sum_d2[b];
        noise_mean /= (sw * shh);
        const double noise_var = noise_sq / (sw * shh) - noise_mean * noise_mean;
        const double luma_strength = is_chroma ? luma_gain * dev_value_at(sh.luma_x, block_mean) : 0;
        const double cl = corr * luma_strength;
        const double t0 = noise_var / 16, t1 = noise_var - cl * cl;
        const double uncorr_std = sqrt(t0 > t1 ? t0 : t1);
        const double noise_std = uncorr_std / noise_gain;
        const double bin = dev_bin_index(block_mean);
        const int i0 = (int)floor(bin);
        sc.e_a[at] = bin - i0;
        sc.e_std[at] = noise_std;
        sc.e_i0[at] = (uint8_t)i0;
        ++at;
        atomicAdd(&sh.cnt[i0], 1u);  // list i0 and list i0 + 1 take the measurement
        if (i0 + 1 < kNumBins) atomicAdd(&sh.cnt[i0 + 1], 1u);
      }
    }
    __threadfence_block();
    __syncthreads();
    K4_TICK(2);
    // ---- the partition: list k = the measurements of bins k - 1 and k, in order (wave w: lists 5 w .. 5 w + 4) ----
    if (tid == 0) {
      uint32_t o = 0;
      for (int k = 0; k < kNumBins; ++k) {
        sh.off2[k] = o;
        o += sh.cnt[k];
      }
      sh.off2[kNumBins] = o;
    }
    __syncthreads();
    {
      uint32_t run[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) run[q] = sh.off2[5 * wave + q];
      const unsigned long long lt = (1ull << lane) - 1ull;
      constexpr int U = 4;  // chunks of 64 measurements whose loads are in flight together
      for (uint32_t base = 0; base < ne; base += 64 * U) {
        int i0v[U];
        double av[U], sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t e = base + 64 * u + lane;
          const bool on = e < ne;
          i0v[u] = on ? (int)sc.e_i0[e] : 255;
          av[u] = on ? sc.e_a[e] : 0.0;
          sv[u] = on ? sc.e_std[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const int k = 5 * wave + q;
            const bool in = i0v[u] == k || i0v[u] == k - 1;
            const unsigned long long m = __ballot(in);
            if (in) {
              const uint32_t pos = run[q] + (uint32_t)__popcll(m & lt);
              // (bin k - 1's measurements carry the sign bit: a >= 0, and -0.0 is told from 0.0 by its bits)
              const double ae = i0v[u] == k ? av[u] : __longlong_as_double(__double_as_longlong(av[u]) | (long long)0x8000000000000000ull);
              sc.s_e[pos] = make_double2(ae, sv[u]);
            }
            run[q] += (uint32_t)__popcll(m);
          }
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    K4_TICK(3);
    // ---- the sums, in raster order: wave 0 lane k the entries (k, k), (k + 1, k) = (k, k + 1) and b[k]; wave 1 the total ----
    if (wave == 0 && lane < kNumBins) {
      const int k = lane;
      double dg = 0, lw_ = 0, bs = 0;
      const bool top = k == kNumBins - 1;  // i1 = i0 = 19: all four entries of the block are (19, 19), both b terms b[19]
      auto step = [&](double ae, double sd) {
        const bool prev = __double_as_longlong(ae) < 0;
        const double a = fabs(ae);
        if (!prev) {  // i0 = k
          const double t10 = a * (1.0 - a);
          dg += (1.0 - a) * (1.0 - a);
          if (top) {
            dg += t10;
            dg += a * a;
            dg += t10;
          } else {
            lw_ += t10;
          }
          bs += (1.0 - a) * sd;
          if (top) bs += a * sd;
        } else {  // i0 = k - 1, i1 = k
          dg += a * a;
          bs += a * sd;
        }
      };
      uint32_t p = sh.off2[k];
      const uint32_t end = sh.off2[k + 1];
      for (; p + 16 <= end; p += 16) {  // (sixteen measurements' loads in flight: the walk is latency bound)
        double2 ev[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) ev[u] = sc.s_e[p + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) step(ev[u].x, ev[u].y);
      }
      for (; p < end; ++p) {
        const double2 e1 = sc.s_e[p];
        step(e1.x, e1.y);
      }
      sh.diag[k] = dg;
      sh.low[k] = lw_;
      sh.bsum[k] = bs;
    } else if (wave == 1) {
      double tot = 0;
      for (uint32_t base = 0; base < ne; base += 64) {
        const double v = base + lane < ne ? sc.e_std[base + lane] : 0.0;
        const int cntc = (int)min(64u, ne - base);
        if (cntc == 64) {
#pragma unroll
          for (int j = 0; j < 64; ++j) tot += readlane_f64(v, j);
        } else {
          for (int j = 0; j < cntc; ++j) tot += __shfl(v, j);
        }
      }
      if (lane == 0) sh.total = tot;
    }
    __syncthreads();
    K4_TICK(4);
    // ---- the strength system, StrengthSolver::solve ----
    for (int e = tid; e < kNumBins * kNumBins; e += kT) {
      const int i = e / kNumBins, j = e - i * kNumBins;
      double v = 0.0;
      if (i == j) v = sh.diag[i];
      else if (i == j + 1) v = sh.low[j];
      else if (j == i + 1) v = sh.low[i];
      sh.SA[e] = v;
    }
    if (tid < kNumBins) {
      const double mean = sh.total / (int)ne;  // apply_regularisation_to_b (kept in b: the reference does not undo it)
      sh.Sb[tid] = sh.bsum[tid] + mean / 8192.;
      sh.Sx[tid] = 0.0;
    }
    __syncthreads();
    if (tid < kNumBins) {
      const int i = tid, nn = kNumBins;
      const double alpha = 2.0 * (double)(int)ne / nn;
      const int lo = max(0, i - 1), hi = min(nn - 1, i + 1);
      double r[kNumBins];
#pragma unroll
      for (int j = 0; j < kNumBins; ++j) r[j] = sh.SA[i * nn + j];
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
#pragma unroll
      for (int j = 0; j < kNumBins; ++j) sh.SAt[i * nn + j] = r[j];
      sh.Sbt[i] = sh.Sb[i];
    }
    __syncthreads();
    const bool st_ok = dev_gauss(kNumBins, sh.SAt, sh.Sbt, sh.Sx, sh.gs, tid);
    double *q = pd + ncm * ncm + 2 * ncm;
    for (int e = tid; e < kNumBins * kNumBins; e += kT) q[e] = sh.SA[e];
    if (tid < kNumBins) {
      q[kNumBins * kNumBins + tid] = sh.Sb[tid];
      q[kNumBins * kNumBins + kNumBins + tid] = sh.Sx[tid];
      if (!is_chroma) sh.luma_x[tid] = sh.Sx[tid];
    }
    if (tid == 0) {
      ph->num_equations = (int32_t)ne;
      ph->total = sh.total;
      if (!st_ok) {
        hdr->status = G1S_ERR_SOLVE;
        put_text(hdr->err, "Solving latest noise strength failed!");
      }
    }
    __syncthreads();
    K4_TICK(5);
    if (!st_ok) return;
  }
}

}  // namespace

size_t latest_scratch_bytes(uint32_t nblocks) { return scratch_bytes_for(nblocks); }
const char *latest_kernel_name() { return "k4_latest"; }

hipError_t launch_latest(const LatestJob &job, int frames, hipStream_t stream) {
  if (frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(k4_latest, dim3(frames), dim3(kT), 0, stream, job);
#ifdef G1S_LATEST_TIMERS
  {
    double t[16];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(t, job.scratch + job.scratch_bytes - 128, sizeof(t), hipMemcpyDeviceToHost);
    fprintf(stderr, "k4_latest phases, us (frame 0, all planes): load %.1f  ar solve %.1f  measure %.1f  partition %.1f  sums %.1f  strength %.1f\n",
            t[0] / 100, t[1] / 100, t[2] / 100, t[3] / 100, t[4] / 100, t[5] / 100);
  }
#endif
  return hipGetLastError();
}

}  // namespace g1s

// fold.cpp -- ordered host fold of per-frame integer records (see fold.h).
//
// Follows av1_grain::DiffGenerator / NoiseModel (crate av1-grain 0.4.2, a port
// of libaom aom_dsp/noise_model.c) as it is driven from the reference at
// src/main.rs:420-427, :442, :524.  Must be compiled with -ffp-contract=off:
// the f64 operation order below is part of the contract.
#include "fold.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace g1s {

namespace {
constexpr double kTiny = 1.0e-16;        // TINY_NEAR_ZERO
constexpr double kNorm2 = 255.0 * 255.0;  // BLOCK_NORMALIZATION^2
constexpr uint16_t kDefaultGrainSeed = 10956;  // av1_grain::DEFAULT_GRAIN_SEED (src/parser/frame.rs:3)
inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

// (this file also passes through the device compiler, which knows no x86 function multiversioning)
#if defined(__HIP_DEVICE_COMPILE__)
#define G1S_HOST_CLONES
#else
#define G1S_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#endif

// ---------------------------------------------------------------- solver ---
// (clones: the row operations are elementwise, AVX2 does four at a time with the same roundings)
__attribute__((target_clones("avx2", "default"))) bool gauss_solve(int n, double *A, double *b, double *x) {
  // forward elimination with the reference's "bubble the larger magnitude up
  // one row at a time" pivoting.  Columns left of the pivot column are never read again
  // (pivoting looks at column k, back substitution at columns >= i), so the row operations
  // run over columns >= k only: every value that is used is bit for bit the reference's.
  // Rows are swapped by pointer: the same values meet in the same operations, nothing is moved.
  double *R[kMaxN];
  for (int i = 0; i < n; ++i) R[i] = A + i * n;
  for (int k = 0; k < n - 1; ++k) {
    for (int i = n - 1; i > k; --i) {
      if (std::fabs(R[i - 1][k]) < std::fabs(R[i][k])) {
        std::swap(R[i - 1], R[i]);
        std::swap(b[i], b[i - 1]);
      }
    }
    const double *__restrict pivot = R[k];
    if (std::fabs(pivot[k]) < kTiny) return false;
    const double pk = pivot[k], bk = b[k];
    for (int i = k; i < n - 1; ++i) {
      double *__restrict row = R[i + 1];
      const double c = row[k] / pk;
      for (int j = k + 1; j < n; ++j) row[j] -= c * pivot[j];
      b[i + 1] -= c * bk;
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    const double *row = R[i];
    if (std::fabs(row[i]) < kTiny) return false;
    double c = 0;
    for (int j = i + 1; j <= n - 1; ++j) c += row[j] * x[j];
    x[i] = (b[i] - c) / row[i];
  }
  return true;
}

void LinearSystem::resize(int n_) {
  n = n_;
  A.assign(size_t(n) * n, 0.0);
  b.assign(n, 0.0);
  x.assign(n, 0.0);
}
void LinearSystem::clear() {
  std::fill(A.begin(), A.end(), 0.0);
  std::fill(b.begin(), b.end(), 0.0);
  std::fill(x.begin(), x.end(), 0.0);
}
// (clones: elementwise, AVX2 does four at a time with the same roundings)
G1S_HOST_CLONES void add_into(double *__restrict dst, const double *__restrict src, int n) {
  for (int i = 0; i < n; ++i) dst[i] += src[i];
}
G1S_HOST_CLONES void sum_into(double *__restrict dst, const double *__restrict a,
                                                                        const double *__restrict b, int n) {
  for (int i = 0; i < n; ++i) dst[i] = a[i] + b[i];
}
void LinearSystem::add(const LinearSystem &o) { add(o.A.data(), o.b.data()); }
void LinearSystem::add(const double *oA, const double *ob) {
  add_into(A.data(), oA, n * n);
  add_into(b.data(), ob, n);
}
// this = a + b (elementwise on A and b; x is left alone): one pass instead of assign + add
void LinearSystem::set_sum(const LinearSystem &a, const LinearSystem &bb) { set_sum(a, bb.A.data(), bb.b.data()); }
void LinearSystem::set_sum(const LinearSystem &a, const double *A2, const double *b2) {
  if (n != a.n) resize(a.n);
  sum_into(A.data(), a.A.data(), A2, n * n);
  sum_into(b.data(), a.b.data(), b2, n);
}
void LinearSystem::assign(const LinearSystem &o) {
  n = o.n;
  A = o.A;
  b = o.b;
  x = o.x;
}
bool LinearSystem::solve() {
  double At[kMaxN * kMaxN], bt[kMaxN];
  std::memcpy(At, A.data(), sizeof(double) * n * n);
  std::memcpy(bt, b.data(), sizeof(double) * n);
  return gauss_solve(n, At, bt, x.data());
}

// ------------------------------------------------------- strength solver ---
StrengthSolver::StrengthSolver() { eq.resize(kNumBins); }
void StrengthSolver::clear() {
  eq.clear();
  num_equations = 0;
  total = 0.0;
}
void StrengthSolver::add(const StrengthSolver &o) {
  eq.add(o.eq);
  num_equations += o.num_equations;
  total += o.total;
}
void StrengthSolver::add(const double *oA, const double *ob, int64_t o_num_equations, double o_total) {
  eq.add(oA, ob);
  num_equations += o_num_equations;
  total += o_total;
}
double StrengthSolver::bin_index(double value) {
  const double val = clampd(value, 0.0, 255.0);
  return (kNumBins - 1) * val / 255.0;
}
double StrengthSolver::value_at(double x) const {
  const double bin = bin_index(x);
  const int i0 = (int)std::floor(bin);
  const int i1 = std::min(kNumBins - 1, i0 + 1);
  const double a = bin - i0;
  return (1.0 - a) * eq.x[i0] + a * eq.x[i1];
}
void StrengthSolver::add_measurement(double block_mean, double noise_std) {
  const double bin = bin_index(block_mean);
  const int i0 = (int)std::floor(bin);
  const int i1 = std::min(kNumBins - 1, i0 + 1);
  const double a = bin - i0;
  const int n = kNumBins;
  eq.A[i0 * n + i0] += (1.0 - a) * (1.0 - a);
  eq.A[i1 * n + i0] += a * (1.0 - a);
  eq.A[i1 * n + i1] += a * a;
  eq.A[i0 * n + i1] += a * (1.0 - a);
  eq.b[i0] += (1.0 - a) * noise_std;
  eq.b[i1] += a * noise_std;
  total += noise_std;
  num_equations++;
}
// m measurements in order: bin[sel[k]] = bin_index(block mean) (in [0, kNumBins - 1]: truncation is the floor), noise_std[k].
// Every element of the system receives the additions add_measurement would make, in the same order; the six elements of
// the bin pair in hand stay in registers while successive blocks fall into the same pair (flat blocks next to each other
// mostly do: through memory every such addition waits for the store before it, ~10 cycles a block instead of an add's 4).
void StrengthSolver::add_measurements(const double *bin, const uint32_t *sel, const double *noise_std, size_t m) {
  const int n = kNumBins;
  double *A = eq.A.data(), *b = eq.b.data();
  int cur = -1;  // the pair (cur, cur + 1) is in registers
  double a00 = 0, a10 = 0, a01 = 0, a11 = 0, b0 = 0, b1 = 0, tot = total;
  auto put = [&] {
    if (cur < 0) return;
    A[cur * n + cur] = a00;
    A[(cur + 1) * n + cur] = a10;
    A[cur * n + cur + 1] = a01;
    A[(cur + 1) * n + cur + 1] = a11;
    b[cur] = b0;
    b[cur + 1] = b1;
    cur = -1;
  };
  for (size_t k = 0; k < m; ++k) {
    const double bn = bin[sel[k]], sd = noise_std[k];
    const int i0 = (int)bn;
    const double a = bn - i0;
    if (i0 >= n - 1) {  // the last bin is its own neighbour: all six additions as the reference makes them
      put();
      const int i1 = n - 1;
      A[i0 * n + i0] += (1.0 - a) * (1.0 - a);
      A[i1 * n + i0] += a * (1.0 - a);
      A[i1 * n + i1] += a * a;
      A[i0 * n + i1] += a * (1.0 - a);
      b[i0] += (1.0 - a) * sd;
      b[i1] += a * sd;
    } else {
      if (i0 != cur) {
        put();
        cur = i0;
        a00 = A[cur * n + cur];
        a10 = A[(cur + 1) * n + cur];
        a01 = A[cur * n + cur + 1];
        a11 = A[(cur + 1) * n + cur + 1];
        b0 = b[cur];
        b1 = b[cur + 1];
      }
      a00 += (1.0 - a) * (1.0 - a);
      a10 += a * (1.0 - a);
      a11 += a * a;
      a01 += a * (1.0 - a);
      b0 += (1.0 - a) * sd;
      b1 += a * sd;
    }
    tot += sd;
  }
  put();
  total = tot;
  num_equations += (int64_t)m;
}
// The same measurements' b side only, with the matrix copied from a system that has accumulated EXACTLY this sequence of bin
// positions (another plane of the frame: the entries of A depend on the bin positions alone, so they are the same sums of the
// same terms in the same order); b and total get this plane's noise_std as in add_measurements.
void StrengthSolver::add_measurements_like(const StrengthSolver &same_bins, const double *bin, const uint32_t *sel, const double *noise_std, size_t m) {
  const int n = kNumBins;
  eq.A = same_bins.eq.A;
  double *b = eq.b.data();
  int cur = -1;
  double b0 = 0, b1 = 0, tot = total;
  for (size_t k = 0; k < m; ++k) {
    const double bn = bin[sel[k]], sd = noise_std[k];
    const int i0 = (int)bn;
    const double a = bn - i0;
    if (i0 >= n - 1) {
      if (cur >= 0) b[cur] = b0, b[cur + 1] = b1, cur = -1;
      b[i0] += (1.0 - a) * sd;
      b[n - 1] += a * sd;
    } else {
      if (i0 != cur) {
        if (cur >= 0) b[cur] = b0, b[cur + 1] = b1;
        cur = i0;
        b0 = b[cur];
        b1 = b[cur + 1];
      }
      b0 += (1.0 - a) * sd;
      b1 += a * sd;
    }
    tot += sd;
  }
  if (cur >= 0) b[cur] = b0, b[cur + 1] = b1;
  total = tot;
  num_equations += (int64_t)m;
}
void StrengthSolver::apply_regularisation_to_b() {
  const double mean = total / num_equations;
  for (int i = 0; i < kNumBins; ++i) eq.b[i] += mean / 8192.;
}
bool StrengthSolver::solve_x_only() {
  // Regularised system on a copy of A (A itself is restored by the reference too)
  const int n = kNumBins;
  const double alpha = 2.0 * (double)num_equations / n;
  double At[kNumBins * kNumBins], bt[kNumBins];
  std::memcpy(At, eq.A.data(), sizeof(At));
  for (int i = 0; i < n; ++i) {
    const int lo = std::max(0, i - 1), hi = std::min(n - 1, i + 1);
    At[i * n + lo] -= alpha;
    At[i * n + i] += 2 * alpha;
    At[i * n + hi] -= alpha;
  }
  for (int i = 0; i < n; ++i) At[i * n + i] += 1.0 / 8192.;
  std::memcpy(bt, eq.b.data(), sizeof(bt));
  return gauss_solve(n, At, bt, eq.x.data());
}
bool StrengthSolver::solve() {
  // b keeps the mean/8192 term (the reference does not restore it either)
  apply_regularisation_to_b();
  return solve_x_only();
}
double StrengthSolver::center(int i) { return ((double)i) / (kNumBins - 1) * 255.0; }

void StrengthSolver::fit_piecewise(int max_points, std::vector<double> &px,
                                   std::vector<double> &py) const {
  const double tolerance = 255.0 * 0.00625 / 255.0;
  const double dxbin = 255. / kNumBins;
  px.resize(kNumBins);
  py.resize(kNumBins);
  for (int i = 0; i < kNumBins; ++i) {
    px[i] = center(i);
    py[i] = eq.x[i];
  }
  std::vector<double> residual(kNumBins, 0.0);
  auto update = [&](int start, int end) {
    const int np = (int)px.size();
    for (int i = std::max(start, 1); i < std::min(end, np - 1); ++i) {
      const int lower = std::max(0, (int)std::floor(bin_index(px[i - 1])));
      const int upper = std::min(kNumBins - 1, (int)std::ceil(bin_index(px[i + 1])));
      double r = 0;
      for (int j = lower; j <= upper; ++j) {
        const double x = center(j);
        if (x < px[i - 1]) continue;
        if (x >= px[i + 1]) continue;
        const double y = eq.x[j];
        const double a = (x - px[i - 1]) / (px[i + 1] - px[i - 1]);
        const double estimate_y = py[i - 1] * (1.0 - a) + py[i + 1] * a;
        r += std::fabs(y - estimate_y);
      }
      residual[i] = r * dxbin;
    }
  };
  update(0, kNumBins);
  while (px.size() > 2) {
    int min_index = 1;
    for (int j = 1; j < (int)px.size() - 1; ++j)
      if (residual[j] < residual[min_index]) min_index = j;
    const double dx = px[min_index + 1] - px[min_index - 1];
    const double avg_residual = residual[min_index] / dx;
    if ((int)px.size() <= max_points && avg_residual > tolerance) break;
    px.erase(px.begin() + min_index);
    py.erase(py.begin() + min_index);
    residual.erase(residual.begin() + min_index);
    update(min_index - 1, min_index + 1);
  }
}

// ------------------------------------------------------------ noise fold ---
NoiseFold::NoiseFold(int64_t fps_num, int64_t fps_den, uint32_t lag)
    : fps_num_(fps_num), fps_den_(fps_den), lag_(lag), n_((int)num_coeffs(lag)) {
  for (int c = 0; c < 3; ++c) {
    latest_[c].ar.resize(n_ + (c > 0));
    combined_[c].ar.resize(n_ + (c > 0));
  }
}

bool ar_solve(PlaneState &s, bool is_chroma) {
  const bool ok = s.ar.solve();
  s.ar_gain = 1.0;
  if (!ok) return false;
  // mean of the diagonal = variance of the correlated noise
  const int n = s.ar.n;
  const int m = n - (is_chroma ? 1 : 0);
  double var = 0;
  for (int i = 0; i < m; ++i) var += s.ar.A[i * n + i] / s.num_observations;
  var /= m;
  // E(y^2) = <b - A(:,end) x(end), x>
  double sum_covar = 0;
  for (int i = 0; i < m; ++i) {
    double bi = s.ar.b[i];
    if (is_chroma) bi -= s.ar.A[i * n + (n - 1)] * s.ar.x[n - 1];
    sum_covar += (bi * s.ar.x[i]) / s.num_observations;
  }
  const double t = var - sum_covar;
  const double noise_var = t > 1e-6 ? t : 1e-6;
  const double q = var / noise_var;
  const double g = std::sqrt(q > 1e-6 ? q : 1e-6);
  s.ar_gain = 1 > g ? 1 : g;
  return true;
}

void chroma_fallback(PlaneState &s) {
  // zero AR coefficients, keep only the luma correlation
  const int nc = s.ar.n, last = nc - 1;
  std::fill(s.ar.x.begin(), s.ar.x.end(), 0.0);
  if (std::fabs(s.ar.A[last * nc + last]) > 1e-6) s.ar.x[last] = s.ar.b[last] / s.ar.A[last * nc + last];
}

static PlaneView view_of_plane(const PlaneState &s) {
  PlaneView v;
  v.A = s.ar.A.data();
  v.b = s.ar.b.data();
  v.x = s.ar.x.data();
  v.sA = s.strength.eq.A.data();
  v.sb = s.strength.eq.b.data();
  v.sx = s.strength.eq.x.data();
  v.n = s.ar.n;
  v.num_observations = s.num_observations;
  v.ar_gain = s.ar_gain;
  v.num_equations = (int)s.strength.num_equations;  // (a view is of one frame)
  v.total = s.strength.total;
  return v;
}
void view_of(const FrameLatest &fl, FrameView &out) {
  for (int c = 0; c < 3; ++c) out.st[c] = view_of_plane(fl.st[c]);
  out.nplanes = fl.nplanes;
  out.status = fl.status;
  out.err = fl.err.c_str();
}
static void load_plane(PlaneState &s, const PlaneView &v) {
  if (s.ar.n != v.n) s.ar.resize(v.n);
  std::memcpy(s.ar.A.data(), v.A, sizeof(double) * v.n * v.n);
  std::memcpy(s.ar.b.data(), v.b, sizeof(double) * v.n);
  std::memcpy(s.ar.x.data(), v.x, sizeof(double) * v.n);
  std::memcpy(s.strength.eq.A.data(), v.sA, sizeof(double) * kNumBins * kNumBins);
  std::memcpy(s.strength.eq.b.data(), v.sb, sizeof(double) * kNumBins);
  std::memcpy(s.strength.eq.x.data(), v.sx, sizeof(double) * kNumBins);
  s.num_observations = v.num_observations;
  s.ar_gain = v.ar_gain;
  s.strength.num_equations = v.num_equations;
  s.strength.total = v.total;
}

bool NoiseFold::is_different() const { return differs(view_of_plane(latest_[0]), combined_[0]); }

bool NoiseFold::differs(const PlaneView &latest, const PlaneState &combined) {
  const LinearSystem &c = combined.ar;
  const double *lx = latest.x;
  double dot = 0, l2 = 0, c2 = 0;
  for (int i = 0; i < c.n; ++i) {
    l2 += lx[i] * lx[i];
    c2 += c.x[i] * c.x[i];
    dot += lx[i] * c.x[i];
  }
  const double corr = dot / (std::sqrt(l2) * std::sqrt(c2));
  if (corr < 0.9) return true;
  const double dx = 1.0 / kNumBins;
  const LinearSystem &cs = combined.strength.eq;
  const int sn = kNumBins;
  double diff = 0, total_weight = 0;
  for (int j = 0; j < sn; ++j) {
    double weight = 0;
    for (int i = 0; i < sn; ++i) weight += latest.sA[i * sn + j];
    weight = std::sqrt(weight);
    diff += weight * std::fabs(latest.sx[j] - cs.x[j]);
    total_weight += weight;
  }
  return diff * dx / total_weight > 0.005;
}

void NoiseFold::save_latest() {
  chroma_dirty_ = false;  // combined := latest, whose x is solved
  for (int c = 0; c < 3; ++c) {
    combined_[c].ar.assign(latest_[c].ar);
    combined_[c].strength.eq.assign(latest_[c].strength.eq);
    combined_[c].strength.num_equations = latest_[c].strength.num_equations;
    // (strength.total is deliberately NOT copied: reference quirk)
    combined_[c].num_observations = latest_[c].num_observations;
    combined_[c].ar_gain = latest_[c].ar_gain;
  }
}

static void set_error(std::string &dst, const char *fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  dst = buf;
}

// compute_latest's elementwise passes (see there).
// A division by a power of two is the multiplication by its reciprocal, bit for bit (the same real number, rounded once):
// block areas and sample counts are powers of two except in a cut last row or column, so the passes multiply by `rcp`
// (1 / count where count is a power of two, else 0) and the few other blocks are divided one by one afterwards.
static inline double pow2_rcp(int count) { return (count & (count - 1)) == 0 ? 1.0 / count : 0.0; }
// mean[k] comes in as the block's luma sum and leaves as its mean; bin[k] = StrengthSolver::bin_index(mean[k]).
G1S_HOST_CLONES static void flat_means(size_t m, double *__restrict mean, const double *__restrict area, const double *__restrict rcp,
                                       double *__restrict bin) {
  for (size_t k = 0; k < m; ++k) bin[k] = mean[k] * rcp[k];
  for (size_t k = 0; k < m; ++k)
    if (rcp[k] == 0.0) bin[k] = mean[k] / area[k];  // (not a power of two)
  for (size_t k = 0; k < m; ++k) {
    const double bm = bin[k];
    mean[k] = bm;
    const double val = bm < 0.0 ? 0.0 : (bm > 255.0 ? 255.0 : bm);
    bin[k] = (kNumBins - 1) * val / 255.0;
  }
}
// noise_var = sum_d2 / count - (sum_d / count)^2
G1S_HOST_CLONES static void noise_variances(size_t m, const double *__restrict sd, const double *__restrict sd2, const double *__restrict cnt,
                                            const double *__restrict rcp, double *__restrict nv) {
  for (size_t k = 0; k < m; ++k) {
    const double noise_mean = sd[k] * rcp[k];
    nv[k] = sd2[k] * rcp[k] - noise_mean * noise_mean;
  }
  for (size_t k = 0; k < m; ++k)
    if (rcp[k] == 0.0) {
      const double noise_mean = sd[k] / cnt[k];
      nv[k] = sd2[k] / cnt[k] - noise_mean * noise_mean;
    }
}
// uncorr_std / noise_gain, uncorr_std = sqrt(max(noise_var / 16, noise_var - (corr * luma_strength)^2))
G1S_HOST_CLONES static void uncorrelated_stds(size_t m, const double *__restrict nv, const double *__restrict ls, double corr, double noise_gain,
                                              double *__restrict out) {
  for (size_t k = 0; k < m; ++k) {
    const double cl = corr * ls[k];
    const double t0 = nv[k] / 16, t1 = nv[k] - cl * cl;
    out[k] = std::sqrt(t0 > t1 ? t0 : t1) / noise_gain;
  }
}

#ifdef G1S_LATEST_PROFILE  // (a measurement build of tools/: where the per-frame half's time goes)
double g_latest_stage_s[8];
#define STAGE(i)                                                                                                   \
  do {                                                                                                             \
    const double t_now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); \
    g_latest_stage_s[i] += t_now - t_stage;                                                                        \
    t_stage = t_now;                                                                                               \
  } while (0)
#else
#define STAGE(i) ((void)0)
#endif

int compute_latest(const uint8_t *rec, size_t size, uint32_t lag, FrameLatest &out) {
#ifdef G1S_LATEST_PROFILE
  double t_stage = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
#endif
  out.status = G1S_OK;
  out.err.clear();
  auto fail = [&](int code, const char *fmt, int arg = 0) {
    char buf[160];
    snprintf(buf, sizeof(buf), fmt, arg);
    out.err = buf;
    out.status = code;
    return code;
  };
  if (size < sizeof(RecHeader)) return fail(G1S_ERR_INVALID, "record too small");
  RecHeader h;
  std::memcpy(&h, rec, sizeof(h));
  if (h.magic != kRecMagic || h.lag != lag || h.size_bytes > size)
    return fail(G1S_ERR_INVALID, "bad record header (magic/lag/size)");
  const RecLayout L = make_layout(h.width, h.height, h.nplanes, h.lag);
  if (L.size != h.size_bytes) return fail(G1S_ERR_INVALID, "record layout mismatch");
  const int nbw = (int)h.nbw, nbh = (int)h.nbh;
  const uint8_t *mask = rec + L.off_mask;
  const uint32_t *luma_sum = reinterpret_cast<const uint32_t *>(rec + L.off_luma_sum);
  const int w = (int)h.width, hh = (int)h.height;
  const int n = (int)num_coeffs(lag);
  out.nplanes = h.nplanes;
  for (int c = 0; c < 3; ++c) {
    if (out.st[c].ar.n != n + (c > 0)) out.st[c].ar.resize(n + (c > 0));
    out.st[c].ar.clear();
    out.st[c].num_observations = 0;
    out.st[c].ar_gain = 1.0;
    out.st[c].strength.clear();
  }
  int num_flat = 0;
  for (int i = 0; i < nbw * nbh; ++i) num_flat += mask[i] != 0;
  if (num_flat <= 1) return fail(G1S_ERR_NOT_ENOUGH_FLAT, "Not enough flat blocks to update noise estimate");
  // the flat blocks in raster order, their luma means and the means' bin positions (every plane's measurements use them)
  {
    const size_t nf = (size_t)num_flat;
    out.scratch_idx.resize(nf);
    out.scratch_pos.resize(nf);
    out.scratch_mean.resize(nf);
    out.scratch_std.resize(2 * nf);  // (here: the luma blocks' areas and their reciprocals)
    out.scratch_bin.resize(nf);
    uint32_t *idx = out.scratch_idx.data(), *pos = out.scratch_pos.data();
    double *sum = out.scratch_mean.data(), *area = out.scratch_std.data(), *rcp = area + nf;
    const double rcp_full = pow2_rcp(kBlock * kBlock);
    size_t k = 0;
    for (int by = 0; by < nbh; ++by) {
      const int lh = std::min(hh - by * kBlock, kBlock);
      const uint8_t *mrow = mask + (size_t)by * nbw;
      for (int bx = 0; bx < nbw; ++bx) {
        if (!mrow[bx]) continue;
        const int bi = by * nbw + bx;
        const int lw = std::min(w - bx * kBlock, kBlock);
        idx[k] = (uint32_t)bi;
        pos[k] = (uint32_t)by << 16 | (uint32_t)bx;
        sum[k] = (double)luma_sum[bi];
        area[k] = (double)(lw * lh);
        rcp[k] = (lw == kBlock && lh == kBlock) ? rcp_full : pow2_rcp(lw * lh);
        ++k;
      }
    }
    flat_means(nf, sum, area, rcp, out.scratch_bin.data());
  }
  STAGE(0);

  for (int c = 0; c < (int)h.nplanes; ++c) {
    const bool is_chroma = c != 0;
    const int sx = is_chroma ? (int)h.xdec : 0, sy = is_chroma ? (int)h.ydec : 0;
    PlaneState &lat = out.st[c];
    const int nc = lat.ar.n;
    // ---- exact integer sums -> f64 normal equations (one rounding each) ----
    {
      const int64_t *S = reinterpret_cast<const int64_t *>(rec + L.off_ar[c]);
      const int64_t *Sb = S + size_t(nc) * nc;
      const double ns = (double)((1 << sx) * (1 << sy));
      for (int i = 0; i < nc; ++i) {
        for (int j = 0; j < nc; ++j) {
          double den = kNorm2;
          if (is_chroma && i == nc - 1) den *= ns;
          if (is_chroma && j == nc - 1) den *= ns;
          // the device fills the upper triangle; the matrix is symmetric
          const int64_t s = i <= j ? S[i * nc + j] : S[j * nc + i];
          lat.ar.A[i * nc + j] = (double)s / den;
        }
        double den = kNorm2;
        if (is_chroma && i == nc - 1) den *= ns;
        lat.ar.b[i] = (double)Sb[i] / den;
      }
      lat.num_observations = Sb[nc];
    }
    STAGE(1);
    if (!ar_solve(lat, is_chroma)) {
      if (is_chroma)
        chroma_fallback(lat);
      else
        return fail(G1S_ERR_SOLVE, "Solving latest noise equation system failed %d!", c);
    }
    STAGE(2);
    // ---- noise strength vs. intensity measurements, block raster order ----
    {
      const int32_t *sum_d = reinterpret_cast<const int32_t *>(rec + L.off_sum_d[c]);
      const uint32_t *sum_d2 = reinterpret_cast<const uint32_t *>(rec + L.off_sum_d2[c]);
      const int bw = kBlock >> sx, bh = kBlock >> sy;
      const double luma_gain = out.st[0].ar_gain;
      const double noise_gain = lat.ar_gain;
      const double corr = is_chroma ? lat.ar.x[n] : 0;
      // Three passes over the plane's flat blocks, raster order every time.  The measurements are independent from block to
      // block and nearly all divisions and square roots (the divider was 150 of this function's 158 us a 4K frame): they run
      // as elementwise loops over compact arrays, which the compiler turns into packed divides -- IEEE division and square
      // root round the same packed or scalar, -ffp-contract=off keeps every product and difference its own operation.  The
      // block mean and its bin position are the luma block's for every plane: worked out once per frame (flat_means).  Then
      // the accumulation into the equation system, whose f64 sums must run in the reference's order.
      // Every operation and its operands are those of the single loop this replaces: same bits.
      std::vector<double> &sc = out.scratch_plane;
      const size_t cap = out.scratch_idx.size();
      sc.resize(7 * cap);
      double *p_sd = sc.data(), *p_sd2 = p_sd + cap, *p_cnt = p_sd2 + cap, *p_rcp = p_cnt + cap, *p_ls = p_rcp + cap, *p_nv = p_ls + cap,
             *p_std = p_nv + cap;
      // the blocks a plane measures, their counts: luma's list in scratch_sel0, the chroma planes' (one list: the two planes
      // have one geometry) in scratch_sel; plane 2 takes plane 1's
      std::vector<uint32_t> &selv = c == 0 ? out.scratch_sel0 : out.scratch_sel;
      selv.resize(cap);
      uint32_t *sel = selv.data();  // position in the frame's flat list of the blocks this plane measures
      const uint32_t *idx = out.scratch_idx.data(), *pos = out.scratch_pos.data();
      const int pw = w >> sx, ph = hh >> sy;
      const int full_bx = pw / bw, full_by = ph / bh;  // blocks left of / above these are whole
      const double cnt_full = (double)(bw * bh), rcp_full = pow2_rcp(bw * bh);
      const bool full_counts = bw * bh > kBlock;
      size_t m = 0;
      // (G1S_FOLD_NO_SHARE=1, a test aid: every plane builds its own list and its own matrix -- what the sharing must equal)
      static const bool no_share = getenv("G1S_FOLD_NO_SHARE") != nullptr;
      if (c == 2 && !no_share) {
        m = out.scratch_m[1];
        for (size_t k = 0; k < m; ++k) {
          const int bi = (int)idx[sel[k]];
          p_sd[k] = (double)sum_d[bi];
          p_sd2[k] = (double)sum_d2[bi];
        }
      } else {
        for (size_t k = 0; k < cap; ++k) {
          const int bi = (int)idx[k];
          const int by = (int)(pos[k] >> 16), bx = (int)(pos[k] & 0xffffu);
          if (bx < full_bx && by < full_by) {
            if (!full_counts) continue;
            p_cnt[m] = cnt_full;
            p_rcp[m] = rcp_full;
          } else {
            const int sh = std::min(ph - by * bh, bh);
            const int sw = std::min(pw - bx * bw, bw);
            if (!(sw * sh > kBlock)) continue;
            p_cnt[m] = (double)(sw * sh);
            p_rcp[m] = pow2_rcp(sw * sh);
          }
          p_sd[m] = (double)sum_d[bi];
          p_sd2[m] = (double)sum_d2[bi];
          sel[m] = (uint32_t)k;
          ++m;
        }
        out.scratch_m[c] = m;
      }
      STAGE(3);
      noise_variances(m, p_sd, p_sd2, p_cnt, p_rcp, p_nv);
      STAGE(7);
      const double *bins = out.scratch_bin.data();
      if (c == 1) {
        const double *lx = out.st[0].strength.eq.x.data();
        for (size_t k = 0; k < m; ++k) {  // luma_gain * luma strength.value_at(block_mean)
          const double bin = bins[sel[k]];
          const int i0 = (int)bin;  // (bin >= 0: the floor)
          const int i1 = std::min(kNumBins - 1, i0 + 1);
          const double a = bin - i0;
          p_ls[k] = luma_gain * ((1.0 - a) * lx[i0] + a * lx[i1]);
        }
      } else if (c == 0) {
        for (size_t k = 0; k < m; ++k) p_ls[k] = 0;
      }  // (c == 2: plane 1's values -- the same blocks, the same luma strength; nothing between the planes writes p_ls)
      uncorrelated_stds(m, p_nv, p_ls, corr, noise_gain, p_std);
      STAGE(4);
      // the matrix side of the accumulation depends on the bin positions alone: a plane that measures the very blocks an
      // earlier plane measured (Cr after Cb always; Cb after luma unless a cut last row or column is too small at chroma
      // resolution) copies that plane's matrix (solve() leaves A as it was) and accumulates its b side only
      const int like = no_share ? -1
                       : c == 2 ? 1
                       : (c == 1 && m == out.scratch_m[0] && std::memcmp(sel, out.scratch_sel0.data(), m * sizeof(uint32_t)) == 0) ? 0 : -1;
      if (like >= 0) lat.strength.add_measurements_like(out.st[like].strength, bins, sel, p_std, m);
      else lat.strength.add_measurements(bins, sel, p_std, m);
      STAGE(5);
    }
    if (!lat.strength.solve()) return fail(G1S_ERR_SOLVE, "Solving latest noise strength failed!");
    STAGE(6);
  }
  return G1S_OK;
}

int NoiseFold::push(const uint8_t *rec, size_t size) {
  FrameLatest fl;
  compute_latest(rec, size, lag_, fl);
  return push_latest(fl);
}

// The combined chroma state is only read at segment boundaries: between them
// the per-frame AR solve is skipped and the strength solve is reduced to its
// side effect on b; this brings x up to date for the current (A, b).
void NoiseFold::finalize_chroma() const {
  if (!chroma_dirty_) return;
  for (int c = 1; c < 3; ++c) {
    PlaneState &com = combined_[c];
    if (!ar_solve(com, true)) chroma_fallback(com);
    com.strength.solve_x_only();
  }
  chroma_dirty_ = false;
}

// ---- FrameLatest <-> blob ----
size_t latest_blob_size(uint32_t lag) { return sizeof(LatestHeader) + 3 * plane_blob_bytes((int)num_coeffs(lag) + 1); }

void latest_to_blob(const FrameLatest &fl, uint32_t lag, uint8_t *blob) {
  const size_t total = latest_blob_size(lag);
  std::memset(blob, 0, total);
  LatestHeader h{};
  h.magic = kLatestMagic;
  h.lag = lag;
  h.nplanes = fl.nplanes;
  h.status = fl.status;
  h.size_bytes = (uint32_t)total;
  std::snprintf(h.err, sizeof(h.err), "%s", fl.err.c_str());
  std::memcpy(blob, &h, sizeof(h));
  const int ncm = (int)num_coeffs(lag) + 1;
  for (int c = 0; c < 3; ++c) {
    uint8_t *p = blob + sizeof(LatestHeader) + c * plane_blob_bytes(ncm);
    const PlaneState &s = fl.st[c];
    LatestPlaneHead ph{};
    ph.num_observations = s.num_observations;
    ph.ar_gain = s.ar_gain;
    ph.num_equations = (int32_t)s.strength.num_equations;
    ph.total = s.strength.total;
    std::memcpy(p, &ph, sizeof(ph));
    double *d = reinterpret_cast<double *>(p + sizeof(ph));
    const int nc = s.ar.n;
    if (nc > 0 && nc <= ncm) {
      std::memcpy(d, s.ar.A.data(), sizeof(double) * nc * nc);
      std::memcpy(d + ncm * ncm, s.ar.b.data(), sizeof(double) * nc);
      std::memcpy(d + ncm * ncm + ncm, s.ar.x.data(), sizeof(double) * nc);
    }
    double *q = d + ncm * ncm + 2 * ncm;
    std::memcpy(q, s.strength.eq.A.data(), sizeof(double) * kNumBins * kNumBins);
    std::memcpy(q + kNumBins * kNumBins, s.strength.eq.b.data(), sizeof(double) * kNumBins);
    std::memcpy(q + kNumBins * kNumBins + kNumBins, s.strength.eq.x.data(), sizeof(double) * kNumBins);
  }
}

int latest_from_blob(const uint8_t *blob, size_t size, uint32_t lag, FrameLatest &out) {
  LatestHeader h;
  if (size < sizeof(h)) return G1S_ERR_INVALID;
  std::memcpy(&h, blob, sizeof(h));
  if (h.magic != kLatestMagic || h.lag != lag || h.size_bytes != latest_blob_size(lag) || h.size_bytes > size || h.nplanes > 3)
    return G1S_ERR_INVALID;
  out.nplanes = h.nplanes;
  out.status = h.status;
  h.err[sizeof(h.err) - 1] = 0;
  out.err = h.err;
  const int n = (int)num_coeffs(lag), ncm = n + 1;
  for (int c = 0; c < 3; ++c) {
    const uint8_t *p = blob + sizeof(LatestHeader) + c * plane_blob_bytes(ncm);
    PlaneState &s = out.st[c];
    LatestPlaneHead ph;
    std::memcpy(&ph, p, sizeof(ph));
    s.num_observations = ph.num_observations;
    s.ar_gain = ph.ar_gain;
    s.strength.num_equations = ph.num_equations;
    s.strength.total = ph.total;
    const int nc = n + (c > 0);
    if (s.ar.n != nc) s.ar.resize(nc);
    const double *d = reinterpret_cast<const double *>(p + sizeof(ph));
    std::memcpy(s.ar.A.data(), d, sizeof(double) * nc * nc);
    std::memcpy(s.ar.b.data(), d + ncm * ncm, sizeof(double) * nc);
    std::memcpy(s.ar.x.data(), d + ncm * ncm + ncm, sizeof(double) * nc);
    const double *q = d + ncm * ncm + 2 * ncm;
    std::memcpy(s.strength.eq.A.data(), q, sizeof(double) * kNumBins * kNumBins);
    std::memcpy(s.strength.eq.b.data(), q + kNumBins * kNumBins, sizeof(double) * kNumBins);
    std::memcpy(s.strength.eq.x.data(), q + kNumBins * kNumBins + kNumBins, sizeof(double) * kNumBins);
  }
  return G1S_OK;
}

int view_of_blob(const uint8_t *blob, size_t size, uint32_t lag, FrameView &out) {
  if (size < sizeof(LatestHeader) || (reinterpret_cast<uintptr_t>(blob) & 7)) return G1S_ERR_INVALID;
  const LatestHeader *h = reinterpret_cast<const LatestHeader *>(blob);
  if (h->magic != kLatestMagic || h->lag != lag || h->size_bytes != latest_blob_size(lag) || h->size_bytes > size || h->nplanes > 3)
    return G1S_ERR_INVALID;
  if (!memchr(h->err, 0, sizeof(h->err))) return G1S_ERR_INVALID;
  out.nplanes = h->nplanes;
  out.status = h->status;
  out.err = h->err;
  const int n = (int)num_coeffs(lag), ncm = n + 1;
  for (int c = 0; c < 3; ++c) {
    const uint8_t *p = blob + sizeof(LatestHeader) + c * plane_blob_bytes(ncm);
    const LatestPlaneHead *ph = reinterpret_cast<const LatestPlaneHead *>(p);
    PlaneView &v = out.st[c];
    v.num_observations = ph->num_observations;
    v.ar_gain = ph->ar_gain;
    v.num_equations = ph->num_equations;
    v.total = ph->total;
    v.n = n + (c > 0);
    const double *d = reinterpret_cast<const double *>(p + sizeof(*ph));
    v.A = d;
    v.b = d + ncm * ncm;
    v.x = d + ncm * ncm + ncm;
    const double *q = d + ncm * ncm + 2 * ncm;
    v.sA = q;
    v.sb = q + kNumBins * kNumBins;
    v.sx = q + kNumBins * kNumBins + kNumBins;
  }
  return G1S_OK;
}

int NoiseFold::push_latest(FrameLatest &fl) {
  if (fl.status != G1S_OK) {
    err_ = fl.err;
    return fl.status;
  }
  for (int c = 0; c < 3; ++c) std::swap(latest_[c], fl.st[c]);
  bool y_model_different = false;
  for (int c = 0; c < (int)fl.nplanes; ++c) {
    const bool is_chroma = c != 0;
    PlaneState &lat = latest_[c];
    if (c == 0 && combined_[0].strength.num_equations > 0 && is_different()) y_model_different = true;
    if (y_model_different) continue;
    PlaneState &com = combined_[c];
    com.num_observations += lat.num_observations;
    com.ar.add(lat.ar);
    com.strength.add(lat.strength);
    if (!is_chroma) {
      if (!ar_solve(com, false)) {
        set_error(err_, "Solving combined noise equation system failed %d!", c);
        return G1S_ERR_SOLVE;
      }
      if (!com.strength.solve()) {
        set_error(err_, "Solving combined noise strength failed!");
        return G1S_ERR_SOLVE;
      }
    } else {
      com.strength.apply_regularisation_to_b();
      chroma_dirty_ = true;
    }
  }
  if (y_model_different) {
    const uint64_t cur = frame_count_ * 10000000ULL * (uint64_t)fps_den_ / (uint64_t)fps_num_;
    table_.push_back(grain_parameters(prev_timestamp_, cur));
    save_latest();
    prev_timestamp_ = cur;
  }
  frame_count_ += 1;
  return G1S_OK;
}

namespace {
struct FoldProfile {  // G1S_FOLD_PROFILE=1: where the ordered merge's time goes (printed when the process ends)
  bool on = getenv("G1S_FOLD_PROFILE") != nullptr;
  double prefix = 0, solves = 0, commit = 0;
  size_t frames = 0;
  ~FoldProfile() {
    if (on && frames)
      fprintf(stderr, "ordered merge, us per frame: prefix sums %.2f, solves %.2f, in-order commit %.2f (%zu frames)\n", prefix * 1e6 / frames,
              solves * 1e6 / frames, commit * 1e6 / frames, frames);
  }
} g_fold_profile;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int NoiseFold::push_latest_many(FrameLatest *fl, size_t n, const ParallelFor &pfor) {
  if (views_.size() < n) views_.resize(n);
  for (size_t i = 0; i < n; ++i) view_of(fl[i], views_[i]);
  return push_latest_many(views_.data(), n, pfor);
}

int NoiseFold::push_latest_many(const FrameView *fl, size_t n, const ParallelFor &pfor) {
  constexpr size_t kWindow = 256;  // (frames per speculative window: a dispatch to the merge pool costs tens of microseconds)
  if (snap_.size() < kWindow) {
    snap_.resize(kWindow);
    snap_ok_.resize(kWindow);
  }
  // one frame's chroma plane into a running combined state (what push_latest does for c > 0, in frame order)
  auto chroma_add = [](PlaneState &com, const PlaneView &lat) {
    com.num_observations += lat.num_observations;
    com.ar.add(lat.A, lat.b);
    com.strength.add(lat.sA, lat.sb, lat.num_equations, lat.total);
    com.strength.apply_regularisation_to_b();
  };
  size_t i = 0;
  while (i < n) {
    // ---- a window of frames, no segment cut assumed ----
    const double t_a = g_fold_profile.on ? now_s() : 0;
    size_t W = std::min(kWindow, n - i);
    for (size_t j = 0; j < W; ++j)
      if (fl[i + j].status != G1S_OK) {  // an error frame ends the window; it is reported when reached
        W = j;
        break;
      }
    if (W == 0) {
      err_ = fl[i].err;
      return fl[i].status;
    }
    // Running sums, each element's sequential in frame order (the additions happen in the reference's order: same bits),
    // elements independent of each other -- so slices of the systems run next to each other: for luma the prefix sums (one
    // state per frame: the solves below need them all), for Cb / Cr the combined state after the window's last frame (nothing
    // reads a chroma state between segment boundaries).  The serial commit below then only swaps pointers: the frames'
    // states were parsed on other cores, and every byte the serial stage does not touch is a cache miss it does not wait for.
    uint32_t cplanes = fl[i].nplanes;
    for (size_t j = 1; j < W; ++j) cplanes = std::min(cplanes, fl[i + j].nplanes);
    // (task t: plane t / 5; parts 0..2 = thirds of the AR matrix, part 0 with b and the observation count; parts 3, 4 = halves
    //  of the strength matrix, part 3 with b, the equation count, the total and the regularisation term they give)
    // (a task's slice of frame j + 4 is asked for while frame j is added: the slices lie a blob apart, no hardware
    //  prefetcher follows that, and the blobs arrive cold)
    auto ahead = [](const double *p, int count) {
      for (int k = 0; k < count; k += 8) __builtin_prefetch(p + k, 0, 0);
    };
    const std::function<void(int)> sums = [&](int t) {
      const int plane = t / 5, part = t % 5;
      if (plane > 0 && (uint32_t)plane >= cplanes) return;
      const bool ar_part = part < 3;
      const int nn = ar_part ? combined_[plane].ar.n * combined_[plane].ar.n : kNumBins * kNumBins;
      const int lo = ar_part ? nn * part / 3 : nn * (part - 3) / 2, hi = ar_part ? nn * (part + 1) / 3 : nn * (part - 2) / 2;
      const bool head = part == 0 || part == 3;
      if (plane == 0) {
        for (size_t j = 0; j < W; ++j) {
          PlaneState &s = snap_[j];
          const PlaneState &prev = j ? snap_[j - 1] : combined_[0];
          const PlaneView &lat = fl[i + j].st[0];
          if (j + 4 < W) ahead((ar_part ? fl[i + j + 4].st[0].A : fl[i + j + 4].st[0].sA) + lo, hi - lo);
          if (ar_part) {
            sum_into(s.ar.A.data() + lo, prev.ar.A.data() + lo, lat.A + lo, hi - lo);
            if (head) {
              sum_into(s.ar.b.data(), prev.ar.b.data(), lat.b, s.ar.n);
              s.num_observations = prev.num_observations + lat.num_observations;
            }
          } else {
            sum_into(s.strength.eq.A.data() + lo, prev.strength.eq.A.data() + lo, lat.sA + lo, hi - lo);
            if (head) {
              sum_into(s.strength.eq.b.data(), prev.strength.eq.b.data(), lat.sb, kNumBins);
              s.strength.num_equations = prev.strength.num_equations + lat.num_equations;
              s.strength.total = prev.strength.total + lat.total;
              s.strength.apply_regularisation_to_b();  // (the first half of StrengthSolver::solve)
            }
          }
        }
      } else {
        PlaneState &acc = csum_[plane];
        const PlaneState &com = combined_[plane];
        if (ar_part) {
          std::memcpy(acc.ar.A.data() + lo, com.ar.A.data() + lo, sizeof(double) * (hi - lo));
          if (head) {
            acc.ar.b = com.ar.b;
            acc.num_observations = com.num_observations;
          }
          for (size_t j = 0; j < W; ++j) {
            const PlaneView &lat = fl[i + j].st[plane];
            if (j + 4 < W) ahead(fl[i + j + 4].st[plane].A + lo, hi - lo);
            add_into(acc.ar.A.data() + lo, lat.A + lo, hi - lo);
            if (head) {
              add_into(acc.ar.b.data(), lat.b, acc.ar.n);
              acc.num_observations += lat.num_observations;
            }
          }
        } else {
          std::memcpy(acc.strength.eq.A.data() + lo, com.strength.eq.A.data() + lo, sizeof(double) * (hi - lo));
          if (head) {
            acc.strength.eq.b = com.strength.eq.b;
            acc.strength.num_equations = com.strength.num_equations;
            acc.strength.total = com.strength.total;
          }
          for (size_t j = 0; j < W; ++j) {
            const PlaneView &lat = fl[i + j].st[plane];
            if (j + 4 < W) ahead(fl[i + j + 4].st[plane].sA + lo, hi - lo);
            add_into(acc.strength.eq.A.data() + lo, lat.sA + lo, hi - lo);
            if (head) {
              add_into(acc.strength.eq.b.data(), lat.sb, kNumBins);
              acc.strength.num_equations += lat.num_equations;
              acc.strength.total += lat.total;
              acc.strength.apply_regularisation_to_b();
            }
          }
        }
      }
    };
    // (sizes settled before the tasks share the states)
    for (size_t j = 0; j < W; ++j)
      if (snap_[j].ar.n != combined_[0].ar.n) snap_[j].ar.resize(combined_[0].ar.n);
    for (int c = 1; c < 3; ++c)
      if (csum_[c].ar.n != combined_[c].ar.n) csum_[c].ar.resize(combined_[c].ar.n);
    if (pfor && W > 8) pfor(15, sums);
    else
      for (int t = 0; t < 15; ++t) sums(t);
    // ---- the solves, independent of each other; and the is_different() test of every frame whose predecessor's state the
    //      same task solved (frame j is tested against the combined model after frame j - 1: snap_[j - 1]) ----
    const double t_b = g_fold_profile.on ? now_s() : 0;
    if (snap_cut_.size() < kWindow) snap_cut_.resize(kWindow);
    auto test_one = [&](size_t j) {
      const PlaneState &before = j ? snap_[j - 1] : combined_[0];
      snap_cut_[j] = (uint8_t)(before.strength.num_equations > 0 && differs(fl[i + j].st[0], before));
    };
    const int T = pfor && W > 1 ? (int)std::min<size_t>((W + 7) / 8, 32) : 1;  // tasks of 8+ solves each: a solve is ~2 us,
                                                                    // waking a thread costs more
    const std::function<void(int)> range = [&](int t) {
      const size_t a = W * t / T, b = W * (t + 1) / T;
      for (size_t j = a; j < b; ++j) {
        const bool ok_ar = ar_solve(snap_[j], false);
        const bool ok_st = snap_[j].strength.solve_x_only();
        snap_ok_[j] = (uint8_t)((ok_ar ? 1 : 0) | (ok_st ? 2 : 0));
        if (j > a) test_one(j);
      }
    };
    if (T > 1) pfor(T, range);
    else range(0);
    // ---- in order: the first segment cut or failed solve of the window, then one commit for the frames before it ----
    const double t_c = g_fold_profile.on ? now_s() : 0;
    for (int t = 0; t < T; ++t) test_one(W * t / T);  // (the first frame of each task's range: its predecessor is solved now)
    size_t m = W;  // frames that go into the combined model as assumed
    int failed = 0;
    bool cut = false;
    for (size_t j = 0; j < W; ++j) {
      if (snap_cut_[j]) {
        m = j;
        cut = true;
        break;
      }
      if ((snap_ok_[j] & 3) != 3) {
        m = j;
        failed = (snap_ok_[j] & 1) ? 2 : 1;
        break;
      }
    }
    if (m > 0) {
      std::swap(combined_[0], snap_[m - 1]);
      frame_count_ += m;
    }
    size_t done = m;
    if (m == W) {  // the window went through: the combined chroma states are the running sums
      for (uint32_t c = 1; c < cplanes; ++c) {
        std::swap(combined_[c].ar, csum_[c].ar);
        std::swap(combined_[c].strength.eq, csum_[c].strength.eq);
        combined_[c].strength.num_equations = csum_[c].strength.num_equations;
        combined_[c].strength.total = csum_[c].strength.total;
        combined_[c].num_observations = csum_[c].num_observations;
        chroma_dirty_ = true;
      }
      // (frames with fewer planes than the window's minimum do not exist: cplanes is the minimum; a frame with MORE planes
      //  than another cannot happen inside one generator)
    } else {  // the chroma planes of the frames before the stop, in order, into the combined model that ends there
      for (size_t q = 0; q < m; ++q)
        for (int c = 1; c < (int)fl[i + q].nplanes; ++c) {
          chroma_add(combined_[c], fl[i + q].st[c]);
          chroma_dirty_ = true;
        }
    }
    if (failed) {
      if (failed == 1) set_error(err_, "Solving combined noise equation system failed %d!", 0);
      else set_error(err_, "Solving combined noise strength failed!");
      return G1S_ERR_SOLVE;
    }
    if (cut) {  // a new segment starts with frame m; the states behind it were built on a combined model that is gone
      for (int c = 0; c < 3; ++c) load_plane(latest_[c], fl[i + m].st[c]);
      const uint64_t cur = frame_count_ * 10000000ULL * (uint64_t)fps_den_ / (uint64_t)fps_num_;
      table_.push_back(grain_parameters(prev_timestamp_, cur));
      save_latest();
      prev_timestamp_ = cur;
      frame_count_ += 1;
      done = m + 1;
    }
    i += done;
    if (g_fold_profile.on) {
      const double t_d = now_s();
      g_fold_profile.prefix += t_b - t_a;
      g_fold_profile.solves += t_c - t_b;
      g_fold_profile.commit += t_d - t_c;
      g_fold_profile.frames += done;
    }
  }
  return G1S_OK;
}

void NoiseFold::finish(std::vector<g1s_segment_t> &out) {
  table_.push_back(grain_parameters(prev_timestamp_, (uint64_t)INT64_MAX));
  out = table_;
}

g1s_segment_t NoiseFold::grain_parameters(uint64_t start_ts, uint64_t end_ts) const {
  finalize_chroma();
  g1s_segment_t g;
  std::memset(&g, 0, sizeof(g));
  g.random_seed = start_ts == 0 ? kDefaultGrainSeed : 0;
  g.start_time = start_ts;
  g.end_time = end_ts;
  g.ar_coeff_lag = (uint8_t)lag_;

  std::vector<double> px[3], py[3];
  combined_[0].strength.fit_piecewise(G1S_NUM_Y_POINTS, px[0], py[0]);
  combined_[1].strength.fit_piecewise(G1S_NUM_UV_POINTS, px[1], py[1]);
  combined_[2].strength.fit_piecewise(G1S_NUM_UV_POINTS, px[2], py[2]);
  double max_scaling_value = 1e-4;
  for (int c = 0; c < 3; ++c) {
    for (size_t i = 0; i < px[c].size(); ++i) {
      px[c][i] = std::min(255.0, px[c][i]);
      py[c][i] = std::min(255.0, py[c][i]);
      if (py[c][i] > max_scaling_value) max_scaling_value = py[c][i];
    }
  }
  const int log2v = clampi((int)std::floor(std::log2(max_scaling_value) + 1), 2, 5);
  g.scaling_shift = (uint8_t)(5 + (8 - log2v));
  const double scale_factor = 1 << (8 - log2v);
  g.num_y_points = (uint8_t)px[0].size();
  g.num_cb_points = (uint8_t)px[1].size();
  g.num_cr_points = (uint8_t)px[2].size();
  uint8_t(*dst[3])[2] = {g.scaling_points_y, g.scaling_points_cb, g.scaling_points_cr};
  for (int c = 0; c < 3; ++c) {
    for (size_t i = 0; i < px[c].size(); ++i) {
      dst[c][i][0] = (uint8_t)clampi((int)(px[c][i] + 0.5), 0, 255);
      dst[c][i][1] = (uint8_t)clampi((int)(scale_factor * py[c][i] + 0.5), 0, 255);
    }
  }

  const int n_coeff = n_;
  double max_coeff = 1e-4, min_coeff = -1e-4;
  double y_corr[2] = {0, 0};
  double avg_luma_strength = 0;
  for (int c = 0; c < 3; ++c) {
    const LinearSystem &eq = combined_[c].ar;
    for (int i = 0; i < n_coeff; ++i) {
      if (eq.x[i] > max_coeff) max_coeff = eq.x[i];
      if (eq.x[i] < min_coeff) min_coeff = eq.x[i];
    }
    const LinearSystem &se = combined_[c].strength.eq;
    double average_strength = 0, total_weight = 0;
    for (int i = 0; i < se.n; ++i) {
      double wgt = 0;
      for (int j = 0; j < se.n; ++j) wgt += se.A[i * se.n + j];
      wgt = std::sqrt(wgt);
      average_strength += se.x[i] * wgt;
      total_weight += wgt;
    }
    if (total_weight == 0)
      average_strength = 1;
    else
      average_strength /= total_weight;
    if (c == 0) {
      avg_luma_strength = average_strength;
    } else {
      y_corr[c - 1] = avg_luma_strength * eq.x[n_coeff] / average_strength;
      if (y_corr[c - 1] > max_coeff) max_coeff = y_corr[c - 1];
      if (y_corr[c - 1] < min_coeff) min_coeff = y_corr[c - 1];
    }
  }
  {
    const double a = 1 + std::floor(std::log2(max_coeff));
    const double b = std::ceil(std::log2(-min_coeff));
    g.ar_coeff_shift = (uint8_t)clampi(7 - (int)(a > b ? a : b), 6, 9);
  }
  const double scale_ar = 1 << g.ar_coeff_shift;
  int8_t *ar[3] = {g.ar_coeffs_y, g.ar_coeffs_cb, g.ar_coeffs_cr};
  for (int c = 0; c < 3; ++c) {
    const LinearSystem &eq = combined_[c].ar;
    for (int i = 0; i < n_coeff; ++i)
      ar[c][i] = (int8_t)clampi((int)std::round(scale_ar * eq.x[i]), -128, 127);
    if (c > 0) ar[c][n_coeff] = (int8_t)clampi((int)std::round(scale_ar * y_corr[c - 1]), -128, 127);
  }
  g.num_y_coeffs = (uint8_t)n_coeff;
  g.num_uv_coeffs = (uint8_t)(n_coeff + 1);
  g.cb_mult = 128;
  g.cb_luma_mult = 192;
  g.cb_offset = 256;
  g.cr_mult = 128;
  g.cr_luma_mult = 192;
  g.cr_offset = 256;
  g.chroma_scaling_from_luma = 0;
  g.grain_scale_shift = 0;
  g.overlap_flag = 1;
  return g;
}

// ---------------------------------------------------------------- .tbl ----
long format_tbl(const g1s_segment_t *segs, size_t n, char *buf, size_t cap) {
  std::string s = "filmgrn1\n";
  char line[512];
  for (size_t k = 0; k < n; ++k) {
    const g1s_segment_t &g = segs[k];
    snprintf(line, sizeof(line), "E %llu %llu 1 %u 1\n", (unsigned long long)g.start_time,
             (unsigned long long)g.end_time, (unsigned)g.random_seed);
    s += line;
    snprintf(line, sizeof(line), "\tp %u %u %u %u %u %u %u %u %u %u %u %u\n", g.ar_coeff_lag,
             g.ar_coeff_shift, g.grain_scale_shift, g.scaling_shift, g.chroma_scaling_from_luma,
             g.overlap_flag, g.cb_mult, g.cb_luma_mult, g.cb_offset, g.cr_mult, g.cr_luma_mult,
             g.cr_offset);
    s += line;
    auto points = [&](const char *tag, const uint8_t(*p)[2], int np) {
      s += tag;
      for (int i = 0; i < np; ++i) {
        snprintf(line, sizeof(line), " %u %u", p[i][0], p[i][1]);
        s += line;
      }
      s += "\n";
    };
    // "\tsY {n} " keeps its trailing space before the points (src/main.rs:659)
    snprintf(line, sizeof(line), "\tsY %u ", g.num_y_points);
    points(line, g.scaling_points_y, g.num_y_points);
    snprintf(line, sizeof(line), "\tsCb %u", g.num_cb_points);
    points(line, g.scaling_points_cb, g.num_cb_points);
    snprintf(line, sizeof(line), "\tsCr %u", g.num_cr_points);
    points(line, g.scaling_points_cr, g.num_cr_points);
    auto coeffs = [&](const char *tag, const int8_t *c, int nc) {
      s += tag;
      for (int i = 0; i < nc; ++i) {
        snprintf(line, sizeof(line), " %d", c[i]);
        s += line;
      }
      s += "\n";
    };
    coeffs("\tcY", g.ar_coeffs_y, g.num_y_coeffs);
    coeffs("\tcCb", g.ar_coeffs_cb, g.num_uv_coeffs);
    coeffs("\tcCr", g.ar_coeffs_cr, g.num_uv_coeffs);
  }
  if (s.size() > cap) return G1S_ERR_CAPACITY;
  std::memcpy(buf, s.data(), s.size());
  return (long)s.size();
}

// The reader of the same text (av1_grain::parse_grain_table as `apply` uses it, src/main.rs:228-241):
// header line, then per segment the E line and the seven tagged lines in any order.
int parse_tbl(const char *text, size_t len, std::vector<g1s_segment_t> &out, std::string &err) {
  out.clear();
  std::vector<std::string> lines;
  {
    std::string cur;
    for (size_t i = 0; i < len; ++i) {
      if (text[i] == '\n') {
        lines.push_back(cur);
        cur.clear();
      } else if (text[i] != '\r') {
        cur.push_back(text[i]);
      }
    }
    if (!cur.empty()) lines.push_back(cur);
  }
  auto blank = [](const std::string &l) { return l.find_first_not_of(" \t") == std::string::npos; };
  auto fields = [](const std::string &l, std::string &tag, std::vector<long long> &v) -> bool {
    tag.clear();
    v.clear();
    size_t i = 0;
    auto skip = [&] { while (i < l.size() && (l[i] == ' ' || l[i] == '\t')) ++i; };
    skip();
    while (i < l.size() && l[i] != ' ' && l[i] != '\t') tag.push_back(l[i++]);
    for (;;) {
      skip();
      if (i >= l.size()) return true;
      char *end = nullptr;
      const long long x = std::strtoll(l.c_str() + i, &end, 10);
      if (end == l.c_str() + i) return false;  // not a number
      v.push_back(x);
      i = (size_t)(end - l.c_str());
    }
  };
  size_t li = 0;
  while (li < lines.size() && blank(lines[li])) ++li;
  {
    std::string tag;
    std::vector<long long> v;
    if (li >= lines.size() || !fields(lines[li], tag, v) || tag != "filmgrn1" || !v.empty()) {
      err = "missing filmgrn1 header";
      return G1S_ERR_INVALID;
    }
    ++li;
  }
  char msg[160];
  while (li < lines.size()) {
    if (blank(lines[li])) {
      ++li;
      continue;
    }
    std::string tag;
    std::vector<long long> e;
    if (!fields(lines[li], tag, e) || tag != "E") {
      snprintf(msg, sizeof msg, "line %zu: expected an E line", li + 1);
      err = msg;
      return G1S_ERR_INVALID;
    }
    if (e.size() != 5) {
      snprintf(msg, sizeof msg, "line %zu: E line needs 5 fields", li + 1);
      err = msg;
      return G1S_ERR_INVALID;
    }
    if (e[2] != 1 || e[4] != 1) {
      snprintf(msg, sizeof msg, "line %zu: apply_grain/update_parameters must be 1", li + 1);
      err = msg;
      return G1S_ERR_INVALID;
    }
    g1s_segment_t g;
    std::memset(&g, 0, sizeof g);
    g.start_time = (uint64_t)e[0];
    g.end_time = (uint64_t)e[1];
    g.random_seed = (uint16_t)e[3];
    std::vector<long long> body[7];
    bool have[7] = {false, false, false, false, false, false, false};
    static const char *const kTags[7] = {"p", "sY", "sCb", "sCr", "cY", "cCb", "cCr"};
    for (int k = 1; k <= 7; ++k) {
      std::vector<long long> v;
      if (li + k >= lines.size() || blank(lines[li + k]) || !fields(lines[li + k], tag, v)) {
        err = "truncated segment";
        return G1S_ERR_INVALID;
      }
      for (int t = 0; t < 7; ++t) {
        if (tag == kTags[t]) {
          body[t] = v;
          have[t] = true;
        }
      }
    }
    li += 8;
    for (int t = 0; t < 7; ++t) {
      if (!have[t]) {
        snprintf(msg, sizeof msg, "segment starting at %llu: missing %s line", (unsigned long long)g.start_time, kTags[t]);
        err = msg;
        return G1S_ERR_INVALID;
      }
    }
    const std::vector<long long> &p = body[0];
    if (p.size() != 12) {
      err = "p line needs 12 fields";
      return G1S_ERR_INVALID;
    }
    if (p[0] < 0 || p[0] > 3) {
      err = "ar_coeff_lag out of range";
      return G1S_ERR_INVALID;
    }
    g.ar_coeff_lag = (uint8_t)p[0];
    g.ar_coeff_shift = (uint8_t)p[1];
    g.grain_scale_shift = (uint8_t)p[2];
    g.scaling_shift = (uint8_t)p[3];
    g.chroma_scaling_from_luma = p[4] != 0;
    g.overlap_flag = p[5] != 0;
    g.cb_mult = (uint8_t)p[6];
    g.cb_luma_mult = (uint8_t)p[7];
    g.cb_offset = (uint16_t)p[8];
    g.cr_mult = (uint8_t)p[9];
    g.cr_luma_mult = (uint8_t)p[10];
    g.cr_offset = (uint16_t)p[11];
    auto points = [&](int t, size_t cap, uint8_t(*dst)[2], uint8_t &n_out) -> bool {
      const std::vector<long long> &v = body[t];
      if (v.empty() || v[0] < 0 || (size_t)v[0] > cap || v.size() != 1 + 2 * (size_t)v[0]) {
        err = std::string(kTags[t]) + ": bad point count";
        return false;
      }
      n_out = (uint8_t)v[0];
      for (size_t j = 0; j < (size_t)v[0]; ++j) {
        dst[j][0] = (uint8_t)v[1 + 2 * j];
        dst[j][1] = (uint8_t)v[2 + 2 * j];
      }
      return true;
    };
    if (!points(1, G1S_NUM_Y_POINTS, g.scaling_points_y, g.num_y_points) ||
        !points(2, G1S_NUM_UV_POINTS, g.scaling_points_cb, g.num_cb_points) ||
        !points(3, G1S_NUM_UV_POINTS, g.scaling_points_cr, g.num_cr_points))
      return G1S_ERR_INVALID;
    const size_t ncoef = 2 * (size_t)g.ar_coeff_lag * (g.ar_coeff_lag + 1);  // src/parser/grain.rs:40-44
    if (body[4].size() != ncoef || body[5].size() != ncoef + 1 || body[6].size() != ncoef + 1) {
      err = "coefficient count does not match ar_coeff_lag";
      return G1S_ERR_INVALID;
    }
    g.num_y_coeffs = (uint8_t)ncoef;
    g.num_uv_coeffs = (uint8_t)(ncoef + 1);
    for (size_t j = 0; j < ncoef; ++j) g.ar_coeffs_y[j] = (int8_t)body[4][j];
    for (size_t j = 0; j < ncoef + 1; ++j) {
      g.ar_coeffs_cb[j] = (int8_t)body[5][j];
      g.ar_coeffs_cr[j] = (int8_t)body[6][j];
    }
    out.push_back(g);
  }
  return G1S_OK;
}

}  // namespace g1s

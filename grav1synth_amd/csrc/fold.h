// fold.h -- the ORDERED part of DiffGenerator, on the host.
//
// The HIP kernels turn each frame pair into a record of exact integers
// (record.h).  What remains of av1_grain::DiffGenerator::diff_frame /
// ::finish (reference call sites src/main.rs:442 and :524) is sequential in
// frame order and tiny (24/25-dim AR solves, 20-bin strength solves, the
// is-the-noise-different test, quantisation): that is this class.  It is the
// product's own implementation; the test oracle under oracle/ is separate.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/g1s_diff.h"
#include "record.h"

namespace g1s {

constexpr int kNumBins = 20;

// Dense square system A x = b with the elimination order of the reference
// solver (libaom linsolve == av1-grain solver::util::linsolve).
struct LinearSystem {
  int n = 0;
  std::vector<double> A, b, x;
  void resize(int n_);
  void clear();
  void add(const LinearSystem &o);
  void add(const double *oA, const double *ob);  // (the other system where it lies: n x n and n doubles)
  void set_sum(const LinearSystem &a, const LinearSystem &b);
  void set_sum(const LinearSystem &a, const double *A2, const double *b2);
  void assign(const LinearSystem &o);
  bool solve();  // works on copies of A and b; writes x
};

bool gauss_solve(int n, double *A, double *b, double *x);
struct PlaneState;
bool ar_solve(PlaneState &s, bool is_chroma);
void chroma_fallback(PlaneState &s);

struct StrengthSolver {
  LinearSystem eq;
  int64_t num_equations = 0;  // (the reference's is a usize: a combined state passes 2^31 block measurements after ~420 k 4K frames)
  double total = 0.0;
  StrengthSolver();
  void clear();
  void add(const StrengthSolver &o);
  void add(const double *oA, const double *ob, int64_t o_num_equations, double o_total);
  static double bin_index(double value);
  double value_at(double x) const;
  void add_measurement(double block_mean, double noise_std);
  void add_measurements(const double *bin, const uint32_t *sel, const double *noise_std, size_t m);  // (in order; fold.cpp)
  void add_measurements_like(const StrengthSolver &same_bins, const double *bin, const uint32_t *sel, const double *noise_std, size_t m);
  bool solve();
  // The two halves of solve(): the reference's solve() both perturbs b
  // (b += mean/8192, never undone) and computes x.  When x is not needed yet
  // (combined chroma state between segment boundaries) only the perturbation is
  // applied per frame and x is computed later from the same (A, b).
  void apply_regularisation_to_b();
  bool solve_x_only();
  static double center(int i);
  // piecewise-linear simplification -> (x, y) points
  void fit_piecewise(int max_points, std::vector<double> &px, std::vector<double> &py) const;
};

struct PlaneState {
  LinearSystem ar;
  StrengthSolver strength;
  int64_t num_observations = 0;
  double ar_gain = 1.0;
};

// Per-frame ("latest") noise state: depends on that frame's record only, so it
// can be computed for many frames concurrently.
struct FrameLatest {
  PlaneState st[3];
  uint32_t nplanes = 0;
  int status = 0;  // G1S_OK or error code
  std::string err;
  // compute_latest: the frame's flat blocks (raster order), their luma means and bin positions; a plane's measurement arrays
  std::vector<double> scratch_mean, scratch_std, scratch_bin, scratch_plane;
  std::vector<uint32_t> scratch_idx, scratch_pos, scratch_sel, scratch_sel0;
  size_t scratch_m[3] = {0, 0, 0};
};
// Thread-safe: record -> latest state (AR solve, measurements, strength solve).
int compute_latest(const uint8_t *rec, size_t size, uint32_t lag, FrameLatest &out);

// A FrameLatest as a flat, fixed-size blob (what frame-shard ranks exchange instead of the ~10x larger
// records: the per-frame half of the fold runs where the frame was processed, only the ordered half
// runs on rank 0).  Doubles are copied bit for bit.
// (the blob's layout, shared with the device kernel that writes the same bytes: latest.hip)
constexpr uint32_t kLatestMagic = 0x4c315347u;  // "GS1L"
struct LatestHeader {
  uint32_t magic, lag, nplanes;
  int32_t status;
  uint32_t size_bytes, reserved;
  char err[104];
};
struct LatestPlaneHead {
  int64_t num_observations;
  double ar_gain;
  int32_t num_equations, reserved;
  double total;
};
// per plane: head, AR A (nc x nc packed at the front of an ncm x ncm field), b (ncm), x (ncm), strength A, b, x
constexpr size_t plane_blob_bytes(int nc_max) {
  return sizeof(LatestPlaneHead) + sizeof(double) * ((size_t)nc_max * nc_max + 2 * (size_t)nc_max + (size_t)kNumBins * kNumBins + 2 * kNumBins);
}
size_t latest_blob_size(uint32_t lag);
void latest_to_blob(const FrameLatest &fl, uint32_t lag, uint8_t *blob);
int latest_from_blob(const uint8_t *blob, size_t size, uint32_t lag, FrameLatest &out);

// A frame's latest state where it lies (in a FrameLatest, or in a blob as it arrived): what the ordered merge reads.
// The merge reads each of a frame's systems exactly once; copying a blob into a FrameLatest first was a second pass over
// 27 KB per frame, and the larger share of the merge's time.
struct PlaneView {
  const double *A = nullptr, *b = nullptr, *x = nullptr;     // AR system (n x n, n, n)
  const double *sA = nullptr, *sb = nullptr, *sx = nullptr;  // strength system (kNumBins)
  int n = 0;
  int64_t num_observations = 0;
  double ar_gain = 1.0;
  int num_equations = 0;
  double total = 0.0;
};
struct FrameView {
  PlaneView st[3];
  uint32_t nplanes = 0;
  int status = 0;
  const char *err = "";  // (points into the blob / the FrameLatest)
};
void view_of(const FrameLatest &fl, FrameView &out);
int view_of_blob(const uint8_t *blob, size_t size, uint32_t lag, FrameView &out);  // header checked, nothing copied (blob 8-aligned)

class NoiseFold {
 public:
  NoiseFold(int64_t fps_num, int64_t fps_den, uint32_t lag);
  // Consumes one record (frame order!).  Returns G1S_OK or an error code;
  // message in error().
  int push(const uint8_t *rec, size_t size);
  // The sequential half: merge one frame's latest state (frame order!).
  int push_latest(FrameLatest &fl);
  // The same for n frames in order, with the solves of the combined luma systems taken out of the
  // serial chain: as long as no frame starts a new segment, the combined (A, b) after frame j are
  // prefix sums of the frames' systems -- cheap, sequential -- and their solves are independent of
  // each other, so a window of them runs through `pfor` (runs fn(i) for i in [0, n), any order, any
  // threads; empty = serial); the is_different() tests then run in order on the solved states, and
  // a segment cut discards the speculative states behind it.  Every solve sees exactly the operands
  // push_latest() would give it: results are identical bit for bit.
  using ParallelFor = std::function<void(int, const std::function<void(int)> &)>;
  int push_latest_many(const FrameView *fl, size_t n, const ParallelFor &pfor);
  int push_latest_many(FrameLatest *fl, size_t n, const ParallelFor &pfor);
  void finish(std::vector<g1s_segment_t> &out);
  const std::string &error() const { return err_; }
  uint64_t frames() const { return frame_count_; }

 private:
  bool is_different() const;
  static bool differs(const PlaneView &latest, const PlaneState &combined);
  void finalize_chroma() const;
  void save_latest();
  g1s_segment_t grain_parameters(uint64_t start_ts, uint64_t end_ts) const;

  int64_t fps_num_, fps_den_;
  uint32_t lag_;
  int n_;
  PlaneState latest_[3];
  mutable PlaneState combined_[3];
  mutable bool chroma_dirty_ = false;  // combined chroma x not yet solved for the current (A, b)
  uint64_t frame_count_ = 0, prev_timestamp_ = 0;
  std::vector<g1s_segment_t> table_;
  std::string err_;
  std::vector<PlaneState> snap_;  // push_latest_many: speculative combined luma states
  std::vector<uint8_t> snap_ok_;
  PlaneState csum_[3];            // ... the combined chroma states after the window (running sums)
  std::vector<uint8_t> snap_cut_;  // ... whether frame j starts a new segment, given no cut before it
  std::vector<FrameView> views_;
};

long format_tbl(const g1s_segment_t *segs, size_t n, char *buf, size_t cap);
int parse_tbl(const char *text, size_t len, std::vector<g1s_segment_t> &out, std::string &err);

}  // namespace g1s

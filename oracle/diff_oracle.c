/*
 * diff_oracle.c -- CPU ORACLE (test infrastructure; see diff_oracle.h header).
 *
 * PARITY UNPINNED (see diff_oracle.h).  Scalar f64, reference operation order,
 * compile with -ffp-contract=off (no FMA contraction).
 *
 * Every function names the reference call site it serves (paths relative to
 * /root/reference) and the av1-grain 0.4.2 / libaom routine it restates.
 * av1-grain file names (src/diff.rs, src/diff/solver.rs,
 * src/diff/solver/util.rs, src/util.rs) are from memory: that crate is not in
 * the reference tree.
 */
#include "diff_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_SIZE 32
#define BLOCK_SIZE_SQ (BLOCK_SIZE * BLOCK_SIZE)
#define LOW_POLY_NUM_PARAMS 3
#define NUM_BINS 20
#define BLOCK_NORMALIZATION 255.0
#define TINY_NEAR_ZERO 1.0E-16
#define MAX_N 25
/* av1_grain::DEFAULT_GRAIN_SEED, imported at src/parser/frame.rs:3 (value from
 * memory of av1-grain's lib.rs). */
#define DEFAULT_GRAIN_SEED 10956

/* ---- pin-sensitivity variants (tools/pin_sensitivity.py; oracle/Makefile `variants`) ------------------------------------
 * The oracle the tests and goldens use is built with NEITHER macro: every product and every sum is its own rounding, and the
 * normal equations are accumulated per sample.  What cannot be checked here is where av1-grain 0.4.2 writes
 * f64::mul_add -- a Rust port kept clean under clippy's `suboptimal_flops` has one at every a * b + c -- so the study builds
 * this file again with fused multiply-adds at groups of sites and counts what moves in the masks, scores, cuts and tables:
 *   ORC_FMA_SITES  bit mask of the groups below; MA(g, a, b, c) is fma(a, b, c) when g is in it and a * b + c otherwise
 *                  (identical code to the plain expression when the mask is 0: -ffp-contract=off).
 *   ORC_DIVIDE_ONCE  the frame's normal equations as ONE division of the exact integer sums (what the HIP path's fold does,
 *                  grav1synth_amd/csrc/fold.cpp `exact integer sums -> f64 normal equations`) instead of a division a sample. */
#ifndef ORC_FMA_SITES
#define ORC_FMA_SITES 0
#endif
#define G_LINSOLVE 1  /* linsolve: row updates, back substitution */
#define G_MATMUL 2    /* multiply_mat, the finder's AtA */
#define G_FINDER 4    /* gradient covariance sums, var - mean^2, det, disc, the weighted score sum */
#define G_STRENGTH 8  /* strength solver: lerps, the measurement's products, piecewise fit */
#define G_NOISE 16    /* block noise variance, the luma-correlated part, ar_equation_system_solve */
#define G_MODEL 32    /* cross correlation, is_different, the quantisation's weighted means */
#define MA(g, a, b, c) (((ORC_FMA_SITES) & (g)) ? fma((a), (b), (c)) : ((a) * (b) + (c)))

/* ------------------------------------------------------------------------ */
/* libaom aom_dsp/mathutils.h linsolve == av1-grain solver/util.rs linsolve  */
/* ------------------------------------------------------------------------ */
static int linsolve(int n, double *A, int stride, double *b, double *x) {
  int i, j, k;
  double c;
  /* Forward elimination */
  for (k = 0; k < n - 1; k++) {
    /* Bring the largest magnitude to the diagonal position */
    for (i = n - 1; i > k; i--) {
      if (fabs(A[(i - 1) * stride + k]) < fabs(A[i * stride + k])) {
        for (j = 0; j < n; j++) {
          c = A[i * stride + j];
          A[i * stride + j] = A[(i - 1) * stride + j];
          A[(i - 1) * stride + j] = c;
        }
        c = b[i];
        b[i] = b[i - 1];
        b[i - 1] = c;
      }
    }
    for (i = k; i < n - 1; i++) {
      if (fabs(A[k * stride + k]) < TINY_NEAR_ZERO) return 0;
      c = A[(i + 1) * stride + k] / A[k * stride + k];
      for (j = 0; j < n; j++) A[(i + 1) * stride + j] = MA(G_LINSOLVE, -c, A[k * stride + j], A[(i + 1) * stride + j]);
      b[i + 1] = MA(G_LINSOLVE, -c, b[k], b[i + 1]);
    }
  }
  /* Backward substitution */
  for (i = n - 1; i >= 0; i--) {
    if (fabs(A[i * stride + i]) < TINY_NEAR_ZERO) return 0;
    c = 0;
    for (j = i + 1; j <= n - 1; j++) c = MA(G_LINSOLVE, A[i * stride + j], x[j], c);
    x[i] = (b[i] - c) / A[i * stride + i];
  }
  return 1;
}

/* naive triple loop, libaom mathutils.h multiply_mat */
static void multiply_mat(const double *m1, const double *m2, double *res,
                         int m1_rows, int inner_dim, int m2_cols) {
  for (int row = 0; row < m1_rows; ++row) {
    for (int col = 0; col < m2_cols; ++col) {
      double sum = 0;
      for (int inner = 0; inner < inner_dim; ++inner)
        sum = MA(G_MATMUL, m1[row * inner_dim + inner], m2[inner * m2_cols + col], sum);
      *(res++) = sum;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* EquationSystem (libaom aom_equation_system_t)                              */
/* ------------------------------------------------------------------------ */
typedef struct {
  int n;
  double A[MAX_N * MAX_N];
  double b[MAX_N];
  double x[MAX_N];
} eqsys;

static void eq_init(eqsys *e, int n) {
  memset(e, 0, sizeof(*e));
  e->n = n;
}
static void eq_clear(eqsys *e) {
  int n = e->n;
  memset(e, 0, sizeof(*e));
  e->n = n;
}
static int eq_solve(eqsys *e) {
  double A[MAX_N * MAX_N], b[MAX_N];
  const int n = e->n;
  memcpy(A, e->A, sizeof(double) * n * n);
  memcpy(b, e->b, sizeof(double) * n);
  return linsolve(n, A, n, b, e->x);
}
static void eq_add(eqsys *dst, const eqsys *src) {
  const int n = dst->n;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) dst->A[i * n + j] += src->A[i * n + j];
    dst->b[i] += src->b[i];
  }
}
static void eq_copy(eqsys *dst, const eqsys *src) {
  const int n = dst->n;
  memcpy(dst->A, src->A, sizeof(double) * n * n);
  memcpy(dst->x, src->x, sizeof(double) * n);
  memcpy(dst->b, src->b, sizeof(double) * n);
}

/* ------------------------------------------------------------------------ */
/* NoiseStrengthSolver (libaom aom_noise_strength_solver_t), 8-bit domain    */
/* ------------------------------------------------------------------------ */
typedef struct {
  eqsys eqns;
  int num_bins;
  long long num_equations; /* (usize in the reference) */
  double total;
} strength_solver;

static void ss_init(strength_solver *s) {
  eq_init(&s->eqns, NUM_BINS);
  s->num_bins = NUM_BINS;
  s->num_equations = 0;
  s->total = 0;
}
static void ss_clear(strength_solver *s) {
  eq_clear(&s->eqns);
  s->num_equations = 0;
  s->total = 0;
}
static void ss_add(strength_solver *dst, const strength_solver *src) {
  eq_add(&dst->eqns, &src->eqns);
  dst->num_equations += src->num_equations;
  dst->total += src->total;
}
static double fclamp(double v, double lo, double hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}
static double ss_bin_index(const strength_solver *s, double value) {
  const double val = fclamp(value, 0.0, 255.0);
  const double range = 255.0 - 0.0;
  return (s->num_bins - 1) * (val - 0.0) / range;
}
static double ss_get_value(const strength_solver *s, double x) {
  const double bin = ss_bin_index(s, x);
  const int bin_i0 = (int)floor(bin);
  const int bin_i1 = (s->num_bins - 1) < (bin_i0 + 1) ? (s->num_bins - 1) : (bin_i0 + 1);
  const double a = bin - bin_i0;
  return MA(G_STRENGTH, 1.0 - a, s->eqns.x[bin_i0], a * s->eqns.x[bin_i1]);
}
static void ss_add_measurement(strength_solver *s, double block_mean, double noise_std) {
  const double bin = ss_bin_index(s, block_mean);
  const int bin_i0 = (int)floor(bin);
  const int bin_i1 = (s->num_bins - 1) < (bin_i0 + 1) ? (s->num_bins - 1) : (bin_i0 + 1);
  const double a = bin - bin_i0;
  const int n = s->num_bins;
  s->eqns.A[bin_i0 * n + bin_i0] = MA(G_STRENGTH, 1.0 - a, 1.0 - a, s->eqns.A[bin_i0 * n + bin_i0]);
  s->eqns.A[bin_i1 * n + bin_i0] = MA(G_STRENGTH, a, 1.0 - a, s->eqns.A[bin_i1 * n + bin_i0]);
  s->eqns.A[bin_i1 * n + bin_i1] = MA(G_STRENGTH, a, a, s->eqns.A[bin_i1 * n + bin_i1]);
  s->eqns.A[bin_i0 * n + bin_i1] = MA(G_STRENGTH, a, 1.0 - a, s->eqns.A[bin_i0 * n + bin_i1]);
  s->eqns.b[bin_i0] = MA(G_STRENGTH, 1.0 - a, noise_std, s->eqns.b[bin_i0]);
  s->eqns.b[bin_i1] = MA(G_STRENGTH, a, noise_std, s->eqns.b[bin_i1]);
  s->total += noise_std;
  s->num_equations++;
}
/* libaom aom_noise_strength_solver_solve: regularised solve; A is restored,
 * b is NOT (the mean/8192 perturbation persists across calls). */
static int ss_solve(strength_solver *s) {
  const int n = s->num_bins;
  const double kAlpha = 2.0 * (double)(s->num_equations) / n;
  double oldA[NUM_BINS * NUM_BINS];
  memcpy(oldA, s->eqns.A, sizeof(oldA));
  double *A = s->eqns.A;
  for (int i = 0; i < n; ++i) {
    const int i_lo = (i - 1) > 0 ? (i - 1) : 0;
    const int i_hi = (n - 1) < (i + 1) ? (n - 1) : (i + 1);
    A[i * n + i_lo] -= kAlpha;
    A[i * n + i] += 2 * kAlpha;
    A[i * n + i_hi] -= kAlpha;
  }
  /* Small regularization to give average noise strength */
  const double mean = s->total / s->num_equations;
  for (int i = 0; i < n; ++i) {
    A[i * n + i] += 1.0 / 8192.;
    s->eqns.b[i] += mean / 8192.;
  }
  const int result = eq_solve(&s->eqns);
  memcpy(s->eqns.A, oldA, sizeof(oldA));
  return result;
}
static double ss_get_center(const strength_solver *s, int i) {
  const double range = 255.0 - 0.0;
  const int n = s->num_bins;
  return ((double)i) / (n - 1) * range + 0.0;
}

typedef struct {
  double points[NUM_BINS][2];
  int num_points;
} strength_lut;

static void update_piecewise_linear_residual(const strength_solver *s,
                                             const strength_lut *lut,
                                             double *residual, int start, int end) {
  const double dx = 255. / s->num_bins;
  const int i_begin = start > 1 ? start : 1;
  const int i_end = end < (lut->num_points - 1) ? end : (lut->num_points - 1);
  for (int i = i_begin; i < i_end; ++i) {
    int lower = (int)floor(ss_bin_index(s, lut->points[i - 1][0]));
    if (lower < 0) lower = 0;
    int upper = (int)ceil(ss_bin_index(s, lut->points[i + 1][0]));
    if (upper > s->num_bins - 1) upper = s->num_bins - 1;
    double r = 0;
    for (int j = lower; j <= upper; ++j) {
      const double x = ss_get_center(s, j);
      if (x < lut->points[i - 1][0]) continue;
      if (x >= lut->points[i + 1][0]) continue;
      const double y = s->eqns.x[j];
      const double a = (x - lut->points[i - 1][0]) /
                       (lut->points[i + 1][0] - lut->points[i - 1][0]);
      const double estimate_y = MA(G_STRENGTH, lut->points[i - 1][1], 1.0 - a, lut->points[i + 1][1] * a);
      r += fabs(y - estimate_y);
    }
    residual[i] = r * dx;
  }
}

/* libaom aom_noise_strength_solver_fit_piecewise */
static void ss_fit_piecewise(const strength_solver *s, int max_output_points,
                             strength_lut *lut) {
  const double kTolerance = 255.0 * 0.00625 / 255.0;
  lut->num_points = s->num_bins;
  for (int i = 0; i < s->num_bins; ++i) {
    lut->points[i][0] = ss_get_center(s, i);
    lut->points[i][1] = s->eqns.x[i];
  }
  if (max_output_points < 0) max_output_points = s->num_bins;
  double residual[NUM_BINS];
  memset(residual, 0, sizeof(residual));
  update_piecewise_linear_residual(s, lut, residual, 0, s->num_bins);
  /* Greedily remove points if there are too many or if it doesn't hurt local
   * approximation (never remove the end points) */
  while (lut->num_points > 2) {
    int min_index = 1;
    for (int j = 1; j < lut->num_points - 1; ++j) {
      if (residual[j] < residual[min_index]) min_index = j;
    }
    const double dx = lut->points[min_index + 1][0] - lut->points[min_index - 1][0];
    const double avg_residual = residual[min_index] / dx;
    if (lut->num_points <= max_output_points && avg_residual > kTolerance) break;
    const int num_remaining = lut->num_points - min_index - 1;
    memmove(lut->points + min_index, lut->points + min_index + 1,
            sizeof(lut->points[0]) * num_remaining);
    memmove(residual + min_index, residual + min_index + 1,
            sizeof(residual[0]) * num_remaining);
    lut->num_points--;
    update_piecewise_linear_residual(s, lut, residual, min_index - 1, min_index + 1);
  }
}

/* ------------------------------------------------------------------------ */
/* FlatBlockFinder (libaom aom_flat_block_finder_*)                           */
/* ------------------------------------------------------------------------ */
typedef struct {
  double A[LOW_POLY_NUM_PARAMS * BLOCK_SIZE_SQ];
  double AtA_inv[LOW_POLY_NUM_PARAMS * LOW_POLY_NUM_PARAMS];
} flat_finder;

static void ff_init(flat_finder *f) {
  eqsys eqns;
  eq_init(&eqns, LOW_POLY_NUM_PARAMS);
  const int n = LOW_POLY_NUM_PARAMS;
  for (int y = 0; y < BLOCK_SIZE; ++y) {
    const double yd = ((double)y - BLOCK_SIZE / 2.) / (BLOCK_SIZE / 2.);
    for (int x = 0; x < BLOCK_SIZE; ++x) {
      const double xd = ((double)x - BLOCK_SIZE / 2.) / (BLOCK_SIZE / 2.);
      const double coords[3] = { yd, xd, 1 };
      const int row = y * BLOCK_SIZE + x;
      f->A[n * row + 0] = yd;
      f->A[n * row + 1] = xd;
      f->A[n * row + 2] = 1;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) eqns.A[n * i + j] = MA(G_MATMUL, coords[i], coords[j], eqns.A[n * i + j]);
    }
  }
  /* Lazy inverse using existing equation solver. */
  for (int i = 0; i < n; ++i) {
    memset(eqns.b, 0, sizeof(double) * n);
    eqns.b[i] = 1;
    eq_solve(&eqns);
    for (int j = 0; j < n; ++j) f->AtA_inv[j * n + i] = eqns.x[j];
  }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void ff_extract_block(const flat_finder *f, const uint8_t *data, int w,
                             int h, int stride, int offsx, int offsy,
                             double *plane, double *block) {
  double plane_coords[LOW_POLY_NUM_PARAMS];
  double AtA_inv_b[LOW_POLY_NUM_PARAMS];
  for (int yi = 0; yi < BLOCK_SIZE; ++yi) {
    const int y = clampi(offsy + yi, 0, h - 1);
    for (int xi = 0; xi < BLOCK_SIZE; ++xi) {
      const int x = clampi(offsx + xi, 0, w - 1);
      block[yi * BLOCK_SIZE + xi] = ((double)data[y * stride + x]) / BLOCK_NORMALIZATION;
    }
  }
  multiply_mat(block, f->A, AtA_inv_b, 1, BLOCK_SIZE_SQ, LOW_POLY_NUM_PARAMS);
  multiply_mat(f->AtA_inv, AtA_inv_b, plane_coords, LOW_POLY_NUM_PARAMS,
               LOW_POLY_NUM_PARAMS, 1);
  multiply_mat(f->A, plane_coords, plane, BLOCK_SIZE_SQ, LOW_POLY_NUM_PARAMS, 1);
  for (int i = 0; i < BLOCK_SIZE_SQ; ++i) block[i] -= plane[i];
}

typedef struct {
  float score;
  int index;
} index_and_score;

static int compare_scores(const void *a, const void *b) {
  const float diff = ((const index_and_score *)a)->score - ((const index_and_score *)b)->score;
  if (diff < 0) return -1;
  if (diff > 0) return 1;
  return 0;
}

/* libaom aom_flat_block_finder_run (8-bit).  Returns num_flat. */
static int ff_run(const flat_finder *f, const uint8_t *data, int w, int h,
                  int stride, uint8_t *flat_blocks, float *scores_out) {
  const double kTraceThreshold = 0.15 / BLOCK_SIZE_SQ;
  const double kRatioThreshold = 1.25;
  const double kNormThreshold = 0.08 / BLOCK_SIZE_SQ;
  const double kVarThreshold = 0.005 / BLOCK_SIZE_SQ;
  const int num_blocks_w = (w + BLOCK_SIZE - 1) / BLOCK_SIZE;
  const int num_blocks_h = (h + BLOCK_SIZE - 1) / BLOCK_SIZE;
  const int num_blocks = num_blocks_w * num_blocks_h;
  int num_flat = 0;
  double plane[BLOCK_SIZE_SQ], block[BLOCK_SIZE_SQ];
  index_and_score *scores = (index_and_score *)malloc(sizeof(*scores) * num_blocks);
  const double norm_factor = (BLOCK_SIZE - 2) * (BLOCK_SIZE - 2);

  for (int by = 0; by < num_blocks_h; ++by) {
    for (int bx = 0; bx < num_blocks_w; ++bx) {
      /* Compute gradient covariance matrix. */
      double Gxx = 0, Gxy = 0, Gyy = 0, var = 0, mean = 0;
      ff_extract_block(f, data, w, h, stride, bx * BLOCK_SIZE, by * BLOCK_SIZE, plane, block);
      for (int yi = 1; yi < BLOCK_SIZE - 1; ++yi) {
        for (int xi = 1; xi < BLOCK_SIZE - 1; ++xi) {
          const double gx = (block[yi * BLOCK_SIZE + xi + 1] - block[yi * BLOCK_SIZE + xi - 1]) / 2;
          const double gy = (block[yi * BLOCK_SIZE + xi + BLOCK_SIZE] - block[yi * BLOCK_SIZE + xi - BLOCK_SIZE]) / 2;
          Gxx = MA(G_FINDER, gx, gx, Gxx);
          Gxy = MA(G_FINDER, gx, gy, Gxy);
          Gyy = MA(G_FINDER, gy, gy, Gyy);
          mean += block[yi * BLOCK_SIZE + xi];
          var = MA(G_FINDER, block[yi * BLOCK_SIZE + xi], block[yi * BLOCK_SIZE + xi], var);
        }
      }
      mean /= norm_factor;
      /* Normalize gradients by block_size. */
      Gxx /= norm_factor;
      Gxy /= norm_factor;
      Gyy /= norm_factor;
      var = MA(G_FINDER, -mean, mean, var / norm_factor);
      {
        const double trace = Gxx + Gyy;
        const double det = MA(G_FINDER, Gxx, Gyy, -(Gxy * Gxy));
        /* av1-grain guards the discriminant with max(.,0) (libaom does not);
         * from memory, see SURVEY.md Appendix A.4. */
        double disc = MA(G_FINDER, trace, trace, -(4 * det));
        if (!(disc > 0.0)) disc = 0.0;
        const double e1 = (trace + sqrt(disc)) / 2.;
        const double e2 = (trace - sqrt(disc)) / 2.;
        const double norm = e1; /* spectral norm */
        const double ratio = e1 / (e2 > 1e-6 ? e2 : 1e-6);
        const int is_flat = (trace < kTraceThreshold) && (ratio < kRatioThreshold) &&
                            (norm < kNormThreshold) && (var > kVarThreshold);
        /* weights: [{var}, {ratio}, {trace}, {norm}, offset] */
        const double weights[5] = { -6682, -0.2056, 13087, -12434, 2.5694 };
        double sum_weights =
            MA(G_FINDER, weights[3], norm, MA(G_FINDER, weights[2], trace, MA(G_FINDER, weights[1], ratio, weights[0] * var))) + weights[4];
        /* clamp the value to [-25.0, 100.0] to prevent overflow */
        sum_weights = fclamp(sum_weights, -25.0, 100.0);
        const float score = (float)(1.0 / (1 + exp(-sum_weights)));
        flat_blocks[by * num_blocks_w + bx] = is_flat ? 255 : 0;
        scores[by * num_blocks_w + bx].score = var > kVarThreshold ? score : 0;
        scores[by * num_blocks_w + bx].index = by * num_blocks_w + bx;
        if (scores_out) scores_out[by * num_blocks_w + bx] = scores[by * num_blocks_w + bx].score;
        num_flat += is_flat;
      }
    }
  }
  qsort(scores, num_blocks, sizeof(*scores), &compare_scores);
  /* union of the thresholded results and the top 10th percentile of scores */
  const int top_nth_percentile = num_blocks * 90 / 100;
  const float score_threshold = scores[top_nth_percentile].score;
  for (int i = 0; i < num_blocks; ++i) {
    if (scores[i].score >= score_threshold) {
      num_flat += flat_blocks[scores[i].index] == 0;
      flat_blocks[scores[i].index] |= 1;
    }
  }
  free(scores);
  return num_flat;
}

/* ------------------------------------------------------------------------ */
/* NoiseModel (libaom aom_noise_model_t)                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
  eqsys eqns;
  strength_solver strength;
  int num_observations;
  double ar_gain;
} noise_state;

typedef struct {
  int lag;
  int n; /* number of AR coords (luma) */
  int coords[MAX_N][2];
  noise_state combined[3];
  noise_state latest[3];
} noise_model;

static void ns_init(noise_state *s, int n) {
  eq_init(&s->eqns, n);
  s->ar_gain = 1.0;
  s->num_observations = 0;
  ss_init(&s->strength);
}

static void nm_init(noise_model *m, int lag) {
  memset(m, 0, sizeof(*m));
  m->lag = lag;
  const int side = 2 * lag + 1;
  m->n = (side * side) / 2;
  for (int c = 0; c < 3; ++c) {
    ns_init(&m->combined[c], m->n + (c > 0));
    ns_init(&m->latest[c], m->n + (c > 0));
  }
  int i = 0;
  for (int y = -lag; y <= 0; ++y) {
    const int max_x = y == 0 ? -1 : lag;
    for (int x = -lag; x <= max_x; ++x) {
      m->coords[i][0] = x;
      m->coords[i][1] = y;
      ++i;
    }
  }
}

static void set_chroma_coefficient_fallback_soln(eqsys *e) {
  const double kTolerance = 1e-6;
  const int last = e->n - 1;
  memset(e->x, 0, sizeof(double) * e->n);
  if (fabs(e->A[last * e->n + last]) > kTolerance)
    e->x[last] = e->b[last] / e->A[last * e->n + last];
}

static int ar_equation_system_solve(noise_state *s, int is_chroma) {
  const int ret = eq_solve(&s->eqns);
  s->ar_gain = 1.0;
  if (!ret) return ret;
  double var = 0;
  const int n = s->eqns.n;
  for (int i = 0; i < (n - is_chroma); ++i) var += s->eqns.A[i * n + i] / s->num_observations;
  var /= (n - is_chroma);
  double sum_covar = 0;
  for (int i = 0; i < n - is_chroma; ++i) {
    double bi = s->eqns.b[i];
    if (is_chroma) bi = MA(G_NOISE, -s->eqns.A[i * n + (n - 1)], s->eqns.x[n - 1], bi);
    sum_covar += (bi * s->eqns.x[i]) / s->num_observations;
  }
  const double t = var - sum_covar;
  const double noise_var = t > 1e-6 ? t : 1e-6;
  const double q = var / noise_var;
  const double g = sqrt(q > 1e-6 ? q : 1e-6);
  s->ar_gain = 1 > g ? 1 : g;
  return ret;
}

typedef struct {
  const uint8_t *data[3], *den[3];
  int stride[3];
  int w, h; /* luma */
  int sub[2]; /* chroma sub_log2 {x,y} */
} planes8;

/* shadow integer records of the last frame (test introspection only) */
typedef struct {
  int64_t S[3][MAX_N * MAX_N];
  int64_t Sb[3][MAX_N];
  int64_t nobs[3];
  uint32_t *luma_sum;    /* [nblocks] */
  int32_t *sum_d[3];     /* [nblocks] */
  uint32_t *sum_d2[3];   /* [nblocks] */
} shadow_rec;

/* libaom add_block_observations + extract_ar_row_lowbd */
static void add_block_observations(noise_model *m, int c, const planes8 *P,
                                   const uint8_t *flat_blocks, int nbw, int nbh,
                                   shadow_rec *sh) {
  const int lag = m->lag;
  const int num_coords = m->n;
  const double normalization = BLOCK_NORMALIZATION;
  noise_state *st = &m->latest[c];
  double *A = st->eqns.A;
  double *b = st->eqns.b;
  const int n = st->eqns.n;
  const int sx = c > 0 ? P->sub[0] : 0, sy = c > 0 ? P->sub[1] : 0;
  const uint8_t *data = P->data[c], *den = P->den[c];
  const int stride = P->stride[c];
  const uint8_t *alt = c > 0 ? P->data[0] : NULL, *alt_den = c > 0 ? P->den[0] : NULL;
  const int alt_stride = P->stride[0];
  const int w = P->w, h = P->h;
  const int bw = BLOCK_SIZE >> sx, bh = BLOCK_SIZE >> sy;
  const int ns = (1 << sx) * (1 << sy);
  double buffer[MAX_N + 1];
  int64_t ibuf[MAX_N + 1];
  for (int by = 0; by < nbh; ++by) {
    const int y_o = by * bh;
    for (int bx = 0; bx < nbw; ++bx) {
      const int x_o = bx * bw;
      if (!flat_blocks[by * nbw + bx]) continue;
      const int y_start = (by > 0 && flat_blocks[(by - 1) * nbw + bx]) ? 0 : lag;
      const int x_start = (bx > 0 && flat_blocks[by * nbw + bx - 1]) ? 0 : lag;
      int y_end = (h >> sy) - by * bh;
      if (y_end > bh) y_end = bh;
      int x_end_a = (w >> sx) - bx * bw - lag;
      int x_end_b = (bx + 1 < nbw && flat_blocks[by * nbw + bx + 1]) ? bw : (bw - lag);
      const int x_end = x_end_a < x_end_b ? x_end_a : x_end_b;
      for (int y = y_start; y < y_end; ++y) {
        for (int x = x_start; x < x_end; ++x) {
          const int X = x + x_o, Y = y + y_o;
          for (int i = 0; i < num_coords; ++i) {
            const int x_i = X + m->coords[i][0], y_i = Y + m->coords[i][1];
            const int d = (int)data[y_i * stride + x_i] - (int)den[y_i * stride + x_i];
            buffer[i] = (double)d;
            ibuf[i] = d;
          }
          const int vd = (int)data[Y * stride + X] - (int)den[Y * stride + X];
          const double val = (double)vd;
          if (alt && alt_den) {
            double avg_data = 0, avg_denoised = 0;
            int num_samples = 0;
            int64_t isum = 0;
            for (int dy_i = 0; dy_i < (1 << sy); dy_i++) {
              const int y_up = (Y << sy) + dy_i;
              for (int dx_i = 0; dx_i < (1 << sx); dx_i++) {
                const int x_up = (X << sx) + dx_i;
                avg_data += alt[y_up * alt_stride + x_up];
                avg_denoised += alt_den[y_up * alt_stride + x_up];
                isum += (int)alt[y_up * alt_stride + x_up] - (int)alt_den[y_up * alt_stride + x_up];
                num_samples++;
              }
            }
            buffer[num_coords] = (avg_data - avg_denoised) / num_samples;
            ibuf[num_coords] = isum; /* == ns * buffer[num_coords] */
          }
          for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) {
              A[i * n + j] += (buffer[i] * buffer[j]) / (normalization * normalization);
            }
            b[i] += (buffer[i] * val) / (normalization * normalization);
          }
          st->num_observations++;
          if (sh) {
            for (int i = 0; i < n; ++i) {
              for (int j = 0; j < n; ++j) sh->S[c][i * n + j] += ibuf[i] * ibuf[j];
              sh->Sb[c][i] += ibuf[i] * vd;
            }
            sh->nobs[c]++;
          }
        }
      }
    }
  }
#ifdef ORC_DIVIDE_ONCE
  /* the frame's system from the exact integer sums, one division an entry (the chroma regressor's sum is ns x the average the
   * per-sample form uses): needs the integer shadow, which every caller of this file passes */
  if (sh) {
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) {
        double den = normalization * normalization;
        if (c > 0 && i == n - 1) den *= ns;
        if (c > 0 && j == n - 1) den *= ns;
        A[i * n + j] = (double)sh->S[c][i * n + j] / den;
      }
      double den = normalization * normalization;
      if (c > 0 && i == n - 1) den *= ns;
      b[i] = (double)sh->Sb[c][i] / den;
    }
  }
#endif
  (void)ns;
}

static double get_block_mean(const uint8_t *data, int w, int h, int stride,
                             int x_o, int y_o, uint32_t *isum) {
  const int max_h = (h - y_o) < BLOCK_SIZE ? (h - y_o) : BLOCK_SIZE;
  const int max_w = (w - x_o) < BLOCK_SIZE ? (w - x_o) : BLOCK_SIZE;
  double block_mean = 0;
  for (int y = 0; y < max_h; ++y)
    for (int x = 0; x < max_w; ++x) block_mean += data[(y_o + y) * stride + x_o + x];
  if (isum) *isum = (uint32_t)block_mean;
  return block_mean / (max_w * max_h);
}

static double get_noise_var(const uint8_t *data, const uint8_t *denoised, int stride,
                            int w, int h, int x_o, int y_o, int block_size_x,
                            int block_size_y, int32_t *isd, uint32_t *isd2) {
  const int max_h = (h - y_o) < block_size_y ? (h - y_o) : block_size_y;
  const int max_w = (w - x_o) < block_size_x ? (w - x_o) : block_size_x;
  double noise_var = 0, noise_mean = 0;
  for (int y = 0; y < max_h; ++y) {
    for (int x = 0; x < max_w; ++x) {
      double noise = (double)data[(y_o + y) * stride + x_o + x] - denoised[(y_o + y) * stride + x_o + x];
      noise_mean += noise;
      noise_var = MA(G_NOISE, noise, noise, noise_var);
    }
  }
  if (isd) *isd = (int32_t)noise_mean;
  if (isd2) *isd2 = (uint32_t)noise_var;
  noise_mean /= (max_w * max_h);
  return MA(G_NOISE, -noise_mean, noise_mean, noise_var / (max_w * max_h));
}

/* libaom add_noise_std_observations */
static void add_noise_std_observations(noise_model *m, int c, const double *coeffs,
                                       const planes8 *P, const uint8_t *flat_blocks,
                                       int nbw, int nbh, shadow_rec *sh) {
  const int num_coords = m->n;
  strength_solver *solver = &m->latest[c].strength;
  const strength_solver *luma_solver = &m->latest[0].strength;
  const double luma_gain = m->latest[0].ar_gain;
  const double noise_gain = m->latest[c].ar_gain;
  const int sx = c > 0 ? P->sub[0] : 0, sy = c > 0 ? P->sub[1] : 0;
  const int bw = BLOCK_SIZE >> sx, bh = BLOCK_SIZE >> sy;
  const int w = P->w, h = P->h;
  for (int by = 0; by < nbh; ++by) {
    const int y_o = by * bh;
    for (int bx = 0; bx < nbw; ++bx) {
      const int x_o = bx * bw;
      if (!flat_blocks[by * nbw + bx]) continue;
      int num_samples_h = (h >> sy) - by * bh;
      if (num_samples_h > bh) num_samples_h = bh;
      int num_samples_w = (w >> sx) - bx * bw;
      if (num_samples_w > bw) num_samples_w = bw;
      /* Make sure that we have a reasonable amount of samples */
      if (num_samples_w * num_samples_h > BLOCK_SIZE) {
        uint32_t ls;
        int32_t sd;
        uint32_t sd2;
        const double block_mean = get_block_mean(P->data[0], w, h, P->stride[0],
                                                 x_o << sx, y_o << sy, &ls);
        const double noise_var = get_noise_var(P->data[c], P->den[c], P->stride[c],
                                               w >> sx, h >> sy, x_o, y_o, bw, bh, &sd, &sd2);
        if (sh) {
          sh->luma_sum[by * nbw + bx] = ls;
          sh->sum_d[c][by * nbw + bx] = sd;
          sh->sum_d2[c][by * nbw + bx] = sd2;
        }
        /* remove the part of the noise correlated with luma */
        const double luma_strength = c > 0 ? luma_gain * ss_get_value(luma_solver, block_mean) : 0;
        const double corr = c > 0 ? coeffs[num_coords] : 0;
        /* don't allow fully correlated noise (hence the max) */
        const double t0 = noise_var / 16;
        const double cl = corr * luma_strength; /* Rust powi(2) == x*x */
        const double t1 = MA(G_NOISE, -cl, cl, noise_var);
        const double uncorr_std = sqrt(t0 > t1 ? t0 : t1);
        /* undo the gain of the IIR filter */
        const double adjusted_strength = uncorr_std / noise_gain;
        ss_add_measurement(solver, block_mean, adjusted_strength);
      }
    }
  }
}

static double normalized_cross_correlation(const double *a, const double *b, int n) {
  double c = 0, a_len = 0, b_len = 0;
  for (int i = 0; i < n; ++i) {
    a_len = MA(G_MODEL, a[i], a[i], a_len);
    b_len = MA(G_MODEL, b[i], b[i], b_len);
    c = MA(G_MODEL, a[i], b[i], c);
  }
  return c / (sqrt(a_len) * sqrt(b_len));
}

static int is_noise_model_different(const noise_model *m) {
  const double kCoeffThreshold = 0.9;
  const double kStrengthThreshold = 0.005;
  const double corr = normalized_cross_correlation(m->latest[0].eqns.x, m->combined[0].eqns.x,
                                                   m->combined[0].eqns.n);
  if (corr < kCoeffThreshold) return 1;
  const double dx = 1.0 / m->latest[0].strength.num_bins;
  const eqsys *le = &m->latest[0].strength.eqns;
  const eqsys *ce = &m->combined[0].strength.eqns;
  double diff = 0, total_weight = 0;
  for (int j = 0; j < le->n; ++j) {
    double weight = 0;
    for (int i = 0; i < le->n; ++i) weight += le->A[i * le->n + j];
    weight = sqrt(weight);
    diff = MA(G_MODEL, weight, fabs(le->x[j] - ce->x[j]), diff);
    total_weight += weight;
  }
  if (diff * dx / total_weight > kStrengthThreshold) return 1;
  return 0;
}

enum { NOISE_OK = 0, NOISE_DIFFERENT = 1, NOISE_ERROR = -1 };

/* libaom aom_noise_model_update */
static int nm_update(noise_model *m, const planes8 *P, int nplanes,
                     const uint8_t *flat_blocks, shadow_rec *sh, char *err, size_t errcap) {
  const int nbw = (P->w + BLOCK_SIZE - 1) / BLOCK_SIZE;
  const int nbh = (P->h + BLOCK_SIZE - 1) / BLOCK_SIZE;
  int y_model_different = 0;
  int num_blocks = 0;
  for (int i = 0; i < 3; ++i) {
    eq_clear(&m->latest[i].eqns);
    m->latest[i].num_observations = 0;
    ss_clear(&m->latest[i].strength);
  }
  for (int i = 0; i < nbw * nbh; ++i)
    if (flat_blocks[i]) num_blocks++;
  if (num_blocks <= 1) {
    snprintf(err, errcap, "Not enough flat blocks to update noise estimate");
    return NOISE_ERROR;
  }
  for (int channel = 0; channel < 3; ++channel) {
    const int is_chroma = channel != 0;
    if (channel >= nplanes || !P->data[channel] || !P->den[channel]) break;
    add_block_observations(m, channel, P, flat_blocks, nbw, nbh, sh);
    if (!ar_equation_system_solve(&m->latest[channel], is_chroma)) {
      if (is_chroma) {
        set_chroma_coefficient_fallback_soln(&m->latest[channel].eqns);
      } else {
        snprintf(err, errcap, "Solving latest noise equation system failed %d!", channel);
        return NOISE_ERROR;
      }
    }
    add_noise_std_observations(m, channel, m->latest[channel].eqns.x, P, flat_blocks, nbw, nbh, sh);
    if (!ss_solve(&m->latest[channel].strength)) {
      snprintf(err, errcap, "Solving latest noise strength failed!");
      return NOISE_ERROR;
    }
    /* Check noise characteristics and return if error. */
    if (channel == 0 && m->combined[channel].strength.num_equations > 0 &&
        is_noise_model_different(m)) {
      y_model_different = 1;
    }
    /* Don't update the combined stats if the y model is different. */
    if (y_model_different) continue;
    m->combined[channel].num_observations += m->latest[channel].num_observations;
    eq_add(&m->combined[channel].eqns, &m->latest[channel].eqns);
    if (!ar_equation_system_solve(&m->combined[channel], is_chroma)) {
      if (is_chroma) {
        set_chroma_coefficient_fallback_soln(&m->combined[channel].eqns);
      } else {
        snprintf(err, errcap, "Solving combined noise equation system failed %d!", channel);
        return NOISE_ERROR;
      }
    }
    ss_add(&m->combined[channel].strength, &m->latest[channel].strength);
    if (!ss_solve(&m->combined[channel].strength)) {
      snprintf(err, errcap, "Solving combined noise strength failed!");
      return NOISE_ERROR;
    }
  }
  return y_model_different ? NOISE_DIFFERENT : NOISE_OK;
}

static void nm_save_latest(noise_model *m) {
  for (int c = 0; c < 3; c++) {
    eq_copy(&m->combined[c].eqns, &m->latest[c].eqns);
    eq_copy(&m->combined[c].strength.eqns, &m->latest[c].strength.eqns);
    m->combined[c].strength.num_equations = m->latest[c].strength.num_equations;
    m->combined[c].num_observations = m->latest[c].num_observations;
    m->combined[c].ar_gain = m->latest[c].ar_gain;
  }
}

/* libaom aom_noise_model_get_grain_parameters -> av1_grain::GrainTableSegment */
static void nm_get_grain_parameters(const noise_model *m, uint64_t start_ts,
                                    uint64_t end_ts, orc_segment *g) {
  memset(g, 0, sizeof(*g));
  g->random_seed = start_ts == 0 ? DEFAULT_GRAIN_SEED : 0;
  g->start_time = start_ts;
  g->end_time = end_ts;
  g->ar_coeff_lag = (uint8_t)m->lag;

  strength_lut scaling_points[3];
  ss_fit_piecewise(&m->combined[0].strength, ORC_MAX_Y_POINTS, &scaling_points[0]);
  ss_fit_piecewise(&m->combined[1].strength, ORC_MAX_UV_POINTS, &scaling_points[1]);
  ss_fit_piecewise(&m->combined[2].strength, ORC_MAX_UV_POINTS, &scaling_points[2]);

  /* bit depth is 8 here: strength_divisor == 1 */
  double max_scaling_value = 1e-4;
  for (int c = 0; c < 3; ++c) {
    for (int i = 0; i < scaling_points[c].num_points; ++i) {
      if (scaling_points[c].points[i][0] > 255) scaling_points[c].points[i][0] = 255;
      if (scaling_points[c].points[i][1] > 255) scaling_points[c].points[i][1] = 255;
      if (scaling_points[c].points[i][1] > max_scaling_value)
        max_scaling_value = scaling_points[c].points[i][1];
    }
  }
  /* Scaling_shift values are in the range [8,11] */
  const int max_scaling_value_log2 = clampi((int)floor(log2(max_scaling_value) + 1), 2, 5);
  g->scaling_shift = (uint8_t)(5 + (8 - max_scaling_value_log2));
  const double scale_factor = 1 << (8 - max_scaling_value_log2);
  g->num_y_points = (uint8_t)scaling_points[0].num_points;
  g->num_cb_points = (uint8_t)scaling_points[1].num_points;
  g->num_cr_points = (uint8_t)scaling_points[2].num_points;
  uint8_t(*dst[3])[2] = { g->scaling_points_y, g->scaling_points_cb, g->scaling_points_cr };
  for (int c = 0; c < 3; c++) {
    for (int i = 0; i < scaling_points[c].num_points; ++i) {
      dst[c][i][0] = (uint8_t)clampi((int)(scaling_points[c].points[i][0] + 0.5), 0, 255);
      dst[c][i][1] = (uint8_t)clampi((int)MA(G_MODEL, scale_factor, scaling_points[c].points[i][1], 0.5), 0, 255);
    }
  }

  /* Convert the ar_coeffs into 8-bit values */
  const int n_coeff = m->combined[0].eqns.n;
  double max_coeff = 1e-4, min_coeff = -1e-4;
  double y_corr[2] = { 0, 0 };
  double avg_luma_strength = 0;
  for (int c = 0; c < 3; c++) {
    const eqsys *eqns = &m->combined[c].eqns;
    for (int i = 0; i < n_coeff; ++i) {
      if (eqns->x[i] > max_coeff) max_coeff = eqns->x[i];
      if (eqns->x[i] < min_coeff) min_coeff = eqns->x[i];
    }
    const strength_solver *solver = &m->combined[c].strength;
    double average_strength = 0, total_weight = 0;
    for (int i = 0; i < solver->eqns.n; ++i) {
      double w = 0;
      for (int j = 0; j < solver->eqns.n; ++j) w += solver->eqns.A[i * solver->eqns.n + j];
      w = sqrt(w);
      average_strength = MA(G_MODEL, solver->eqns.x[i], w, average_strength);
      total_weight += w;
    }
    if (total_weight == 0)
      average_strength = 1;
    else
      average_strength /= total_weight;
    if (c == 0) {
      avg_luma_strength = average_strength;
    } else {
      y_corr[c - 1] = avg_luma_strength * eqns->x[n_coeff] / average_strength;
      if (y_corr[c - 1] > max_coeff) max_coeff = y_corr[c - 1];
      if (y_corr[c - 1] < min_coeff) min_coeff = y_corr[c - 1];
    }
  }
  /* Shift value: AR coeffs range (values 6-9) */
  {
    const double a = 1 + floor(log2(max_coeff));
    const double bb = ceil(log2(-min_coeff));
    g->ar_coeff_shift = (uint8_t)clampi(7 - (int)(a > bb ? a : bb), 6, 9);
  }
  const double scale_ar_coeff = 1 << g->ar_coeff_shift;
  int8_t *ar[3] = { g->ar_coeffs_y, g->ar_coeffs_cb, g->ar_coeffs_cr };
  for (int c = 0; c < 3; ++c) {
    const eqsys *eqns = &m->combined[c].eqns;
    for (int i = 0; i < n_coeff; ++i)
      ar[c][i] = (int8_t)clampi((int)round(scale_ar_coeff * eqns->x[i]), -128, 127);
    if (c > 0)
      ar[c][n_coeff] = (int8_t)clampi((int)round(scale_ar_coeff * y_corr[c - 1]), -128, 127);
  }
  g->num_y_coeffs = (uint8_t)n_coeff;
  g->num_uv_coeffs = (uint8_t)(n_coeff + 1);
  g->cb_mult = 128;
  g->cb_luma_mult = 192;
  g->cb_offset = 256;
  g->cr_mult = 128;
  g->cr_luma_mult = 192;
  g->cr_offset = 256;
  g->chroma_scaling_from_luma = 0;
  g->grain_scale_shift = 0;
  g->overlap_flag = 1;
}

/* ------------------------------------------------------------------------ */
/* DiffGenerator (reference call sites src/main.rs:420-427, :442, :524)       */
/* ------------------------------------------------------------------------ */
struct orc_diff {
  int64_t fps_num, fps_den;
  int src_bd, den_bd;
  int chroma;
  uint64_t frame_count;
  uint64_t prev_timestamp;
  flat_finder finder;
  noise_model model;
  orc_segment *table;
  int ntable, captable;
  char err[256];
  /* last-frame introspection */
  uint8_t *flat;
  float *scores;
  int nbw, nbh, nblocks_alloc;
  shadow_rec sh;
  /* 8-bit scratch planes */
  uint8_t *s8[3], *d8[3];
  size_t s8cap[3], d8cap[3];
};

orc_diff *orc_diff_new(int64_t fps_num, int64_t fps_den, int src_bd, int den_bd,
                       int lag, int chroma) {
  if (lag < 1 || lag > 3) return NULL;
  orc_diff *g = (orc_diff *)calloc(1, sizeof(*g));
  g->fps_num = fps_num;
  g->fps_den = fps_den;
  g->src_bd = src_bd;
  g->den_bd = den_bd;
  g->chroma = chroma;
  ff_init(&g->finder);
  nm_init(&g->model, lag);
  return g;
}

void orc_diff_free(orc_diff *g) {
  if (!g) return;
  free(g->table);
  free(g->flat);
  free(g->scores);
  free(g->sh.luma_sum);
  for (int c = 0; c < 3; ++c) {
    free(g->sh.sum_d[c]);
    free(g->sh.sum_d2[c]);
    free(g->s8[c]);
    free(g->d8[c]);
  }
  free(g);
}

const char *orc_diff_last_error(const orc_diff *g) { return g->err; }

/* av1-grain util.rs frame_into_u8: truncating right shift by (bit_depth - 8).
 * Always packs into a tight u8 scratch plane (stride == w) so that source and
 * denoised share one stride per channel, as libaom's interface assumes. */
static const uint8_t *plane_into_u8(const void *data, size_t stride_bytes, int bps,
                                    int bd, int w, int h, uint8_t **scratch,
                                    size_t *cap) {
  const size_t need = (size_t)w * h;
  if (*cap < need) {
    free(*scratch);
    *scratch = (uint8_t *)malloc(need);
    *cap = need;
  }
  if (bps == 1) {
    for (int y = 0; y < h; ++y)
      memcpy(*scratch + (size_t)y * w, (const uint8_t *)data + (size_t)y * stride_bytes, w);
    return *scratch;
  }
  const int shift = bd - 8;
  for (int y = 0; y < h; ++y) {
    const uint16_t *row = (const uint16_t *)((const uint8_t *)data + (size_t)y * stride_bytes);
    for (int x = 0; x < w; ++x) (*scratch)[(size_t)y * w + x] = (uint8_t)(row[x] >> shift);
  }
  return *scratch;
}

static void push_segment(orc_diff *g, const orc_segment *s) {
  if (g->ntable == g->captable) {
    g->captable = g->captable ? 2 * g->captable : 4;
    g->table = (orc_segment *)realloc(g->table, sizeof(orc_segment) * g->captable);
  }
  g->table[g->ntable++] = *s;
}

int orc_diff_frame(orc_diff *g, const orc_frame *src, const orc_frame *den) {
  /* verify_dimensions_match */
  if (src->width != den->width || src->height != den->height ||
      src->xdec != den->xdec || src->ydec != den->ydec || src->nplanes != den->nplanes) {
    snprintf(g->err, sizeof(g->err), "Source and denoised frame dimensions do not match");
    return -1;
  }
  const int w = (int)src->width, h = (int)src->height;
  const int nplanes = g->chroma ? src->nplanes : 1;
  planes8 P;
  memset(&P, 0, sizeof(P));
  P.w = w;
  P.h = h;
  P.sub[0] = src->xdec;
  P.sub[1] = src->ydec;
  for (int c = 0; c < nplanes; ++c) {
    const int pw = c ? (w >> src->xdec) : w, ph = c ? (h >> src->ydec) : h;
    P.data[c] = plane_into_u8(src->data[c], src->stride_bytes[c], src->bytes_per_sample,
                              g->src_bd, pw, ph, &g->s8[c], &g->s8cap[c]);
    P.den[c] = plane_into_u8(den->data[c], den->stride_bytes[c], den->bytes_per_sample,
                             g->den_bd, pw, ph, &g->d8[c], &g->d8cap[c]);
    P.stride[c] = pw;
  }

  const int nbw = (w + BLOCK_SIZE - 1) / BLOCK_SIZE, nbh = (h + BLOCK_SIZE - 1) / BLOCK_SIZE;
  const int nblocks = nbw * nbh;
  if (g->nblocks_alloc < nblocks) {
    g->flat = (uint8_t *)realloc(g->flat, nblocks);
    g->scores = (float *)realloc(g->scores, sizeof(float) * nblocks);
    g->sh.luma_sum = (uint32_t *)realloc(g->sh.luma_sum, sizeof(uint32_t) * nblocks);
    for (int c = 0; c < 3; ++c) {
      g->sh.sum_d[c] = (int32_t *)realloc(g->sh.sum_d[c], sizeof(int32_t) * nblocks);
      g->sh.sum_d2[c] = (uint32_t *)realloc(g->sh.sum_d2[c], sizeof(uint32_t) * nblocks);
    }
    g->nblocks_alloc = nblocks;
  }
  g->nbw = nbw;
  g->nbh = nbh;
  memset(g->sh.S, 0, sizeof(g->sh.S));
  memset(g->sh.Sb, 0, sizeof(g->sh.Sb));
  memset(g->sh.nobs, 0, sizeof(g->sh.nobs));
  memset(g->sh.luma_sum, 0, sizeof(uint32_t) * nblocks);
  for (int c = 0; c < 3; ++c) {
    memset(g->sh.sum_d[c], 0, sizeof(int32_t) * nblocks);
    memset(g->sh.sum_d2[c], 0, sizeof(uint32_t) * nblocks);
  }

  ff_run(&g->finder, P.data[0], w, h, P.stride[0], g->flat, g->scores);
  const int status = nm_update(&g->model, &P, nplanes, g->flat, &g->sh, g->err, sizeof(g->err));
  if (status == NOISE_ERROR) return -2;
  if (status == NOISE_DIFFERENT) {
    const uint64_t cur_timestamp =
        g->frame_count * 10000000ULL * (uint64_t)g->fps_den / (uint64_t)g->fps_num;
    orc_segment s;
    nm_get_grain_parameters(&g->model, g->prev_timestamp, cur_timestamp, &s);
    push_segment(g, &s);
    nm_save_latest(&g->model);
    g->prev_timestamp = cur_timestamp;
  }
  g->frame_count += 1;
  return 0;
}

int orc_diff_finish(orc_diff *g, orc_segment *out, int cap) {
  orc_segment s;
  nm_get_grain_parameters(&g->model, g->prev_timestamp, (uint64_t)INT64_MAX, &s);
  push_segment(g, &s);
  if (g->ntable > cap) return -1;
  memcpy(out, g->table, sizeof(orc_segment) * g->ntable);
  return g->ntable;
}

/* ---- checkpoint of a generator between two frames (tests/golden/make_golden.py long: an hour-long job that must survive an
 * interrupted session).  The state that crosses frames is the noise model (plain arrays), the frame counter, the last cut's
 * timestamp and the segments emitted so far; everything else is scratch of the frame in hand. ---- */
size_t orc_diff_state_size(const orc_diff *g) {
  return sizeof(uint64_t) * 3 + sizeof(noise_model) + sizeof(orc_segment) * (size_t)g->ntable;
}
long orc_diff_save(const orc_diff *g, void *buf, size_t cap) {
  const size_t need = orc_diff_state_size(g);
  if (cap < need) return -1;
  uint8_t *p = (uint8_t *)buf;
  const uint64_t head[3] = { g->frame_count, g->prev_timestamp, (uint64_t)g->ntable };
  memcpy(p, head, sizeof(head));
  p += sizeof(head);
  memcpy(p, &g->model, sizeof(noise_model));
  p += sizeof(noise_model);
  if (g->ntable) memcpy(p, g->table, sizeof(orc_segment) * (size_t)g->ntable);
  return (long)need;
}
int orc_diff_restore(orc_diff *g, const void *buf, size_t size) {
  const uint8_t *p = (const uint8_t *)buf;
  uint64_t head[3];
  if (size < sizeof(head) + sizeof(noise_model)) return -1;
  memcpy(head, p, sizeof(head));
  p += sizeof(head);
  if (size != sizeof(head) + sizeof(noise_model) + sizeof(orc_segment) * (size_t)head[2]) return -1;
  noise_model m;
  memcpy(&m, p, sizeof(m));
  if (m.lag != g->model.lag) return -2;
  p += sizeof(noise_model);
  g->model = m;
  g->frame_count = head[0];
  g->prev_timestamp = head[1];
  g->ntable = 0;
  for (uint64_t i = 0; i < head[2]; ++i) {
    orc_segment sgm;
    memcpy(&sgm, p + sizeof(orc_segment) * i, sizeof(sgm));
    push_segment(g, &sgm);
  }
  return 0;
}

const uint8_t *orc_last_flat_mask(const orc_diff *g, int *nbw, int *nbh) {
  if (nbw) *nbw = g->nbw;
  if (nbh) *nbh = g->nbh;
  return g->flat;
}
const float *orc_last_scores(const orc_diff *g) { return g->scores; }
int orc_last_ar_sums(const orc_diff *g, int c, int64_t *S, int64_t *Sb, int64_t *nobs) {
  const int n = g->model.latest[c].eqns.n;
  if (S) memcpy(S, g->sh.S[c], sizeof(int64_t) * n * n);
  if (Sb) memcpy(Sb, g->sh.Sb[c], sizeof(int64_t) * n);
  if (nobs) *nobs = g->sh.nobs[c];
  return n;
}
void orc_last_block_stats(const orc_diff *g, int c, uint32_t *luma_sum, int32_t *sum_d,
                          uint32_t *sum_d2) {
  const int nb = g->nbw * g->nbh;
  if (luma_sum) memcpy(luma_sum, g->sh.luma_sum, sizeof(uint32_t) * nb);
  if (sum_d) memcpy(sum_d, g->sh.sum_d[c], sizeof(int32_t) * nb);
  if (sum_d2) memcpy(sum_d2, g->sh.sum_d2[c], sizeof(uint32_t) * nb);
}
int orc_num_segments(const orc_diff *g) { return g->ntable; }

/* ------------------------------------------------------------------------ */
/* .tbl text: src/main.rs:525 ("filmgrn1") + write_film_grain_segment :631-696 */
/* ------------------------------------------------------------------------ */
static int appendf(char **p, char *end, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  const int k = vsnprintf(*p, (size_t)(end - *p), fmt, ap);
  va_end(ap);
  if (k < 0 || *p + k >= end) return -1;
  *p += k;
  return 0;
}

long orc_format_tbl(const orc_segment *segs, int n, char *buf, size_t cap) {
  char *p = buf, *end = buf + cap;
#define AP(...) do { if (appendf(&p, end, __VA_ARGS__)) return -1; } while (0)
  AP("filmgrn1\n");
  for (int s = 0; s < n; ++s) {
    const orc_segment *g = &segs[s];
    AP("E %llu %llu 1 %u 1\n", (unsigned long long)g->start_time,
       (unsigned long long)g->end_time, (unsigned)g->random_seed);
    AP("\tp %u %u %u %u %u %u %u %u %u %u %u %u\n", g->ar_coeff_lag, g->ar_coeff_shift,
       g->grain_scale_shift, g->scaling_shift, g->chroma_scaling_from_luma, g->overlap_flag,
       g->cb_mult, g->cb_luma_mult, g->cb_offset, g->cr_mult, g->cr_luma_mult, g->cr_offset);
    AP("\tsY %u ", g->num_y_points); /* trailing space: src/main.rs:659 */
    for (int i = 0; i < g->num_y_points; ++i)
      AP(" %u %u", g->scaling_points_y[i][0], g->scaling_points_y[i][1]);
    AP("\n");
    AP("\tsCb %u", g->num_cb_points);
    for (int i = 0; i < g->num_cb_points; ++i)
      AP(" %u %u", g->scaling_points_cb[i][0], g->scaling_points_cb[i][1]);
    AP("\n");
    AP("\tsCr %u", g->num_cr_points);
    for (int i = 0; i < g->num_cr_points; ++i)
      AP(" %u %u", g->scaling_points_cr[i][0], g->scaling_points_cr[i][1]);
    AP("\n");
    AP("\tcY");
    for (int i = 0; i < g->num_y_coeffs; ++i) AP(" %d", g->ar_coeffs_y[i]);
    AP("\n");
    AP("\tcCb");
    for (int i = 0; i < g->num_uv_coeffs; ++i) AP(" %d", g->ar_coeffs_cb[i]);
    AP("\n");
    AP("\tcCr");
    for (int i = 0; i < g->num_uv_coeffs; ++i) AP(" %d", g->ar_coeffs_cr[i]);
    AP("\n");
  }
#undef AP
  return (long)(p - buf);
}

#define _DEFAULT_SOURCE /* M_PI */
/* oracle/resize_oracle.c -- CPU restatement of the `resize` source filter of `grav1synth diff --filters`
 * (/root/reference/src/filters.rs:150-178: video_resize::resize::<T, {BicubicHermite, BicubicCatmullRom, BicubicMitchell,
 * Lanczos3, Spline36}>(frame, ResizeDimensions { width, height }, source_bd)).
 *
 * TEST INFRASTRUCTURE: only tests/ may load this.  PARITY UNPINNED: the arithmetic lives in the third-party crate
 * video-resize 0.2.0 (Cargo.lock), which is not in /root/reference and cannot be built here (no Rust toolchain); the
 * reference holds no vectors for it.  This file restates the published algorithm that crate ports (zimg's separable
 * resampler) under the assumptions listed in grav1synth_amd/csrc/resize.hip: per plane a horizontal pass then a vertical
 * pass; output sample i of an axis sits at (i + 0.5) / scale in input coordinates; window of 2 ceil(support / min(scale, 1))
 * taps from floor(pos - size / 2 + 0.5); positions outside the plane mirrored back; taps = kernel((tap - pos) min(scale, 1))
 * normalised to 1 in f64, applied in f32 in ascending tap order (no fused multiply-add: compile with -ffp-contract=off);
 * each pass rounds half up and clamps to 0 .. 2^bit_depth - 1 into the sample type. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double k_sinc(double x) {
  if (x == 0.0) return 1.0;
  x *= M_PI;
  return sin(x) / x;
}
static double k_bicubic(double x, double b, double c) {
  x = fabs(x);
  if (x < 1.0) return ((12.0 - 9.0 * b - 6.0 * c) * x * x * x + (-18.0 + 12.0 * b + 6.0 * c) * x * x + (6.0 - 2.0 * b)) / 6.0;
  if (x < 2.0) return ((-b - 6.0 * c) * x * x * x + (6.0 * b + 30.0 * c) * x * x + (-12.0 * b - 48.0 * c) * x + (8.0 * b + 24.0 * c)) / 6.0;
  return 0.0;
}
static double k_spline36(double x) {
  x = fabs(x);
  if (x < 1.0) return ((13.0 / 11.0 * x - 453.0 / 209.0) * x - 3.0 / 209.0) * x + 1.0;
  if (x < 2.0) {
    x -= 1.0;
    return ((-6.0 / 11.0 * x + 270.0 / 209.0) * x - 156.0 / 209.0) * x;
  }
  if (x < 3.0) {
    x -= 2.0;
    return ((1.0 / 11.0 * x - 45.0 / 209.0) * x + 26.0 / 209.0) * x;
  }
  return 0.0;
}
/* alg: 0 hermite (B 0, C 0), 1 catmullrom (0, 1/2), 2 mitchell (1/3, 1/3), 3 lanczos (3 lobes), 4 spline36 */
static int alg_of(const char *name) {
  static const char *n[] = {"hermite", "catmullrom", "mitchell", "lanczos", "spline36"};
  for (int i = 0; i < 5; ++i)
    if (strcmp(name, n[i]) == 0) return i;
  return -1;
}
static double kernel(int alg, double x) {
  switch (alg) {
    case 0: return k_bicubic(x, 0.0, 0.0);
    case 1: return k_bicubic(x, 0.0, 0.5);
    case 2: return k_bicubic(x, 1.0 / 3.0, 1.0 / 3.0);
    case 3: return fabs(x) < 3.0 ? k_sinc(x) * k_sinc(x / 3.0) : 0.0;
    default: return k_spline36(x);
  }
}

/* taps of an axis; returns the tap count; idx / coef hold dst * taps entries (taps that mirror onto one sample merged) */
int orc_resize_plan(const char *alg_name, int src, int dst, int *idx, float *coef, size_t cap) {
  const int alg = alg_of(alg_name);
  if (alg < 0 || src <= 0 || dst <= 0) return -1;
  const double scale = (double)dst / (double)src, step = scale < 1.0 ? scale : 1.0;
  const double support = (alg <= 2 ? 2.0 : 3.0) / step;
  int fs = (int)ceil(support);
  if (fs < 1) fs = 1;
  fs *= 2;
  if ((size_t)dst * (size_t)fs > cap) return fs;
  double *w = (double *)malloc(sizeof(double) * (size_t)fs);
  int *ix = (int *)malloc(sizeof(int) * (size_t)fs);
  for (int i = 0; i < dst; ++i) {
    const double pos = ((double)i + 0.5) / scale;
    const double begin = floor(pos - (double)fs / 2.0 + 0.5) + 0.5;
    double total = 0.0;
    for (int k = 0; k < fs; ++k) {
      w[k] = kernel(alg, (begin + (double)k - pos) * step);
      total += w[k];
    }
    int n = 0;
    for (int k = 0; k < fs; ++k) {
      double xp = begin + (double)k;
      if (xp < 0.0) xp = -xp;
      else if (xp >= (double)src) xp = 2.0 * (double)src - xp;
      int j = (int)floor(xp);
      if (j < 0) j = 0;
      if (j > src - 1) j = src - 1;
      const double wk = w[k] / total;
      int at = -1;
      for (int m = 0; m < n; ++m)
        if (ix[m] == j) at = m;
      if (at < 0) {
        ix[n] = j;
        w[n] = wk;
        ++n;
      } else {
        w[at] += wk;
      }
    }
    for (int k = 0; k < fs; ++k) {
      idx[(size_t)i * fs + k] = k < n ? ix[k] : ix[0];
      coef[(size_t)i * fs + k] = k < n ? (float)w[k] : 0.0f;
    }
  }
  free(w);
  free(ix);
  return fs;
}

static float sample_at(const void *plane, int bps, size_t stride, int x, int y) {
  const uint8_t *row = (const uint8_t *)plane + (size_t)y * stride;
  return bps == 1 ? (float)row[x] : (float)((const uint16_t *)row)[x];
}
static void store_at(void *plane, int bps, size_t stride, int x, int y, float acc, float maxv) {
  float v = floorf(acc + 0.5f);
  if (v < 0.0f) v = 0.0f;
  if (v > maxv) v = maxv;
  uint8_t *row = (uint8_t *)plane + (size_t)y * stride;
  if (bps == 1) row[x] = (uint8_t)v;
  else ((uint16_t *)row)[x] = (uint16_t)v;
}

/* one plane, sw x sh -> dw x dh; 0 on success */
int orc_resize_plane(const char *alg_name, const void *in, int bps, size_t in_stride, int sw, int sh, void *out, size_t out_stride, int dw,
                     int dh, int bit_depth) {
  if (alg_of(alg_name) < 0 || (bps != 1 && bps != 2)) return -1;
  const float maxv = (float)((1u << bit_depth) - 1u);
  const int th = orc_resize_plan(alg_name, sw, dw, NULL, NULL, 0), tv = orc_resize_plan(alg_name, sh, dh, NULL, NULL, 0);
  int *hi = (int *)malloc(sizeof(int) * (size_t)dw * th), *vi = (int *)malloc(sizeof(int) * (size_t)dh * tv);
  float *hc = (float *)malloc(sizeof(float) * (size_t)dw * th), *vc = (float *)malloc(sizeof(float) * (size_t)dh * tv);
  orc_resize_plan(alg_name, sw, dw, hi, hc, (size_t)dw * th);
  orc_resize_plan(alg_name, sh, dh, vi, vc, (size_t)dh * tv);
  const size_t tstride = (size_t)dw * bps;
  void *tmp = malloc(tstride * (size_t)sh);
  for (int y = 0; y < sh; ++y)
    for (int x = 0; x < dw; ++x) {
      float acc = 0.0f;
      for (int k = 0; k < th; ++k) acc = acc + hc[(size_t)x * th + k] * sample_at(in, bps, in_stride, hi[(size_t)x * th + k], y);
      store_at(tmp, bps, tstride, x, y, acc, maxv);
    }
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      float acc = 0.0f;
      for (int k = 0; k < tv; ++k) acc = acc + vc[(size_t)y * tv + k] * sample_at(tmp, bps, tstride, x, vi[(size_t)y * tv + k]);
      store_at(out, bps, out_stride, x, y, acc, maxv);
    }
  free(tmp);
  free(hi);
  free(vi);
  free(hc);
  free(vc);
  return 0;
}

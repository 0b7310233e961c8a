// estimate.hip -- N4: the single-source noise estimator of `grav1synth estimate` (feature "unstable").
//
// The command (/root/reference/src/main.rs:534-608) calls av1_grain::estimate_plane_noise(&frame.y_plane, bit_depth) on
// every frame and writes "filmgrn1" and one "{:.3}" line per frame (-1 for None).  The estimator is av1-grain's port of
// libaom's av1_estimate_noise_from_single_plane: over the interior pixels of the luma plane, a Sobel gradient decides
// whether the pixel is smooth (|Gx| + |Gy|, rounded down to 8-bit scale, below 50); the smooth pixels' |Laplacian|
// (rounded to 8-bit scale) is averaged: sigma = accum / (6 count) * sqrt(pi / 2), None when fewer than 16 pixels are
// smooth.  One pass over one plane, a 3x3 stencil and two integer sums: HBM bound (2 bytes per luma pixel at 10 bit).
//
// Kernel: a wave owns 62 x 8 output columns of a strip of rows and walks the strip top to bottom with the three rows
// of the stencil in registers (each row is loaded once per strip, as one 8-sample word per lane: 16-byte loads at
// 10 bit); lanes 0 and 63 only carry halo columns, the horizontal neighbours of a lane's word come from the adjacent
// lanes.  Exact integers: the two sums of a frame are the same for any strip / wave / batch partition; the host turns
// them into the f64 with the reference's three operations.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/g1s_diff.h"

namespace {

constexpr int kEdgeThreshold = 50;  // EDGE_THRESHOLD
constexpr int kStripRows = 32;      // output rows per wave strip (+ 2 halo rows)
constexpr int kWavesPerWg = 4;
constexpr int kColsPerWave = 62 * 8;

struct EstFrame {
  const uint8_t *y;
  uint32_t stride;  // bytes
};

struct EstParams {
  const EstFrame *frames;
  unsigned long long *sums;  // [frames][2]: accum, count
  int W, H, bps, shift;
  int col_strips, row_strips;
};

// the 8 samples of a word, widened; outside the plane: zeros (never used by an output pixel that counts)
template <int BPS>
__device__ __forceinline__ void est_load(const uint8_t *row, int x0, int W, bool row_ok, int (&p)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = 0;
  if (!row_ok || x0 >= W || x0 + 8 <= 0) return;
  if (x0 >= 0 && x0 + 8 <= W) {
    if (BPS == 2) {
      // (rows of a frame are at least 2-byte aligned; a word is read with the widest loads its address allows)
      const uint8_t *a = row + (size_t)x0 * 2;
      uint32_t w[4];
      if (((uintptr_t)a & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(a);
        w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
      } else if (((uintptr_t)a & 3) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = reinterpret_cast<const uint32_t *>(a)[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (uint32_t)reinterpret_cast<const uint16_t *>(a)[2 * k] | ((uint32_t)reinterpret_cast<const uint16_t *>(a)[2 * k + 1] << 16);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        p[2 * k] = (int)(w[k] & 0xffffu);
        p[2 * k + 1] = (int)(w[k] >> 16);
      }
    } else {
      const uint8_t *a = row + x0;
      uint32_t w[2];
      if (((uintptr_t)a & 7) == 0) {
        const uint2 v = *reinterpret_cast<const uint2 *>(a);
        w[0] = v.x, w[1] = v.y;
      } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) w[k] = (uint32_t)a[4 * k] | ((uint32_t)a[4 * k + 1] << 8) | ((uint32_t)a[4 * k + 2] << 16) | ((uint32_t)a[4 * k + 3] << 24);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // a word over the right (or left) edge of the plane
    const int x = x0 + k;
    if (x >= 0 && x < W) p[k] = BPS == 2 ? (int)reinterpret_cast<const uint16_t *>(row)[x] : (int)row[x];
  }
}

template <int BPS>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_estimate(EstParams ep) {
  const int frame = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = blockIdx.x * kWavesPerWg + wave;
  if (strip >= ep.col_strips * ep.row_strips) return;
  const int cs = strip % ep.col_strips, rs = strip / ep.col_strips;
  const EstFrame fr = ep.frames[frame];
  const int W = ep.W, H = ep.H, shift = ep.shift, half = shift ? 1 << (shift - 1) : 0;
  // output columns of the wave: [cs * 496, cs * 496 + 496) (less the plane's first and last column); lane l holds the
  // word at x0 = cs * 496 - 8 + 8 l -- a multiple of 8 samples: aligned 16-byte loads -- lanes 0 and 63 the halo words
  const int x0 = cs * kColsPerWave - 8 + 8 * lane;
  const int y_first = 1 + rs * kStripRows, y_last = min(y_first + kStripRows, H - 1);  // output rows [y_first, y_last)
  int a[8], b[8], c[8];  // rows y - 1, y, y + 1
  est_load<BPS>(fr.y + (size_t)(y_first - 1) * fr.stride, x0, W, true, a);
  est_load<BPS>(fr.y + (size_t)y_first * fr.stride, x0, W, y_first < H, b);
  uint32_t accum = 0, count = 0;
  const bool out_lane = lane >= 1 && lane <= 62;
  for (int y = y_first; y < y_last; ++y) {
    est_load<BPS>(fr.y + (size_t)(y + 1) * fr.stride, x0, W, y + 1 < H, c);
    // the column left of the word (the neighbour lane's last sample) and right of it (its first), for the three rows
    const int al = __shfl_up(a[7], 1), bl = __shfl_up(b[7], 1), cl = __shfl_up(c[7], 1);
    const int ar = __shfl_down(a[0], 1), br = __shfl_down(b[0], 1), cr = __shfl_down(c[0], 1);
    if (out_lane) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int x = x0 + k;
        const int m00 = k ? a[k - 1] : al, m01 = a[k], m02 = k < 7 ? a[k + 1] : ar;
        const int m10 = k ? b[k - 1] : bl, m11 = b[k], m12 = k < 7 ? b[k + 1] : br;
        const int m20 = k ? c[k - 1] : cl, m21 = c[k], m22 = k < 7 ? c[k + 1] : cr;
        const int gx = (m00 - m02) + (m20 - m22) + 2 * (m10 - m12);
        const int gy = (m00 - m20) + (m02 - m22) + 2 * (m01 - m21);
        const int ga = (abs(gx) + abs(gy) + half) >> shift;
        const int v = 4 * m11 - 2 * (m01 + m21 + m10 + m12) + (m00 + m02 + m20 + m22);
        const bool on = x >= 1 && x < W - 1 && ga < kEdgeThreshold;
        accum += on ? (uint32_t)((abs(v) + half) >> shift) : 0u;
        count += on ? 1u : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] = b[k];
      b[k] = c[k];
    }
  }
  // wave sums -> one pair of atomics per wave
  for (int o = 32; o; o >>= 1) {
    accum += __shfl_down(accum, o);
    count += __shfl_down(count, o);
  }
  if (lane == 0 && (accum | count)) {
    atomicAdd(&ep.sums[2 * frame], (unsigned long long)accum);
    atomicAdd(&ep.sums[2 * frame + 1], (unsigned long long)count);
  }
}

constexpr double kSqrtPiBy2 = 1.2533141373155003;  // SQRT_PI_BY_2

}  // namespace

struct g1s_estimate {
  int device = 0;
  uint32_t bit_depth = 8;
  uint32_t W = 0, H = 0, bps = 0;
  uint32_t batch = 32;
  hipStream_t stream = nullptr;
  std::vector<EstFrame> h_frames;  // the batch being filled
  EstFrame *d_frames = nullptr;
  unsigned long long *d_sums = nullptr, *h_sums = nullptr;
  uint8_t *d_stage = nullptr;      // device copies of host frames
  size_t stage_frame = 0;
  std::vector<double> estimates;   // one per frame: sigma, or -1 (None)
  std::string err;
  uint64_t frames_kernel = 0;
  double ms_kernel = 0;
  bool timing = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  int fail(int code, const std::string &m) {
    err = m;
    return code;
  }
  int flush();
};

#define EST_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(G1S_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_)); \
  } while (0)

int g1s_estimate::flush() {
  const uint32_t B = (uint32_t)h_frames.size();
  if (!B) return G1S_OK;
  EST_TRY(hipMemcpyAsync(d_frames, h_frames.data(), sizeof(EstFrame) * B, hipMemcpyHostToDevice, stream));
  EST_TRY(hipMemsetAsync(d_sums, 0, sizeof(unsigned long long) * 2 * B, stream));
  EstParams ep;
  ep.frames = d_frames;
  ep.sums = d_sums;
  ep.W = (int)W;
  ep.H = (int)H;
  ep.bps = (int)bps;
  ep.shift = (int)bit_depth - 8;
  ep.col_strips = ((int)W - 1 + kColsPerWave - 1) / kColsPerWave;
  ep.row_strips = ((int)H - 2 + kStripRows - 1) / kStripRows;
  if (W >= 3 && H >= 3) {
    const dim3 grid((ep.col_strips * ep.row_strips + kWavesPerWg - 1) / kWavesPerWg, B);
    if (timing) EST_TRY(hipEventRecord(ev0, stream));
    if (bps == 2) hipLaunchKernelGGL(k_estimate<2>, grid, dim3(64 * kWavesPerWg), 0, stream, ep);
    else hipLaunchKernelGGL(k_estimate<1>, grid, dim3(64 * kWavesPerWg), 0, stream, ep);
    if (timing) EST_TRY(hipEventRecord(ev1, stream));
  }
  EST_TRY(hipMemcpyAsync(h_sums, d_sums, sizeof(unsigned long long) * 2 * B, hipMemcpyDeviceToHost, stream));
  EST_TRY(hipStreamSynchronize(stream));
  EST_TRY(hipGetLastError());
  if (timing && W >= 3 && H >= 3) {
    float ms = 0;
    EST_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    ms_kernel += ms;
    frames_kernel += B;
  }
  for (uint32_t i = 0; i < B; ++i) {
    const unsigned long long accum = h_sums[2 * i], count = h_sums[2 * i + 1];
    // (count < 16) ? None : accum as f64 / (6 * count) as f64 * SQRT_PI_BY_2
    estimates.push_back(count < 16 ? -1.0 : (double)accum / (double)(6 * count) * kSqrtPiBy2);
  }
  h_frames.clear();
  return G1S_OK;
}

extern "C" {

g1s_estimate_t *g1s_estimate_new(uint32_t bit_depth, int32_t device, uint32_t batch_frames) {
  if (bit_depth < 8 || bit_depth > 16) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;  // no CPU fallback
  g1s_estimate *e = new g1s_estimate;
  if (device < 0) (void)hipGetDevice(&device);
  e->device = device;
  e->bit_depth = bit_depth;
  e->bps = bit_depth > 8 ? 2 : 1;
  e->batch = batch_frames ? std::min(batch_frames, 256u) : 32u;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void **)&e->d_frames, sizeof(EstFrame) * e->batch) != hipSuccess ||
      hipMalloc((void **)&e->d_sums, sizeof(unsigned long long) * 2 * e->batch) != hipSuccess ||
      hipHostMalloc((void **)&e->h_sums, sizeof(unsigned long long) * 2 * e->batch, hipHostMallocDefault) != hipSuccess ||
      hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
    g1s_estimate_free(e);
    return nullptr;
  }
  return e;
}

int g1s_estimate_frame(g1s_estimate_t *e, const g1s_frame_t *f) {
  if (!e || !f || !f->data[0]) return G1S_ERR_INVALID;
  (void)hipSetDevice(e->device);
  if ((f->bytes_per_sample == 1) != (e->bit_depth == 8) || (f->bytes_per_sample != 1 && f->bytes_per_sample != 2))
    return e->fail(G1S_ERR_INVALID, "bytes_per_sample does not match the bit depth given to g1s_estimate_new");
  if (f->width < 1 || f->height < 1) return e->fail(G1S_ERR_INVALID, "empty frame");
  if (!e->W) {
    e->W = f->width;
    e->H = f->height;
    e->stage_frame = (((size_t)e->W * e->bps + 15) & ~size_t(15)) * e->H;
  } else if (e->W != f->width || e->H != f->height) {
    return e->fail(G1S_ERR_DIM_MISMATCH, "frame geometry changed mid-stream");
  }
  EstFrame ef;
  if (f->on_device == 1) {
    ef.y = static_cast<const uint8_t *>(f->data[0]);
    ef.stride = (uint32_t)f->stride_bytes[0];
  } else {  // host planes: copied before the call returns (the `&frame.y_plane` borrow)
    if (!e->d_stage && hipMalloc((void **)&e->d_stage, e->stage_frame * e->batch) != hipSuccess)
      return e->fail(G1S_ERR_HIP, "hipMalloc of the staging buffer failed");
    const size_t row = ((size_t)e->W * e->bps + 15) & ~size_t(15);
    uint8_t *dst = e->d_stage + e->stage_frame * e->h_frames.size();
    if (hipMemcpy2D(dst, row, f->data[0], f->stride_bytes[0], (size_t)e->W * e->bps, e->H, hipMemcpyHostToDevice) != hipSuccess)
      return e->fail(G1S_ERR_HIP, "hipMemcpy2D of a host frame failed");
    ef.y = dst;
    ef.stride = (uint32_t)row;
  }
  e->h_frames.push_back(ef);
  return e->h_frames.size() == e->batch ? e->flush() : G1S_OK;
}

int g1s_estimate_finish(g1s_estimate_t *e, double *out, size_t cap, size_t *n_out) {
  if (!e) return G1S_ERR_INVALID;
  (void)hipSetDevice(e->device);
  const int rc = e->flush();
  if (rc) return rc;
  if (n_out) *n_out = e->estimates.size();
  if (e->estimates.size() > cap || (!out && !e->estimates.empty())) return e->fail(G1S_ERR_CAPACITY, "estimate buffer too small");
  if (!e->estimates.empty()) std::memcpy(out, e->estimates.data(), sizeof(double) * e->estimates.size());
  return G1S_OK;
}

int g1s_estimate_set_timing(g1s_estimate_t *e, int enable, double *ms_kernel, uint64_t *frames) {
  if (!e) return G1S_ERR_INVALID;
  e->timing = enable != 0;
  if (ms_kernel) *ms_kernel = e->ms_kernel;
  if (frames) *frames = e->frames_kernel;
  return G1S_OK;
}

const char *g1s_estimate_last_error(const g1s_estimate_t *e) { return e ? e->err.c_str() : ""; }

void g1s_estimate_free(g1s_estimate_t *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->d_frames) (void)hipFree(e->d_frames);
  if (e->d_sums) (void)hipFree(e->d_sums);
  if (e->h_sums) (void)hipHostFree(e->h_sums);
  if (e->d_stage) (void)hipFree(e->d_stage);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

long g1s_format_estimates(const double *estimates, size_t n, char *buf, size_t cap) {
  // writeln!("filmgrn1"), then writeln!("{:.3}", estimate.unwrap_or(-1f64)) per frame (src/main.rs:597-600)
  std::string s = "filmgrn1\n";
  char line[64];
  for (size_t i = 0; i < n; ++i) {
    snprintf(line, sizeof(line), "%.3f\n", estimates[i]);
    s += line;
  }
  if (s.size() > cap) return G1S_ERR_CAPACITY;
  std::memcpy(buf, s.data(), s.size());
  return (long)s.size();
}

}  // extern "C"

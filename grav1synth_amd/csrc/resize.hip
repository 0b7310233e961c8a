// resize.hip -- the `resize` filter of `grav1synth diff --filters` (N3) on the device.
//
// The reference applies Filter::Resize to the SOURCE frame before every diff_frame (/root/reference/src/filters.rs:150-178,
// src/main.rs:615-629): video_resize::resize::<T, {BicubicHermite, BicubicCatmullRom, BicubicMitchell, Lanczos3, Spline36}>
// (frame, ResizeDimensions { width, height }, source_bd).  The crate (video-resize 0.2.0, Cargo.lock) is not in
// /root/reference: PARITY UNPINNED.  What is built here is the published algorithm that crate ports (zimg's separable
// resampler): per plane a horizontal pass and then a vertical one; output sample i of an axis sits at input position
// (i + 0.5) / scale, its window is filter_size = 2 ceil(support / min(scale, 1)) taps wide starting at
// floor(pos - filter_size / 2 + 0.5), positions outside the plane mirror back onto it, the taps are the kernel's values at
// (tap - pos) * min(scale, 1) normalised to sum 1.  ASSUMED (to be checked when the crate can be built): coefficients are
// formed in f64 and applied in f32 in ascending tap order without fused multiply-adds, each pass rounds half up and clamps
// to 0 .. 2^source_bd - 1 into the sample type, chroma planes scale to (width >> xdec, height >> ydec).
// oracle/resize_oracle.c is the scalar restatement (same plan, same order): the -m gpu tests compare bit for bit.
// Two plain kernels, one output sample a thread: this filter is not on the hot path and is not tuned.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/g1s_diff.h"
#include "resize.h"

namespace g1s {

namespace {
double sinc(double x) {
  if (x == 0.0) return 1.0;
  const double a = M_PI * x;
  return std::sin(a) / a;
}
double bicubic(double x, double b, double c) {
  x = std::fabs(x);
  if (x < 1.0) return ((12.0 - 9.0 * b - 6.0 * c) * x * x * x + (-18.0 + 12.0 * b + 6.0 * c) * x * x + (6.0 - 2.0 * b)) / 6.0;
  if (x < 2.0) return ((-b - 6.0 * c) * x * x * x + (6.0 * b + 30.0 * c) * x * x + (-12.0 * b - 48.0 * c) * x + (8.0 * b + 24.0 * c)) / 6.0;
  return 0.0;
}
double spline36(double x) {
  x = std::fabs(x);
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
}  // namespace

int resize_alg_id(const char *alg) {
  static const char *names[] = {"hermite", "catmullrom", "mitchell", "lanczos", "spline36"};
  for (int i = 0; i < 5; ++i)
    if (std::strcmp(alg, names[i]) == 0) return i;
  return -1;
}
double resize_kernel(int alg, double x) {
  switch (alg) {
    case 0: return bicubic(x, 0.0, 0.0);
    case 1: return bicubic(x, 0.0, 0.5);
    case 2: return bicubic(x, 1.0 / 3.0, 1.0 / 3.0);
    case 3: return std::fabs(x) < 3.0 ? sinc(x) * sinc(x / 3.0) : 0.0;
    default: return spline36(x);
  }
}
double resize_support(int alg) { return alg <= 2 ? 2.0 : 3.0; }

// taps of one axis: output i = sum over k < taps of coef[i * taps + k] * in[idx[i * taps + k]], k ascending
void resize_plan(int alg, int src, int dst, ResizePlan &p) {
  const double scale = (double)dst / (double)src, step = std::min(scale, 1.0), support = resize_support(alg) / step;
  const int fs = std::max((int)std::ceil(support), 1) * 2;
  p.src = src;
  p.dst = dst;
  p.taps = fs;
  p.idx.assign((size_t)dst * fs, 0);
  p.coef.assign((size_t)dst * fs, 0.0f);
  std::vector<double> w(fs);
  std::vector<int> ix(fs);
  for (int i = 0; i < dst; ++i) {
    const double pos = ((double)i + 0.5) / scale;
    const double begin = std::floor(pos - (double)fs / 2.0 + 0.5) + 0.5;  // centre of the window's first input sample
    double total = 0.0;
    for (int k = 0; k < fs; ++k) {
      w[k] = resize_kernel(alg, (begin + k - pos) * step);
      total += w[k];
    }
    for (int k = 0; k < fs; ++k) {
      double xp = begin + k;  // (sample centres: x + 0.5) -> mirrored onto the plane, then clamped
      if (xp < 0.0) xp = -xp;
      else if (xp >= (double)src) xp = 2.0 * (double)src - xp;
      int j = (int)std::floor(xp);
      j = std::min(std::max(j, 0), src - 1);
      ix[k] = j;
      w[k] /= total;
    }
    // taps that landed on one input sample (mirroring) are added up; the freed slots keep a zero coefficient
    int n = 0;
    for (int k = 0; k < fs; ++k) {
      int at = -1;
      for (int m = 0; m < n; ++m)
        if (ix[m] == ix[k]) at = m;
      if (at < 0) {
        ix[n] = ix[k];
        w[n] = w[k];
        ++n;
      } else {
        w[at] += w[k];
      }
    }
    for (int k = 0; k < fs; ++k) {
      p.idx[(size_t)i * fs + k] = k < n ? ix[k] : ix[0];
      p.coef[(size_t)i * fs + k] = k < n ? (float)w[k] : 0.0f;
    }
  }
}

template <typename T>
__global__ void k_resize_h(const T *__restrict__ in, size_t in_stride, T *__restrict__ out, size_t out_stride, int dst_w, int h,
                           int taps, const int *__restrict__ idx, const float *__restrict__ coef, float maxv) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dst_w || y >= h) return;
  const T *row = reinterpret_cast<const T *>(reinterpret_cast<const uint8_t *>(in) + (size_t)y * in_stride);
  float acc = 0.0f;
  for (int k = 0; k < taps; ++k) acc = acc + coef[(size_t)x * taps + k] * (float)row[idx[(size_t)x * taps + k]];
  float v = floorf(acc + 0.5f);
  v = v < 0.0f ? 0.0f : (v > maxv ? maxv : v);
  reinterpret_cast<T *>(reinterpret_cast<uint8_t *>(out) + (size_t)y * out_stride)[x] = (T)v;
}
template <typename T>
__global__ void k_resize_v(const T *__restrict__ in, size_t in_stride, T *__restrict__ out, size_t out_stride, int w, int dst_h,
                           int taps, const int *__restrict__ idx, const float *__restrict__ coef, float maxv) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= dst_h) return;
  float acc = 0.0f;
  for (int k = 0; k < taps; ++k) {
    const T *row = reinterpret_cast<const T *>(reinterpret_cast<const uint8_t *>(in) + (size_t)idx[(size_t)y * taps + k] * in_stride);
    acc = acc + coef[(size_t)y * taps + k] * (float)row[x];
  }
  float v = floorf(acc + 0.5f);
  v = v < 0.0f ? 0.0f : (v > maxv ? maxv : v);
  reinterpret_cast<T *>(reinterpret_cast<uint8_t *>(out) + (size_t)y * out_stride)[x] = (T)v;
}

// ---- device-side state of a resize filter (plans and buffers of one geometry) ------------------------------------
struct DevPlan {
  ResizePlan host;
  int *d_idx = nullptr;
  float *d_coef = nullptr;
};
struct ResizeState::Impl {
  int device = -1;
  hipStream_t stream = nullptr;
  std::map<std::pair<int, int>, DevPlan> plans;  // (src, dst) -> taps on the device
  uint8_t *d_in = nullptr, *d_tmp = nullptr;     // host input staged here / the horizontal pass's output
  size_t in_cap = 0, tmp_cap = 0;
  std::vector<uint8_t *> ring;                   // output frames
  size_t out_bytes = 0;
};

ResizeState::ResizeState(int alg_) : alg(alg_), im(new Impl) {}
ResizeState::~ResizeState() {
  if (im->device >= 0) (void)hipSetDevice(im->device);
  for (auto &kv : im->plans) {
    (void)hipFree(kv.second.d_idx);
    (void)hipFree(kv.second.d_coef);
  }
  (void)hipFree(im->d_in);
  (void)hipFree(im->d_tmp);
  for (uint8_t *p : im->ring) (void)hipFree(p);
  if (im->stream) (void)hipStreamDestroy(im->stream);
  delete im;
}

static const DevPlan *dev_plan(ResizeState::Impl *im, int alg, int src, int dst) {
  auto it = im->plans.find({src, dst});
  if (it != im->plans.end()) return &it->second;
  DevPlan dp;
  resize_plan(alg, src, dst, dp.host);
  const size_t n = dp.host.idx.size();
  if (hipMalloc((void **)&dp.d_idx, n * sizeof(int)) != hipSuccess || hipMalloc((void **)&dp.d_coef, n * sizeof(float)) != hipSuccess) return nullptr;
  (void)hipMemcpy(dp.d_idx, dp.host.idx.data(), n * sizeof(int), hipMemcpyHostToDevice);
  (void)hipMemcpy(dp.d_coef, dp.host.coef.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return &im->plans.emplace(std::make_pair(src, dst), dp).first->second;
}

// in (host or device planes) -> out: device planes inside ring slot `slot` (grown on demand), rows 256-byte aligned
int ResizeState::run(const g1s_frame_t &in, uint32_t bit_depth, uint32_t out_w, uint32_t out_h, int device, int slot, g1s_frame_t &out,
                     std::string &err) {
  if (bit_depth < 8 || bit_depth > 16 || (in.bytes_per_sample != 1 && in.bytes_per_sample != 2) || (in.bytes_per_sample == 1 && bit_depth != 8)) {
    err = "resize: unsupported sample format";
    return G1S_ERR_UNSUPPORTED;
  }
  if (in.nplanes == 3 && ((out_w & ((1u << in.xdec) - 1u)) || (out_h & ((1u << in.ydec) - 1u)))) {
    err = "resize: width and height must be multiples of the chroma subsampling";
    return G1S_ERR_INVALID;
  }
  if (device < 0) (void)hipGetDevice(&device);
  if (hipSetDevice(device) != hipSuccess) {
    err = "resize: no HIP device (the filter runs on the device: no CPU fallback)";
    return G1S_ERR_NO_DEVICE;
  }
  if (im->device >= 0 && im->device != device) {
    err = "resize: one filter chain serves one device";
    return G1S_ERR_STATE;
  }
  im->device = device;
  if (!im->stream && hipStreamCreateWithFlags(&im->stream, hipStreamNonBlocking) != hipSuccess) {
    err = "resize: hipStreamCreate failed";
    return G1S_ERR_HIP;
  }
  const size_t bps = in.bytes_per_sample;
  auto align = [](size_t v) { return (v + 255) & ~size_t(255); };
  size_t off_out[3], stride_out[3], total = 0, tmp_need = 0, in_need = 0;
  uint32_t pw[3], ph[3], ow[3], oh[3];
  for (uint32_t c = 0; c < in.nplanes; ++c) {
    pw[c] = c ? in.width >> in.xdec : in.width;
    ph[c] = c ? in.height >> in.ydec : in.height;
    ow[c] = c ? out_w >> in.xdec : out_w;
    oh[c] = c ? out_h >> in.ydec : out_h;
    stride_out[c] = align(ow[c] * bps);
    off_out[c] = total;
    total += stride_out[c] * oh[c];
    tmp_need = std::max(tmp_need, stride_out[c] * ph[c]);
    in_need = std::max(in_need, align(pw[c] * bps) * ph[c]);
  }
  if (total != im->out_bytes) {  // a new geometry: the ring starts over
    // (every slot's frame is gone with it: a caller that still has frames of the old geometry inside a generator must
    //  g1s_diff_sync first -- the device is drained here so that at least no running kernel reads freed memory)
    if (!im->ring.empty()) (void)hipDeviceSynchronize();
    for (uint8_t *p : im->ring) (void)hipFree(p);
    im->ring.clear();
    im->out_bytes = total;
  }
  while ((int)im->ring.size() <= slot) {
    uint8_t *p = nullptr;
    if (hipMalloc((void **)&p, total) != hipSuccess) {
      err = "resize: out of device memory";
      return G1S_ERR_HIP;
    }
    im->ring.push_back(p);
  }
  if (tmp_need > im->tmp_cap) {
    (void)hipFree(im->d_tmp);
    im->d_tmp = nullptr;
    im->tmp_cap = 0;
    if (hipMalloc((void **)&im->d_tmp, tmp_need) != hipSuccess) {
      err = "resize: out of device memory (intermediate plane)";
      return G1S_ERR_HIP;
    }
    im->tmp_cap = tmp_need;
  }
  const bool host_in = in.on_device != 1;  // (0: pageable host, 2: pinned host)
  if (host_in && in_need > im->in_cap) {
    (void)hipFree(im->d_in);
    im->d_in = nullptr;
    im->in_cap = 0;
    if (hipMalloc((void **)&im->d_in, in_need) != hipSuccess) {
      err = "resize: out of device memory (input staging)";
      return G1S_ERR_HIP;
    }
    im->in_cap = in_need;
  }
  out = in;
  out.width = out_w;
  out.height = out_h;
  out.on_device = 1;
  const float maxv = (float)((1u << bit_depth) - 1u);
  for (uint32_t c = 0; c < in.nplanes; ++c) {
    const DevPlan *hp = dev_plan(im, alg, (int)pw[c], (int)ow[c]), *vp = dev_plan(im, alg, (int)ph[c], (int)oh[c]);
    if (!hp || !vp) {
      err = "resize: out of device memory";
      return G1S_ERR_HIP;
    }
    const uint8_t *src = static_cast<const uint8_t *>(in.data[c]);
    size_t sstride = in.stride_bytes[c];
    if (host_in) {  // (the filter works on the device: stage the host plane)
      sstride = align(pw[c] * bps);
      if (hipMemcpy2DAsync(im->d_in, sstride, in.data[c], in.stride_bytes[c], pw[c] * bps, ph[c], hipMemcpyHostToDevice, im->stream) != hipSuccess) {
        err = "resize: staging a host plane failed (hipMemcpy2DAsync)";
        return G1S_ERR_HIP;
      }
      src = im->d_in;
    }
    uint8_t *dst = im->ring[slot] + off_out[c];
    const dim3 bh(256), gh((ow[c] + 255) / 256, ph[c]), gv((ow[c] + 255) / 256, oh[c]);
    if (bps == 1) {
      hipLaunchKernelGGL(k_resize_h<uint8_t>, gh, bh, 0, im->stream, src, sstride, im->d_tmp, stride_out[c], (int)ow[c], (int)ph[c], hp->host.taps,
                         hp->d_idx, hp->d_coef, maxv);
      hipLaunchKernelGGL(k_resize_v<uint8_t>, gv, bh, 0, im->stream, im->d_tmp, stride_out[c], dst, stride_out[c], (int)ow[c], (int)oh[c],
                         vp->host.taps, vp->d_idx, vp->d_coef, maxv);
    } else {
      hipLaunchKernelGGL(k_resize_h<uint16_t>, gh, bh, 0, im->stream, (const uint16_t *)src, sstride, (uint16_t *)im->d_tmp, stride_out[c],
                         (int)ow[c], (int)ph[c], hp->host.taps, hp->d_idx, hp->d_coef, maxv);
      hipLaunchKernelGGL(k_resize_v<uint16_t>, gv, bh, 0, im->stream, (const uint16_t *)im->d_tmp, stride_out[c], (uint16_t *)dst, stride_out[c],
                         (int)ow[c], (int)oh[c], vp->host.taps, vp->d_idx, vp->d_coef, maxv);
    }
    if (host_in && hipStreamSynchronize(im->stream) != hipSuccess) {  // (d_in is reused by the next plane)
      err = "resize: a plane's kernels failed";
      return G1S_ERR_HIP;
    }
    out.data[c] = dst;
    out.stride_bytes[c] = stride_out[c];
  }
  if (hipStreamSynchronize(im->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
    err = "resize: kernel launch failed";
    return G1S_ERR_HIP;
  }
  return G1S_OK;
}

}  // namespace g1s

extern "C" {

int g1s_resize_plan(const char *alg, uint32_t src, uint32_t dst, uint32_t *taps, int32_t *idx, float *coef, size_t cap) {
  const int a = alg ? g1s::resize_alg_id(alg) : -1;
  if (a < 0 || src == 0 || dst == 0 || !taps) return G1S_ERR_INVALID;
  g1s::ResizePlan p;
  g1s::resize_plan(a, (int)src, (int)dst, p);
  *taps = (uint32_t)p.taps;
  if (p.idx.size() > cap) return G1S_ERR_CAPACITY;
  if (idx) std::memcpy(idx, p.idx.data(), p.idx.size() * sizeof(int32_t));
  if (coef) std::memcpy(coef, p.coef.data(), p.coef.size() * sizeof(float));
  return G1S_OK;
}

int g1s_resize_frame_to_host(const char *alg, const g1s_frame_t *in, uint32_t bit_depth, uint32_t out_w, uint32_t out_h, int32_t device,
                             void *const out_planes[3], const size_t out_stride_bytes[3], char *err, size_t errcap) {
  const int a = alg ? g1s::resize_alg_id(alg) : -1;
  if (a < 0 || !in || !out_planes || !out_stride_bytes) return G1S_ERR_INVALID;
  g1s::ResizeState st(a);
  g1s_frame_t out;
  std::string why;
  const int rc = st.run(*in, bit_depth, out_w, out_h, device, 0, out, why);
  if (rc) {
    if (err && errcap) snprintf(err, errcap, "%s", why.c_str());
    return rc;
  }
  for (uint32_t c = 0; c < in->nplanes; ++c) {
    const uint32_t w = c ? out_w >> in->xdec : out_w, h = c ? out_h >> in->ydec : out_h;
    if (hipMemcpy2D(out_planes[c], out_stride_bytes[c], out.data[c], out.stride_bytes[c], (size_t)w * in->bytes_per_sample, h,
                    hipMemcpyDeviceToHost) != hipSuccess)
      return G1S_ERR_HIP;
  }
  return G1S_OK;
}

}  // extern "C"

// resize.h -- the device resize filter (resize.hip) as filters.cpp / ingest.cpp see it.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/g1s_diff.h"

namespace g1s {

struct ResizePlan {
  int src = 0, dst = 0, taps = 0;
  std::vector<int> idx;     // [dst][taps] input sample of a tap
  std::vector<float> coef;  // [dst][taps]
};
int resize_alg_id(const char *alg);  // hermite 0, catmullrom 1, mitchell 2, lanczos 3, spline36 4; -1 unknown
double resize_kernel(int alg, double x);
double resize_support(int alg);
void resize_plan(int alg, int src, int dst, ResizePlan &p);

// The state of one resize filter of a chain: the taps of the geometries it has seen, staging buffers, and a ring of output
// frames on the device (slot s stays valid until run() is called with s again).
struct ResizeState {
  struct Impl;
  int alg;
  Impl *im;
  explicit ResizeState(int alg);
  ~ResizeState();
  ResizeState(const ResizeState &) = delete;
  ResizeState &operator=(const ResizeState &) = delete;
  int run(const g1s_frame_t &in, uint32_t bit_depth, uint32_t out_w, uint32_t out_h, int device, int slot, g1s_frame_t &out, std::string &err);
};

}  // namespace g1s

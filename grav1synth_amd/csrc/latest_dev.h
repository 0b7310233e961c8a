// latest_dev.h -- the per-frame half of the fold on the device (latest.hip): a frame's integer record, where the
// accumulation kernels left it in HBM, -> the frame's latest noise state as the 27 KB blob of fold.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "record.h"

namespace g1s {

struct LatestJob {
  const uint8_t *records;  // device, frame i at records + i * L.size (the AR sums' upper triangles, as k3m_finish writes them)
  RecLayout L;
  uint8_t *blobs;  // device, frame i at blobs + i * blob_bytes
  size_t blob_bytes;
  uint8_t *scratch;  // device, frame i at scratch + i * scratch_bytes (latest_scratch_bytes)
  size_t scratch_bytes;
  int lag, n, nplanes;
  int W, H, xdec, ydec, nbw, nbh;
};

size_t latest_scratch_bytes(uint32_t nblocks);
// one workgroup per frame; nothing else is read or written
hipError_t launch_latest(const LatestJob &job, int frames, hipStream_t stream);
const char *latest_kernel_name();

}  // namespace g1s

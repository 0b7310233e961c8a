// record.h -- layout of the per-frame integer record produced by the HIP
// kernels and consumed by the ordered host fold (and exchanged between GPUs in
// frame-shard mode).  Everything in it is an exact integer (or the f32 score /
// mask bytes of the flat-block finder), so it is independent of how the frame's
// pixels were partitioned over workgroups, waves or GPUs.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace g1s {

constexpr uint32_t kRecMagic = 0x31533147u;  // "G1S1"
constexpr int kBlock = 32;                   // av1-grain BLOCK_SIZE
constexpr int kMaxN = 25;                    // NUM_UV_COEFFS

struct RecHeader {
  uint32_t magic;
  uint32_t lag;
  uint32_t width, height;
  uint32_t xdec, ydec, nplanes;
  uint32_t nbw, nbh;
  uint32_t n;  // luma AR coefficients: ((2*lag+1)^2)/2
  uint32_t status;  // 0 ok; filled by the device: number of flat blocks
  uint32_t reserved;
  uint64_t size_bytes;
};
static_assert(sizeof(RecHeader) == 56, "RecHeader layout");

struct RecLayout {
  uint32_t nblocks;
  uint32_t n;
  uint32_t nplanes;
  size_t off_ar[3];      // int64: S[nc*nc], Sb[nc], nobs
  size_t off_luma_sum;   // u32[nblocks]
  size_t off_sum_d[3];   // i32[nblocks]
  size_t off_sum_d2[3];  // u32[nblocks]
  size_t off_scores;     // f32[nblocks]
  size_t off_mask;       // u8[nblocks]
  size_t size;
};

inline uint32_t num_coeffs(uint32_t lag) {
  const uint32_t s = 2 * lag + 1;
  return (s * s) / 2;
}

inline size_t align8(size_t v) { return (v + 7) & ~size_t(7); }

inline RecLayout make_layout(uint32_t width, uint32_t height, uint32_t nplanes, uint32_t lag) {
  RecLayout L{};
  const uint32_t nbw = (width + kBlock - 1) / kBlock, nbh = (height + kBlock - 1) / kBlock;
  L.nblocks = nbw * nbh;
  L.n = num_coeffs(lag);
  L.nplanes = nplanes;
  size_t off = align8(sizeof(RecHeader));
  for (uint32_t c = 0; c < 3; ++c) {
    L.off_ar[c] = off;
    if (c < nplanes) {
      const size_t nc = L.n + (c > 0);
      off += sizeof(int64_t) * (nc * nc + nc + 1);
    }
  }
  L.off_luma_sum = off;
  off += align8(sizeof(uint32_t) * L.nblocks);
  for (uint32_t c = 0; c < 3; ++c) {
    L.off_sum_d[c] = off;
    if (c < nplanes) off += align8(sizeof(int32_t) * L.nblocks);
    L.off_sum_d2[c] = off;
    if (c < nplanes) off += align8(sizeof(uint32_t) * L.nblocks);
  }
  L.off_scores = off;
  off += align8(sizeof(float) * L.nblocks);
  L.off_mask = off;
  off += align8(L.nblocks);
  L.size = align8(off);
  return L;
}

}  // namespace g1s

/* estimate_oracle.c -- CPU restatement of the single-source noise estimator (TEST INFRASTRUCTURE: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never does).
 *
 * `grav1synth estimate` (/root/reference/src/main.rs:534-608, feature "unstable") calls
 * av1_grain::estimate_plane_noise(&frame.y_plane, bit_depth) per frame (:567, :575).  The crate (av1-grain 0.4.2,
 * Cargo.lock:92-95) is absent from the reference tree: PARITY UNPINNED.  Its estimator is a port of libaom's
 * av1_estimate_noise_from_single_plane (av1/encoder/temporal_filter.c), restated here from the published algorithm:
 *   for every interior pixel: Sobel Gx, Gy over the 3x3 neighbourhood; Ga = ROUND_POWER_OF_TWO(|Gx| + |Gy|, bd - 8);
 *   if Ga < 50 (EDGE_THRESHOLD): accum += ROUND_POWER_OF_TWO(|Laplacian|, bd - 8), count += 1;
 *   count < 16 -> None (-1 in the command's output), else accum / (6 count) * SQRT_PI_BY_2.
 * Scalar, one pixel at a time, the matrix built like the C source builds it. */
#include <stdint.h>
#include <stdlib.h>

#define EDGE_THRESHOLD 50
#define SQRT_PI_BY_2 1.2533141373155003
#define ROUND_POWER_OF_TWO(value, n) (((value) + (((1 << (n)) >> 1))) >> (n))

double orc_estimate_plane_noise(const void *plane, size_t stride_bytes, uint32_t width, uint32_t height, uint32_t bit_depth) {
  const int shift = (int)bit_depth - 8;
  int64_t accum = 0;
  uint64_t count = 0;
  for (uint32_t i = 1; i + 1 < height; ++i) {
    for (uint32_t j = 1; j + 1 < width; ++j) {
      int mat[3][3];
      for (int ii = -1; ii <= 1; ++ii) {
        const uint8_t *row = (const uint8_t *)plane + (size_t)(i + ii) * stride_bytes;
        for (int jj = -1; jj <= 1; ++jj)
          mat[ii + 1][jj + 1] = bit_depth > 8 ? (int)((const uint16_t *)row)[j + jj] : (int)row[j + jj];
      }
      const int gx = (mat[0][0] - mat[0][2]) + (mat[2][0] - mat[2][2]) + 2 * (mat[1][0] - mat[1][2]);
      const int gy = (mat[0][0] - mat[2][0]) + (mat[0][2] - mat[2][2]) + 2 * (mat[0][1] - mat[2][1]);
      const int ga = ROUND_POWER_OF_TWO(abs(gx) + abs(gy), shift);
      if (ga < EDGE_THRESHOLD) {
        const int v = 4 * mat[1][1] - 2 * (mat[0][1] + mat[2][1] + mat[1][0] + mat[1][2]) +
                      (mat[0][0] + mat[0][2] + mat[2][0] + mat[2][2]);
        accum += ROUND_POWER_OF_TWO(abs(v), shift);
        ++count;
      }
    }
  }
  return count < 16 ? -1.0 : (double)accum / (double)(6 * count) * SQRT_PI_BY_2;
}

// tools/shard_native.cpp -- the frame-shard round protocol (include/g1s_diff.h, "frame-shard rounds") driven natively:
// ONE process, one generator per visible device (or the first N of them), the per-round gather of the fixed-size messages
// over RCCL (ncclSend / ncclRecv inside a group; communicators from ncclCommInitAll: no bootstrap, no MPI).  This is the
// loop of INTEGRATION.md section 3 as a program that compiles, links and runs: the library itself links no communication
// library -- the transport belongs to the host application, and this is one.
//
//   shard_native [N devices, default all] [frames, default 40] [batch_frames, default 4] [W H, default 352 224]
//
// It feeds a deterministic synthetic 10-bit 4:2:0 video (flat field + noise on the source side, a scene change halfway so
// that the table has two segments), batch j to device j % N, runs rounds + flush rounds until every frame is merged, merges on device 0's host, and
// compares the table, byte for byte, with the one a single generator gives for the same frames.  Exit code 0 = identical.
// Build: hipcc --offload-arch=gfx950 -O2 -I include tools/shard_native.cpp -L grav1synth_amd -lg1s_diff -lrccl -o tools/shard_native
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "g1s_diff.h"

#define CHECK_HIP(e)                                                                  \
  do {                                                                                \
    hipError_t r_ = (e);                                                              \
    if (r_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (%s:%d)\n", #e, hipGetErrorString(r_), __FILE__, __LINE__); \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)
#define CHECK_NCCL(e)                                                                  \
  do {                                                                                 \
    ncclResult_t r_ = (e);                                                             \
    if (r_ != ncclSuccess) {                                                           \
      fprintf(stderr, "%s: %s (%s:%d)\n", #e, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)

namespace {

struct HostFrame {
  std::vector<uint16_t> y, u, v;
};

// flat field + per-pixel noise (source) / flat field (denoised); the level changes at the scene cut
void make_pair(int W, int H, int k, int cut, HostFrame &s, HostFrame &d) {
  const int cw = W / 2, ch = H / 2;
  s.y.resize((size_t)W * H), d.y.resize((size_t)W * H);
  s.u.resize((size_t)cw * ch), d.u.resize((size_t)cw * ch);
  s.v.resize((size_t)cw * ch), d.v.resize((size_t)cw * ch);
  uint32_t x = 12345u + 7919u * (uint32_t)k;
  auto rnd = [&]() {
    x = x * 1664525u + 1013904223u;
    return (int)((x >> 16) & 0xff);
  };
  const int amp = k < cut ? 12 : 40, base = k < cut ? 400 : 560;
  for (size_t i = 0; i < s.y.size(); ++i) {
    const int n = (rnd() * amp >> 8) - amp / 2 + (rnd() * amp >> 8) - amp / 2;
    d.y[i] = (uint16_t)base;
    s.y[i] = (uint16_t)(base + n);
  }
  for (size_t i = 0; i < s.u.size(); ++i) {
    d.u[i] = 512, d.v[i] = 512;
    s.u[i] = (uint16_t)(512 + (rnd() * amp >> 9) - amp / 4);
    s.v[i] = (uint16_t)(512 + (rnd() * amp >> 9) - amp / 4);
  }
}

g1s_frame_t describe(const HostFrame &f, int W, int H) {
  g1s_frame_t r;
  std::memset(&r, 0, sizeof r);
  r.width = (uint32_t)W, r.height = (uint32_t)H;
  r.bytes_per_sample = 2, r.xdec = 1, r.ydec = 1, r.nplanes = 3;
  r.data[0] = f.y.data(), r.data[1] = f.u.data(), r.data[2] = f.v.data();
  r.stride_bytes[0] = (size_t)W * 2, r.stride_bytes[1] = r.stride_bytes[2] = (size_t)(W / 2) * 2;
  r.on_device = 0;  // host planes: copied before the call returns
  return r;
}

std::string table_of(const std::vector<g1s_segment_t> &segs) {
  std::string s(1 << 20, '\0');
  const long n = g1s_format_tbl(segs.data(), segs.size(), &s[0], s.size());
  s.resize(n > 0 ? (size_t)n : 0);
  return s;
}

}  // namespace

int main(int argc, char **argv) {
  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  int N = argc > 1 ? atoi(argv[1]) : ndev;
  if (N <= 0 || N > ndev) N = ndev;
  const int frames = argc > 2 ? atoi(argv[2]) : 40;
  const uint32_t B = argc > 3 ? (uint32_t)atoi(argv[3]) : 4u;
  const int W = argc > 5 ? atoi(argv[4]) : 352, H = argc > 5 ? atoi(argv[5]) : 224;
  const int cut = frames / 2 + 1;
  const int64_t fps_num = 24000, fps_den = 1001;

  std::vector<HostFrame> src(frames), den(frames);
  std::vector<g1s_frame_t> fs(frames), fd(frames);
  for (int k = 0; k < frames; ++k) {
    make_pair(W, H, k, cut, src[k], den[k]);
    fs[k] = describe(src[k], W, H);
    fd[k] = describe(den[k], W, H);
  }

  // ---- the reference: one generator, the whole video ----
  std::string want;
  {
    g1s_opts_t o = {sizeof(g1s_opts_t), 0, 3, 0, B, 0};
    g1s_diff_t *g = g1s_diff_new(fps_num, fps_den, 10, 10, &o);
    if (!g) {
      fprintf(stderr, "g1s_diff_new: %s\n", g1s_last_global_error());
      return 2;
    }
    if (g1s_diff_frames(g, fs.data(), fd.data(), (size_t)frames)) {
      fprintf(stderr, "single generator: %s\n", g1s_diff_last_error(g));
      return 2;
    }
    std::vector<g1s_segment_t> segs(64);
    size_t n = 0;
    if (g1s_diff_finish(g, segs.data(), segs.size(), &n)) {
      fprintf(stderr, "single generator finish: %s\n", g1s_diff_last_error(g));
      return 2;
    }
    segs.resize(n);
    want = table_of(segs);
    g1s_diff_free(g);
  }

  // ---- N ranks in this process: a generator, a stream and an RCCL communicator per device ----
  std::vector<int> devs(N);
  for (int r = 0; r < N; ++r) devs[r] = r;
  std::vector<ncclComm_t> comm(N);
  CHECK_NCCL(ncclCommInitAll(comm.data(), N, devs.data()));
  const size_t msg_bytes = g1s_shard_msg_size(3, B);
  std::vector<g1s_diff_t *> gen(N);
  std::vector<hipStream_t> stream(N);
  std::vector<uint8_t *> d_msg(N), h_msg(N);
  uint8_t *d_all = nullptr, *h_all = nullptr;
  for (int r = 0; r < N; ++r) {
    CHECK_HIP(hipSetDevice(r));
    g1s_opts_t o = {sizeof(g1s_opts_t), r, 3, 0, B, /* records_only = */ 2};
    gen[r] = g1s_diff_new(fps_num, fps_den, 10, 10, &o);
    if (!gen[r]) {
      fprintf(stderr, "g1s_diff_new on device %d: %s\n", r, g1s_last_global_error());
      return 2;
    }
    CHECK_HIP(hipStreamCreateWithFlags(&stream[r], hipStreamNonBlocking));
    CHECK_HIP(hipMalloc((void **)&d_msg[r], msg_bytes));
    CHECK_HIP(hipHostMalloc((void **)&h_msg[r], msg_bytes, hipHostMallocDefault));
  }
  CHECK_HIP(hipSetDevice(0));
  CHECK_HIP(hipMalloc((void **)&d_all, msg_bytes * N));
  CHECK_HIP(hipHostMalloc((void **)&h_all, msg_bytes * N, hipHostMallocDefault));
  g1s_fold_t *fold = g1s_fold_new(fps_num, fps_den, 3);

  const size_t nbatches = ((size_t)frames + B - 1) / B, rounds = (nbatches + N - 1) / N;
  // (flush rounds behind the feeding rounds until every frame fed has been merged -- what is still in the generators'
  //  pipelines: usually 4 rounds; a count that does not arrive is an error, not a shorter table)
  size_t flush_rounds = 0;
  for (size_t k = 0; k < rounds || (g1s_fold_frames(fold) < (uint64_t)frames && flush_rounds < 64); ++k) {
    if (k >= rounds) ++flush_rounds;
    for (int r = 0; r < N; ++r) {
      CHECK_HIP(hipSetDevice(r));
      const size_t j = k * N + r;
      if (k < rounds && j < nbatches) {
        const size_t first = j * B, cnt = std::min<size_t>(B, (size_t)frames - first);
        if (g1s_diff_frames(gen[r], fs.data() + first, fd.data() + first, cnt)) {
          fprintf(stderr, "rank %d: %s\n", r, g1s_diff_last_error(gen[r]));
          return 2;
        }
      }
      if (g1s_shard_pack(gen[r], /* flush = */ k >= rounds, h_msg[r], msg_bytes)) {
        fprintf(stderr, "rank %d pack: %s\n", r, g1s_diff_last_error(gen[r]));
        return 2;
      }
      CHECK_HIP(hipMemcpyAsync(d_msg[r], h_msg[r], msg_bytes, hipMemcpyHostToDevice, stream[r]));
    }
    // the transport: ONE rooted gather of msg_bytes per rank, over RCCL
    CHECK_NCCL(ncclGroupStart());
    for (int r = 0; r < N; ++r) {
      CHECK_NCCL(ncclSend(d_msg[r], msg_bytes, ncclUint8, 0, comm[r], stream[r]));
      CHECK_NCCL(ncclRecv(d_all + (size_t)r * msg_bytes, msg_bytes, ncclUint8, r, comm[0], stream[0]));
    }
    CHECK_NCCL(ncclGroupEnd());
    for (int r = 0; r < N; ++r) {
      CHECK_HIP(hipSetDevice(r));
      CHECK_HIP(hipStreamSynchronize(stream[r]));
    }
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMemcpy(h_all, d_all, msg_bytes * N, hipMemcpyDeviceToHost));
    if (g1s_shard_merge(fold, h_all, msg_bytes, (uint32_t)N)) {  // global frame order
      fprintf(stderr, "merge: %s\n", g1s_fold_last_error(fold));
      return 2;
    }
  }
  if (g1s_fold_frames(fold) != (uint64_t)frames) {
    fprintf(stderr, "merged %llu of %d frames after %zu flush rounds\n", (unsigned long long)g1s_fold_frames(fold), frames, flush_rounds);
    return 2;
  }
  std::vector<g1s_segment_t> segs(64);
  size_t n = 0;
  if (g1s_fold_finish(fold, segs.data(), segs.size(), &n)) {
    fprintf(stderr, "fold finish: %s\n", g1s_fold_last_error(fold));
    return 2;
  }
  segs.resize(n);
  const std::string got = table_of(segs);
  for (int r = 0; r < N; ++r) {
    g1s_diff_free(gen[r]);
    ncclCommDestroy(comm[r]);
  }
  g1s_fold_free(fold);
  const bool same = got == want;
  printf("shard_native: %d device(s), %d frames in batches of %u, %zu rounds + %zu flush rounds: %zu segment(s), table %s the single generator's (%zu bytes)\n",
         N, frames, B, rounds, flush_rounds, n, same ? "IDENTICAL to" : "DIFFERS from", got.size());
  if (!same) fprintf(stderr, "--- sharded\n%s--- single\n%s", got.c_str(), want.c_str());
  return same ? 0 : 1;
}

// ingest.cpp -- the caller side of the diff path: the frame-pair loop of `grav1synth diff`
// (reference: src/main.rs:414-531, get_filtered_frame_pair at :615-629) over pull-model frame
// sources, and a YUV4MPEG2 source that stands where the reference's libav BitstreamReader
// (src/reader.rs:37-212) stands: raw planar frames from a file into PINNED host memory, read
// ahead by a thread per file, so that file IO, the H2D copies of g1s_diff_frame and the kernels
// of earlier frames overlap.
//
// Plain C ABI (include/g1s_diff.h).  No pixel arithmetic happens here.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g1s_diff.h"

namespace {

void set_err(char *err, size_t cap, const std::string &msg) {
  if (!err || !cap) return;
  std::snprintf(err, cap, "%s", msg.c_str());
}

struct PinnedBuf {
  uint8_t *p = nullptr;
  bool pinned = false;
  void alloc(size_t n) {
    // pinned when a HIP device is there (DMA straight out of it); plain memory otherwise, so that
    // the reader itself also works on a machine without a GPU (header / frame-count tools, tests)
    void *q = nullptr;
    if (hipHostMalloc(&q, n, hipHostMallocDefault) == hipSuccess && q) {
      p = static_cast<uint8_t *>(q);
      pinned = true;
    } else {
      (void)hipGetLastError();
      p = static_cast<uint8_t *>(std::malloc(n));
      pinned = false;
    }
  }
  void release() {
    if (!p) return;
    if (pinned) (void)hipHostFree(p);
    else std::free(p);
    p = nullptr;
  }
};

constexpr int kRing = 6;  // pinned frame buffers per file: read ahead + lent to the generator until their copies have run
constexpr int kMaxReadThreads = 16;
// positional reads in flight per frame (G1S_Y4M_READ_THREADS: tuning aid)
static const int kReadThreads = [] {
  const char *e = getenv("G1S_Y4M_READ_THREADS");
  const int n = e ? atoi(e) : 3;  // (measured: 1 -> 22, 2 -> 25..32, 3 -> 31, 4 -> 28, 8 -> 26 GB/s through two readers)
  return n < 1 ? 1 : (n > kMaxReadThreads ? kMaxReadThreads : n);
}();

}  // namespace

struct g1s_y4m {
  FILE *f = nullptr;
  g1s_y4m_info_t info{};
  size_t plane_bytes[3] = {0, 0, 0}, plane_off[3] = {0, 0, 0}, frame_bytes = 0;
  size_t row_bytes[3] = {0, 0, 0};
  PinnedBuf ring[kRing];
  // reader thread -> consumer
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<int> ready;  // ring indices holding a frame
  int free_slots = kRing;
  int held = -1;          // ring index lent to the consumer by the last g1s_y4m_next
  // bound to a generator (g1s_y4m_bind): frames go out as on_device = 2 and stay lent until their copies have run
  g1s_diff_t *bound = nullptr;
  std::deque<std::pair<int, uint64_t>> lent;  // (ring index, frame pairs the generator must have copied before it is free)
  uint64_t frames_out = 0;
  bool eof = false, stop = false, failed = false;
  std::string error;
  uint64_t frames_read = 0;

  bool read_frame_into(uint8_t *dst) {
    // "FRAME" [ params ] '\n'
    char line[256];
    int c = std::fgetc(f);
    if (c == EOF) return false;  // clean end of stream
    std::ungetc(c, f);
    if (!std::fgets(line, sizeof line, f) || std::strncmp(line, "FRAME", 5) != 0) {
      failed = true;
      error = "y4m: FRAME marker expected at frame " + std::to_string(frames_read);
      return false;
    }
    // the payload: big frames in kReadThreads positional reads side by side (one fread of a 25 MB frame out of
    // the page cache is a single-core memcpy, ~8 GB/s; the H2D copy behind it takes 50+)
    const off_t at = ftello(f);
    bool ok = true;
    if (frame_bytes >= (size_t)kReadThreads << 20 && at >= 0) {
      const int fd = fileno(f);
      const size_t chunk = ((frame_bytes + kReadThreads - 1) / kReadThreads + 4095) & ~size_t(4095);
      bool good[kMaxReadThreads];
      auto part = [&](int t) {
        size_t o = std::min(frame_bytes, chunk * (size_t)t), end = std::min(frame_bytes, o + chunk);
        good[t] = true;
        while (o < end) {
          const ssize_t n = pread(fd, dst + o, end - o, at + (off_t)o);
          if (n <= 0) {
            good[t] = false;
            return;
          }
          o += (size_t)n;
        }
      };
      std::thread helpers[kMaxReadThreads - 1];
      for (int t = 1; t < kReadThreads; ++t) helpers[t - 1] = std::thread(part, t);
      part(0);
      for (int t = 1; t < kReadThreads; ++t) helpers[t - 1].join();
      for (int t = 0; t < kReadThreads; ++t) ok = ok && good[t];
      if (ok && fseeko(f, at + (off_t)frame_bytes, SEEK_SET) != 0) ok = false;
    } else {
      ok = std::fread(dst, 1, frame_bytes, f) == frame_bytes;
    }
    if (!ok) {
      failed = true;
      error = "y4m: truncated frame " + std::to_string(frames_read);
      return false;
    }
    return true;
  }

  void reader_main() {
    int next = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || free_slots > 0; });
        if (stop) return;
        --free_slots;
      }
      const bool ok = read_frame_into(ring[next].p);
      {
        std::lock_guard<std::mutex> lk(m);
        if (ok) {
          ready.push_back(next);
          ++frames_read;
        } else {
          eof = true;
        }
      }
      cv.notify_all();
      if (!ok) return;
      next = (next + 1) % kRing;
    }
  }
};

extern "C" {

g1s_y4m_t *g1s_y4m_open(const char *path, char *err, size_t errcap) {
  if (!path) {
    set_err(err, errcap, "y4m: null path");
    return nullptr;
  }
  FILE *f = std::fopen(path, "rb");
  if (!f) {
    set_err(err, errcap, std::string("y4m: cannot open ") + path);
    return nullptr;
  }
  char hdr[1024];
  if (!std::fgets(hdr, sizeof hdr, f) || std::strncmp(hdr, "YUV4MPEG2", 9) != 0) {
    std::fclose(f);
    set_err(err, errcap, std::string("y4m: not a YUV4MPEG2 stream: ") + path);
    return nullptr;
  }
  g1s_y4m_info_t info{};
  info.fps_num = 25;  // the format's defaults when a field is absent
  info.fps_den = 1;
  info.bit_depth = 8;
  info.xdec = 1;
  info.ydec = 1;
  info.nplanes = 3;
  std::string cs = "420";
  for (char *tok = std::strtok(hdr + 9, " \n\r"); tok; tok = std::strtok(nullptr, " \n\r")) {
    switch (tok[0]) {
      case 'W': info.width = (uint32_t)std::strtoul(tok + 1, nullptr, 10); break;
      case 'H': info.height = (uint32_t)std::strtoul(tok + 1, nullptr, 10); break;
      case 'F': {
        long long n = 0, d = 0;
        if (std::sscanf(tok + 1, "%lld:%lld", &n, &d) == 2 && n > 0 && d > 0) {
          info.fps_num = n;
          info.fps_den = d;
        }
        break;
      }
      case 'C': cs = tok + 1; break;
      default: break;  // interlacing, aspect, comments: not needed by the estimator
    }
  }
  // colour space tag -> subsampling and depth: C420jpeg / C420mpeg2 / C420paldv / C420 / C422 /
  // C444 / Cmono, with an optional p9..p16 depth suffix (the pixel formats src/reader.rs:51-85 maps)
  std::string base = cs;
  const size_t pp = cs.find('p', 3);
  if (cs.compare(0, 4, "mono") == 0) {
    info.nplanes = 1;
    info.xdec = info.ydec = 0;
    if (cs.size() > 4) info.bit_depth = (uint32_t)std::strtoul(cs.c_str() + 4, nullptr, 10);
  } else {
    if (pp != std::string::npos && pp + 1 < cs.size() && cs[pp + 1] >= '0' && cs[pp + 1] <= '9') {
      info.bit_depth = (uint32_t)std::strtoul(cs.c_str() + pp + 1, nullptr, 10);
      base = cs.substr(0, pp);
    }
    if (base.compare(0, 3, "420") == 0) info.xdec = 1, info.ydec = 1;
    else if (base.compare(0, 3, "422") == 0) info.xdec = 1, info.ydec = 0;
    else if (base.compare(0, 3, "444") == 0) info.xdec = 0, info.ydec = 0;
    else {
      std::fclose(f);
      set_err(err, errcap, "y4m: unsupported colour space C" + cs);
      return nullptr;
    }
  }
  if (info.width == 0 || info.height == 0 || info.bit_depth < 8 || info.bit_depth > 16) {
    std::fclose(f);
    set_err(err, errcap, "y4m: bad header (size or bit depth)");
    return nullptr;
  }
  g1s_y4m *y = new g1s_y4m;
  y->f = f;
  y->info = info;
  const size_t bps = info.bit_depth > 8 ? 2 : 1;
  size_t off = 0;
  for (uint32_t c = 0; c < info.nplanes; ++c) {
    // chroma planes: ceil(width / 2^xdec) x ceil(height / 2^ydec) samples in the file
    const size_t fw = c ? (info.width + (1u << info.xdec) - 1) >> info.xdec : info.width;
    const size_t fh = c ? (info.height + (1u << info.ydec) - 1) >> info.ydec : info.height;
    y->row_bytes[c] = fw * bps;
    y->plane_bytes[c] = fw * fh * bps;
    y->plane_off[c] = off;
    off += y->plane_bytes[c];
  }
  y->frame_bytes = off;
  for (auto &b : y->ring) {
    b.alloc(y->frame_bytes);
    if (!b.p) {
      set_err(err, errcap, "y4m: out of memory");
      g1s_y4m_close(y);
      return nullptr;
    }
  }
  y->th = std::thread([y] { y->reader_main(); });
  return y;
}

int g1s_y4m_get_info(const g1s_y4m_t *y, g1s_y4m_info_t *out) {
  if (!y || !out) return G1S_ERR_INVALID;
  *out = y->info;
  return G1S_OK;
}

int g1s_y4m_next(void *user, g1s_frame_t *out) {
  g1s_y4m *y = static_cast<g1s_y4m *>(user);
  if (!y || !out) return G1S_ERR_INVALID;
  int idx;
  {
    std::unique_lock<std::mutex> lk(y->m);
    if (y->held >= 0) {  // the frame lent by the previous call goes back to the reader ...
      if (y->bound) {
        y->lent.emplace_back(y->held, y->frames_out);  // ... once the generator has copied it (frame pair frames_out - 1)
      } else {
        ++y->free_slots;
        y->cv.notify_all();
      }
      y->held = -1;
    }
    if (y->bound) {
      // (a) what has been copied meanwhile; (b) never sit on more than half of the ring: wait for the oldest copy
      lk.unlock();
      uint64_t copied = g1s_diff_frames_copied(y->bound, 0);
      lk.lock();
      for (;;) {
        while (!y->lent.empty() && y->lent.front().second <= copied) {
          y->lent.pop_front();
          ++y->free_slots;
          y->cv.notify_all();
        }
        if ((int)y->lent.size() <= kRing / 2) break;
        const uint64_t need = y->lent.front().second;
        lk.unlock();
        copied = g1s_diff_frames_copied(y->bound, need);
        lk.lock();
        if (copied < need) break;  // (the frame never reached the generator: nothing to wait for)
      }
    }
    y->cv.wait(lk, [&] { return !y->ready.empty() || y->eof; });
    if (y->ready.empty()) return y->failed ? G1S_ERR_INVALID : 0;  // 0: end of stream
    idx = y->ready.front();
    y->ready.pop_front();
    y->held = idx;
    ++y->frames_out;
  }
  std::memset(out, 0, sizeof *out);
  out->width = y->info.width;
  out->height = y->info.height;
  out->bytes_per_sample = y->info.bit_depth > 8 ? 2 : 1;
  out->xdec = (uint8_t)y->info.xdec;
  out->ydec = (uint8_t)y->info.ydec;
  out->nplanes = (uint8_t)y->info.nplanes;
  for (uint32_t c = 0; c < y->info.nplanes; ++c) {
    out->data[c] = y->ring[idx].p + y->plane_off[c];
    out->stride_bytes[c] = y->row_bytes[c];
  }
  out->on_device = y->bound ? 2 : 0;
  return 1;
}

int g1s_y4m_bind(g1s_y4m_t *y, g1s_diff_t *g) {
  if (!y) return G1S_ERR_INVALID;
  std::lock_guard<std::mutex> lk(y->m);
  if (y->frames_out && g != y->bound) return G1S_ERR_STATE;  // (frame i of the reader = frame pair i of the generator)
  y->bound = g;
  return G1S_OK;
}

const char *g1s_y4m_last_error(const g1s_y4m_t *y) { return y ? y->error.c_str() : ""; }

void g1s_y4m_close(g1s_y4m_t *y) {
  if (!y) return;
  {
    std::lock_guard<std::mutex> lk(y->m);
    y->stop = true;
  }
  y->cv.notify_all();
  if (y->th.joinable()) y->th.join();
  for (auto &b : y->ring) b.release();
  if (y->f) std::fclose(y->f);
  delete y;
}

// (engine.hip) replaces the generator's error text: the loop below prefixes errors with the index of the frame pair
extern "C" void g1s_diff_set_error_text_(g1s_diff_t *, const char *);
extern "C" uint32_t g1s_diff_source_bit_depth_(const g1s_diff_t *);
extern "C" int32_t g1s_diff_device_(const g1s_diff_t *);
extern "C" uint32_t g1s_diff_frames_in_flight_max_(const g1s_diff_t *);

int g1s_diff_run_filtered(g1s_diff_t *g, g1s_next_frame_fn source, void *source_user, g1s_next_frame_fn denoised,
                          void *denoised_user, const g1s_filters_t *filters, uint64_t *frames_out, int *unequal_out) {
  if (!g || !source || !denoised) return G1S_ERR_INVALID;
  uint64_t frames = 0;
  int unequal = 0;
  int rc = G1S_OK;
  const bool resizes = filters && g1s_filters_has_resize(filters);
  // resized frames that may be inside the generator at once: its slots x its launch group, and one more group so that a slot is
  // rarely waited for (4K: 5 x 64 frames = 8 GB of device memory, 8K 4:4:4: 5 x 16 = 16 GB; a fixed 512 was 12.7 / 100 GB).
  // Frame 0 goes to slot 0 before the generator has chosen its group: any ring agrees on that.
  uint64_t kResizeRing = 512;
  auto note = [&](const std::string &what) {
    g1s_diff_set_error_text_(g, ("frame " + std::to_string(frames) + ": " + what).c_str());
  };
  for (;;) {
    // get_filtered_frame_pair (src/main.rs:615-629): one frame from each reader, source first; the filter chain on
    // the source frame only
    g1s_frame_t s, d;
    const int rs = source(source_user, &s);
    const int rd = denoised(denoised_user, &d);
    if (rs < 0 || rd < 0) {
      rc = rs < 0 ? rs : rd;
      note(std::string(rs < 0 ? "source" : "denoised") + " reader failed");
      break;
    }
    if (rs == 0 && rd == 0) break;  // (None, None)
    if (rs == 0 || rd == 0) {       // "Videos did not have equal frame counts. Resulting grain table may
      unequal = 1;                  //  not be as expected." -- a warning, then the loop ends (src/main.rs:449-455)
      break;
    }
    if (filters) {
      char ferr[320] = "";
      g1s_frame_t cropped;
      // a resized frame lives in slot (frame index mod ring) of the chain's ring; the generator reads device frames in
      // place, so a slot is taken again only when the frame that was there has been released
      uint32_t slot = 0;
      if (resizes) {
        if (frames == 1) {
          const uint32_t inside = g1s_diff_frames_in_flight_max_(g);
          if (inside) kResizeRing = std::min<uint64_t>(512, inside + inside / 4);
        }
        slot = (uint32_t)(frames % kResizeRing);
        if (frames >= kResizeRing && g1s_diff_frames_released(g) + kResizeRing <= frames) {
          rc = g1s_diff_sync(g);
          if (rc) {
            note(std::string("diff_frame: ") + g1s_diff_last_error(g));
            break;
          }
        }
      }
      rc = g1s_filters_apply_bd(filters, &s, g1s_diff_source_bit_depth_(g), g1s_diff_device_(g), slot, &cropped, ferr, sizeof(ferr));
      if (rc) {
        note(ferr);
        break;
      }
      s = cropped;
    }
    rc = g1s_diff_frame(g, &s, &d);  // `?`: the first error ends the command
    if (rc) {
      // (an error of an EARLIER, queued batch surfaces here too: the text says which call, the engine's says what)
      note(std::string("diff_frame: ") + g1s_diff_last_error(g));
      break;
    }
    ++frames;
  }
  if (frames_out) *frames_out = frames;
  if (unequal_out) *unequal_out = unequal;
  return rc;
}

int g1s_diff_run(g1s_diff_t *g, g1s_next_frame_fn source, void *source_user, g1s_next_frame_fn denoised,
                 void *denoised_user, uint64_t *frames_out, int *unequal_out) {
  return g1s_diff_run_filtered(g, source, source_user, denoised, denoised_user, nullptr, frames_out, unequal_out);
}

int g1s_diff_y4m_files(const char *source_path, const char *denoised_path, const char *out_tbl_path,
                       const g1s_opts_t *opts, uint64_t *frames_out, int *unequal_out, char *err, size_t errcap) {
  return g1s_diff_y4m_files_filtered(source_path, denoised_path, out_tbl_path, opts, nullptr, frames_out, unequal_out, err, errcap);
}

int g1s_diff_y4m_files_filtered(const char *source_path, const char *denoised_path, const char *out_tbl_path,
                                const g1s_opts_t *opts, const char *filter_text, uint64_t *frames_out, int *unequal_out,
                                char *err, size_t errcap) {
  g1s_filters_t *filters = nullptr;
  if (filter_text && *filter_text) {  // src/main.rs:370-380: "Invalid filter chain: {e}", nothing is opened
    char ferr[256] = "";
    filters = g1s_filters_new(filter_text, ferr, sizeof(ferr));
    if (!filters) {
      set_err(err, errcap, std::string("Invalid filter chain: ") + ferr);
      return G1S_ERR_INVALID;
    }
  }
  g1s_y4m_t *ys = g1s_y4m_open(source_path, err, errcap);
  if (!ys) {
    g1s_filters_free(filters);
    return G1S_ERR_INVALID;
  }
  g1s_y4m_t *yd = g1s_y4m_open(denoised_path, err, errcap);
  if (!yd) {
    g1s_y4m_close(ys);
    g1s_filters_free(filters);
    return G1S_ERR_INVALID;
  }
  int rc = G1S_OK;
  g1s_diff_t *g = nullptr;
  std::vector<g1s_segment_t> segs(64);
  size_t n = 0;
  // the frame rate and the bit depths come from the readers (src/main.rs:414-427)
  g = g1s_diff_new(ys->info.fps_num, ys->info.fps_den, ys->info.bit_depth, yd->info.bit_depth, opts);
  if (!g) {
    set_err(err, errcap, g1s_last_global_error());
    rc = G1S_ERR_NO_DEVICE;
    goto done;
  }
  // the readers' pinned rings feed the generator directly: copies are queued, the loop goes on reading
  static const bool sync_ingest = getenv("G1S_INGEST_SYNC") != nullptr;  // comparison aid (tools/bench_ingest.py)
  if (!sync_ingest) {
    g1s_y4m_bind(ys, g);
    g1s_y4m_bind(yd, g);
  }
  rc = g1s_diff_run_filtered(g, g1s_y4m_next, ys, g1s_y4m_next, yd, filters, frames_out, unequal_out);
  if (rc) {
    std::string e = g1s_diff_last_error(g);  // "frame N: ..."
    if (ys->failed) e += " (" + ys->error + ")";
    else if (yd->failed) e += " (" + yd->error + ")";
    set_err(err, errcap, e);
    goto done;
  }
  rc = g1s_diff_finish(g, segs.data(), segs.size(), &n);
  if (rc == G1S_ERR_CAPACITY) {  // more scene cuts than the first guess: the segments are still there, ask again
    segs.resize(n);
    rc = g1s_diff_finish(g, segs.data(), segs.size(), &n);
  }
  if (rc) {
    set_err(err, errcap, g1s_diff_last_error(g));
    goto done;
  }
  rc = g1s_write_tbl(out_tbl_path, segs.data(), n);
  if (rc) set_err(err, errcap, std::string("cannot write ") + out_tbl_path);
done:
  g1s_filters_free(filters);
  if (g) g1s_diff_free(g);
  g1s_y4m_close(ys);
  g1s_y4m_close(yd);
  return rc;
}


// `grav1synth diff` over several devices (north_star: "frames shard naturally across the GPUs ... so it drops in for that
// subcommand"; the loop of src/main.rs:414-531).  ONE process: a records_only = 2 generator per entry of `devices` (an
// ordinal may repeat: two generators on one device), the two readers' frame pairs dealt batch by batch (batch j of the
// video -> generator j % n), per round one g1s_shard_pack per generator and one g1s_shard_merge -- the round protocol of
// include/g1s_diff.h with the host as the transport (the messages never leave this process).  The table is the single
// generator's, byte for byte.
int g1s_diff_y4m_files_sharded(const char *source_path, const char *denoised_path, const char *out_tbl_path,
                               const g1s_opts_t *opts, const char *filter_text, const int32_t *devices, uint32_t n_devices,
                               uint64_t *frames_out, int *unequal_out, char *err, size_t errcap) {
  if (!devices || n_devices == 0) {
    set_err(err, errcap, "no devices");
    return G1S_ERR_INVALID;
  }
  g1s_filters_t *filters = nullptr;
  if (filter_text && *filter_text) {
    char ferr[256] = "";
    filters = g1s_filters_new(filter_text, ferr, sizeof(ferr));
    if (!filters) {
      set_err(err, errcap, std::string("Invalid filter chain: ") + ferr);
      return G1S_ERR_INVALID;
    }
  }
  g1s_y4m_t *ys = g1s_y4m_open(source_path, err, errcap);
  if (!ys) {
    g1s_filters_free(filters);
    return G1S_ERR_INVALID;
  }
  g1s_y4m_t *yd = g1s_y4m_open(denoised_path, err, errcap);
  if (!yd) {
    g1s_y4m_close(ys);
    g1s_filters_free(filters);
    return G1S_ERR_INVALID;
  }
  const uint32_t N = n_devices;
  g1s_opts_t o{};
  o.struct_size = sizeof(g1s_opts_t);
  if (opts) o = *opts;
  o.struct_size = sizeof(g1s_opts_t);
  if (o.ar_coeff_lag == 0) o.ar_coeff_lag = 3;
  if (o.batch_frames == 0) {  // (the engine's default is sized by the frame: ask one generator what it would take)
    const uint64_t px = (uint64_t)ys->info.width * ys->info.height;
    o.batch_frames = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(1, (530ull << 20) / std::max<uint64_t>(px, 1)));
  }
  o.records_only = 2;
  const uint32_t B = o.batch_frames;
  int rc = G1S_OK;
  uint64_t frames = 0;
  int unequal = 0;
  std::vector<g1s_diff_t *> gen(N, nullptr);
  g1s_fold_t *fold = nullptr;
  const size_t msg_bytes = g1s_shard_msg_size(o.ar_coeff_lag, B);
  std::vector<uint8_t> msgs(msg_bytes * N);
  std::vector<g1s_segment_t> segs(64);
  size_t n = 0;
  bool ended = false;
  auto round = [&](int flush) -> int {  // every generator's message of the round, merged in global batch order
    for (uint32_t r = 0; r < N; ++r) {
      const int prc = g1s_shard_pack(gen[r], flush, msgs.data() + (size_t)r * msg_bytes, msg_bytes);
      if (prc) {
        set_err(err, errcap, std::string("generator ") + std::to_string(r) + ": " + g1s_diff_last_error(gen[r]));
        return prc;
      }
    }
    const int mrc = g1s_shard_merge(fold, msgs.data(), msg_bytes, N);
    if (mrc) set_err(err, errcap, std::string("merge: ") + g1s_fold_last_error(fold));
    return mrc;
  };
  for (uint32_t r = 0; r < N; ++r) {
    o.device = devices[r];
    gen[r] = g1s_diff_new(ys->info.fps_num, ys->info.fps_den, ys->info.bit_depth, yd->info.bit_depth, &o);
    if (!gen[r]) {
      set_err(err, errcap, std::string("device ") + std::to_string(devices[r]) + ": " + g1s_last_global_error());
      rc = G1S_ERR_NO_DEVICE;
      goto done;
    }
  }
  fold = g1s_fold_new(ys->info.fps_num, ys->info.fps_den, o.ar_coeff_lag);
  if (!fold) {
    set_err(err, errcap, "g1s_fold_new failed");
    rc = G1S_ERR_INVALID;
    goto done;
  }
  // rounds: generator r takes batch (round * N + r) of the video -- B frame pairs read from the two files -- then every
  // generator packs, the fold merges.  The readers lend host buffers: a frame is copied before g1s_diff_frame returns.
  while (!ended && rc == G1S_OK) {
    for (uint32_t r = 0; r < N && !ended && rc == G1S_OK; ++r) {
      for (uint32_t i = 0; i < B; ++i) {
        g1s_frame_t s, d;
        const int rs = g1s_y4m_next(ys, &s), rd = g1s_y4m_next(yd, &d);
        if (rs < 0 || rd < 0) {
          rc = rs < 0 ? rs : rd;
          set_err(err, errcap, "frame " + std::to_string(frames) + ": " + (rs < 0 ? "source" : "denoised") + " reader failed (" +
                                   (rs < 0 ? ys->error : yd->error) + ")");
          break;
        }
        if (rs == 0 || rd == 0) {  // (None, None), or the warning of src/main.rs:449-455
          unequal = (rs == 0) != (rd == 0);
          ended = true;
          break;
        }
        if (filters) {
          char ferr[320] = "";
          g1s_frame_t cropped;
          if (g1s_filters_has_resize(filters)) {  // (one chain, one device: the sharded command does not resize)
            rc = G1S_ERR_UNSUPPORTED;
            set_err(err, errcap, "frame " + std::to_string(frames) + ": the resize filter is served on one device only (drop --gpus / --devices)");
            break;
          }
          rc = g1s_filters_apply(filters, &s, &cropped, ferr, sizeof(ferr));
          if (rc) {
            set_err(err, errcap, "frame " + std::to_string(frames) + ": " + ferr);
            break;
          }
          s = cropped;
        }
        rc = g1s_diff_frame(gen[r], &s, &d);
        if (rc) {
          set_err(err, errcap, "frame " + std::to_string(frames) + ": diff_frame: " + g1s_diff_last_error(gen[r]));
          break;
        }
        ++frames;
      }
    }
    if (rc == G1S_OK) rc = round(0);
  }
  // what is still in the generators' pipelines: flush rounds until every frame fed has been merged (a generator holds at most
  // its slots' worth of batches: a handful of rounds; the bound only stops a protocol error from spinning)
  for (int k = 0; k < 64 && rc == G1S_OK && g1s_fold_frames(fold) < frames; ++k) rc = round(1);
  if (rc == G1S_OK && g1s_fold_frames(fold) != frames) {
    set_err(err, errcap, "sharded diff: " + std::to_string(g1s_fold_frames(fold)) + " of " + std::to_string(frames) + " frames merged after the flush rounds");
    rc = G1S_ERR_STATE;
  }
  if (rc) goto done;
  rc = g1s_fold_finish(fold, segs.data(), segs.size(), &n);
  if (rc == G1S_ERR_CAPACITY) {
    segs.resize(n);
    rc = g1s_fold_finish(fold, segs.data(), segs.size(), &n);
  }
  if (rc) {
    set_err(err, errcap, g1s_fold_last_error(fold));
    goto done;
  }
  rc = g1s_write_tbl(out_tbl_path, segs.data(), n);
  if (rc) set_err(err, errcap, std::string("cannot write ") + out_tbl_path);
done:
  if (frames_out) *frames_out = frames;
  if (unequal_out) *unequal_out = unequal;
  for (g1s_diff_t *g : gen)
    if (g) g1s_diff_free(g);
  if (fold) g1s_fold_free(fold);
  g1s_filters_free(filters);
  g1s_y4m_close(ys);
  g1s_y4m_close(yd);
  return rc;
}

}  // extern "C"

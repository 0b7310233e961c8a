// engine.hip -- host engine + C-ABI of libg1s_diff.so (see include/g1s_diff.h).
//
// Frames are queued into a slot of `batch_frames` pairs; a full slot is one
// K1 -> K2 -> K3 launch group on the engine's HIP stream followed by one D2H
// copy of the slot's integer records.  Two slots alternate, so the GPU works on
// batch k+1 while the host folds batch k in frame order (fold.cpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/g1s_diff.h"
#include "fold.h"
#include "kernels.hip.h"
#include "k1f.hip.h"
#include "k3m.hip.h"
#include "k3s_params.hip.h"
#include "k3s.hip.h"
#include "k3w.hip.h"
#include "latest_dev.h"
#include "record.h"

using namespace g1s;

namespace {

thread_local std::string g_global_error;

constexpr uint32_t kDefaultBatch = 32;
constexpr int kK3Chunks = 48;

// The AR accumulation is an exact int8 SYRK on the matrix cores.  G1S_K3 selects the chain:
//   wide (default)    k3w.hip.h: 128-sample units, residuals in 32-bit SWAR, windows as masks on the A operand at multiply time,
//                     entries parked in LDS, ghost units instead of halo loads; the launches scatter the block statistics
//                     themselves, k3w_tail = the partial-system reduction + the exact kernel.  Serves aligned planes of equal depth;
//   stream            k3s.hip.h: round 3's form of the same pass (two-block units, two tile buffers); what `wide` falls back on for
//                     unaligned planes, widths that are not a multiple of 8 samples and mixed depths.
// Both are bit-exact against the oracle; tests/test_gpu_selfcheck.py compares them with each other.  (Rounds 1 and 2's chains --
// the lag-structured v_dot4 kernels behind the pixel pass K0, K0's planes in front of the matrix-core kernel, the 32x32x32 form
// of the fused pass -- were removed in round 4: git history)
// timing experiments: G1S_DBG_SKIP=name[,name...] leaves kernels out (wrong results; never set in tests or bench lines)
bool dbg_skip(const char *name) {
  static const std::string v = getenv("G1S_DBG_SKIP") ? std::string(",") + getenv("G1S_DBG_SKIP") + "," : std::string();
  return !v.empty() && v.find(std::string(",") + name + ",") != std::string::npos;
}
int k3_mode() {
  static const int v = [] {
    const char *e = getenv("G1S_K3");
    if (e && std::strcmp(e, "stream") == 0) return 3;
    return 4;
  }();
  return v;
}
bool use_wide() { return k3_mode() == 4;}
constexpr int kMTargetWgs = 1024;  // accumulation workgroups per launch: 4 per CU, one round
// workgroups per frame for a launch of B frames: enough to fill the chip, and few enough units each for int32.
// kind 0: the luma launch, 1: the chroma launch.  The luma launch likes workgroups of ~64 units of the list (4 096 - 6 144
// workgroups for 64 4K frames: 518 / 510 us against 534 at 2 048), the chroma launch 2 048 (372 us against 377 - 383 at 4 096):
// profiles/r03_wgs_sweep.txt.
int m_wgs_per_frame(int nunits, int B, int kind = 1) {
  const int gmin = (nunits + (kMMaxUnits - 16) - 1) / (kMMaxUnits - 16);  // (two lists, each dealt with its own rounding)
  const char *e = getenv("G1S_F_WGS");  // tuning / test aid (read at every call: a test sets it for its own generator)
  const char *el = kind == 0 ? getenv("G1S_F_WGS_L") : nullptr;  // ... the luma launch alone
  // (at least 32 workgroups to a frame: 64-frame launches are two resident rounds)
  int target = std::max(kMTargetWgs, 32 * B);
  if (kind == 0) target = std::max(target, std::min(64 * B, (int)((long long)B * nunits / 64)));
  if (e) target = std::max(8, atoi(e));
  if (el) target = std::max(8, atoi(el));
  return (std::max(gmin, (target + B - 1) / std::max(B, 1)) + 7) & ~7;  // (a multiple of 8: workgroup b of a frame on XCD b % 8)
}

// the wide chain: workgroups per frame.  A workgroup walks a contiguous slice of the frame's raster-ordered list and forms a
// ghost unit at either end: longer slices, fewer ghosts; at least one resident round (4 to a CU) a launch
int w_wgs_per_frame(int ncell, int B, int kind) {
  const int cap = kind == 1 ? kWMaxUnitsC : kWMaxUnits;
  const int gmin = (ncell + cap - 1) / cap;
  const char *e = getenv(kind ? "G1S_W_WGS_C" : "G1S_W_WGS");  // tuning / test aid
  int target = std::max(kMTargetWgs, (kind ? 16 : 32) * B);
  if (e) target = std::max(8, atoi(e));
  int G = (std::max(gmin, (target + B - 1) / std::max(B, 1)) + 7) & ~7;
  // (luma launch of a small frame: slices of at least 64 cells -- a slice pays two ghost units, its entries' parking and a partial
  //  system whatever its length -- as long as a launch still has a workgroup for every slot of the chip.  1080p, 128 frames a
  //  launch: 32 -> 8 workgroups a frame, luma launch 178 -> 168 us, k3w_tail 15 -> 11; 4K (32 a frame) and 8K are where they were.
  //  profiles/r05n_wgs_small_frames.txt)
  if (!e && kind == 0) {
    const int by_cells = ((ncell + 63) / 64 + 7) & ~7, by_slots = ((kMTargetWgs + B - 1) / std::max(B, 1) + 7) & ~7;
    G = std::max(gmin, std::min(G, std::max(by_cells, by_slots)));
    G = (G + 7) & ~7;
  }
  return G;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      char b_[384];                                                                        \
      snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return fail_hip(b_);                                                                 \
    }                                                                                      \
  } while (0)

// Minimal persistent worker pool: the per-frame half of the fold (AR solve,
// block measurements, strength solve) is independent across frames.
class Pool {
 public:
  explicit Pool(unsigned n) {
    for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
  }
  // runs fn(i) for i in [0, n); the caller participates.  Every call has its own job object (function, count, claim and
  // completion counters): a worker that comes late to an earlier job holds THAT job, finds it exhausted and goes back to
  // sleep -- it can never claim an index of a newer job or run the newer function with an older count.
  void parallel_for(int n, const std::function<void(int)> &fn) {
    if (n <= 0) return;
    auto job = std::make_shared<Job>();
    job->fn = &fn;
    job->n = n;
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = job;
      ++epoch_;
    }
    cv_.notify_all();
    work(*job);
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return job->done.load() == job->n; });
    if (job_ == job) job_.reset();
  }

 private:
  struct Job {
    const std::function<void(int)> *fn = nullptr;
    int n = 0;
    std::atomic<int> next{0}, done{0};
  };
  void work(Job &job) {
    int mine = 0;
    for (;;) {
      const int i = job.next.fetch_add(1);
      if (i >= job.n) break;
      (*job.fn)(i);  // (fn outlives the job: parallel_for returns only when done == n)
      ++mine;
    }
    if (mine && job.done.fetch_add(mine) + mine == job.n) {
      std::lock_guard<std::mutex> lk(m_);  // (the waiter checks under this lock: no lost wake-up)
      cv_done_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
        job = job_;
      }
      if (job) work(*job);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, cv_done_;
  std::shared_ptr<Job> job_;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};

// The cores this process may really use: the hardware threads, cut to the cgroup's CPU quota where there is one (a 1-GPU box
// of the pool this was measured on shows 256 hardware threads and `cpu.max` = 16 cores: 32 pool threads there do 80 k frames/s
// of the per-frame half where 16 do 95 k -- the quota's throttling stops every thread of the group, the launching one included;
// profiles/r04_host_budget_8ranks.txt).
unsigned usable_cpus() {
  unsigned hw = std::thread::hardware_concurrency();
  if (!hw) hw = 1;
  long long quota = -1, period = 100000;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    char q[32] = {0};
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
    fclose(f1);
    if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(f2, "%lld", &period) != 1) period = 100000;
      fclose(f2);
    }
  }
  if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
  return hw;
}

// One pool per process (creating 30 threads per generator would dominate short jobs).
Pool *shared_pool() {
  static Pool *p = [] {
    unsigned hw = usable_cpus();
    if (const char *e = getenv("G1S_FOLD_THREADS")) hw = (unsigned)atoi(e);
    if (hw > 32) hw = 32;
    return hw > 1 ? new Pool(hw - 1) : nullptr;  // the calling thread participates
  }();
  return p;
}
std::mutex g_pool_mutex;  // parallel_for is not re-entrant: one fold batch at a time
// The ordered merge of exchanged latest states (rank 0 of a frame-shard job) has a small pool of its own:
// on that rank the shared pool is busy half of the time with the per-frame half of the rank's own batches,
// and a merge that waits for it falls behind the eight GPUs it serves.
Pool *merge_pool() {
  static Pool *p = [] {
    unsigned hw = usable_cpus();
    if (const char *e = getenv("G1S_FOLD_THREADS")) hw = (unsigned)atoi(e);
    unsigned n = std::min(8u, hw / 2);  // (8: 1.5 - 1.6 us a frame, steady; 16 reaches 1.0 but swings to 2 - 8 on a busy host: profiles/r03_fold_budget.txt)
    if (const char *e = getenv("G1S_MERGE_POOL")) n = (unsigned)atoi(e);  // (measurement: the pool's size itself)
    return n > 1 ? new Pool(n - 1) : nullptr;
  }();
  return p;
}
std::mutex g_merge_pool_mutex;

constexpr bool kDeviceLatestDefault = false;
constexpr int kSlots = 6;  // batches in flight: being filled, pixel pass + finder, accumulation, (the per-frame half on the device,) D2H, fold

struct Slot {
  FramePlanes *h_planes = nullptr;  // pinned
  FramePlanes *d_planes = nullptr;
  uint8_t *d_records = nullptr;
  uint8_t *h_records = nullptr;  // pinned
  uint8_t *d_flags = nullptr;
  int32_t *d_partials = nullptr;   // k3_interior chunk partials, then k3_mixed chunk partials
  uint8_t *d_defer = nullptr;      // area classes, bad-block flags, area lists and their counts
  uint8_t *d_k0 = nullptr;         // K0 int8 planes [batch] x PlaneSet::frame_bytes
  int32_t *d_k1 = nullptr;         // flat-block fast path: moments [batch][nblocks][16], literal list [batch][nblocks], counts [batch]
  uint32_t *d_pgl = nullptr;       // partial-group lists [batch][2][pg_cap] + counts [batch][2]
  uint8_t *d_mu = nullptr;         // MFMA path: unit lists [batch][nunits], unit counts, deferred-block flags
  long long *d_mpart = nullptr;    // MFMA path: partial systems of the accumulation workgroups
  uint8_t *d_lplane = nullptr;     // MFMA path: L at chroma resolution, int8 (luma launch -> chroma launch)
  uint8_t *d_wu = nullptr;         // wide chain: unit lists, counts, per-unit statistics records, L-outside-int8 flags
  uint8_t *d_stage = nullptr;  // device copies of host-resident frames
  // the per-frame half of the fold on the device (latest.hip): the frames' latest-state blobs and the kernel's scratch
  uint8_t *d_latest = nullptr, *h_latest = nullptr /* pinned */, *d_lscratch = nullptr;
  size_t latest_cap = 0, lscratch_cap = 0;
  size_t stage_bytes_per_frame = 0;
  hipEvent_t done = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // timing: finder start, k2 start, k3 start, k3 end, k0 end, k0 start
  uint32_t count = 0;
  bool timed = false;
  bool chain = false;  // a timed batch with ONE pair of events, around the batch's whole chain of kernels (g1s_diff_set_timing(g, 2))
  bool async_in = false;  // the batch holds frames whose H2D copies were queued on the upload stream
  // per-kernel timing (g1s_diff_set_timing): an event before each launch, the name of the kernel it precedes
  std::vector<hipEvent_t> kev;
  std::vector<std::string> kname;
  std::vector<hipStream_t> kstream;
  size_t nk = 0;
};

static void free_slot(Slot &sl) {
  if (sl.h_planes) (void)hipHostFree(sl.h_planes);
  if (sl.d_planes) (void)hipFree(sl.d_planes);
  if (sl.d_records) (void)hipFree(sl.d_records);
  if (sl.h_records) (void)hipHostFree(sl.h_records);
  if (sl.d_flags) (void)hipFree(sl.d_flags);
  if (sl.d_k1) (void)hipFree(sl.d_k1);
  if (sl.d_partials) (void)hipFree(sl.d_partials);
  if (sl.d_defer) (void)hipFree(sl.d_defer);
  if (sl.d_k0) (void)hipFree(sl.d_k0);
  if (sl.d_pgl) (void)hipFree(sl.d_pgl);
  if (sl.d_mu) (void)hipFree(sl.d_mu);
  if (sl.d_mpart) (void)hipFree(sl.d_mpart);
  if (sl.d_lplane) (void)hipFree(sl.d_lplane);
  if (sl.d_wu) (void)hipFree(sl.d_wu);
  if (sl.d_stage) (void)hipFree(sl.d_stage);
  if (sl.d_latest) (void)hipFree(sl.d_latest);
  if (sl.h_latest) (void)hipHostFree(sl.h_latest);
  if (sl.d_lscratch) (void)hipFree(sl.d_lscratch);
  if (sl.done) (void)hipEventDestroy(sl.done);
  for (auto &e : sl.ev)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : sl.kev) (void)hipEventDestroy(e);
  sl = Slot{};
}

// Process-wide cache of slot buffers: pinned-host and device allocations cost
// hundreds of microseconds each; consecutive generators of the same geometry
// (one per video, or one per bench step) reuse them.
struct SlotKey {
  int device;
  size_t planes, records, flags, partials, defer, stage, k0, pgl, mu, mpart, lplane;
  int W, H, xdec, ydec, nplanes;  // the zeroed padding of the K0 planes depends on the exact geometry
  bool operator==(const SlotKey &o) const {
    return device == o.device && planes == o.planes && records == o.records && flags == o.flags &&
           partials == o.partials && defer == o.defer && stage == o.stage && k0 == o.k0 && pgl == o.pgl && mu == o.mu &&
           mpart == o.mpart && lplane == o.lplane &&
           W == o.W && H == o.H && xdec == o.xdec && ydec == o.ydec && nplanes == o.nplanes;
  }
};
struct CachedSlot {
  SlotKey key;
  Slot slot;
};
std::mutex g_cache_mutex;
std::vector<CachedSlot> g_slot_cache;

// Streams and their events are process-wide too (0.1-0.2 ms to create each); a generator borrows a set and
// hands it back when it is freed:
//   compute  main stream: pixel pass (K0) and the four lag kernels of alternating batches, back to back
//   flat     side stream (high priority): finder chain, window planes, area lists of a batch, next to the
//            lag kernels of the batch before
//   copy     tail of the accumulation (partial-group kernel, reducer, generic kernel) next to the pixel pass
//            of the batch after next, then the records D2H
//   upload   the frame tables (72 bytes a frame pair), ahead of everything
struct StreamSet {
  int device = -1;
  hipStream_t compute = nullptr, copy = nullptr, flat = nullptr, flat2 = nullptr, upload = nullptr;
  // k4_latest of the even / odd slots (made when a generator first runs the half on the device): the kernel is a few serial
  // chains per frame and twice as long next to the accumulation launches as alone -- on ONE stream, with the blobs' copy behind
  // it, a batch's half would only start when the half of the batch before had been copied out (period >= 1 040 us at 4K)
  hipStream_t latest = nullptr, latest2 = nullptr;
  hipStream_t mom = nullptr;  // (G1S_MOM_STREAM: a tuning aid)
  hipStream_t coread = nullptr;  // (G1S_DBG_COREAD: a measurement aid)
  hipEvent_t coread_go = nullptr;
  hipEvent_t latest_done[kSlots] = {};
  int prio_side = 0;
  hipEvent_t kernels_done[kSlots] = {};
  hipEvent_t mask_done[kSlots] = {};
  hipEvent_t pix_done[kSlots] = {};
  hipEvent_t k0_done[kSlots] = {};
  hipEvent_t table_done[kSlots] = {};
};
std::vector<StreamSet> g_stream_cache;
bool acquire_streams(int device, StreamSet &out) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    for (size_t i = 0; i < g_stream_cache.size(); ++i) {
      if (g_stream_cache[i].device == device) {
        out = g_stream_cache[i];
        g_stream_cache.erase(g_stream_cache.begin() + i);
        return true;
      }
    }
  }
  out = StreamSet{};
  out.device = device;
  // the side stream (the finder chain: small latency-bound kernels) outranks the main stream's big kernels, next to which it
  // runs.  What matters more than the order of the two: that the streams do not share a hardware queue.  The runtime gives a
  // process 4 by default (GPU_MAX_HW_QUEUES); the 4 - 6 streams here plus the host application's own then share, by creation
  // order, and kernels of different streams wait for each other: the same build read 544 - 593 k Mpx/s from run to run, 590 -
  // 595 k with GPU_MAX_HW_QUEUES=8, and 393 - 406 k with every stream in one priority class on 4 queues
  // (profiles/r04_streams.txt).  bench.py and the command set the variable before the runtime starts; the streams only some
  // jobs use (the second side stream, the device half's) are made when they are first needed.
  {
    static std::atomic<bool> said{false};
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    if ((!q || atoi(q) < 8) && !said.exchange(true))
      fprintf(stderr, "g1s: GPU_MAX_HW_QUEUES is %s: the generator's streams will share hardware queues (slower, not wrong); set "
                      "GPU_MAX_HW_QUEUES=8 before the process' first HIP call\n", q ? q : "unset");
  }
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // numerically lower = more urgent
  if (const char *e = getenv("G1S_PRIO")) {  // tuning aid: 1 = the main stream outranks the side stream, 2 = no priorities
    const int m = atoi(e);
    if (m == 1) std::swap(prio_lo, prio_hi);
    else if (m == 2) prio_lo = prio_hi = (prio_lo + prio_hi) / 2;
  }
  static const bool want_flat2 = getenv("G1S_SIDE2") && atoi(getenv("G1S_SIDE2")) != 0;
  out.prio_side = prio_hi;
  bool ok = hipStreamCreateWithPriority(&out.compute, hipStreamNonBlocking, prio_lo) == hipSuccess &&
            hipStreamCreateWithFlags(&out.copy, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithPriority(&out.flat, hipStreamNonBlocking, prio_hi) == hipSuccess &&
            (!want_flat2 || hipStreamCreateWithPriority(&out.flat2, hipStreamNonBlocking, prio_hi) == hipSuccess) &&
            hipStreamCreateWithFlags(&out.upload, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < kSlots && ok; ++i)
    ok = hipEventCreateWithFlags(&out.kernels_done[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&out.mask_done[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&out.pix_done[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&out.k0_done[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&out.table_done[i], hipEventDisableTiming) == hipSuccess;
  return ok;
}
void release_streams(StreamSet &ss) {
  if (!ss.compute) return;
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  g_stream_cache.push_back(ss);
  ss = StreamSet{};
}

}  // namespace

struct g1s_diff {
  int64_t fps_num, fps_den;
  uint32_t src_bd, den_bd;
  uint32_t lag, n;
  bool luma_only, records_only;
  bool latest_only = false;  // keep the per-frame latest states (blobs) instead of folding them here
  bool device_latest = false;  // the per-frame half of the fold runs on the device (k4_latest): blobs come back, not records
  uint32_t batch;
  bool batch_auto = false;  // no batch size asked for: sized to the frames at the first frame pair
  int device = 0;
  hipStream_t stream = nullptr;       // == ss.compute
  StreamSet ss;                       // borrowed from the process-wide cache
  bool geometry_set = false;
  g1s_frame_t shape{};  // geometry of the first frame
  Geom geom{};
  RecLayout L{};
  FlatConsts fc{};
  double *d_lut = nullptr;
  size_t defer_bytes = 0;
  uint32_t m_lpitch = 0, m_lframe = 0;  // MFMA path: L plane geometry
  int m_nunits = 0;          // MFMA path: chunks per frame
  size_t m_wg_cap = 0;       // ... workgroups (partial systems) the slots hold
  size_t m_only_bytes = 0;   // ... deferred-block flags [batch][3][nblocks] (one list a plane), 16-byte rounded
  // the wide chain (k3w.hip.h): blocks a chroma unit, cells a block row / a frame per kind, L geometry, layout of Slot::d_wu
  int w_ub_c = 4, w_gx[2] = {0, 0}, w_ncell[2] = {0, 0};
  uint32_t w_lpitch = 0, w_lframe = 0;
  size_t w_off_units[2] = {0, 0}, w_off_count = 0, w_off_lbad = 0, w_lbad_bytes = 0, w_bytes = 0;
  bool wide_ok(const Geom &g) const;
  MParams make_mparams(const Slot &sl) const;
  SlotKey slot_key{};
  Slot slots[kSlots];
  int cur = 0;
  int pending = -1;  // slot whose front half is queued and whose back half is not
  // The API thread queues frames and launches batches; the drainer thread waits for a batch's records,
  // runs the fold on them and frees the slot.  Everything below dm is shared between the two.
  std::thread drainer;  // waits for a batch's records, runs the per-frame half of the fold on the pool
  std::thread folder;   // the ordered half (merge in frame order), then frees the slot
  std::deque<int> fold_q;  // batches whose per-frame half is done
  std::condition_variable cv_fold;
  bool folder_stop = false;
  std::vector<FrameLatest> latest_s[kSlots];  // per slot: two batches are in the fold at a time
  std::vector<FrameView> views_s[kSlots];     // device_latest: the slot's blobs, read where the copy put them
  std::vector<uint32_t> nflat_s[kSlots];
  std::vector<uint8_t> stage_s[kSlots];
  double ms_fold_front = 0, ms_fold_back = 0;  // (one writer each)
  bool front_failed[kSlots] = {};               // a HIP error in the front stage: the batch is not folded
  std::mutex dm;
  std::condition_variable cv_work, cv_free;
  std::deque<int> in_flight;       // submitted, not yet picked up by the drainer
  bool slot_busy[kSlots] = {};     // submitted and not yet drained
  uint64_t submitted = 0, drained = 0;  // batches
  uint64_t frames_released = 0;         // frame pairs of the drained batches (their inputs are no longer read)
  // asynchronous host -> device copies of pinned frames (on_device == 2): one event per frame pair, in order
  std::mutex h2d_mutex;
  std::deque<std::pair<uint64_t, hipEvent_t>> h2d_pending;  // (frame pairs handed over up to and including this one, copies done)
  std::vector<hipEvent_t> h2d_free;
  uint64_t frames_appended = 0;
  hipEvent_t h2d_order = nullptr;  // copies -> table upload when the two run on different streams
  bool drainer_stop = false;
  NoiseFold *fold = nullptr;
  Pool *pool = nullptr;
  std::vector<uint8_t> records_out;
  size_t records_out_frames = 0;
  std::vector<uint8_t> latest_out;  // latest_only: blobs of the drained frames, in frame order
  size_t latest_out_frames = 0;
  std::deque<uint32_t> latest_batches;  // frames per drained, not yet delivered batch
  uint64_t delivered = 0;               // batches handed out by g1s_diff_take_latest
  std::vector<uint8_t> last_record;
  std::string err;
  int deferred = G1S_OK;
  // a fold error, a HIP error or a batch that failed to launch or drain kills the generator: every later call, finish
  // included, returns it (the reference `?`-propagates out of main; a table with frames silently missing is worse)
  std::atomic<int> sticky{G1S_OK};
  bool finished = false;
  std::vector<g1s_segment_t> final_segs;  // what finish() returned (kept: a too-small buffer can be retried)
  bool timing = false;
  bool timing_chain = false;  // timed batches carry one pair of events (first kernel's start, last kernel's end) instead of one a kernel
  int flat_literal = 0;  // flat-block finder: literal f64 evaluation of every block (1: lane per block, 2: wave per block)
  g1s_stats_t stats{};
  std::map<std::string, std::pair<double, uint64_t>> ktimes;  // timed batches: kernel name -> (ms, launches)
  std::mutex ktimes_mutex;
  // timed batches run on one stream: an event before each launch (and one after the last), named after the kernel
  int kmark(Slot &sl, hipStream_t st, const char *name) {
    if (sl.chain) return G1S_OK;  // (a chain-timed batch: nothing between two of its launches, trace mode or not)
    if (!sl.timed && !trace) return G1S_OK;
    if (sl.nk == sl.kev.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return fail(G1S_ERR_HIP, "hipEventCreate failed");
      sl.kev.push_back(e);
      sl.kname.emplace_back();
      sl.kstream.push_back(nullptr);
    }
    if (hipEventRecord(sl.kev[sl.nk], st) != hipSuccess) return fail(G1S_ERR_HIP, "hipEventRecord failed");
    sl.kstream[sl.nk] = st;
    sl.kname[sl.nk++] = name ? name : "";
    return G1S_OK;
  }
  // G1S_TRACE=file (a measurement aid): the PIPELINED job's own timeline -- an event in front of every launch on the stream it
  // is launched on (its end = the next event of that stream), the host's time at every submit; written at finish
  bool trace = false;
  hipEvent_t trace_base = nullptr;
  std::chrono::steady_clock::time_point trace_host0;
  std::vector<std::string> trace_lines;
  std::mutex trace_mutex;
  double trace_now() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - trace_host0).count(); }
  void trace_host(const char *what, int si) {
    if (!trace) return;
    char b[96];
    snprintf(b, sizeof(b), "H %12.1f %s slot %d", trace_now(), what, si);
    std::lock_guard<std::mutex> lk(trace_mutex);
    trace_lines.emplace_back(b);
  }

  int fail(int code, const std::string &msg) {
    err = msg;
    return code;
  }
  int fail_hip(const char *msg) {
    err = msg;
    int ok = G1S_OK;
    sticky.compare_exchange_strong(ok, G1S_ERR_HIP);
    return G1S_ERR_HIP;
  }

  int set_geometry(const g1s_frame_t *s, const g1s_frame_t *d);
  int set_geometry_alloc(const g1s_frame_t *s, const g1s_frame_t *d);
  int append(const g1s_frame_t *s, const g1s_frame_t *d);
  uint64_t frames_copied(uint64_t wait_for);
  int submit(int si);        // front half now; back half now or with the next batch's front half
  int launch_front(int si);  // zero, pixel pass, flat-block finder, window planes, area lists
  int launch_back(int si);   // accumulation kernels, records D2H, hand-over to the drainer
  static bool wide_gen(const Geom &g);  // the wide chain's general residual form (mixed sample sizes / shifts)
  int flush_pending();
  Geom batch_geom(const Slot &sl) const;
  int drain_front(int si);  // drainer thread
  int drain_back(int si);   // folder thread
  void drainer_main();
  void folder_main();
  int drain_all();          // API thread: wait until everything submitted is drained
  void wait_drained(uint64_t upto);
  void release();
};

// The 3x3 inverse of A^T A for the plane fit, built the way
// FlatBlockFinder::new does (three solves of the normal equations).
static void make_flat_consts(FlatConsts &fc) {
  double AtA[9] = {0};
  for (int y = 0; y < kBlock; ++y) {
    const double yd = ((double)y - kBlock / 2.) / (kBlock / 2.);
    for (int x = 0; x < kBlock; ++x) {
      const double xd = ((double)x - kBlock / 2.) / (kBlock / 2.);
      const double co[3] = {yd, xd, 1};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) AtA[3 * i + j] += co[i] * co[j];
    }
  }
  for (int i = 0; i < 3; ++i) {
    double A[9], b[3] = {0, 0, 0}, x[3] = {0, 0, 0};
    std::memcpy(A, AtA, sizeof(A));
    b[i] = 1;
    gauss_solve(3, A, b, x);
    for (int j = 0; j < 3; ++j) fc.ata_inv[j * 3 + i] = x[j];
  }
}

// (an allocation that fails half way leaves no half-built slot behind: a retry starts from nothing)
int g1s_diff::set_geometry(const g1s_frame_t *s, const g1s_frame_t *d) {
  const int rc = set_geometry_alloc(s, d);
  if (rc)
    for (Slot &sl : slots) free_slot(sl);
  return rc;
}
int g1s_diff::set_geometry_alloc(const g1s_frame_t *s, const g1s_frame_t *d) {
  shape = *s;
  if (batch_auto) {
    // about 530 Mpixels a launch group (64 4K frames; measured: 32 -> 64 frames a launch +12 % on the 4K job, no gain
    // beyond): the per-launch costs of the small kernels are the same for small frames, so they get more frames per
    // launch (1080p: 128, +6 % over 64 and +13 % over 256; 8K: 32 -- 16 is 5 % slower, 64 the same; 4K: 64 -- 96 and 128 are
    // 5 - 7 % slower: profiles/r04_other_workloads.txt, tools/r4_batch.sh)
    const uint64_t px = (uint64_t)s->width * s->height;
    batch = (uint32_t)std::min<uint64_t>(128, std::max<uint64_t>(32, (530000000ull + px / 2) / std::max<uint64_t>(px, 1)));
  }
  const uint32_t np = luma_only ? 1u : (uint32_t)s->nplanes;
  L = make_layout(s->width, s->height, np, lag);
  Geom &g = geom;
  g.W = (int)s->width;
  g.H = (int)s->height;
  g.xdec = s->xdec;
  g.ydec = s->ydec;
  g.nplanes = (int)np;
  g.nbw = (g.W + kBlock - 1) / kBlock;
  g.nbh = (g.H + kBlock - 1) / kBlock;
  g.nblocks = g.nbw * g.nbh;
  g.src_bps = s->bytes_per_sample;
  g.den_bps = d->bytes_per_sample;
  g.src_shift = s->bytes_per_sample == 2 ? (int)src_bd - 8 : 0;
  g.den_shift = d->bytes_per_sample == 2 ? (int)den_bd - 8 : 0;
  g.lag = (int)lag;
  g.n = (int)n;
  g.rec_size = (uint32_t)L.size;
  for (int c = 0; c < 3; ++c) {
    g.off_ar[c] = (uint32_t)L.off_ar[c];
    g.off_sum_d[c] = (uint32_t)L.off_sum_d[c];
    g.off_sum_d2[c] = (uint32_t)L.off_sum_d2[c];
  }
  g.off_luma_sum = (uint32_t)L.off_luma_sum;
  g.off_scores = (uint32_t)L.off_scores;
  g.off_mask = (uint32_t)L.off_mask;

  size_t frame_bytes = 0;  // tight device copy of one frame pair (host-resident input)
  for (uint32_t c = 0; c < np; ++c) {
    const size_t pw = c ? (s->width >> s->xdec) : s->width, ph = c ? (s->height >> s->ydec) : s->height;
    frame_bytes += ((pw * s->bytes_per_sample + 15) & ~size_t(15)) * ph;
    frame_bytes += ((pw * d->bytes_per_sample + 15) & ~size_t(15)) * ph;
  }
  const size_t partial_bytes = 0, k0_bytes = 0, pgl_bytes = 0;  // (buffers of the chains removed in round 4: the slot key keeps its fields)
  defer_bytes = 0;
  m_nunits = ((g.nbw + kMUnitBlocks - 1) / kMUnitBlocks) * g.nbh;
  m_only_bytes = ((size_t)g.nblocks * 3 * batch + 15) & ~size_t(15);
  // [units][unit counts][any-deferred flags][deferred-block flags]
  // ... [per-unit statistics records]
  const size_t mu_bytes = sizeof(uint32_t) * ((size_t)batch * m_nunits * kMUnitDwords + 3 * (size_t)batch) + m_only_bytes +
                          sizeof(int32_t) * (size_t)batch * m_nunits * kMStatInts;
  // one partial system per accumulation workgroup and plane: the most workgroups a launch of 1 .. batch frames asks for
  m_wg_cap = 0;
  for (uint32_t b = 1; b <= batch; ++b)
    m_wg_cap = std::max(m_wg_cap, (size_t)b * std::max(m_wgs_per_frame(m_nunits, (int)b, 0), m_wgs_per_frame(m_nunits, (int)b, 1)));
  // L plane of a frame: block rows x chunk columns at chroma resolution (+ a slack row)
  m_lpitch = g.nplanes == 3 ? (uint32_t)((((g.nbw + kMUnitBlocks - 1) / kMUnitBlocks) * kMUnitBlocks * (kBlock >> g.xdec) + 15) & ~15) : 0u;
  m_lframe = m_lpitch * (uint32_t)(g.nbh * (kBlock >> g.ydec) + 1);
  // the wide chain (k3w.hip.h): its own lists, statistics records and L geometry
  w_ub_c = g.nplanes == 3 ? (kWUnitW / (kBlock >> g.xdec)) : 4;
  w_gx[0] = (g.nbw + 3) / 4;
  w_gx[1] = (g.nbw + w_ub_c - 1) / w_ub_c;
  w_ncell[0] = w_gx[0] * g.nbh;
  w_ncell[1] = g.nplanes == 3 ? w_gx[1] * g.nbh : 0;
  w_lpitch = g.nplanes == 3 ? (uint32_t)((std::max(w_gx[0] * 4 * (kBlock >> g.xdec), w_gx[1] * kWUnitW) + 15) & ~15) : 0u;
  w_lframe = w_lpitch * (uint32_t)(g.nbh * (kBlock >> g.ydec) + 1);
  {
    size_t o = 64;  // (a workgroup parks the entry in front of its slice too)
    w_off_units[0] = o, o += sizeof(uint32_t) * (size_t)batch * w_ncell[0] * kWEntry;
    w_off_units[1] = o, o += sizeof(uint32_t) * (size_t)batch * w_ncell[1] * kWEntry;
    w_off_count = o, o += sizeof(uint32_t) * 2 * (size_t)batch;
    o = (o + 15) & ~size_t(15);
    o += 256;  // (a workgroup parks three entries past its slice)
    w_off_lbad = o, w_lbad_bytes = ((size_t)batch * w_ncell[0] + 15) & ~size_t(15), o += w_lbad_bytes;
    w_bytes = use_wide() ? o : 0;
  }
  if (use_wide())
    for (uint32_t b = 1; b <= batch; ++b)
      m_wg_cap = std::max(m_wg_cap, (size_t)b * std::max(w_wgs_per_frame(w_ncell[0], (int)b, 0), w_wgs_per_frame(std::max(w_ncell[1], 1), (int)b, 1)));
  const size_t mpart_bytes2 = sizeof(long long) * 3 * kMRec * m_wg_cap;
  const size_t lplane_bytes = (size_t)std::max(m_lframe, use_wide() ? w_lframe : 0u) * batch;
  slot_key = SlotKey{device, sizeof(FramePlanes) * batch, L.size * batch, (size_t)g.nblocks * batch,
                     partial_bytes, defer_bytes, frame_bytes * batch, k0_bytes, pgl_bytes, mu_bytes + w_bytes, mpart_bytes2, lplane_bytes,
                     g.W, g.H, g.xdec, g.ydec, g.nplanes};
  for (Slot &sl : slots) {
    {
      std::lock_guard<std::mutex> lk(g_cache_mutex);
      for (size_t i = 0; i < g_slot_cache.size(); ++i) {
        if (g_slot_cache[i].key == slot_key) {
          sl = g_slot_cache[i].slot;
          g_slot_cache.erase(g_slot_cache.begin() + i);
          break;
        }
      }
    }
    sl.stage_bytes_per_frame = frame_bytes;
    sl.count = 0;
    if (sl.h_planes) continue;  // reused
    HIP_TRY(hipHostMalloc((void **)&sl.h_planes, slot_key.planes, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&sl.d_planes, slot_key.planes));
    HIP_TRY(hipMalloc((void **)&sl.d_records, slot_key.records));
    HIP_TRY(hipHostMalloc((void **)&sl.h_records, slot_key.records, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&sl.d_flags, slot_key.flags));
    HIP_TRY(hipMalloc((void **)&sl.d_k1, sizeof(int32_t) * ((size_t)g.nblocks * batch * (kMomInts + 1) + batch)));
    if (mu_bytes) HIP_TRY(hipMalloc((void **)&sl.d_mu, mu_bytes));
    if (mpart_bytes2) HIP_TRY(hipMalloc((void **)&sl.d_mpart, mpart_bytes2));
    if (w_bytes) HIP_TRY(hipMalloc((void **)&sl.d_wu, w_bytes));
    if (lplane_bytes) HIP_TRY(hipMalloc((void **)&sl.d_lplane, lplane_bytes));
    // (blocking: the drainer SLEEPS until a batch's records have landed instead of spinning on the event -- a core a rank, which an
    //  8-rank node inside a 16-core quota does not have; five more batches are in flight, the wake-up costs the job nothing)
    HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync));
    for (auto &e : sl.ev) HIP_TRY(hipEventCreate(&e));
  }
  geometry_set = true;
  return G1S_OK;
}

static bool same_shape(const g1s_frame_t &a, const g1s_frame_t &b) {
  return a.width == b.width && a.height == b.height && a.xdec == b.xdec && a.ydec == b.ydec &&
         a.nplanes == b.nplanes;
}

int g1s_diff::append(const g1s_frame_t *s, const g1s_frame_t *d) {
  if (!s || !d) return fail(G1S_ERR_INVALID, "null frame");
  if (!same_shape(*s, *d))
    return fail(G1S_ERR_DIM_MISMATCH, "Source and denoised frame dimensions do not match");
  if ((s->bytes_per_sample != 1 && s->bytes_per_sample != 2) ||
      (d->bytes_per_sample != 1 && d->bytes_per_sample != 2) || (s->nplanes != 1 && s->nplanes != 3) ||
      s->width < 1 || s->height < 1 || s->xdec > 1 || s->ydec > 1)
    return fail(G1S_ERR_INVALID, "unsupported frame format");
  if ((s->bytes_per_sample == 1) != (src_bd == 8) || (d->bytes_per_sample == 1) != (den_bd == 8))
    return fail(G1S_ERR_INVALID, "bytes_per_sample does not match the bit depth given to g1s_diff_new");
  if (!geometry_set) {
    const int rc = set_geometry(s, d);
    if (rc) return rc;
  } else if (!same_shape(shape, *s) || shape.bytes_per_sample != s->bytes_per_sample) {
    return fail(G1S_ERR_DIM_MISMATCH, "frame geometry changed mid-stream");
  }
  Slot &sl = slots[cur];
  if (sl.count >= batch) return fail(G1S_ERR_STATE, "the previous batch failed to launch");
  FramePlanes &fp = sl.h_planes[sl.count];
  std::memset(&fp, 0, sizeof(fp));
  const uint32_t np = (uint32_t)geom.nplanes;
  const bool any_host = s->on_device != 1 || d->on_device != 1;
  bool any_async = false;
  if (any_host && !sl.d_stage)
    HIP_TRY(hipMalloc((void **)&sl.d_stage, sl.stage_bytes_per_frame * batch));
  uint8_t *stage = any_host ? sl.d_stage + sl.stage_bytes_per_frame * sl.count : nullptr;
  for (int side = 0; side < 2; ++side) {
    const g1s_frame_t *f = side ? d : s;
    for (uint32_t c = 0; c < np; ++c) {
      const size_t pw = c ? (f->width >> f->xdec) : f->width, ph = c ? (f->height >> f->ydec) : f->height;
      const uint8_t *ptr;
      uint32_t stride;
      if (f->on_device == 1) {
        ptr = (const uint8_t *)f->data[c];
        stride = (uint32_t)f->stride_bytes[c];
      } else {
        const size_t row = (pw * f->bytes_per_sample + 15) & ~size_t(15);
        if (f->on_device == 2) {
          // pinned host memory the caller keeps valid until g1s_diff_frames_copied() covers this frame: queued on the
          // upload stream (the table upload of the batch follows on that stream, the kernels wait for that), the call
          // returns at once -- file reads, copies and the kernels of earlier batches overlap
          hipStream_t cs = ss.upload ? ss.upload : stream;
          if (f->stride_bytes[c] == row && pw * f->bytes_per_sample == row)
            HIP_TRY(hipMemcpyAsync(stage, f->data[c], row * ph, hipMemcpyHostToDevice, cs));
          else
            HIP_TRY(hipMemcpy2DAsync(stage, row, f->data[c], f->stride_bytes[c], pw * f->bytes_per_sample, ph, hipMemcpyHostToDevice, cs));
          any_async = true;
        } else {
          // the `&Frame` borrow ends when this call returns: copy now
          HIP_TRY(hipMemcpy2D(stage, row, f->data[c], f->stride_bytes[c], pw * f->bytes_per_sample, ph,
                              hipMemcpyHostToDevice));
        }
        ptr = stage;
        stride = (uint32_t)row;
        stage += row * ph;
      }
      if (side) {
        fp.den[c] = ptr;
        fp.den_stride[c] = stride;
      } else {
        fp.src[c] = ptr;
        fp.src_stride[c] = stride;
      }
    }
  }
  {
    std::lock_guard<std::mutex> lk(h2d_mutex);
    ++frames_appended;
    if (any_async) {
      hipEvent_t e;
      if (!h2d_free.empty()) {
        e = h2d_free.back();
        h2d_free.pop_back();
      } else {
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
      HIP_TRY(hipEventRecord(e, ss.upload ? ss.upload : stream));
      h2d_pending.emplace_back(frames_appended, e);
      sl.async_in = true;
    }
  }
  sl.count++;
  if (sl.count == batch) return submit(cur);
  return G1S_OK;
}

// how many frame pairs' host planes are no longer needed (every on_device == 0 frame at once; on_device == 2 frames when
// their queued copies have run); wait_for > 0: block until that many are
uint64_t g1s_diff::frames_copied(uint64_t wait_for) {
  std::lock_guard<std::mutex> lk(h2d_mutex);
  wait_for = std::min(wait_for, frames_appended);
  while (!h2d_pending.empty()) {
    const auto front = h2d_pending.front();
    if (front.first <= wait_for) (void)hipEventSynchronize(front.second);
    else if (hipEventQuery(front.second) != hipSuccess) break;
    h2d_free.push_back(front.second);
    h2d_pending.pop_front();
  }
  return h2d_pending.empty() ? frames_appended : h2d_pending.front().first - 1;
}

Geom g1s_diff::batch_geom(const Slot &sl) const {
  const uint32_t B = sl.count;
  Geom g = geom;
  int fast = 1;
  for (uint32_t i = 0; i < B; ++i) {
    if (((uintptr_t)sl.h_planes[i].src[0] & 15) || (sl.h_planes[i].src_stride[0] & 15)) fast = 0;
  }
  g.fast_rows = fast;
  int vec_mask = 0x3f;
  for (uint32_t i = 0; i < B; ++i) {
    for (int c = 0; c < g.nplanes; ++c) {
      if (((uintptr_t)sl.h_planes[i].src[c] & 15) || (sl.h_planes[i].src_stride[c] & 15)) vec_mask &= ~(1 << c);
      if (((uintptr_t)sl.h_planes[i].den[c] & 15) || (sl.h_planes[i].den_stride[c] & 15)) vec_mask &= ~(8 << c);
    }
  }
  g.vec_mask = vec_mask;
  return g;
}

// A batch runs in two halves.  Front: zero fills and the pixel pass (K0, HBM bound) on the main stream,
// then the flat-block finder, the window planes and the area lists -- small latency-bound kernels -- on the
// side stream.  Back: the accumulation kernels on the main stream, the records D2H, the hand-over to the
// drainer.  The back half of batch N is queued behind the front half of batch N + 1: the main stream runs
// K0(N + 1), accumulation(N), K0(N + 2), ... back to back (the big kernels never share the chip, which
// only stretches them), and the side stream's chain of N + 1 -- a few thousand waves, ~0.1 ms of latency --
// runs next to accumulation(N) and is long done when accumulation(N + 1) comes up.
int g1s_diff::submit(int si) {
  Slot &sl = slots[si];
  if (sl.count == 0) return G1S_OK;
  static const bool one_stream = getenv("G1S_ONE_STREAM") != nullptr;  // debugging aid
  static const bool no_defer = getenv("G1S_NO_DEFER") != nullptr;     // debugging aid
  {
    std::lock_guard<std::mutex> lk(dm);
    slot_busy[si] = true;  // until the drainer has folded it
  }
  trace_host("submit", si);
  int rc = launch_front(si);
  if (rc) return rc;
  trace_host("front queued", si);
  const int prev = pending;
  pending = si;
  if (prev >= 0) {
    rc = launch_back(prev);
    if (rc) return rc;
    trace_host("back queued", prev);
  }
  if (one_stream || no_defer || timing || !ss.flat) {
    rc = flush_pending();
    if (rc) return rc;
  }
  {
    // move on to the next slot; wait if the drainer has not freed it yet (back-pressure)
    std::unique_lock<std::mutex> lk(dm);
    cur = (si + 1) % kSlots;
    cv_free.wait(lk, [&] { return !slot_busy[cur]; });
  }
  trace_host("next slot free", cur);
  return G1S_OK;
}

int g1s_diff::flush_pending() {
  if (pending < 0) return G1S_OK;
  const int si = pending;
  pending = -1;
  return launch_back(si);
}

int g1s_diff::launch_front(int si) {
  Slot &sl = slots[si];
  const uint32_t B = sl.count;
  Geom g = batch_geom(sl);
  static const bool one_stream = getenv("G1S_ONE_STREAM") != nullptr;  // debugging aid
  hipStream_t stream = ss.compute;                                        // main stream (shadows the member)
  // Two side streams, a batch's chain on the one of its slot's parity: the latency-bound tail of a chain (certify, the
  // literal blocks, select, unit lists) then runs next to the moments kernel of the batch after it -- one waits on dependent
  // loads, the other streams through HBM -- instead of in front of it (G1S_SIDE2=1; the timeline shows them overlap either way)
  static const bool side2 = getenv("G1S_SIDE2") && atoi(getenv("G1S_SIDE2")) != 0;  // (measured: no gain, profiles/r03b; off)
  hipStream_t fstream = (one_stream || timing || !ss.flat) ? stream : ((si & 1) && side2 && ss.flat2 ? ss.flat2 : ss.flat);  // per-kernel timing: one stream
  // the pixel pass of round 1's chain (K0) runs on the main stream; the fused pass has no K0: its only pixel pass before
  // the mask is the finder's luma-source moments kernel, which joins the finder chain on the side stream and runs next to
  // the accumulation of the batch before
  // (measured and dropped: k1_moments on the main stream in front of the accumulation of the batch before, the rest of the
  //  finder chain beside that accumulation: -3 to -10 % at 4K, +11 % at 1080p, -4 % at 8K; profiles/r04_streams.txt)
  hipStream_t pstream = fstream;
  // G1S_MOM_STREAM=1|2 (tuning aid): k1_moments on a stream of its own in the main stream's priority class (1) / the default
  // class (2), the latency-bound rest of the finder chain alone on the high-priority side stream
  static const int mom_mode = getenv("G1S_MOM_STREAM") ? atoi(getenv("G1S_MOM_STREAM")) : 0;
  if (mom_mode && fstream != stream) {
    if (!ss.mom) {
      int plo = 0, phi = 0;
      (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
      if (mom_mode == 1) HIP_TRY(hipStreamCreateWithPriority(&ss.mom, hipStreamNonBlocking, plo));
      else HIP_TRY(hipStreamCreateWithFlags(&ss.mom, hipStreamNonBlocking));
    }
    pstream = ss.mom;
  }
  // the frame table: pinned host copy -> device, on the upload stream (idle: done long before the main
  // stream gets here); per-kernel timing / one-stream mode: in line
  FrameTable ft;
  ft.f = reinterpret_cast<const FramePlanes *>(sl.d_planes);
  hipStream_t up = (fstream == stream || !ss.upload) ? stream : ss.upload;
  if (sl.async_in) {  // queued frame copies: on the upload stream; everything below waits for `up`
    hipStream_t cs = ss.upload ? ss.upload : stream;
    if (cs != up) {
      if (!h2d_order) HIP_TRY(hipEventCreateWithFlags(&h2d_order, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(h2d_order, cs));
      HIP_TRY(hipStreamWaitEvent(up, h2d_order, 0));
    }
    sl.async_in = false;
  }
  sl.timed = timing;
  sl.chain = timing && timing_chain;
  sl.nk = 0;
  if (!sl.timed) kmark(sl, up, "table H2D");  // (trace mode only: the timed batches' table of kernels stays what it was)
  HIP_TRY(hipMemcpyAsync(sl.d_planes, sl.h_planes, sizeof(FramePlanes) * B, hipMemcpyHostToDevice, up));
  if (!sl.timed) kmark(sl, up, "k_zero");
  {
    // all per-batch zero fills in one launch: records, lag / masked accumulators, bad flags + list counters
    ZeroJob z{};
    z.ptr[0] = reinterpret_cast<uint32_t *>(sl.d_records);
    z.ndw[0] = (uint32_t)(L.size * B / 4);
    z.ptr[1] = reinterpret_cast<uint32_t *>(sl.d_mu) + (size_t)batch * m_nunits * kMUnitDwords;  // unit counts (2 lists), any-deferred flags
    z.ndw[1] = 3 * (uint32_t)batch;
    z.ptr[3] = reinterpret_cast<uint32_t *>(sl.d_mu) + (size_t)batch * m_nunits * kMUnitDwords + 3 * (size_t)batch;  // deferred-block flags
    z.ndw[3] = (uint32_t)(m_only_bytes / 4);
    if (use_wide() && sl.d_wu) {
      z.ptr[6] = reinterpret_cast<uint32_t *>(sl.d_wu + w_off_lbad);  // luma units whose L left int8
      z.ndw[6] = (uint32_t)(w_lbad_bytes / 4);
    }
    z.ptr[5] = reinterpret_cast<uint32_t *>(sl.d_k1) + (size_t)g.nblocks * batch * (kMomInts + 1);  // literal-list counts
    z.ndw[5] = (uint32_t)batch;
    // (on the upload stream too: the slot is free, its buffers can be zeroed while the main stream is still busy
    //  with earlier batches)
    hipLaunchKernelGGL(k_zero, dim3(256), dim3(256), 0, up, z);
  }
  if (!sl.timed) kmark(sl, up, nullptr);
  if (up != stream) {
    HIP_TRY(hipEventRecord(ss.table_done[si], up));
    HIP_TRY(hipStreamWaitEvent(stream, ss.table_done[si], 0));
    if (pstream != stream) HIP_TRY(hipStreamWaitEvent(pstream, ss.table_done[si], 0));
  }
  if (sl.timed) HIP_TRY(hipEventRecord(sl.ev[0], pstream));
  // (the fused pass serves every format; round 1's chain has no structured path for 4:4:0)
  {
    // flat-block features: integer moments + certified evaluation; the literal f64 kernel only for
    // the blocks the certificate leaves open (G1S_K1_LITERAL=1 / g1s_diff_set_flat_finder: for every block)
    static const int env_literal = getenv("G1S_K1_LITERAL") ? atoi(getenv("G1S_K1_LITERAL")) : 0;
    const int literal_mode = flat_literal ? flat_literal : env_literal;
    const int force_literal = literal_mode ? 1 : 0;
    int32_t *mom = sl.d_k1;
    bool pix_recorded = false;
    CertifyLists cl;
    cl.list = reinterpret_cast<uint32_t *>(sl.d_k1) + (size_t)g.nblocks * batch * kMomInts;
    cl.count = cl.list + (size_t)g.nblocks * batch;
    cl.global = literal_mode == 0 ? 1 : 0;  // (the default chain: one sequence for the launch; "every block literally": per-frame lists)
    {
      // the finder's moments of the luma source: the only pass over pixels that are not in a flat block's tile
      if (sl.timed && !sl.chain) HIP_TRY(hipEventRecord(sl.ev[5], pstream));
      {  // (also when every block is evaluated literally: the record's luma_sum comes from the moments)
        const dim3 mg((g.nblocks + 7) / 8, B);
        kmark(sl, pstream, g.src_bps == 1 ? "k1_moments<1>" : "k1_moments<2>");
        if (dbg_skip("moments")) {
        } else if (g.src_bps == 1) hipLaunchKernelGGL(k1_moments<1>, mg, dim3(256), 0, pstream, ft, g, mom);
        else hipLaunchKernelGGL(k1_moments<2>, mg, dim3(256), 0, pstream, ft, g, mom);
      }
      if (sl.timed && !sl.chain) HIP_TRY(hipEventRecord(sl.ev[4], pstream));
    }
    if (pstream != fstream) {  // the finder chain: on the side stream, behind the pixel pass (its luma half)
      if (!pix_recorded) HIP_TRY(hipEventRecord(ss.pix_done[si], pstream));
      HIP_TRY(hipStreamWaitEvent(fstream, ss.pix_done[si], 0));
    }
    kmark(sl, fstream, "k1_certify");
    if (!dbg_skip("certify")) hipLaunchKernelGGL(k1_certify, dim3((g.nblocks + 255) / 256, B), dim3(256), 0, fstream, g, fc, (const int32_t *)mom,
                       sl.d_records, sl.d_flags, cl, force_literal);
    kmark(sl, fstream, literal_mode == 1 ? "k1_flat_features" : "k1_flat_block");
    if (literal_mode == 1) {  // every block: one lane per block
      dim3 grid((g.nblocks + 63) / 64, B);
      if (g.src_bps == 1)
        hipLaunchKernelGGL((k1_flat_features<1, true>), grid, dim3(64), 0, fstream, ft, g, fc, d_lut, sl.d_records, sl.d_flags,
                           (const uint32_t *)cl.list, (const uint32_t *)cl.count);
      else
        hipLaunchKernelGGL((k1_flat_features<2, true>), grid, dim3(64), 0, fstream, ft, g, fc, d_lut, sl.d_records, sl.d_flags,
                           (const uint32_t *)cl.list, (const uint32_t *)cl.count);
    } else if (!dbg_skip("flatblock")) {  // the few blocks the certificate leaves open (mode 2, a test aid: every block): one wave per block
      dim3 grid(kFbGrid);
#define G1S_FB(BP, GL) hipLaunchKernelGGL((k1_flat_block<BP, GL>), grid, dim3(64), 0, fstream, ft, g, fc, d_lut, sl.d_records, sl.d_flags, \
                                          (const uint32_t *)cl.list, (const uint32_t *)cl.count, (int)B)
      if (cl.global) {
        if (g.src_bps == 1) G1S_FB(1, true);
        else G1S_FB(2, true);
      } else {
        if (g.src_bps == 1) G1S_FB(1, false);
        else G1S_FB(2, false);
      }
#undef G1S_FB
    }
  }
  if (sl.timed && !sl.chain) HIP_TRY(hipEventRecord(sl.ev[1], fstream));
  const bool w_lists = wide_ok(g);  // the wide chain: the unit lists come out of the select kernel
  WUnitParams wup{};
  if (w_lists) {
    for (int k = 0; k < 2; ++k) {
      wup.units[k] = reinterpret_cast<uint32_t *>(sl.d_wu + w_off_units[k]);
      wup.ncell[k] = w_ncell[k];
      wup.gx[k] = w_gx[k];
    }
    wup.count = reinterpret_cast<uint32_t *>(sl.d_wu + w_off_count);
    wup.ub[0] = 4;
    wup.ub[1] = w_ub_c;
  }
  kmark(sl, fstream, w_lists ? "k2w_select_units" : "k2_flat_select");
  if (w_lists) hipLaunchKernelGGL(k2w_select_units, dim3(B, g.nplanes == 3 ? 2 : 1), dim3(kK2Threads), 0, fstream, g, sl.d_records, (const uint8_t *)sl.d_flags, wup);
  else if (!dbg_skip("k2")) hipLaunchKernelGGL(k2_flat_select, dim3(B), dim3(kK2Threads), 0, fstream, g, sl.d_records, sl.d_flags);
  if (sl.timed && !sl.chain) HIP_TRY(hipEventRecord(sl.ev[2], fstream));
  if (!w_lists) {
    // the unit lists (chunks with a flat block) need the flat mask (the wide chain: k2w_select_units has built them)
    const MParams mp = make_mparams(sl);
    kmark(sl, fstream, "k3m_units");
    hipLaunchKernelGGL(k3m_units, dim3((m_nunits + 255) / 256, B), dim3(256), 0, fstream, g, (const uint8_t *)sl.d_records, mp);
  }
  kmark(sl, fstream, nullptr);
  if (fstream != stream) HIP_TRY(hipEventRecord(ss.mask_done[si], fstream));
  HIP_TRY(hipGetLastError());
  return G1S_OK;
}

// the wide chain serves: equal sample widths, every plane's rows 16-byte aligned, whole 8-sample words in every plane,
// (unaligned planes, odd widths and mixed depths run the stream chain)
bool g1s_diff::wide_ok(const Geom &g) const {
  if (!use_wide()) return false;
  static const bool off = getenv("G1S_W_OFF") != nullptr;  // debugging aid
  if (off) return false;
  if (g.lag < 1) return false;
  // inputs of one sample size and one narrowing shift <= 4: the residual in place (w_residual); any other pair of depths: the
  // general form (w_residual_gen), built for 4:2:0 and for frames without chroma planes (the other subsamplings of such a pair
  // run the stream chain)
  if (wide_gen(g) && !(g.nplanes != 3 || (g.xdec == 1 && g.ydec == 1))) return false;
  const int need = g.nplanes == 3 ? 0x3f : 0x09;
  if ((g.vec_mask & need) != need) return false;
  if ((g.W & 7) != 0 || (g.nplanes == 3 && ((g.W >> g.xdec) & 7) != 0)) return false;
  if (g.nbw > 1023 * 4 || g.nbh > 4095) return false;
  return true;
}

bool g1s_diff::wide_gen(const Geom &g) { return g.src_bps != g.den_bps || g.src_shift != g.den_shift || g.src_shift > 4; }

MParams g1s_diff::make_mparams(const Slot &sl) const {
  MParams mp;
  // (the fused pass finds the residuals outside int8 itself; K0 flags them per block)
  mp.bad = nullptr;  // (no pixel pass flags residuals outside int8: the accumulation launches find them themselves)
  mp.units = reinterpret_cast<uint32_t *>(sl.d_mu);
  mp.unit_count = mp.units + (size_t)batch * m_nunits * kMUnitDwords;
  mp.only_any = mp.unit_count + 2 * batch;
  mp.only = reinterpret_cast<uint8_t *>(mp.only_any + batch);
  mp.partials = sl.d_mpart;
  mp.nunits = m_nunits;
  return mp;
}

int g1s_diff::launch_back(int si) {
  Slot &sl = slots[si];
  const uint32_t B = sl.count;
  Geom g = batch_geom(sl);
  static const bool one_stream = getenv("G1S_ONE_STREAM") != nullptr;  // debugging aid
  hipStream_t stream = ss.compute;
  const bool side = !(one_stream || sl.timed || !ss.flat);
  const bool acc_aside = false;
  if (side) HIP_TRY(hipStreamWaitEvent(stream, ss.mask_done[si], 0));  // the mask, the unit lists
  FrameTable ft;
  ft.f = reinterpret_cast<const FramePlanes *>(sl.d_planes);  // (uploaded by the front half)
  if (wide_ok(g)) {
    // the wide chain (k3w.hip.h): luma launch (leaves L behind), chroma launch, the reducer, the exact kernel for deferred blocks
    const MParams mp = make_mparams(sl);
    const bool chroma = g.nplanes == 3;
    WParams wq;
    wq.ft = ft;
    wq.records = sl.d_records;
    wq.partials = mp.partials;
    wq.only = mp.only;
    wq.only_any = mp.only_any;
    wq.lbad = sl.d_wu + w_off_lbad;
    wq.lplane = sl.d_lplane;
    wq.lpitch = w_lpitch;
    wq.lframe_bytes = w_lframe;
    wq.ncell_y = w_ncell[0];
    wq.gx_y = w_gx[0];
    wq.frames = (int)B;
    static const int w_dbg = getenv("G1S_W_DBG") ? atoi(getenv("G1S_W_DBG")) : 0;  // timing experiments (a -DG1S_W_DBG_BUILD library)
    wq.dbg = w_dbg;
    static const int w_rev = getenv("G1S_W_REV") ? atoi(getenv("G1S_W_REV")) : 0;  // tuning aid: bit 0 the luma launch, bit 1 the chroma launch walk the frames last to first
    int Gk[2] = {w_wgs_per_frame(w_ncell[0], (int)B, 0), w_wgs_per_frame(std::max(w_ncell[1], 1), (int)B, 1)};
    for (int k = 0; k < 2; ++k) {
      if ((size_t)Gk[k] * B <= m_wg_cap) continue;
      // the environment (G1S_W_WGS / G1S_W_WGS_C) changed after the slots were sized: what the slots hold -- but never fewer
      // workgroups than the int32 accumulators and the parked entries of a workgroup allow
      Gk[k] = (int)(m_wg_cap / B) & ~7;
      const int ncell_k = k == 0 ? w_ncell[0] : std::max(w_ncell[1], 1), cap_k = k == 1 ? kWMaxUnitsC : kWMaxUnits;
      if (Gk[k] <= 0 || Gk[k] < (ncell_k + cap_k - 1) / cap_k)
        return fail(G1S_ERR_STATE, "the wide launches' workgroup count was raised (G1S_W_WGS / G1S_W_WGS_C) after this generator's buffers were sized");
    }
    const int G_cap = std::max(Gk[0], Gk[1]);
    wq.wg_cap = G_cap;
    auto set_kind = [&](int k) {
      wq.units = reinterpret_cast<const uint32_t *>(sl.d_wu + w_off_units[k]);
      wq.count = reinterpret_cast<const uint32_t *>(sl.d_wu + w_off_count) + k;  // (stride 2: see the kernel)
      wq.ncell = w_ncell[k];
      wq.wgs = Gk[k];
    };
#define G1S_WG(KIND, BP, SX, SY, BD, GEN)                                                                              \
  do {                                                                                                                 \
    constexpr int lds_ = w_lds_bytes(KIND, WShape<KIND, SX, SY>::BH);                                                  \
    static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k3w_pass<KIND, BP, SX, SY, BD, GEN>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_);             \
    (void)attr_;                                                                                                       \
    char kn_[64];                                                                                                      \
    if (GEN) snprintf(kn_, sizeof(kn_), "k3w_pass<%d, %d, %d, %d, %d, true>", KIND, BP, SX, SY, BD);                   \
    else snprintf(kn_, sizeof(kn_), "k3w_pass<%d, %d, %d, %d>", KIND, BP, SX, SY);                                     \
    kmark(sl, stream, kn_);                                                                                            \
    set_kind(KIND);                                                                                                    \
    wq.rev = (w_rev >> KIND) & 1;                                                                                      \
    hipLaunchKernelGGL((k3w_pass<KIND, BP, SX, SY, BD, GEN>), dim3((uint32_t)Gk[KIND] * B), dim3(kWThreads), lds_, stream, g, wq); \
  } while (0)
#define G1S_W(KIND, BP, SX, SY) G1S_WG(KIND, BP, SX, SY, BP, false)
#define G1S_WB(KIND, SX, SY)                   \
  do {                                         \
    if (g.src_bps == 2) G1S_W(KIND, 2, SX, SY); \
    else G1S_W(KIND, 1, SX, SY);               \
  } while (0)
  // (the general form: wide_ok lets it through for 4:2:0 and for frames without chroma planes)
#define G1S_WGEN(KIND, SX, SY)                                              \
  do {                                                                      \
    if (g.src_bps == 2 && g.den_bps == 2) G1S_WG(KIND, 2, SX, SY, 2, true); \
    else if (g.src_bps == 2) G1S_WG(KIND, 2, SX, SY, 1, true);              \
    else G1S_WG(KIND, 1, SX, SY, 2, true); /* (two 8-bit inputs are never general) */ \
  } while (0)
#define G1S_WK(KIND)                                 \
  do {                                               \
    if (g.xdec == 1 && g.ydec == 1) G1S_WB(KIND, 1, 1); \
    else if (g.xdec == 1) G1S_WB(KIND, 1, 0);        \
    else if (g.ydec == 1) G1S_WB(KIND, 0, 1);        \
    else G1S_WB(KIND, 0, 0);                         \
  } while (0)
    const bool gen = wide_gen(g);
    // G1S_DBG_COREAD=1|2|3 (a measurement aid, profiles/r06b_coread.txt): a pass over the batch's luma source by a kernel of few
    // registers on a stream of its own, started with the luma launch (1), the chroma launch (2) or both (3): what a finder pass
    // costs UNDER an accumulation launch when one of its waves fits beside the launch's four on a SIMD -- three times its own length
    static const int coread = getenv("G1S_DBG_COREAD") ? atoi(getenv("G1S_DBG_COREAD")) : 0;
    auto coread_with_next_launch = [&]() -> int {
      if (!ss.coread) {
        HIP_TRY(hipStreamCreateWithFlags(&ss.coread, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ss.coread_go, hipEventDisableTiming));
      }
      HIP_TRY(hipEventRecord(ss.coread_go, stream));
      HIP_TRY(hipStreamWaitEvent(ss.coread, ss.coread_go, 0));
      hipLaunchKernelGGL(k_dbg_coread, dim3((g.nblocks + 7) / 8, B), dim3(256), 0, ss.coread, ft, g, sl.d_k1);
      return G1S_OK;
    };
    if (coread & 1) {
      const int rc = coread_with_next_launch();
      if (rc) return rc;
    }
    if (!chroma) {
      if (gen) G1S_WGEN(0, -1, -1);
      else G1S_WB(0, -1, -1);
    } else {
      if (gen) G1S_WGEN(0, 1, 1);
      else G1S_WK(0);
      if (coread & 2) {
        const int rc = coread_with_next_launch();
        if (rc) return rc;
      }
      // The chroma launch stays on the main stream behind the luma launch.  Round 3's chain moved it (and what follows) to the
      // copy stream, next to the luma launch of the batch after; with this chain both launches fill every register of the
      // chip and only stretch each other: serial is +2 - 5 % on the 4K job, +10 % at 8K 4:4:4 (profiles/r04_streams.txt).
      static const bool chroma_aside = getenv("G1S_W_ASIDE") != nullptr;  // tuning aid: round 3's placement
      if (side && chroma_aside) {  // the chroma launch and what follows: next to the luma launch of the batch after
        HIP_TRY(hipEventRecord(ss.kernels_done[si], stream));
        HIP_TRY(hipStreamWaitEvent(ss.copy, ss.kernels_done[si], 0));
        stream = ss.copy;
      }
      if (gen) G1S_WGEN(1, 1, 1);
      else G1S_WK(1);
    }
#undef G1S_WGEN
#undef G1S_WG
#undef G1S_WK
#undef G1S_WB
#undef G1S_W
    {
      // debugging aid (G1S_DBG_ONLY=1): how many flat blocks the accumulation launches left to the exact kernel
      static const bool count_only = getenv("G1S_DBG_ONLY") != nullptr;
      if (count_only) {
        std::vector<uint8_t> h(m_only_bytes);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), mp.only, m_only_bytes, hipMemcpyDeviceToHost);
        size_t n[3] = {0, 0, 0};
        for (uint32_t f = 0; f < B; ++f)
          for (int c = 0; c < 3; ++c)
            for (int b = 0; b < g.nblocks; ++b) n[c] += h[((size_t)f * 3 + c) * g.nblocks + b] != 0;
        fprintf(stderr, "deferred to k3_ar_generic: %zu luma, %zu Cb, %zu Cr blocks of %u frames x %d blocks\n", n[0], n[1], n[2], B, g.nblocks);
        // ... and what the wide lists hold: units, plain units, units off the plane's interior, flat blocks in them
        for (int k = 0; k < (chroma ? 2 : 1); ++k) {
          std::vector<uint32_t> cnt(2 * B), ent((size_t)B * w_ncell[k] * kWEntry);
          (void)hipMemcpy(cnt.data(), sl.d_wu + w_off_count, cnt.size() * 4, hipMemcpyDeviceToHost);
          (void)hipMemcpy(ent.data(), sl.d_wu + w_off_units[k], ent.size() * 4, hipMemcpyDeviceToHost);
          size_t units = 0, plain = 0, border = 0, flat = 0, top = 0, runs = 0;
          for (uint32_t f = 0; f < B; ++f)
            for (uint32_t u = 0; u < cnt[2 * f + k]; ++u) {
              const uint32_t *e = &ent[((size_t)f * w_ncell[k] + u) * kWEntry];
              ++units, plain += (e[0] >> 24) & 1u, border += !((e[0] >> 25) & 1u), top += (e[0] >> 26) & 1u, runs += !((e[0] >> 22) & 1u);
              flat += (size_t)__builtin_popcount(e[1]);
            }
          fprintf(stderr, "wide list %d: %zu units (%zu cells), %zu plain, %zu off the interior, %zu with a top halo, %zu runs, %zu flat blocks\n", k, units,
                  (size_t)B * w_ncell[k], plain, border, top, runs, flat);
        }
      }
    }
    // (the record's block statistics and AR sums: k3_ar_generic adds to / overwrites what the launches and the reduction wrote,
    //  and the exact kernel reads the frame number relative to the launch: frame0 is 0 here)
    kmark(sl, stream, "k3w_tail");
    hipLaunchKernelGGL(k3w_tail, dim3(kWTailParts + std::min(kWTailChunks, g.nblocks), g.nplanes, B), dim3(kK3Threads), 0, stream, ft, g,
                       sl.d_records, (const uint8_t *)mp.only, (const uint32_t *)mp.only_any, (const long long *)mp.partials, G_cap, Gk[0], Gk[1]);
  } else {
    // the fused pass: planes of the flat blocks' tiles -> residuals, block statistics, exact int8 SYRK on the matrix
    // cores, one partial system per workgroup; the reducer; then the exact int32 kernel for the few blocks next to a
    // residual outside int8
    const MParams mp = make_mparams(sl);
    FParams fq;
    fq.ft = ft;
    fq.units = mp.units;
    fq.unit_count = mp.unit_count;
    fq.partials = mp.partials;
    fq.ustats = reinterpret_cast<int32_t *>(mp.only + m_only_bytes);
    fq.nunits = m_nunits;
    int G_kind[2] = {m_wgs_per_frame(m_nunits, (int)B, 0), m_wgs_per_frame(m_nunits, (int)B, 1)};  // luma launch, chroma launch
    for (int &Gk : G_kind)
      if ((size_t)Gk * B > m_wg_cap) Gk = m_wgs_per_frame(m_nunits, 1 << 20);  // (G1S_F_WGS raised after the slots were sized: the fewest that hold the units)
    const int G_cap = std::max(G_kind[0], G_kind[1]);
    int G = G_kind[0];
    fq.phase_cycles = nullptr;
    const int cbw = g.nplanes == 3 ? (kBlock >> g.xdec) : 0, cbh = g.nplanes == 3 ? (kBlock >> g.ydec) : 0;
    fq.lplane = sl.d_lplane;
    fq.lpitch = m_lpitch;
    fq.lframe_bytes = m_lframe;
    static const size_t lds_pad = getenv("G1S_F_LDS_PAD") ? (size_t)atoi(getenv("G1S_F_LDS_PAD")) : 0;  // tuning aid: fewer workgroups to a CU
    fq.frames = (int)B;
    fq.wgs = G;
    fq.wg_cap = G_cap;
    // units to workgroups: contiguous runs of the lists (measured +2..4 % over round-robin in the pipelined job, although a
    // kernel alone on the chip is 5 % slower: the runs' loads disturb the kernels next to it less)
    static const int deal_env = getenv("G1S_F_DEAL") ? atoi(getenv("G1S_F_DEAL")) : 1;  // tuning aid
    fq.deal = deal_env;
    { const char *e = getenv("G1S_F_REUSE"); fq.reuse = e ? atoi(e) : 1; }  // test / tuning aid (0: every halo word is read)
    static const int s_dbg = getenv("G1S_S_DBG") ? atoi(getenv("G1S_S_DBG")) : 0;  // timing experiments: parts of k3s_fused left out (wrong results)
    fq.dbg = s_dbg;
    dim3 gr((uint32_t)G * B);
    const int bpsm = g.src_bps == g.den_bps ? g.src_bps : 0;  // bytes per sample at compile time unless the depths are mixed
    // two launches: the luma plane (which leaves L behind), then the two chroma planes
    // (the stream chain is what the wide chain falls back on -- unaligned planes, odd widths, mixed or deep bit depths: ONE
    //  instantiation per format and launch, sample widths at run time)
#define G1S_FS(CW, CH, PL)                                                                                           \
  do {                                                                                                               \
    static const hipError_t attr_rs = hipFuncSetAttribute(reinterpret_cast<const void *>(&k3s_fused<CW, CH, 0, PL>), \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);   \
    (void)attr_rs;                                                                                                   \
    const size_t lds = std::min((size_t)s_lds_bytes(CW, CH, PL) + lds_pad, (size_t)144 * 1024);                      \
    char kn_[64];                                                                                                    \
    snprintf(kn_, sizeof(kn_), "k3s_fused<%d, %d, %d, %d>", CW, CH, 0, PL);                                          \
    kmark(sl, stream, kn_);                                                                                          \
    G = G_kind[PL ? 1 : 0];                                                                                          \
    fq.wgs = G;                                                                                                      \
    gr = dim3((uint32_t)G * B);                                                                                      \
    hipLaunchKernelGGL((k3s_fused<CW, CH, 0, PL>), gr, dim3(kFThreads), lds, stream, g, fq);                         \
  } while (0)
    // (the chroma launch, the finisher and what follows go to the copy stream -- next to the luma launch of the batch after)
    static const bool chroma_aside = getenv("G1S_F_SERIAL") == nullptr;  // tuning aid
#define G1S_FP(CW, CH)                                                    \
  do {                                                                    \
    G1S_FS(CW, CH, 0);                                                    \
    if (side && chroma_aside && !acc_aside) {                             \
      HIP_TRY(hipEventRecord(ss.kernels_done[si], stream));               \
      HIP_TRY(hipStreamWaitEvent(ss.copy, ss.kernels_done[si], 0));       \
      stream = ss.copy;                                                   \
    }                                                                     \
    G1S_FS(CW, CH, 1);                                                    \
  } while (0)
    if (cbw == 0) G1S_FS(0, 0, 0);
    else if (cbw == 16 && cbh == 16) G1S_FP(16, 16);
    else if (cbw == 16) G1S_FP(16, 32);
    else if (cbh == 32) G1S_FP(32, 32);
    else G1S_FP(32, 16);
#undef G1S_FP
#undef G1S_FS
    kmark(sl, stream, "k3m_finish");
    if (!dbg_skip("finish")) hipLaunchKernelGGL(k3m_finish, dim3(kMFinishParts * g.nplanes + kMFinishWgs, B), dim3(256), 0, stream, g, mp, G_kind[0], G_kind[1], G_cap,
                       (const int32_t *)fq.ustats, sl.d_records);
    {
      // debugging aid (G1S_DBG_ONLY=1): how many flat blocks the accumulation launches left to the exact kernel
      static const bool count_only = getenv("G1S_DBG_ONLY") != nullptr;
      if (count_only) {
        std::vector<uint8_t> h(m_only_bytes);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), mp.only, m_only_bytes, hipMemcpyDeviceToHost);
        size_t n[3] = {0, 0, 0};
        for (uint32_t f = 0; f < B; ++f)
          for (int c = 0; c < 3; ++c)
            for (int b = 0; b < g.nblocks; ++b) n[c] += h[((size_t)f * 3 + c) * g.nblocks + b] != 0;
        fprintf(stderr, "deferred to k3_ar_generic: %zu luma, %zu Cb, %zu Cr blocks of %u frames x %d blocks\n", n[0], n[1], n[2], B, g.nblocks);
      }
    }
    kmark(sl, stream, "k3_ar_generic");
    if (!dbg_skip("generic"))
      hipLaunchKernelGGL(k3_ar_generic, dim3(std::min(kK3Chunks, g.nblocks), g.nplanes, B), dim3(kK3Threads), 0, stream, ft, g,
                         sl.d_records, (const uint8_t *)mp.only, (const uint32_t *)mp.only_any);
  }
  kmark(sl, stream, nullptr);
  if (sl.timed) HIP_TRY(hipEventRecord(sl.ev[3], stream));
  HIP_TRY(hipGetLastError());
  // records D2H on the copy stream (behind the tail kernels): the main stream goes straight on to the next batch
  HIP_TRY(hipEventRecord(ss.kernels_done[si], stream));
  HIP_TRY(hipStreamWaitEvent(ss.copy, ss.kernels_done[si], 0));
  if (device_latest) {
    // the per-frame half of the fold where the records lie: the host gets 27 KB of latest state a frame instead of the record
    // (the batch's last record still comes back: g1s_diff_last_record)
    const size_t blob = latest_blob_size(lag), scr = latest_scratch_bytes((uint32_t)L.nblocks);
    if (sl.latest_cap < blob * batch) {
      if (sl.d_latest) (void)hipFree(sl.d_latest);
      if (sl.h_latest) (void)hipHostFree(sl.h_latest);
      sl.d_latest = sl.h_latest = nullptr;
      sl.latest_cap = 0;
      HIP_TRY(hipMalloc((void **)&sl.d_latest, blob * batch));
      HIP_TRY(hipHostMalloc((void **)&sl.h_latest, blob * batch, hipHostMallocDefault));
      sl.latest_cap = blob * batch;
    }
    if (sl.lscratch_cap < scr * batch) {
      if (sl.d_lscratch) (void)hipFree(sl.d_lscratch);
      sl.d_lscratch = nullptr;
      sl.lscratch_cap = 0;
      HIP_TRY(hipMalloc((void **)&sl.d_lscratch, scr * batch));
      sl.lscratch_cap = scr * batch;
    }
    LatestJob job{};
    job.records = sl.d_records;
    job.L = L;
    job.blobs = sl.d_latest;
    job.blob_bytes = blob;
    job.scratch = sl.d_lscratch;
    job.scratch_bytes = scr;
    job.lag = (int)lag;
    job.n = (int)n;
    job.nplanes = g.nplanes;
    job.W = g.W;
    job.H = g.H;
    job.xdec = g.xdec;
    job.ydec = g.ydec;
    job.nbw = g.nbw;
    job.nbh = g.nbh;
    if (sl.timed) {  // (per-kernel timing: everything on the one stream)
      kmark(sl, stream, latest_kernel_name());
      HIP_TRY(launch_latest(job, B, stream));
      kmark(sl, stream, nullptr);
      HIP_TRY(hipEventRecord(ss.kernels_done[si], stream));
      HIP_TRY(hipStreamWaitEvent(ss.copy, ss.kernels_done[si], 0));
    } else {
      if (!ss.latest) {
        // the main stream's priority class (the least urgent: the kernel fills in; measured 4 % better than the runtime's default
        // class, profiles/r05_device_latest.txt).  G1S_LATEST_PRIO (tuning aid): 0 the runtime's default class, 1 the main
        // stream's, 2 the side stream's
        static const int lp = getenv("G1S_LATEST_PRIO") ? atoi(getenv("G1S_LATEST_PRIO")) : 1;
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
        // (made into locals and handed to the stream set -- which goes back to the process-wide cache -- only when ALL of them
        //  exist: a failure half way must not leave a set that looks complete with a null stream or event in it)
        hipStream_t made[2] = {nullptr, nullptr};
        hipEvent_t made_ev[kSlots] = {};
        bool ok = true;
        for (hipStream_t &st : made)
          ok = ok && (lp == 0 ? hipStreamCreateWithFlags(&st, hipStreamNonBlocking) : hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lp == 1 ? plo : phi)) == hipSuccess;
        for (int i = 0; i < kSlots; ++i) ok = ok && hipEventCreateWithFlags(&made_ev[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
          for (hipStream_t st : made)
            if (st) (void)hipStreamDestroy(st);
          for (hipEvent_t e : made_ev)
            if (e) (void)hipEventDestroy(e);
          return fail_hip("the device half's streams / events could not be created");
        }
        ss.latest = made[0], ss.latest2 = made[1];
        for (int i = 0; i < kSlots; ++i) ss.latest_done[i] = made_ev[i];
      }
      static const bool one_latest = getenv("G1S_LATEST_ONE_STREAM") != nullptr;  // (tuning aid: round 4's placement)
      hipStream_t lst = (si & 1) && !one_latest ? ss.latest2 : ss.latest;
      HIP_TRY(hipStreamWaitEvent(lst, ss.kernels_done[si], 0));
      kmark(sl, lst, latest_kernel_name());  // (trace mode)
      HIP_TRY(launch_latest(job, B, lst));
      kmark(sl, lst, nullptr);
      // the blobs' copy: on the copy stream, behind the kernel
      HIP_TRY(hipEventRecord(ss.latest_done[si], lst));
      HIP_TRY(hipStreamWaitEvent(ss.copy, ss.latest_done[si], 0));
    }
    hipStream_t ls = ss.copy;
    if (!sl.timed) kmark(sl, ls, "blobs D2H");
    HIP_TRY(hipMemcpyAsync(sl.h_latest, sl.d_latest, blob * B, hipMemcpyDeviceToHost, ls));
    HIP_TRY(hipMemcpyAsync(sl.h_records + L.size * (B - 1), sl.d_records + L.size * (B - 1), L.size, hipMemcpyDeviceToHost, ls));
    if (!sl.timed) kmark(sl, ls, nullptr);
    HIP_TRY(hipEventRecord(sl.done, ls));
  } else {
    HIP_TRY(hipMemcpyAsync(sl.h_records, sl.d_records, L.size * B, hipMemcpyDeviceToHost, ss.copy));
    HIP_TRY(hipEventRecord(sl.done, ss.copy));
  }
  // profiling aid (G1S_D2H_SYNC=1, with G1S_ONE_STREAM=1): the records copy has ended before the next batch's first kernel
  // starts -- under rocprofv3 the copy is a blit kernel that otherwise shares the chip with k1_moments and doubles its time
  // A timed batch (g1s_diff_set_timing: every kernel between two events, "alone on the chip") waits for it too: the copy of
  // batch N next to the kernels of batch N + 1 costs the luma launch 4 % at 8K, and on some boxes of the pool the event pair of
  // the batch's last kernel read 280 - 500 us instead of 45 - 75 with it in flight (profiles/r04_rot.txt vs r04_hwq.txt).
  static const bool d2h_sync = getenv("G1S_D2H_SYNC") != nullptr;
  if (d2h_sync) HIP_TRY(hipStreamSynchronize(ss.copy));
  else if (sl.timed) HIP_TRY(hipEventSynchronize(sl.done));  // (the copy's end, whichever stream carried it)
  stats.launches_flat_features++;
  stats.launches_flat_select++;
  stats.launches_ar_accumulate++;
  {
    std::unique_lock<std::mutex> lk(dm);
    in_flight.push_back(si);
    ++submitted;
    cv_work.notify_one();
  }
  return G1S_OK;
}

void g1s_diff::drainer_main() {
  (void)hipSetDevice(device);
  for (;;) {
    int si;
    {
      std::unique_lock<std::mutex> lk(dm);
      cv_work.wait(lk, [&] { return drainer_stop || !in_flight.empty(); });
      if (in_flight.empty()) return;  // stop requested and nothing left
      si = in_flight.front();
      in_flight.pop_front();
    }
    const int rc = drain_front(si);
    {
      std::lock_guard<std::mutex> lk(dm);
      if (rc && deferred == G1S_OK) deferred = rc;
      int ok = G1S_OK;
      if (rc) sticky.compare_exchange_strong(ok, rc);  // (the batch is missing from the fold: no table from here on)
      front_failed[si] = rc != G1S_OK;
      fold_q.push_back(si);
    }
    cv_fold.notify_one();
  }
}

void g1s_diff::folder_main() {
  for (;;) {
    int si;
    {
      std::unique_lock<std::mutex> lk(dm);
      cv_fold.wait(lk, [&] { return folder_stop || !fold_q.empty(); });
      if (fold_q.empty()) return;  // stop requested and nothing left
      si = fold_q.front();
      fold_q.pop_front();
    }
    const uint32_t nframes = slots[si].count;  // (drain_back hands the slot back empty)
    const int rc = drain_back(si);
    {
      std::lock_guard<std::mutex> lk(dm);
      if (rc && deferred == G1S_OK) deferred = rc;
      int ok = G1S_OK;
      if (rc) sticky.compare_exchange_strong(ok, rc);
      slot_busy[si] = false;
      ++drained;
      frames_released += nframes;
    }
    cv_free.notify_all();
  }
}

void g1s_diff::wait_drained(uint64_t upto) {
  std::unique_lock<std::mutex> lk(dm);
  cv_free.wait(lk, [&] { return drained >= upto; });
}

int g1s_diff::drain_front(int si) {
  Slot &sl = slots[si];
  std::vector<FrameLatest> &latest = latest_s[si];
  std::vector<uint32_t> &nflat_v = nflat_s[si];
  std::vector<uint8_t> &latest_stage = stage_s[si];
  HIP_TRY(hipEventSynchronize(sl.done));
  if (trace && !sl.timed && trace_base) {
    std::lock_guard<std::mutex> lk(trace_mutex);
    for (size_t i = 0; i < sl.nk; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, trace_base, sl.kev[i]) != hipSuccess) continue;
      char b[160];
      snprintf(b, sizeof(b), "G %12.1f slot %d stream %p %s", ms * 1e3, si, (void *)sl.kstream[i], sl.kname[i].empty() ? "-" : sl.kname[i].c_str());
      trace_lines.emplace_back(b);
    }
    trace_lines.emplace_back(std::string("H ") + std::to_string(trace_now()) + " drained slot " + std::to_string(si));
  }
  if (sl.timed && sl.chain) {
    // (one pair of events around the batch's chain: what the chain takes alone on the chip with nothing between its kernels
    //  but their own dependencies -- the per-kernel events below each put a barrier packet and a signal between two launches)
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, sl.ev[0], sl.ev[3]));
    stats.ms_chain += ms;
    stats.chain_batches += 1;
  } else if (sl.timed) {
    float ms = 0;
    float ms_mom = 0;  // the finder's moments pass
    HIP_TRY(hipEventElapsedTime(&ms_mom, sl.ev[5], sl.ev[4]));
    HIP_TRY(hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]));
    stats.ms_flat_features += ms;
    stats.ms_residual += ms_mom;
    HIP_TRY(hipEventElapsedTime(&ms, sl.ev[1], sl.ev[2]));
    stats.ms_flat_select += ms;
    HIP_TRY(hipEventElapsedTime(&ms, sl.ev[2], sl.ev[3]));
    stats.ms_ar_accumulate += ms;
    HIP_TRY(hipEventElapsedTime(&ms, sl.ev[0], sl.ev[3]));
    stats.ms_total_gpu += ms;
    {
      std::lock_guard<std::mutex> lk(ktimes_mutex);
      for (size_t i = 0; i + 1 < sl.nk; ++i) {
        if (sl.kname[i].empty()) continue;  // (the gap between the two halves of a batch)
        HIP_TRY(hipEventElapsedTime(&ms, sl.kev[i], sl.kev[i + 1]));
        auto &kt = ktimes[sl.kname[i]];
        kt.first += ms;
        kt.second += 1;
      }
    }
    {
      std::vector<uint32_t> cnt(batch);
      HIP_TRY(hipMemcpy(cnt.data(), reinterpret_cast<uint32_t *>(sl.d_k1) + (size_t)geom.nblocks * batch * (kMomInts + 1),
                        sizeof(uint32_t) * batch, hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < sl.count; ++i) stats.literal_blocks += cnt[i];
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (latest.size() < sl.count) latest.resize(sl.count);
  nflat_v.assign(sl.count, 0);
  // ---- per-frame half, concurrent: header, symmetric mirror, latest noise state ----
  const size_t blob = latest_only ? latest_blob_size(lag) : 0;
  if (latest_only) latest_stage.resize(blob * sl.count);
  auto finish_record = [&](int i) {
    uint8_t *rec = sl.h_records + L.size * i;
    RecHeader h{};
    h.magic = kRecMagic;
    h.lag = lag;
    h.width = shape.width;
    h.height = shape.height;
    h.xdec = shape.xdec;
    h.ydec = shape.ydec;
    h.nplanes = (uint32_t)geom.nplanes;
    h.nbw = (uint32_t)geom.nbw;
    h.nbh = (uint32_t)geom.nbh;
    h.n = n;
    h.size_bytes = L.size;
    const uint8_t *mask = rec + L.off_mask;
    uint32_t nflat = 0;
    for (uint32_t b = 0; b < L.nblocks; ++b) nflat += mask[b] != 0;
    h.status = nflat;
    nflat_v[i] = nflat;
    std::memcpy(rec, &h, sizeof(h));
    // mirror the symmetric AR sums so consumers see full matrices
    for (int c = 0; c < geom.nplanes; ++c) {
      int64_t *S = reinterpret_cast<int64_t *>(rec + L.off_ar[c]);
      const int nc = (int)n + (c > 0);
      for (int a = 0; a < nc; ++a)
        for (int b = a + 1; b < nc; ++b) S[b * nc + a] = S[a * nc + b];
    }
    return rec;
  };
  if (device_latest) {
    // the latest states were computed on the device (k4_latest): nothing per frame is left but reading the headers
    const size_t bl = latest_blob_size(lag);
    std::vector<FrameView> &views = views_s[si];
    if (views.size() < sl.count) views.resize(sl.count);
    int rc = G1S_OK;
    for (uint32_t i = 0; i < sl.count; ++i) {
      uint8_t *b = sl.h_latest + bl * i;
      LatestHeader *h = reinterpret_cast<LatestHeader *>(b);
      nflat_v[i] = h->reserved;
      h->reserved = 0;  // (the kernel's note to this function; the blob a rank sends is the blob the host half would make)
      if (!latest_only && view_of_blob(b, bl, lag, views[i]) != G1S_OK) rc = G1S_ERR_INVALID;
    }
    if (latest_only) latest_stage.assign(sl.h_latest, sl.h_latest + bl * sl.count);
    if (sl.count) finish_record((int)sl.count - 1);
    ms_fold_front += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc) {
      std::lock_guard<std::mutex> lk(dm);
      err = "k4_latest wrote a blob the fold does not recognise";
    }
    return rc;
  }
  auto per_frame = [&](int i) {
    uint8_t *rec = finish_record(i);
    if (!records_only) compute_latest(rec, L.size, lag, latest[i]);
    if (latest_only) latest_to_blob(latest[i], lag, latest_stage.data() + (size_t)i * blob);
  };
  if (pool && sl.count > 1) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    pool->parallel_for((int)sl.count, per_frame);
  } else {
    for (uint32_t i = 0; i < sl.count; ++i) per_frame((int)i);
  }
  ms_fold_front += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return G1S_OK;
}

// ---- ordered half, serial in frame order; runs next to the per-frame half of the following batch ----
int g1s_diff::drain_back(int si) {
  Slot &sl = slots[si];
  std::vector<FrameLatest> &latest = latest_s[si];
  const std::vector<uint32_t> &nflat_v = nflat_s[si];
  const std::vector<uint8_t> &latest_stage = stage_s[si];
  const size_t blob = latest_only ? latest_blob_size(lag) : 0;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = G1S_OK;
  if (front_failed[si]) {
    sl.count = 0;
    return G1S_OK;  // (the error is already in `deferred`)
  }
  for (uint32_t i = 0; i < sl.count; ++i) {
    uint8_t *rec = sl.h_records + L.size * i;
    stats.frames++;
    stats.blocks += L.nblocks;
    stats.flat_blocks += nflat_v[i];
    if (records_only) {
      std::lock_guard<std::mutex> lk(dm);
      records_out.insert(records_out.end(), rec, rec + L.size);
      records_out_frames++;
    } else if (latest_only) {
      std::lock_guard<std::mutex> lk(dm);
      latest_out.insert(latest_out.end(), latest_stage.begin() + blob * i, latest_stage.begin() + blob * (i + 1));
      latest_out_frames++;
    }
  }
  if (!records_only && !latest_only && sticky == G1S_OK && sl.count) {
    // the ordered merge of the batch: combined-model solves in parallel, tests and commits in order
    Pool *mp = merge_pool();  // (the shared pool is busy with the next batch's per-frame half)
    const NoiseFold::ParallelFor pfor = [&](int m, const std::function<void(int)> &fn) {
      if (mp) {
        std::lock_guard<std::mutex> lk(g_merge_pool_mutex);
        mp->parallel_for(m, fn);
      } else {
        for (int i = 0; i < m; ++i) fn(i);
      }
    };
    rc = device_latest ? fold->push_latest_many(views_s[si].data(), sl.count, pfor) : fold->push_latest_many(latest.data(), sl.count, pfor);
    if (rc) {
      std::lock_guard<std::mutex> lk(dm);
      err = fold->error();
      sticky = rc;
    }
  }
  if (latest_only && sl.count) {
    std::lock_guard<std::mutex> lk(dm);
    latest_batches.push_back(sl.count);
  }
  if (sl.count) last_record.assign(sl.h_records + L.size * (sl.count - 1), sl.h_records + L.size * sl.count);
  sl.count = 0;
  ms_fold_back += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int g1s_diff::drain_all() {
  uint64_t upto;
  {
    std::lock_guard<std::mutex> lk(dm);
    upto = submitted;
  }
  wait_drained(upto);
  return G1S_OK;  // errors of drained batches are in `deferred` / `sticky`
}

void g1s_diff::release() {
  if (drainer.joinable()) {
    {
      std::lock_guard<std::mutex> lk(dm);
      drainer_stop = true;
    }
    cv_work.notify_all();
    drainer.join();  // (it drains whatever was still queued first)
  }
  if (folder.joinable()) {
    {
      std::lock_guard<std::mutex> lk(dm);
      folder_stop = true;
    }
    cv_fold.notify_all();
    folder.join();
  }
  if (ss.compute) (void)hipStreamSynchronize(ss.compute);
  if (ss.copy) (void)hipStreamSynchronize(ss.copy);
  if (ss.latest) (void)hipStreamSynchronize(ss.latest);
  if (ss.latest2) (void)hipStreamSynchronize(ss.latest2);
  if (ss.flat) (void)hipStreamSynchronize(ss.flat);
  if (ss.flat2) (void)hipStreamSynchronize(ss.flat2);
  if (ss.mom) (void)hipStreamSynchronize(ss.mom);
  if (ss.upload) (void)hipStreamSynchronize(ss.upload);
  if (ss.coread) (void)hipStreamSynchronize(ss.coread);
  // (the trace: written when nothing can add to it any more -- the drainer and the folder are joined, the streams idle)
  if (trace) {
    std::lock_guard<std::mutex> lk(trace_mutex);
    if (!trace_lines.empty()) {
      if (FILE *f = fopen(getenv("G1S_TRACE"), "a")) {
        for (const auto &l : trace_lines) fprintf(f, "%s\n", l.c_str());
        fclose(f);
      }
      trace_lines.clear();
    }
  }
  release_streams(ss);
  for (Slot &sl : slots) {
    if (sl.h_planes && geometry_set) {  // park the buffers for the next generator of this geometry
      std::lock_guard<std::mutex> lk(g_cache_mutex);
      if (g_slot_cache.size() < 8) {
        sl.count = 0;
        g_slot_cache.push_back(CachedSlot{slot_key, sl});
        sl = Slot{};
        continue;
      }
    }
    free_slot(sl);
  }
  for (auto &pe : h2d_pending) (void)hipEventDestroy(pe.second);
  h2d_pending.clear();
  for (auto &e : h2d_free) (void)hipEventDestroy(e);
  h2d_free.clear();
  if (h2d_order) (void)hipEventDestroy(h2d_order);
  h2d_order = nullptr;
  d_lut = nullptr;  // shared per device
  stream = nullptr;
  delete fold;
  fold = nullptr;
  pool = nullptr;  // shared
}

// =============================================================== C ABI =====
extern "C" {

const char *g1s_last_global_error(void) { return g_global_error.c_str(); }

g1s_diff_t *g1s_diff_new(int64_t fps_num, int64_t fps_den, uint32_t source_bit_depth,
                         uint32_t denoised_bit_depth, const g1s_opts_t *opts) {
  g_global_error.clear();
  if (fps_num <= 0 || fps_den <= 0) {
    g_global_error = "frame rate must be positive";
    return nullptr;
  }
  // src/main.rs:515-517: "Bit depths not between 8-16 are not currently supported"
  if (source_bit_depth < 8 || source_bit_depth > 16 || denoised_bit_depth < 8 || denoised_bit_depth > 16) {
    g_global_error = "Bit depths not between 8-16 are not currently supported";
    return nullptr;
  }
  uint32_t lag = 3, batch = kDefaultBatch;
  bool batch_auto = true;
  bool luma_only = false, records_only = false, latest_only = false;
  int device = -1;
  if (opts) {
    if (opts->struct_size != sizeof(g1s_opts_t)) {
      g_global_error = "g1s_opts_t.struct_size mismatch";
      return nullptr;
    }
    if (opts->ar_coeff_lag) lag = opts->ar_coeff_lag;
    if (opts->batch_frames) {
      batch = std::min<uint32_t>(opts->batch_frames, (uint32_t)kMaxBatch);
      batch_auto = false;
    }
    luma_only = opts->luma_only != 0;
    records_only = opts->records_only == 1;
    latest_only = opts->records_only == 2;
    device = opts->device;
  }
  if (lag < 1 || lag > 3) {
    g_global_error = "ar_coeff_lag must be 1..3";
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_global_error = "no HIP device available: the diff estimator has no CPU fallback";
    return nullptr;
  }
  if (device >= 0) {
    if (hipSetDevice(device) != hipSuccess) {
      g_global_error = "hipSetDevice failed";
      return nullptr;
    }
  } else if (hipGetDevice(&device) != hipSuccess) {
    g_global_error = "hipGetDevice failed";
    return nullptr;
  }
  g1s_diff *g = new g1s_diff();
  g->fps_num = fps_num;
  g->fps_den = fps_den;
  g->src_bd = source_bit_depth;
  g->den_bd = denoised_bit_depth;
  g->lag = lag;
  g->n = num_coeffs(lag);
  g->luma_only = luma_only;
  g->records_only = records_only;
  g->latest_only = latest_only;
  {
    // G1S_LATEST=host|device: where the per-frame half of the fold runs (records_only generators hand out records: host)
    const char *e = getenv("G1S_LATEST");
    const bool dev = e ? std::string(e) == "device" : kDeviceLatestDefault;
    g->device_latest = dev && !records_only;
  }
  g->batch = batch;
  g->batch_auto = batch_auto;
  g->device = device;
  make_flat_consts(g->fc);
  bool streams_ok = acquire_streams(device, g->ss);
  g->stream = g->ss.compute;
  // the p/255 table is the same for every generator: one device copy per device, kept
  {
    static std::mutex lut_mutex;
    static double *lut_dev[64] = {nullptr};
    std::lock_guard<std::mutex> lk(lut_mutex);
    const int di = device & 63;
    if (!lut_dev[di]) {
      double lut[256];
      for (int i = 0; i < 256; ++i) lut[i] = ((double)i) / 255.0;  // block normalisation, on the host
      if (hipMalloc((void **)&lut_dev[di], sizeof(lut)) != hipSuccess ||
          hipMemcpy(lut_dev[di], lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess) {
        lut_dev[di] = nullptr;
        streams_ok = false;
      }
    }
    g->d_lut = lut_dev[di];
  }
  if (!streams_ok) {
    g_global_error = std::string("HIP initialisation failed: ") + hipGetErrorString(hipGetLastError());
    g->release();
    delete g;
    return nullptr;
  }
  if (!records_only && !latest_only) g->fold = new NoiseFold(fps_num, fps_den, lag);
  g->pool = shared_pool();
  if (getenv("G1S_TRACE")) {
    g->trace = hipEventCreate(&g->trace_base) == hipSuccess && hipEventRecord(g->trace_base, g->ss.compute) == hipSuccess &&
               hipEventSynchronize(g->trace_base) == hipSuccess;
    g->trace_host0 = std::chrono::steady_clock::now();
  }
  g->drainer = std::thread([g] { g->drainer_main(); });
  g->folder = std::thread([g] { g->folder_main(); });
  return g;
}

// errors of queued frames surface once, on a later call (the drainer thread records them)
static int take_deferred(g1s_diff *g) {
  std::lock_guard<std::mutex> lk(g->dm);
  if (g->sticky) return g->sticky;
  const int rc = g->deferred;
  g->deferred = G1S_OK;
  return rc;
}

int g1s_diff_frame(g1s_diff_t *g, const g1s_frame_t *source, const g1s_frame_t *denoised) {
  if (!g) return G1S_ERR_INVALID;
  if (g->finished) return g->fail(G1S_ERR_STATE, "generator already finished");
  {
    const int pending = take_deferred(g);
    if (pending) return pending;
  }
  (void)hipSetDevice(g->device);
  const int rc = g->append(source, denoised);
  if (rc) return rc;
  return take_deferred(g);
}

int g1s_diff_frames(g1s_diff_t *g, const g1s_frame_t *source, const g1s_frame_t *denoised, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const int rc = g1s_diff_frame(g, source + i, denoised + i);
    if (rc) return rc;
  }
  return G1S_OK;
}

int g1s_diff_sync(g1s_diff_t *g) {
  if (!g) return G1S_ERR_INVALID;
  (void)hipSetDevice(g->device);
  if (g->geometry_set) {
    int rc = g->submit(g->cur);
    if (rc) return rc;
    rc = g->flush_pending();
    if (rc) return rc;
    (void)g->drain_all();
  }
  return take_deferred(g);
}

int g1s_diff_finish(g1s_diff_t *g, g1s_segment_t *out, size_t cap, size_t *n_out) {
  if (!g) return G1S_ERR_INVALID;
  if (g->records_only || g->latest_only)
    return g->fail(G1S_ERR_STATE, "records_only / latest_only generator: use g1s_diff_take_* + g1s_fold_*");
  // The segments stay in the object: a call whose buffer is too small reports the count and loses nothing -- the caller
  // sizes the buffer from *n_out and calls again (the reference's Vec has no cap: src/main.rs:524).
  if (!g->finished) {
    const int rc = g1s_diff_sync(g);
    if (rc) return rc;
    g->fold->finish(g->final_segs);
    g->finished = true;  // no more frames
  }
  const std::vector<g1s_segment_t> &segs = g->final_segs;
  if (n_out) *n_out = segs.size();
  if (segs.size() > cap || (!out && !segs.empty())) return g->fail(G1S_ERR_CAPACITY, "segment buffer too small");
  if (!segs.empty()) std::memcpy(out, segs.data(), sizeof(g1s_segment_t) * segs.size());
  return G1S_OK;
}

void g1s_diff_free(g1s_diff_t *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  g->release();
  delete g;
}

const char *g1s_diff_last_error(const g1s_diff_t *g) { return g ? g->err.c_str() : ""; }
// (internal, ingest.cpp: the frame-pair loop prefixes errors with the index of the pair)
uint32_t g1s_diff_source_bit_depth_(const g1s_diff_t *g) { return g ? g->src_bd : 0; }
int32_t g1s_diff_device_(const g1s_diff_t *g) { return g ? g->device : -1; }
// frames the generator can hold at once (its slots x the launch group); 0 until the first frame has set the geometry
uint32_t g1s_diff_frames_in_flight_max_(const g1s_diff_t *g) { return g && g->shape.width ? (uint32_t)kSlots * g->batch : 0; }
void g1s_diff_set_error_text_(g1s_diff_t *g, const char *msg) {
  if (g && msg) g->err = msg;
}

size_t g1s_record_size(uint32_t width, uint32_t height, uint32_t xdec, uint32_t ydec, uint32_t nplanes,
                       uint32_t lag) {
  (void)xdec;
  (void)ydec;
  return make_layout(width, height, nplanes, lag).size;
}

int g1s_record_init(void *rec, size_t cap_bytes, uint32_t width, uint32_t height, uint32_t xdec,
                    uint32_t ydec, uint32_t nplanes, uint32_t lag) {
  if (!rec || lag < 1 || lag > 3 || (nplanes != 1 && nplanes != 3)) return G1S_ERR_INVALID;
  const RecLayout L = make_layout(width, height, nplanes, lag);
  if (L.size > cap_bytes) return G1S_ERR_CAPACITY;
  std::memset(rec, 0, L.size);
  RecHeader h{};
  h.magic = kRecMagic;
  h.lag = lag;
  h.width = width;
  h.height = height;
  h.xdec = xdec;
  h.ydec = ydec;
  h.nplanes = nplanes;
  h.nbw = (width + kBlock - 1) / kBlock;
  h.nbh = (height + kBlock - 1) / kBlock;
  h.n = num_coeffs(lag);
  h.size_bytes = L.size;
  std::memcpy(rec, &h, sizeof(h));
  return G1S_OK;
}

int g1s_diff_take_records(g1s_diff_t *g, void *buf, size_t cap_bytes, size_t *n_frames) {
  if (!g) return G1S_ERR_INVALID;
  if (!g->records_only) return g->fail(G1S_ERR_STATE, "not a records_only generator");
  const int rc = g1s_diff_sync(g);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g->dm);
  if (n_frames) *n_frames = g->records_out_frames;
  if (g->records_out.size() > cap_bytes) return g->fail(G1S_ERR_CAPACITY, "record buffer too small");
  if (!g->records_out.empty()) std::memcpy(buf, g->records_out.data(), g->records_out.size());
  g->records_out.clear();
  g->records_out_frames = 0;
  return G1S_OK;
}

int g1s_diff_take_latest(g1s_diff_t *g, int sync, void *buf, size_t cap_bytes, size_t *n_frames) {
  if (!g) return G1S_ERR_INVALID;
  if (!g->latest_only) return g->fail(G1S_ERR_STATE, "not a latest_only generator");
  if (sync) {
    const int rc = g1s_diff_sync(g);
    if (rc) return rc;
  } else {
    // everything but the two most recently queued batches has been delivered when this returns
    uint64_t upto;
    {
      std::lock_guard<std::mutex> lk(g->dm);
      upto = g->submitted >= 2 ? g->submitted - 2 : 0;
    }
    g->wait_drained(upto);
  }
  std::lock_guard<std::mutex> lk(g->dm);
  // whole batches, in order: all of them after a sync, otherwise exactly those before the two most recent
  // (so that ranks that run ahead by different amounts still deliver the same batches in the same round)
  const uint64_t upto_batch = sync ? g->delivered + g->latest_batches.size()
                                   : std::min<uint64_t>(g->delivered + g->latest_batches.size(),
                                                        g->submitted >= 2 ? g->submitted - 2 : 0);
  size_t frames = 0;
  uint64_t nb = 0;
  while (g->delivered + nb < upto_batch) frames += g->latest_batches[nb++];
  const size_t bs = latest_blob_size(g->lag);
  if (n_frames) *n_frames = frames;
  if (frames * bs > cap_bytes) return g->fail(G1S_ERR_CAPACITY, "latest buffer too small");
  if (frames) std::memcpy(buf, g->latest_out.data(), frames * bs);
  g->latest_out.erase(g->latest_out.begin(), g->latest_out.begin() + frames * bs);
  g->latest_out_frames -= frames;
  for (uint64_t i = 0; i < nb; ++i) g->latest_batches.pop_front();
  g->delivered += nb;
  return G1S_OK;
}

// ---- frame-shard rounds: the exchange protocol (what goes into a round's message, which batch, in which order the root
//      merges) lives here; the transport (RCCL / MPI / torch.distributed gather of fixed-size buffers) stays with the host
namespace {
constexpr uint32_t kShardMagic = 0x4d315347u;  // "GS1M"
constexpr uint32_t kShardNoIndex = 0xffffffffu;  // a message without a batch index: merged in arrival order
struct ShardHeader {
  uint32_t magic, count, lag, batch_frames;
  // which of the SENDING rank's batches this is (0, 1, ...): global batch = local_batch * world + rank.  The root merges
  // by this index, not by arrival: ranks that have fed different numbers of batches (an idle rank in a short last round)
  // send different local batches in the same round
  uint32_t local_batch, reserved;
};
}  // namespace
size_t g1s_shard_msg_size(uint32_t ar_coeff_lag, uint32_t batch_frames) {
  return ar_coeff_lag >= 1 && ar_coeff_lag <= 3 ? sizeof(ShardHeader) + (size_t)batch_frames * latest_blob_size(ar_coeff_lag) : 0;
}
int g1s_shard_msg_from_latest_at(const void *blobs, size_t n, uint32_t ar_coeff_lag, uint32_t batch_frames, uint64_t local_batch,
                                 void *msg, size_t cap_bytes) {
  if (!msg || (!blobs && n) || ar_coeff_lag < 1 || ar_coeff_lag > 3 || n > batch_frames) return G1S_ERR_INVALID;
  if (local_batch != G1S_SHARD_NO_INDEX && local_batch >= kShardNoIndex) return G1S_ERR_INVALID;
  const size_t total = g1s_shard_msg_size(ar_coeff_lag, batch_frames), bs = latest_blob_size(ar_coeff_lag);
  if (cap_bytes < total) return G1S_ERR_CAPACITY;
  std::memset(msg, 0, total);
  const ShardHeader h{kShardMagic, (uint32_t)n, ar_coeff_lag, batch_frames,
                      local_batch == G1S_SHARD_NO_INDEX ? kShardNoIndex : (uint32_t)local_batch, 0u};
  std::memcpy(msg, &h, sizeof(h));
  if (n) std::memcpy((uint8_t *)msg + sizeof(h), blobs, n * bs);
  return G1S_OK;
}
int g1s_shard_msg_from_latest(const void *blobs, size_t n, uint32_t ar_coeff_lag, uint32_t batch_frames, void *msg, size_t cap_bytes) {
  return g1s_shard_msg_from_latest_at(blobs, n, ar_coeff_lag, batch_frames, G1S_SHARD_NO_INDEX, msg, cap_bytes);
}
int g1s_shard_pack(g1s_diff_t *g, int flush, void *msg, size_t cap_bytes) {
  if (!g || !msg) return G1S_ERR_INVALID;
  if (!g->latest_only) return g->fail(G1S_ERR_STATE, "not a latest_only generator (records_only = 2)");
  const size_t total = g1s_shard_msg_size(g->lag, g->batch), bs = latest_blob_size(g->lag);
  if (cap_bytes < total) return g->fail(G1S_ERR_CAPACITY, "shard message buffer too small");
  // flush = 0: whatever is ready goes out, nothing is waited for -- the root orders by the batch index in the message, so the
  // ranks need not send the same batch in the same round (they did, and waited for it, when the root merged by arrival: the
  // feeding thread then ran at most two batches ahead of the drain, 3.5 % of a rank's throughput).  A rank is never more than
  // the generator's slots (g1s_shard_flush_rounds) behind with its messages: a round sends nothing only when every unsent batch is still in a slot.
  if (flush) {
    const int rc = g1s_diff_sync(g);
    if (rc) return rc;
  } else {
    const int pending = take_deferred(g);
    if (pending) return pending;
  }
  std::lock_guard<std::mutex> lk(g->dm);
  size_t n = 0;
  uint64_t local_batch = G1S_SHARD_NO_INDEX;
  if (!g->latest_batches.empty()) {  // ONE batch a round, the oldest not sent yet
    n = g->latest_batches.front();
    g->latest_batches.pop_front();
    local_batch = g->delivered;  // (the message says which one: the root orders by it)
    g->delivered += 1;
  }
  const int rc = g1s_shard_msg_from_latest_at(n ? g->latest_out.data() : nullptr, n, g->lag, g->batch, local_batch, msg, cap_bytes);
  if (rc) return rc;
  g->latest_out.erase(g->latest_out.begin(), g->latest_out.begin() + n * bs);
  g->latest_out_frames -= n;
  return G1S_OK;
}

unsigned g1s_shard_flush_rounds(void) { return (unsigned)kSlots; }

size_t g1s_latest_size(uint32_t ar_coeff_lag) { return ar_coeff_lag >= 1 && ar_coeff_lag <= 3 ? latest_blob_size(ar_coeff_lag) : 0; }

int g1s_latest_from_record(const void *record, size_t size_bytes, uint32_t ar_coeff_lag, void *blob, size_t cap_bytes) {
  if (!record || !blob || ar_coeff_lag < 1 || ar_coeff_lag > 3) return G1S_ERR_INVALID;
  if (cap_bytes < latest_blob_size(ar_coeff_lag)) return G1S_ERR_CAPACITY;
  FrameLatest fl;
  compute_latest((const uint8_t *)record, size_bytes, ar_coeff_lag, fl);  // a failure travels inside the blob
  latest_to_blob(fl, ar_coeff_lag, (uint8_t *)blob);
  return G1S_OK;
}

// The per-frame half of a batch of records on the process' per-frame pool (G1S_FOLD_THREADS; the calling thread takes part):
// what a generator's drainer does with a batch, as a call of its own -- a host that runs the half next to a foreign transport,
// and tools/host_budget_8ranks.py, which replays eight ranks' worth of it.
int g1s_latest_from_records(const void *records, size_t stride_bytes, size_t n, uint32_t ar_coeff_lag, void *blobs, size_t blob_stride_bytes) {
  if ((!records || !blobs) && n) return G1S_ERR_INVALID;
  if (ar_coeff_lag < 1 || ar_coeff_lag > 3) return G1S_ERR_INVALID;
  const size_t bs = latest_blob_size(ar_coeff_lag);
  if (blob_stride_bytes < bs) return G1S_ERR_CAPACITY;
  auto one = [&](int i) {
    static thread_local FrameLatest fl;  // (kept per thread: its vectors are sized once, not once a frame)
    compute_latest((const uint8_t *)records + (size_t)i * stride_bytes, stride_bytes, ar_coeff_lag, fl);
    latest_to_blob(fl, ar_coeff_lag, (uint8_t *)blobs + (size_t)i * blob_stride_bytes);
  };
  Pool *p = shared_pool();
  if (p && n > 1) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    p->parallel_for((int)n, one);
  } else {
    for (size_t i = 0; i < n; ++i) one((int)i);
  }
  return G1S_OK;
}

unsigned g1s_usable_cpus(void) { return usable_cpus(); }

struct g1s_fold {
  NoiseFold fold;
  uint32_t lag;
  std::string err;
  bool finished = false;
  std::vector<g1s_segment_t> final_segs;  // what finish() returned (kept: a too-small buffer can be retried)
  Pool *pool = nullptr;
  std::vector<FrameLatest> latest;
  std::vector<FrameView> views;  // g1s_fold_push_latest: the blobs of a pass, read in place
  // g1s_shard_merge: indexed batches that arrived ahead of the next one in the global order (global batch -> its states)
  std::map<uint64_t, std::vector<uint8_t>> early;
  uint64_t next_batch = 0;
  g1s_fold(int64_t a, int64_t b, uint32_t lag_) : fold(a, b, lag_), lag(lag_) {}
};

g1s_fold_t *g1s_fold_new(int64_t fps_num, int64_t fps_den, uint32_t lag) {
  if (fps_num <= 0 || fps_den <= 0 || lag < 1 || lag > 3) return nullptr;
  return new g1s_fold(fps_num, fps_den, lag);
}
int g1s_fold_push(g1s_fold_t *f, const void *record, size_t size_bytes) {
  if (!f || !record) return G1S_ERR_INVALID;
  if (f->finished) return G1S_ERR_STATE;
  const int rc = f->fold.push((const uint8_t *)record, size_bytes);
  if (rc) f->err = f->fold.error();
  return rc;
}
int g1s_fold_push_many(g1s_fold_t *f, const void *records, size_t stride_bytes, size_t n) {
  if (!f || (!records && n)) return G1S_ERR_INVALID;
  if (f->finished) return G1S_ERR_STATE;
  if (!f->pool) f->pool = shared_pool();
  const uint8_t *base = (const uint8_t *)records;
  const size_t chunk = 64;
  for (size_t o = 0; o < n; o += chunk) {
    const size_t m = std::min(chunk, n - o);
    if (f->latest.size() < m) f->latest.resize(m);
    auto one = [&](int i) { compute_latest(base + (o + i) * stride_bytes, stride_bytes, f->lag, f->latest[i]); };
    if (f->pool && m > 1) {
      std::lock_guard<std::mutex> lk(g_pool_mutex);
      f->pool->parallel_for((int)m, one);
    } else
      for (size_t i = 0; i < m; ++i) one((int)i);
    for (size_t i = 0; i < m; ++i) {
      const int rc = f->fold.push_latest(f->latest[i]);
      if (rc) {
        f->err = f->fold.error();
        return rc;
      }
    }
  }
  return G1S_OK;
}
// Runs of latest-state blobs, merged in the order given.  The frames of ALL runs are taken in windows of kChunk frames (the
// solves of a window run on the merge pool, fold.cpp: push_latest_many): a round of a frame-shard job -- eight messages of
// one batch each -- is merged as two windows of 256, not eight of 64 (the pool's hand-over per window is what a small
// window pays: 3.8 -> 5.5 us a frame single-threaded at 64, profiles/r04_host_budget_8ranks.txt).
struct BlobRun {
  const uint8_t *base;
  size_t stride, n;
};
static int fold_push_runs(g1s_fold_t *f, const BlobRun *runs, size_t nruns) {
  if (f->finished) return G1S_ERR_STATE;
  Pool *mp = merge_pool();
  constexpr size_t kChunk = 256;  // frames parsed and merged per pass (bounds the staging memory)
  const NoiseFold::ParallelFor pfor = [&](int m, const std::function<void(int)> &fn) {
    if (mp && m > 1) {
      std::lock_guard<std::mutex> lk(g_merge_pool_mutex);
      mp->parallel_for(m, fn);
    } else {
      for (int i = 0; i < m; ++i) fn(i);
    }
  };
  static struct ParseProfile {  // G1S_FOLD_PROFILE=1: the whole call next to the fold's own stage timers
    bool on = getenv("G1S_FOLD_PROFILE") != nullptr;
    double s = 0, all = 0;
    size_t frames = 0;
    ~ParseProfile() {
      if (on && frames) fprintf(stderr, "ordered merge, us per frame: blob headers %.2f, whole call %.2f (%zu frames)\n", s * 1e6 / frames, all * 1e6 / frames, frames);
    }
  } pp;
  if (f->views.size() < kChunk) f->views.resize(kChunk);
  size_t run = 0, at = 0;  // the next frame to take: frame `at` of run `run`
  for (;;) {
    while (run < nruns && at == runs[run].n) {
      ++run;
      at = 0;
    }
    if (run == nruns) return G1S_OK;
    const auto t_p0 = std::chrono::steady_clock::now();
    size_t good = 0;
    int bad_rc = G1S_OK;
    while (good < kChunk && run < nruns) {
      if (at == runs[run].n) {
        ++run;
        at = 0;
        continue;
      }
      const BlobRun &R = runs[run];
      const uint8_t *b = R.base + at * R.stride;
      // The blobs are read where they lie (fold.h, FrameView); only a caller's unaligned buffer is copied first.
      int rc;
      if (!((reinterpret_cast<uintptr_t>(R.base) | R.stride) & 7)) {
        rc = view_of_blob(b, R.stride, f->lag, f->views[good]);
      } else {
        if (f->latest.size() < kChunk) f->latest.resize(kChunk);
        rc = latest_from_blob(b, R.stride, f->lag, f->latest[good]);
        if (!rc) view_of(f->latest[good], f->views[good]);
      }
      if (rc) {
        bad_rc = rc;
        break;
      }
      ++good;
      ++at;
    }
    const auto t_p1 = std::chrono::steady_clock::now();
    const int rc = f->fold.push_latest_many(f->views.data(), good, pfor);  // (the frames before a bad blob still count)
    if (pp.on) {
      pp.s += std::chrono::duration<double>(t_p1 - t_p0).count();
      pp.all += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_p0).count();
      pp.frames += good;
    }
    if (rc) {
      f->err = f->fold.error();
      return rc;
    }
    if (bad_rc) {
      f->err = "bad latest blob";
      return bad_rc;
    }
  }
}
int g1s_fold_push_latest(g1s_fold_t *f, const void *blobs, size_t stride_bytes, size_t n) {
  if (!f || (!blobs && n)) return G1S_ERR_INVALID;
  const BlobRun r{(const uint8_t *)blobs, stride_bytes, n};
  return fold_push_runs(f, &r, 1);
}
int g1s_fold_finish(g1s_fold_t *f, g1s_segment_t *out, size_t cap, size_t *n_out) {
  if (!f) return G1S_ERR_INVALID;
  if (!f->early.empty()) {  // (a caller that stopped before the flush rounds, or a rank that skipped a batch)
    f->err = "frame-shard merge: batch " + std::to_string(f->next_batch) + " never arrived (" + std::to_string(f->early.size()) +
             " later batch(es) are waiting for it)";
    return G1S_ERR_STATE;
  }
  if (!f->finished) {
    f->fold.finish(f->final_segs);
    f->finished = true;  // no more records; the segments stay here, so a too-small buffer can be retried
  }
  const std::vector<g1s_segment_t> &segs = f->final_segs;
  if (n_out) *n_out = segs.size();
  if (segs.size() > cap || (!out && !segs.empty())) {
    f->err = "segment buffer too small";
    return G1S_ERR_CAPACITY;
  }
  if (!segs.empty()) std::memcpy(out, segs.data(), sizeof(g1s_segment_t) * segs.size());
  return G1S_OK;
}
int g1s_shard_merge(g1s_fold_t *f, const void *msgs, size_t stride_bytes, uint32_t world) {
  if (!f || !msgs || !world) return G1S_ERR_INVALID;
  // Global frame order: batch j of the video went to rank j % world, and a message says which of its rank's batches it
  // carries, so global batch = local_batch * world + rank.  Batches are merged strictly in that order; one that arrives
  // before its predecessors (a rank that has fed fewer batches sends an older local batch in the same round) waits here.
  // Messages without an index (g1s_shard_msg_from_latest) are merged as they come: rounds in order, ranks in order.
  for (uint32_t r = 0; r < world; ++r) {  // (validate the whole round before merging any of it)
    const uint8_t *m = (const uint8_t *)msgs + (size_t)r * stride_bytes;
    ShardHeader h;
    std::memcpy(&h, m, sizeof(h));
    if (h.magic != kShardMagic || h.lag != f->lag || h.count > h.batch_frames ||
        stride_bytes < g1s_shard_msg_size(h.lag, h.batch_frames)) {
      f->err = "bad shard message from rank " + std::to_string(r);
      return G1S_ERR_INVALID;
    }
  }
  const size_t bs = latest_blob_size(f->lag);
  // The batches of this round that are next in the global order -- straight from the messages, or from `early` once their
  // predecessors have come -- are collected as runs and merged in one go (fold_push_runs: windows across messages).
  std::vector<BlobRun> runs;
  uint64_t next = f->next_batch;  // the global batch the next run must be
  size_t from_early = 0;          // how many of `early`'s first entries are in `runs`
  auto drain_early = [&] {
    auto it = f->early.begin();
    std::advance(it, from_early);
    while (it != f->early.end() && it->first == next) {
      runs.push_back(BlobRun{it->second.data(), bs, it->second.size() / bs});
      ++it;
      ++from_early;
      ++next;
    }
  };
  auto flush = [&]() -> int {
    const int rc = runs.empty() ? G1S_OK : fold_push_runs(f, runs.data(), runs.size());
    runs.clear();
    f->next_batch = next;
    for (; from_early; --from_early) f->early.erase(f->early.begin());
    return rc;
  };
  for (uint32_t r = 0; r < world; ++r) {
    const uint8_t *m = (const uint8_t *)msgs + (size_t)r * stride_bytes;
    ShardHeader h;
    std::memcpy(&h, m, sizeof(h));
    if (!h.count) continue;
    if (h.local_batch == kShardNoIndex) {
      if (const int rc = flush()) return rc;
      if (!f->early.empty()) {
        f->err = "frame-shard merge: a message without a batch index while indexed batches are waiting";
        return G1S_ERR_STATE;
      }
      const int rc = g1s_fold_push_latest(f, m + sizeof(h), bs, h.count);
      if (rc) return rc;
      continue;
    }
    const uint64_t j = (uint64_t)h.local_batch * world + r;
    if (j < next || f->early.count(j)) {
      flush();
      f->err = "frame-shard merge: batch " + std::to_string(j) + " arrived twice (rank " + std::to_string(r) + ")";
      return G1S_ERR_STATE;
    }
    if (j == next) {
      runs.push_back(BlobRun{m + sizeof(h), bs, h.count});
      ++next;
    } else {
      f->early.emplace(j, std::vector<uint8_t>(m + sizeof(h), m + sizeof(h) + (size_t)h.count * bs));
    }
    drain_early();
  }
  return flush();
}
void g1s_fold_free(g1s_fold_t *f) { delete f; }
const char *g1s_fold_last_error(const g1s_fold_t *f) { return f ? f->err.c_str() : ""; }
uint64_t g1s_fold_frames(const g1s_fold_t *f) { return f ? f->fold.frames() : 0; }

long g1s_format_tbl(const g1s_segment_t *segs, size_t n, char *buf, size_t cap) {
  return format_tbl(segs, n, buf, cap);
}
int g1s_parse_tbl(const char *text, size_t len, g1s_segment_t *out, size_t cap, size_t *n_out, char *err, size_t errcap) {
  if (!text && len) return G1S_ERR_INVALID;
  std::vector<g1s_segment_t> segs;
  std::string msg;
  const int rc = parse_tbl(text, len, segs, msg);
  if (rc) {
    if (err && errcap) snprintf(err, errcap, "%s", msg.c_str());
    return rc;
  }
  if (n_out) *n_out = segs.size();
  if (segs.size() > cap) return G1S_ERR_CAPACITY;
  if (!segs.empty()) std::memcpy(out, segs.data(), sizeof(g1s_segment_t) * segs.size());
  return G1S_OK;
}
long g1s_tbl_segment_for(g1s_segment_t *segs, size_t n, uint64_t packet_ts) {
  if (!segs) return -1;
  for (size_t i = 0; i < n; ++i) {
    if (segs[i].start_time <= packet_ts && packet_ts < segs[i].end_time) {
      segs[i].random_seed = (uint16_t)(segs[i].random_seed + 10956u);  // DEFAULT_GRAIN_SEED, wrapping
      return (long)i;
    }
  }
  return -1;
}
int g1s_write_tbl(const char *path, const g1s_segment_t *segs, size_t n) {
  std::vector<char> buf(1024 + 2048 * n);
  const long k = format_tbl(segs, n, buf.data(), buf.size());
  if (k < 0) return (int)k;
  FILE *f = fopen(path, "wb");
  if (!f) return G1S_ERR_INVALID;
  const size_t w = fwrite(buf.data(), 1, (size_t)k, f);
  const int c = fclose(f);
  return (w == (size_t)k && c == 0) ? G1S_OK : G1S_ERR_INVALID;
}

int g1s_diff_get_stats(const g1s_diff_t *g, g1s_stats_t *out) {
  if (!g || !out) return G1S_ERR_INVALID;
  *out = g->stats;
  out->ms_host_fold = g->ms_fold_front + g->ms_fold_back;  // (the two stages overlap across batches)
  return G1S_OK;
}
uint64_t g1s_diff_frames_copied(g1s_diff_t *g, uint64_t wait_for) {
  if (!g) return 0;
  (void)hipSetDevice(g->device);
  return g->frames_copied(wait_for);
}
uint64_t g1s_diff_frames_released(g1s_diff_t *g) {
  if (!g) return 0;
  std::lock_guard<std::mutex> lk(g->dm);
  return g->frames_released;
}
long g1s_diff_kernel_times(g1s_diff_t *g, char *buf, size_t cap) {
  if (!g || (!buf && cap)) return G1S_ERR_INVALID;
  std::string out;
  {
    std::lock_guard<std::mutex> lk(g->ktimes_mutex);
    for (const auto &kv : g->ktimes) {
      char line[256];
      snprintf(line, sizeof(line), "%s\t%.6f\t%llu\n", kv.first.c_str(), kv.second.first, (unsigned long long)kv.second.second);
      out += line;
    }
  }
  if (out.size() > cap) return G1S_ERR_CAPACITY;
  std::memcpy(buf, out.data(), out.size());
  return (long)out.size();
}
int g1s_diff_set_flat_finder(g1s_diff_t *g, int mode) {
  if (!g) return G1S_ERR_INVALID;
  if (mode < 0 || mode > 2) return g->fail(G1S_ERR_INVALID, "flat finder mode must be 0, 1 or 2");
  g->flat_literal = mode;
  return G1S_OK;
}
int g1s_diff_set_timing(g1s_diff_t *g, int enable) {
  if (!g) return G1S_ERR_INVALID;
  g->timing = enable != 0;
  g->timing_chain = enable == 2;
  return G1S_OK;
}

int g1s_diff_last_record(const g1s_diff_t *g, void *buf, size_t cap_bytes) {
  if (!g || !buf) return G1S_ERR_INVALID;
  if (g->last_record.empty()) return G1S_ERR_STATE;
  if (g->last_record.size() > cap_bytes) return G1S_ERR_CAPACITY;
  std::memcpy(buf, g->last_record.data(), g->last_record.size());
  return G1S_OK;
}

static bool rec_layout(const void *rec, RecHeader &h, RecLayout &L) {
  if (!rec) return false;
  std::memcpy(&h, rec, sizeof(h));
  if (h.magic != kRecMagic) return false;
  L = make_layout(h.width, h.height, h.nplanes, h.lag);
  return L.size == h.size_bytes;
}
int g1s_record_geometry(const void *rec, uint32_t *nbw, uint32_t *nbh, uint32_t *nplanes, uint32_t *lag) {
  RecHeader h;
  RecLayout L;
  if (!rec_layout(rec, h, L)) return G1S_ERR_INVALID;
  if (nbw) *nbw = h.nbw;
  if (nbh) *nbh = h.nbh;
  if (nplanes) *nplanes = h.nplanes;
  if (lag) *lag = h.lag;
  return G1S_OK;
}
const uint8_t *g1s_record_flat_mask(const void *rec) {
  RecHeader h;
  RecLayout L;
  if (!rec_layout(rec, h, L)) return nullptr;
  return (const uint8_t *)rec + L.off_mask;
}
const float *g1s_record_scores(const void *rec) {
  RecHeader h;
  RecLayout L;
  if (!rec_layout(rec, h, L)) return nullptr;
  return reinterpret_cast<const float *>((const uint8_t *)rec + L.off_scores);
}
int g1s_record_ar_sums(const void *rec, uint32_t c, const int64_t **S, const int64_t **Sb, int64_t *nobs) {
  RecHeader h;
  RecLayout L;
  if (!rec_layout(rec, h, L) || c >= h.nplanes) return G1S_ERR_INVALID;
  const int nc = (int)h.n + (c > 0);
  const int64_t *p = reinterpret_cast<const int64_t *>((const uint8_t *)rec + L.off_ar[c]);
  if (S) *S = p;
  if (Sb) *Sb = p + (size_t)nc * nc;
  if (nobs) *nobs = p[(size_t)nc * nc + nc];
  return nc;
}
int g1s_record_block_stats(const void *rec, uint32_t c, const uint32_t **luma_sum, const int32_t **sum_d,
                           const uint32_t **sum_d2) {
  RecHeader h;
  RecLayout L;
  if (!rec_layout(rec, h, L) || c >= h.nplanes) return G1S_ERR_INVALID;
  const uint8_t *r = (const uint8_t *)rec;
  if (luma_sum) *luma_sum = reinterpret_cast<const uint32_t *>(r + L.off_luma_sum);
  if (sum_d) *sum_d = reinterpret_cast<const int32_t *>(r + L.off_sum_d[c]);
  if (sum_d2) *sum_d2 = reinterpret_cast<const uint32_t *>(r + L.off_sum_d2[c]);
  return (int)L.nblocks;
}

}  // extern "C"

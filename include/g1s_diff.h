/*
 * g1s_diff.h -- C-ABI of the MI355X-native `diff` film-grain estimator.
 *
 * Drop-in boundary: the three-method object API of av1_grain::DiffGenerator as
 * grav1synth uses it (paths relative to the reference tree):
 *     DiffGenerator::new(fps, source_bd, denoised_bd)   src/main.rs:420-427
 *     differ.diff_frame(&source_frame, &denoised_frame) src/main.rs:442,462,482,502
 *     differ.finish() -> Vec<GrainTableSegment>         src/main.rs:524
 * and the `.tbl` writer fed from it (src/main.rs:525-529, 631-696).  A Rust
 * `grav1synth` binds these symbols with an `extern "C"` block.
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * All pixel work runs as HIP kernels for gfx950; the library fails loudly
 * (G1S_ERR_NO_DEVICE) when no HIP device is usable -- there is no CPU fallback.
 */
#ifndef G1S_DIFF_H
#define G1S_DIFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G1S_ABI_VERSION 1

/* av1_grain::{NUM_Y_POINTS, NUM_UV_POINTS, NUM_Y_COEFFS, NUM_UV_COEFFS}
 * (imported at src/parser/grain.rs:2; capacities at :27-31 and :46-50). */
#define G1S_NUM_Y_POINTS 14
#define G1S_NUM_UV_POINTS 10
#define G1S_NUM_Y_COEFFS 24
#define G1S_NUM_UV_COEFFS 25

enum {
  G1S_OK = 0,
  G1S_ERR_INVALID = -1,       /* bad argument / unsupported format */
  G1S_ERR_DIM_MISMATCH = -2,  /* source and denoised frame geometry differ */
  G1S_ERR_NOT_ENOUGH_FLAT = -3, /* "Not enough flat blocks to update noise estimate" */
  G1S_ERR_SOLVE = -4,         /* luma AR / strength equation system is singular */
  G1S_ERR_NO_DEVICE = -5,     /* no usable HIP device or kernel image */
  G1S_ERR_HIP = -6,           /* a HIP runtime call failed (see last_error) */
  G1S_ERR_STATE = -7,         /* call not legal in this state (e.g. after finish) */
  G1S_ERR_CAPACITY = -8,      /* output buffer too small */
  G1S_ERR_UNSUPPORTED = -9    /* valid request this path does not serve (a resize of deep samples without their bit depth) */
};

/* One decoded frame == v_frame::Frame<T> as produced by
 * BitstreamReader::decode_frame (src/reader.rs:172-212): planar Y,U,V; u8 for
 * 8-bit, native-endian u16 for 9..16-bit (src/reader.rs:51-67); chroma planes
 * decimated by (xdec, ydec) (src/reader.rs:69-85). */
typedef struct {
  uint32_t width, height;   /* luma plane size in samples */
  uint8_t bytes_per_sample; /* 1 or 2 */
  uint8_t xdec, ydec;       /* chroma subsampling log2: 4:2:0=(1,1) 4:2:2=(1,0) 4:4:4=(0,0) */
  uint8_t nplanes;          /* 1 (monochrome) or 3 */
  const void *data[3];
  size_t stride_bytes[3];
  int32_t on_device;        /* 0: host memory, copied before the call returns (the
                               `&Frame` borrow of src/main.rs:442).
                               1: device (HIP) memory of this process; must stay valid and
                               unmodified until g1s_diff_frames_released() covers the frame
                               (g1s_diff_sync()/finish at the latest).
                               2: PINNED host memory (hipHostMalloc / hipHostRegister): the copy
                               is queued and the call returns at once; must stay valid and
                               unmodified until g1s_diff_frames_copied() covers the frame */
} g1s_frame_t;

/* POD mirror of av1_grain::GrainTableSegment, field for field as consumed by
 * `impl From<av1_grain::GrainTableSegment>` at src/parser/grain.rs:108-133 and
 * src/main.rs:705-713. */
typedef struct {
  uint64_t start_time, end_time;
  uint16_t random_seed;
  uint8_t num_y_points, num_cb_points, num_cr_points;
  uint8_t scaling_points_y[G1S_NUM_Y_POINTS][2];
  uint8_t scaling_points_cb[G1S_NUM_UV_POINTS][2];
  uint8_t scaling_points_cr[G1S_NUM_UV_POINTS][2];
  uint8_t scaling_shift;
  uint8_t ar_coeff_lag;
  uint8_t num_y_coeffs, num_uv_coeffs; /* 2*lag*(lag+1) and that + 1 (src/parser/grain.rs:40-44) */
  int8_t ar_coeffs_y[G1S_NUM_Y_COEFFS];
  int8_t ar_coeffs_cb[G1S_NUM_UV_COEFFS];
  int8_t ar_coeffs_cr[G1S_NUM_UV_COEFFS];
  uint8_t ar_coeff_shift;
  uint8_t cb_mult, cb_luma_mult;
  uint16_t cb_offset;
  uint8_t cr_mult, cr_luma_mult;
  uint16_t cr_offset;
  uint8_t chroma_scaling_from_luma;
  uint8_t grain_scale_shift;
  uint8_t overlap_flag;
} g1s_segment_t;

/* Options; NULL == reference behaviour (lag 3, chroma estimated when present). */
typedef struct {
  uint32_t struct_size;   /* sizeof(g1s_opts_t), for forward compatibility */
  int32_t device;         /* HIP device ordinal; -1 = current device */
  uint32_t ar_coeff_lag;  /* 1..3; 0 = default (3, the reference's NOISE_MODEL_LAG) */
  uint32_t luma_only;     /* 1 = skip chroma planes (extension; reference: 0) */
  uint32_t batch_frames;  /* frames per kernel batch (<= 256); 0 = default: about 530 Mpixels' worth
                             (64 at 4K, 128 at 1080p) */
  uint32_t records_only;  /* frame-shard mode: do not fold here (the ordered fold runs
                             after the exchange, see g1s_fold_*).
                             0 = fold locally (single GPU);
                             1 = emit the per-frame records;
                             2 = run the per-frame half of the fold here as well and emit
                                 the per-frame "latest" states (~27 KB instead of ~250 KB a
                                 4K frame): only the ordered merge is left for rank 0. */
} g1s_opts_t;

typedef struct g1s_diff g1s_diff_t;

/* DiffGenerator::new (src/main.rs:420-427).  Returns NULL on failure; the
 * reason is then available from g1s_last_global_error(). */
g1s_diff_t *g1s_diff_new(int64_t fps_num, int64_t fps_den, uint32_t source_bit_depth,
                         uint32_t denoised_bit_depth, const g1s_opts_t *opts);
const char *g1s_last_global_error(void);

/* DiffGenerator::diff_frame (src/main.rs:442).  Frames are consumed in call
 * order.  Work is queued and may still be running when the call returns;
 * errors of queued frames (not enough flat blocks, singular system) surface on
 * a later diff_frame / sync / finish call, exactly once. */
int g1s_diff_frame(g1s_diff_t *, const g1s_frame_t *source, const g1s_frame_t *denoised);
/* n frame pairs in one call (same semantics as n diff_frame calls). */
int g1s_diff_frames(g1s_diff_t *, const g1s_frame_t *source, const g1s_frame_t *denoised, size_t n);
/* Drain all queued work (kernels + ordered fold). */
int g1s_diff_sync(g1s_diff_t *);
/* How many frame pairs, counted in the order they were handed over, the generator is done reading: the planes of
 * on_device frames (and of host frames, which are copied at the call anyway) before that count may be freed or
 * overwritten.  Monotonic; reaches the number of frames handed over after g1s_diff_sync / g1s_diff_finish.  A caller
 * that streams device-resident frames polls this instead of keeping every frame alive until the end. */
uint64_t g1s_diff_frames_released(g1s_diff_t *);
/* Same count for the HOST planes of frames handed over with on_device == 2 (pinned, copied asynchronously): frame
 * pairs before the returned count have been copied to the device.  wait_for > 0: block until at least that many have
 * (clamped to the frames handed over).  Frames with on_device 0 or 1 count as copied when their call returns. */
uint64_t g1s_diff_frames_copied(g1s_diff_t *, uint64_t wait_for);
/* DiffGenerator::finish (src/main.rs:524).  Ends the stream: afterwards no more frames are accepted.  *n_out = the
 * number of segments; if cap is too small the call returns G1S_ERR_CAPACITY with *n_out set and NOTHING is lost --
 * call again with a buffer of *n_out segments (the reference's Vec has no cap). */
int g1s_diff_finish(g1s_diff_t *, g1s_segment_t *out, size_t cap, size_t *n_out);
void g1s_diff_free(g1s_diff_t *);
/* anyhow::Error text of the last failure ("" if none). */
const char *g1s_diff_last_error(const g1s_diff_t *);

/* ---- frame-shard mode (multi-GPU): records out, ordered fold after exchange ---- */
/* Size in bytes of one per-frame record for this geometry (all exact integers:
 * AR normal-equation sums, per-block noise statistics, flat mask). */
size_t g1s_record_size(uint32_t width, uint32_t height, uint32_t xdec, uint32_t ydec,
                       uint32_t nplanes, uint32_t lag);
/* Zero a record buffer of g1s_record_size() bytes and write its header. */
int g1s_record_init(void *rec, size_t cap_bytes, uint32_t width, uint32_t height, uint32_t xdec,
                    uint32_t ydec, uint32_t nplanes, uint32_t lag);
/* Copies the records of all frames queued so far (frame order) into buf and
 * clears the internal list.  records_only generators only. */
int g1s_diff_take_records(g1s_diff_t *, void *buf, size_t cap_bytes, size_t *n_frames);
/* records_only == 2: latest states (g1s_latest_size() bytes each, frame order), whole batches:
 * sync = 0: exactly the batches queued before the two most recent ones that were not handed out yet
 * (waits for them if need be; deterministic, so that ranks deliver the same batches in the same
 * round); sync = 1: everything queued so far. */
int g1s_diff_take_latest(g1s_diff_t *, int sync, void *buf, size_t cap_bytes, size_t *n_frames);

/* ---- frame-shard rounds (one process per GPU): the exchange protocol behind the ABI, the transport with the host ----
 * The video is dealt to the N ranks batch by batch (batch j of batch_frames frame pairs -> rank j % N).  The job runs in
 * ROUNDS; in a round every rank (1) feeds its next batch, if the video still has one for it, (2) calls g1s_shard_pack,
 * (3) takes part in ONE gather of the fixed-size messages to rank 0 -- ncclAllGather / ncclSend+Recv on RCCL, MPI_Gather,
 * torch.distributed.gather: whatever the host application owns; this library links no communication library --
 * and rank 0 hands the N gathered messages, rank order, to g1s_shard_merge.  After the last feeding round every rank runs
 * g1s_shard_flush_rounds() more rounds with flush = 1 (the batches still in the generator's pipeline).  Rank 0's fold then finishes the table:
 * identical, byte for byte, to one generator fed the whole video (the states are exact; only their merge is ordered).
 * Message = 24-byte header + batch_frames latest states (g1s_latest_size() each, ~27 KB): N x 0.9 MB a round at 4K.
 * The header says WHICH of the sending rank's batches the message carries; the root merges by that index (global batch =
 * local batch * N + rank), so ranks that have fed different numbers of batches -- the idle rank of a short last round is
 * one feed behind -- may send different local batches in the same round: a batch that arrives before its predecessors
 * waits inside the fold, and g1s_fold_finish refuses (G1S_ERR_STATE) while one is still missing. */
size_t g1s_shard_msg_size(uint32_t ar_coeff_lag, uint32_t batch_frames);
/* This rank's message of the round: the latest states of ONE batch -- the oldest one not sent yet among those the generator
 * has finished (flush = 0: never waits; the message is empty when every unsent batch is still in the pipeline, which holds
 * at most g1s_shard_flush_rounds()) or among all (flush = 1: drains the generator first) -- or an empty message.  records_only = 2 generators. */
int g1s_shard_pack(g1s_diff_t *, int flush, void *msg, size_t cap_bytes);
/* How many batches a generator can still hold unsent when its last frame has been fed (its slots): the flush rounds every rank
 * runs behind its last feeding round. */
unsigned g1s_shard_flush_rounds(void);
/* A message from latest states made elsewhere (g1s_latest_from_record): n <= batch_frames.  Without a batch index: the
 * root merges such messages as they come (rounds in order, ranks in order) -- the caller keeps its ranks in lock step. */
int g1s_shard_msg_from_latest(const void *blobs, size_t n, uint32_t ar_coeff_lag, uint32_t batch_frames, void *msg, size_t cap_bytes);
/* The same with the index of the batch among the sending rank's batches (0, 1, ...; G1S_SHARD_NO_INDEX: none). */
#define G1S_SHARD_NO_INDEX UINT64_MAX
int g1s_shard_msg_from_latest_at(const void *blobs, size_t n, uint32_t ar_coeff_lag, uint32_t batch_frames, uint64_t local_batch,
                                 void *msg, size_t cap_bytes);
size_t g1s_latest_size(uint32_t ar_coeff_lag);
/* The per-frame half of the fold on the host: record -> latest state.  Thread-safe.  A frame
 * that fails (not enough flat blocks, singular system) yields a blob that carries the error;
 * it surfaces when the blob is pushed. */
int g1s_latest_from_record(const void *record, size_t size_bytes, uint32_t ar_coeff_lag, void *blob,
                           size_t cap_bytes);
/* The same for n records a stride apart, on the process' per-frame pool (G1S_FOLD_THREADS threads, the caller included). */
int g1s_latest_from_records(const void *records, size_t stride_bytes, size_t n, uint32_t ar_coeff_lag, void *blobs,
                            size_t blob_stride_bytes);
/* The cores this process may use: hardware threads cut to the cgroup CPU quota (the default size of the per-frame pool;
 * a launcher of several local ranks divides it among them: bench.py). */
unsigned g1s_usable_cpus(void);

typedef struct g1s_fold g1s_fold_t;
/* The sequential part of DiffGenerator (noise-model update, segmentation,
 * quantisation) over records, in frame order.  Pure host code. */
g1s_fold_t *g1s_fold_new(int64_t fps_num, int64_t fps_den, uint32_t lag);
int g1s_fold_push(g1s_fold_t *, const void *record, size_t size_bytes);
/* n records, stride_bytes apart, in frame order: the per-frame half runs on a
 * host thread pool, the ordered half serially.  Same result as n pushes. */
int g1s_fold_push_many(g1s_fold_t *, const void *records, size_t stride_bytes, size_t n);
/* n latest states, stride_bytes apart, in frame order: the ordered half only. */
int g1s_fold_push_latest(g1s_fold_t *, const void *blobs, size_t stride_bytes, size_t n);
/* Rank 0: the `world` gathered messages of one round, stride_bytes apart, rank order -> the ordered merge (by the batch
 * index the messages carry; messages without one: rounds must be merged in order). */
int g1s_shard_merge(g1s_fold_t *, const void *msgs, size_t stride_bytes, uint32_t world);
/* Same contract as g1s_diff_finish: G1S_ERR_CAPACITY leaves the segments in place for a second call. */
int g1s_fold_finish(g1s_fold_t *, g1s_segment_t *out, size_t cap, size_t *n_out);
void g1s_fold_free(g1s_fold_t *);
const char *g1s_fold_last_error(const g1s_fold_t *);
/* Frames folded so far (pushed or merged in order): a driver of the round protocol checks it against the frames it fed
 * before g1s_fold_finish -- a batch still inside a generator's pipeline is a frame missing here. */
uint64_t g1s_fold_frames(const g1s_fold_t *);

/* ---- `.tbl` text, byte for byte what src/main.rs:525-529,631-696 writes ---- */
/* Returns the number of bytes written (no NUL), or G1S_ERR_CAPACITY. */
long g1s_format_tbl(const g1s_segment_t *segs, size_t n, char *buf, size_t cap);
int g1s_write_tbl(const char *path, const g1s_segment_t *segs, size_t n);
/* The reader of the same text: av1_grain::parse_grain_table as `apply` calls it (src/main.rs:228-241).
 * *n_out = number of segments (also when cap is too small: G1S_ERR_CAPACITY); reason of a failure in err. */
int g1s_parse_tbl(const char *text, size_t len, g1s_segment_t *out, size_t cap, size_t *n_out, char *err, size_t errcap);
/* The segment `apply` stamps on a frame with presentation time packet_ts (src/parser/frame.rs:617-633): the
 * first one with start_time <= ts < end_time, or -1.  Like the reference, every hit advances that segment's
 * random_seed by DEFAULT_GRAIN_SEED (wrapping) before it is used. */
long g1s_tbl_segment_for(g1s_segment_t *segs, size_t n, uint64_t packet_ts);

/* ---- measurement hooks (bench.py) ---- */
typedef struct {
  uint64_t frames;            /* frame pairs processed by the kernels */
  uint64_t blocks, flat_blocks;
  /* HIP-event time per kernel family, summed over launches, milliseconds */
  double ms_flat_features, ms_flat_select, ms_ar_accumulate, ms_total_gpu;
  uint64_t launches_flat_features, launches_flat_select, launches_ar_accumulate;
  double ms_host_fold;        /* wall time spent in the ordered host fold */
  double ms_residual;         /* part of ms_ar_accumulate: the K0 residual pass over the input planes */
  uint64_t literal_blocks;    /* timed batches only: blocks the certified flat-block fast path left to the literal f64 kernel */
  double ms_chain;            /* set_timing(2): HIP-event time from the first kernel's start to the last kernel's end, summed over batches */
  uint64_t chain_batches;     /* ... the batches in that sum */
} g1s_stats_t;
int g1s_diff_get_stats(const g1s_diff_t *, g1s_stats_t *out);
/* Enable HIP-event timing of the batches (off by default: events serialise batches; a timed batch runs on one stream, alone on
 * the chip).  1: an event in front of every kernel (g1s_diff_kernel_times, the ms_* family sums); 2: ONE pair of events around the
 * batch's whole chain of kernels (ms_chain / chain_batches) -- no barrier packet between two of its launches. */
int g1s_diff_set_timing(g1s_diff_t *, int enable);
/* Timed batches: one line per kernel, "name\tmilliseconds\tlaunches\n" (HIP events around each launch, on the
 * stream the kernel runs on; the names are the ones rocprofv3 --kernel-trace prints).  Returns the number of bytes
 * written (no NUL) or G1S_ERR_CAPACITY. */
long g1s_diff_kernel_times(g1s_diff_t *, char *buf, size_t cap);
/* Flat-block finder: 0 (default) = integer moments + certified evaluation, literal f64 evaluation (one
 * wave per block) only for the blocks the certificate leaves open; 1 = literal evaluation of every
 * block, one lane per block; 2 = of every block, one wave per block (all three must agree bit for
 * bit: tests/test_gpu_parity.py).  Takes effect from the next batch. */
int g1s_diff_set_flat_finder(g1s_diff_t *, int mode);

/* ---- introspection of the most recently *completed* frame (parity tests) ---- */
/* Copies the frame's record (layout: g1s_record_* accessors below). */
int g1s_diff_last_record(const g1s_diff_t *, void *buf, size_t cap_bytes);
/* Record field accessors (so that bindings need not know the layout). */
int g1s_record_geometry(const void *rec, uint32_t *nbw, uint32_t *nbh, uint32_t *nplanes, uint32_t *lag);
const uint8_t *g1s_record_flat_mask(const void *rec);
const float *g1s_record_scores(const void *rec);
/* n x n sums S[i*n+j] and Sb[i] of plane c (chroma regressor n-1 pre-scaled by
 * ns = (1<<xdec)*(1<<ydec)); returns n. */
int g1s_record_ar_sums(const void *rec, uint32_t c, const int64_t **S, const int64_t **Sb, int64_t *nobs);
int g1s_record_block_stats(const void *rec, uint32_t c, const uint32_t **luma_sum,
                           const int32_t **sum_d, const uint32_t **sum_d2);

/* ---- the caller of the path: `grav1synth diff`'s frame-pair loop and a raw-video frame source ---- */
/* A frame source == BitstreamReader::get_frame (src/reader.rs:122-170) as the diff loop uses it:
 * returns 1 and fills *out (host or device planes, valid until the next call on this source),
 * 0 at end of stream (`None`), a negative G1S_ERR_* on failure (`Err`, `?`-propagated). */
typedef int (*g1s_next_frame_fn)(void *user, g1s_frame_t *out);
/* The loop of Commands::Diff (src/main.rs:432-521) with get_filtered_frame_pair (src/main.rs:615-629):
 * one frame from each source (source first), diff_frame on the pair, until a source ends.
 * Both ended together: normal end.  Only one ended: *unequal = 1 -- the reference warns "Videos did
 * not have equal frame counts. Resulting grain table may not be as expected." and stops there too.
 * The first error ends the loop and is returned.  g1s_diff_finish() is left to the caller
 * (src/main.rs:524).  *frames = pairs handed to diff_frame. */
int g1s_diff_run(g1s_diff_t *, g1s_next_frame_fn source, void *source_user, g1s_next_frame_fn denoised,
                 void *denoised_user, uint64_t *frames, int *unequal);

/* ---- `--filters` (N3): FilterChain of /root/reference/src/filters.rs ---- */
/* FilterChain::new (src/filters.rs:16-110): "name:arg=value,...;name:..." with the filters crop (top, bottom, left,
 * right) and resize (width, height, alg = hermite | catmullrom | mitchell | lanczos | spline36; default catmullrom).
 * Same grammar, same error texts ("Invalid filter syntax in \"..\"", "Unrecognized filter \"..\"", "Unrecognized crop
 * arg \"..\"", "invalid digit found in string", "Both width and height must be provided to resize filter", ...).
 * NULL + reason in err on a parse error; "" is the empty chain. */
typedef struct g1s_filters g1s_filters_t;
typedef struct {
  uint32_t kind;                     /* 0 = crop, 1 = resize */
  uint64_t top, bottom, left, right; /* crop */
  uint64_t width, height;            /* resize */
  char alg[16];                      /* resize */
} g1s_filter_desc_t;
g1s_filters_t *g1s_filters_new(const char *text, char *err, size_t errcap);
size_t g1s_filters_len(const g1s_filters_t *);
int g1s_filters_get(const g1s_filters_t *, size_t i, g1s_filter_desc_t *out);
/* FilterChain::apply (src/filters.rs:112-116) on a frame descriptor.  crop is extent arithmetic: *out points into
 * *in's planes (host or device), same strides, smaller width / height -- no sample is touched, so it costs nothing on
 * the device.  Crop amounts must be multiples of the chroma subsampling and leave at least one sample
 * (G1S_ERR_INVALID otherwise).  A chain with a resize filter is served by g1s_filters_apply_bd below (this call, without a
 * bit depth, takes 8-bit samples only and answers G1S_ERR_UNSUPPORTED for deeper ones).  filters == NULL: *out = *in. */
int g1s_filters_apply(const g1s_filters_t *, const g1s_frame_t *in, g1s_frame_t *out, char *err, size_t errcap);
/* FilterChain::apply(frame, source_bd) (src/filters.rs:112-116) with the resize filter served: crop as above; resize
 * (src/filters.rs:150-178: hermite / catmullrom / mitchell / lanczos / spline36) runs ON THE DEVICE `device` (-1: the
 * current one; host planes are staged there) and the resized frame comes back as device planes (on_device = 1) inside
 * buffer `slot` of a ring the chain owns -- valid until the chain is applied with the same slot again, or freed.  No CPU
 * fallback.  bit_depth = the source bit depth (8..16).  g1s_filters_apply = this with bit_depth 8 for 8-bit samples and a
 * refusal (G1S_ERR_UNSUPPORTED) for deeper ones when the chain resizes. */
int g1s_filters_apply_bd(const g1s_filters_t *, const g1s_frame_t *in, uint32_t bit_depth, int32_t device, uint32_t slot,
                         g1s_frame_t *out, char *err, size_t errcap);
int g1s_filters_has_resize(const g1s_filters_t *);
/* The taps of one axis of a resize (test / documentation aid): output i = sum over k < *taps of coef[i * taps + k] *
 * in[idx[i * taps + k]], k ascending, in f32 without fused multiply-adds; cap = entries idx / coef hold (dst * taps). */
int g1s_resize_plan(const char *alg, uint32_t src, uint32_t dst, uint32_t *taps, int32_t *idx, float *coef, size_t cap);
/* One frame (host or device planes) through the device resize, result into host planes (tests, the Python FilterChain). */
int g1s_resize_frame_to_host(const char *alg, const g1s_frame_t *in, uint32_t bit_depth, uint32_t out_w, uint32_t out_h,
                             int32_t device, void *const out_planes[3], const size_t out_stride_bytes[3], char *err, size_t errcap);
void g1s_filters_free(g1s_filters_t *);
/* g1s_diff_run with get_filtered_frame_pair's filter step (src/main.rs:615-629): the chain is applied to every SOURCE
 * frame before the pair is handed to diff_frame; the denoised frame is taken as it comes.  A frame index goes with
 * every error text.  filters == NULL: exactly g1s_diff_run. */
int g1s_diff_run_filtered(g1s_diff_t *, g1s_next_frame_fn source, void *source_user, g1s_next_frame_fn denoised,
                          void *denoised_user, const g1s_filters_t *filters, uint64_t *frames, int *unequal);

/* YUV4MPEG2 frame source: stands where the libav reader stands in the reference (what reaches the
 * estimator is the same planar Y,U,V u8 / little-endian u16 frame, src/reader.rs:172-212).  Frames
 * are read ahead by a thread (big frames: four positional reads at a time) into pinned host memory (a ring
 * of 4), so file IO, the H2D copies of
 * g1s_diff_frame and the kernels of earlier frames overlap.  Colour spaces: C420* / C422 / C444 /
 * Cmono with an optional p9..p16 depth suffix (8-, 10-, 12-bit 4:2:0 / 4:2:2 / 4:4:4 are what
 * src/reader.rs:51-85 accepts). */
typedef struct g1s_y4m g1s_y4m_t;
typedef struct {
  uint32_t width, height, bit_depth, xdec, ydec, nplanes;
  int64_t fps_num, fps_den;
} g1s_y4m_info_t;
g1s_y4m_t *g1s_y4m_open(const char *path, char *err, size_t errcap); /* NULL on failure, reason in err */
int g1s_y4m_get_info(const g1s_y4m_t *, g1s_y4m_info_t *out);
int g1s_y4m_next(void *y4m, g1s_frame_t *out); /* a g1s_next_frame_fn; user = the g1s_y4m_t* */
/* Bind the reader to the generator its frames go to (frame i of the reader = frame pair i of the generator): frames
 * then come out with on_device = 2 -- g1s_diff_frame queues their copies straight from the reader's pinned ring and
 * returns -- and the reader recycles a ring buffer only when g1s_diff_frames_copied() covers its frame.  Close the
 * reader after the generator has synced / finished.  NULL unbinds. */
int g1s_y4m_bind(g1s_y4m_t *, g1s_diff_t *);
const char *g1s_y4m_last_error(const g1s_y4m_t *);
void g1s_y4m_close(g1s_y4m_t *);
/* `grav1synth diff SOURCE DENOISED -o OUT` for two .y4m files (src/main.rs:414-531): frame rate from
 * the source, bit depths from each file, the loop above, finish, "filmgrn1" table to out_tbl. */
int g1s_diff_y4m_files(const char *source, const char *denoised, const char *out_tbl, const g1s_opts_t *opts,
                       uint64_t *frames, int *unequal, char *err, size_t errcap);
/* The same with `-f FILTERS` (src/main.rs:370-380): filters = NULL or "" for none.  A chain that does not parse:
 * G1S_ERR_INVALID and err = "Invalid filter chain: <reason>" (the reference logs that line and exits). */
int g1s_diff_y4m_files_filtered(const char *source, const char *denoised, const char *out_tbl, const g1s_opts_t *opts,
                                const char *filters, uint64_t *frames, int *unequal, char *err, size_t errcap);
/* The same command over several devices (north_star: frames shard across the GPUs of a node; the loop of src/main.rs:414-531):
 * ONE process, a generator per entry of `devices` (an ordinal may repeat), the two files' frame pairs dealt batch by batch
 * (batch j -> generator j % n_devices), the frame-shard rounds above with the host as the transport.  Same table, byte for
 * byte, as one generator; same return values and error texts as g1s_diff_y4m_files_filtered. */
int g1s_diff_y4m_files_sharded(const char *source, const char *denoised, const char *out_tbl, const g1s_opts_t *opts,
                               const char *filter_text, const int32_t *devices, uint32_t n_devices, uint64_t *frames_out,
                               int *unequal_out, char *err, size_t errcap);

/* ---- N4: `grav1synth estimate` (feature "unstable", src/main.rs:534-608): the single-source noise estimator ---- */
/* av1_grain::estimate_plane_noise(&frame.y_plane, bit_depth) per frame (the port of libaom's
 * av1_estimate_noise_from_single_plane: Sobel-gated mean |Laplacian| of the luma plane), on the device: one pass over
 * the luma plane, exact integer sums, the f64 formed on the host with the reference's three operations.
 * g1s_estimate_new: NULL without a HIP device (no CPU fallback) or for a bit depth outside 8..16. */
typedef struct g1s_estimate g1s_estimate_t;
g1s_estimate_t *g1s_estimate_new(uint32_t bit_depth, int32_t device, uint32_t batch_frames);
/* One frame (only data[0], the luma plane, is read; on_device 0 = host, copied before the call returns; 1 = device,
 * valid until the next g1s_estimate_finish or until batch_frames more frames have been handed over). */
int g1s_estimate_frame(g1s_estimate_t *, const g1s_frame_t *frame);
/* frame_estimates (src/main.rs:563-590): one f64 per frame, -1.0 where the reference has None (fewer than 16 smooth
 * pixels).  G1S_ERR_CAPACITY leaves them in place (*n_out = count); more frames may follow. */
int g1s_estimate_finish(g1s_estimate_t *, double *out, size_t cap, size_t *n_out);
/* HIP-event time of the kernel launches so far (enable = 1 from the next batch on). */
int g1s_estimate_set_timing(g1s_estimate_t *, int enable, double *ms_kernel, uint64_t *frames);
const char *g1s_estimate_last_error(const g1s_estimate_t *);
void g1s_estimate_free(g1s_estimate_t *);
/* The command's output (src/main.rs:596-603): "filmgrn1\n" then "{:.3}\n" per frame.  Bytes written or G1S_ERR_CAPACITY. */
long g1s_format_estimates(const double *estimates, size_t n, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* G1S_DIFF_H */

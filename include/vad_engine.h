/*
 * vad_engine.h — C ABI of the MI355X-native batched Silero-VAD engine.
 *
 * This is the drop-in boundary for ONE hot path of Picurit/cutter-vad: the per-frame model
 * operator and its pre/post steps.  Every entry point cites the reference interface it
 * replaces (paths relative to /root/reference/src/real_time_vad/).  The reference is pure
 * Python over onnxruntime; what it binds today is
 *
 *     ort.InferenceSession(path, sess_options, providers)          core/silero_model.py:321-325
 *     session.run(None, {'input','state','sr'} | {'input','h','c','sr'})   core/silero_model.py:433
 *
 * one 512-sample frame, one stream, batch 1.  The engine keeps that contract per stream and
 * adds the stream-batch axis: n independent streams advance one frame in one launch.
 *
 * Conventions
 *   - plain C, no C++/torch types; all sizes explicit; pointers are host pointers unless
 *     the parameter name starts with d_ (device pointer, same GPU as the engine).
 *   - return value: 0 = VAD_OK, negative = vad_status; the message for the last failure on
 *     an engine is vad_last_error(e); for a failed vad_engine_create it is
 *     vad_last_create_error() (thread-local).
 *   - there is NO CPU fallback: if no HIP device is usable, vad_engine_create fails with
 *     VAD_ERR_NO_DEVICE.
 *   - ownership: the engine owns per-slot recurrent state (h,c) in device HBM and a private
 *     copy of the weights; callers own every buffer they pass, for the duration of the call.
 *   - threading: calls on one engine are serialised by an internal mutex; a slot may appear
 *     at most once per step call (the same rule the reference's per-wrapper lock gives,
 *     core/vad_wrapper.py:560).
 */
#ifndef VAD_ENGINE_H
#define VAD_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VAD_API __attribute__((visibility("default")))
#else
#define VAD_API
#endif

#define VAD_ABI_VERSION 4
#define VAD_FRAME_SAMPLES 512   /* core/silero_model.py:464-468: frames are padded/truncated to 512 (Silero V5 8 kHz engines: 256, see vad_info) */
#define VAD_STATE_FLOATS 256    /* V5: state[2][1][128]; V4: h[2][1][64] then c[2][1][64]  (silero_model.py:391-401) */

typedef enum vad_status {
    VAD_OK = 0,
    VAD_ERR_INVALID_ARG = -1,   /* -> AudioProcessingError / ConfigurationError on the Python side */
    VAD_ERR_NO_DEVICE = -2,     /* no usable HIP device: the product has no CPU path */
    VAD_ERR_BAD_WEIGHTS = -3,   /* -> ModelInitializationError("Failed to load model ...") silero_model.py:330-334 */
    VAD_ERR_HIP = -4,           /* a HIP runtime call failed -> AudioProcessingError("Model prediction failed: ...") :444-447 */
    VAD_ERR_NO_SLOT = -5,       /* stream pool exhausted */
    VAD_ERR_BAD_SLOT = -6,      /* slot not open / out of range / duplicated within one step */
    VAD_ERR_UNSUPPORTED = -7,   /* e.g. an input rate the resampler has no operator for */
    VAD_ERR_BUSY = -8           /* vad_step_submit: every pipeline buffer holds an uncollected ticket */
} vad_status;

typedef enum vad_frame_format {
    VAD_FMT_F32 = 0,            /* float32 in [-1,1]: what VADWrapper.process_audio_data hands down (vad_wrapper.py:598) */
    VAD_FMT_I16_32767 = 1,      /* int16 PCM scaled by 1/32767 (websocket server convention, vad_websocket_server.py:341) */
    VAD_FMT_I16_32768 = 2       /* int16 PCM scaled by 1/32768 (AudioUtils.pcm_to_float32, utils/audio.py:308) */
} vad_frame_format;

/* event bits produced by the per-stream hysteresis state machine (core/silero_model.py:790-949) */
enum { VAD_EV_START = 1, VAD_EV_END = 2, VAD_EV_CONTINUE = 4 };

typedef struct vad_engine vad_engine;

/* Replaces SileroVADModel.__init__/_load_model (core/silero_model.py:276-334). */
typedef struct vad_engine_desc {
    uint32_t struct_size;       /* sizeof(vad_engine_desc) */
    int32_t model_version;      /* 4 | 5  (core/config.py:23-26) */
    const void *weights;        /* SVW blob: the tensors of the .onnx file's 16 kHz branch (tools/extract_weights.py) */
    size_t weights_len;
    int32_t device_id;          /* HIP device ordinal; one engine drives one GPU */
    int32_t max_streams;        /* capacity of the per-GPU stream pool (slots) */
    int32_t sample_rate;        /* the graph's `sr` input (core/silero_model.py:491): 16000, or - with the blob of the graph's 8 kHz
                                   sub-model - V4: 8000 / 24000 / 48000 (same 512-sample frames), V5: 8000 (native 8 kHz audio in
                                   256-sample frames: every [512] below reads [vad_info.frame_samples]) */
    uint32_t flags;             /* VAD_ENGINE_* bits */
} vad_engine_desc;
/* Another engine's kernels run on this GPU at the same time (e.g. a Silero V4 and a V5 pool side by side: BASELINE configs[4]).
 * A Silero V5 engine normally serves one-frame calls (and multi-frame calls of <= 4 096 streams) on 16-stream tiles, which spreads
 * them over up to all 256 CUs, two workgroups per CU above 4 096 streams (24.5 us per step for 1 024 streams, 44.5 for 8 192, instead
 * of 44 - 46 us on 32-stream tiles) - and leaves no CU to a co-tenant.  With this flag it keeps to 32-stream tiles: a call of n
 * streams occupies n / 32 CUs and the other engine's workgroups run beside it.  A Silero V4 engine (16-stream tiles) then always
 * places two workgroups on a CU instead of spreading a small call over one CU per tile: n / 32 CUs as well. */
#define VAD_ENGINE_SHARED_GPU 1u

typedef struct vad_info {
    uint32_t struct_size;
    int32_t abi_version;
    int32_t model_version;
    int32_t device_id;
    int32_t max_streams;
    int32_t open_streams;
    int32_t compute_units;
    int32_t streams_per_workgroup;
    int64_t weight_bytes_device;   /* packed weight streams resident in HBM */
    int64_t state_bytes_device;
    int64_t steps;                 /* launches so far (SileroVADModel.prediction_count analogue, silero_model.py:440) */
    int64_t frames;                /* frames processed so far */
    char device_name[64];
    char arch[32];                 /* "gfx950..." */
    int32_t frame_samples;         /* samples per model step: 512 (core/silero_model.py:464-468); 256 for Silero V5's 8 kHz sub-model */
    int32_t sample_rate;           /* the `sr` the engine was created for */
} vad_info;

/* thresholds of one stream's state machine: VADConfig fields core/config.py:54-94 */
typedef struct vad_thresholds {
    double start_probability;   /* vad_start_probability (Python float: compared as double, like the reference) */
    double end_probability;     /* vad_end_probability   */
    double start_ratio;         /* voice_start_ratio (dead logic in the reference, kept: SURVEY a10) */
    double end_ratio;           /* voice_end_ratio */
    int32_t start_frame_count;  /* voice_start_frame_count */
    int32_t end_frame_count;    /* voice_end_frame_count */
} vad_thresholds;

/* ---- lifetime ----------------------------------------------------------------------- */

/* SileroVADModel(model_path, model_version)  core/silero_model.py:276-301 */
VAD_API int vad_engine_create(const vad_engine_desc *desc, vad_engine **out);
/* session release (the reference lets the GC drop the ORT session; vad_wrapper.py:747-762 cleanup) */
VAD_API void vad_engine_destroy(vad_engine *e);
VAD_API const char *vad_last_error(const vad_engine *e);
VAD_API const char *vad_last_create_error(void);
/* SileroVADModel.get_model_info  core/silero_model.py:548-566 */
VAD_API int vad_engine_info(const vad_engine *e, vad_info *info);

/* ---- per-stream recurrent state (ModelState, core/silero_model.py:33-83) ------------- */

/* one VADWrapper/VADProcessor/SileroVADModel per client in the reference
 * (websocket_service/server/vad_websocket_server.py:277) == one slot here */
VAD_API int vad_stream_open(vad_engine *e, int64_t *slot);
/* n streams at once (one device synchronisation for the lot); slots_out [n] */
VAD_API int vad_stream_open_many(vad_engine *e, int64_t n, int64_t *slots_out);
VAD_API int vad_stream_close(vad_engine *e, int64_t slot);
/* SileroVADModel.reset / _reset_states  core/silero_model.py:384-401, 539-546 (also resets the slot's state machine) */
VAD_API int vad_stream_reset(vad_engine *e, const int64_t *slots, int64_t n);
/* ModelState.state / hidden_state+cell_state as ONNX lays them out: 256 floats */
VAD_API int vad_stream_get_state(vad_engine *e, int64_t slot, float *hc);
VAD_API int vad_stream_set_state(vad_engine *e, int64_t slot, const float *hc);
/* VADWrapper.set_thresholds  core/vad_wrapper.py:367-419: values only (validation lives in the host mirror);
 * the counters/history of the slot are NOT reset here - the wrapper calls vad_stream_reset next, as :412-413 does */
VAD_API int vad_stream_set_thresholds(vad_engine *e, int64_t slot, const vad_thresholds *t);
/* the same for n slots in ONE launch: t holds nt = 1 (shared by all) or nt = n (one per slot) entries.  vad_stream_reset,
 * vad_stream_open_many and this call cost one small copy + one kernel + one synchronisation whatever n is (a shared-pool
 * server resets / reconfigures thousands of sessions per tick: websocket_service/server/vad_websocket_server.py:277, 420-470) */
VAD_API int vad_stream_set_thresholds_many(vad_engine *e, const int64_t *slots, int64_t n, const vad_thresholds *t, int64_t nt);
/* Everything a stream is between two frames, as one opaque blob: (h, c) + the state machine's thresholds,
 * counters and history (VAD_STREAM_SAVE_BYTES).  No reference counterpart: the reference processes a chunk's frames
 * one by one and a callback that raises leaves the later frames unprocessed (core/vad_wrapper.py:638-647); the host
 * mirror runs a chunk's frames in ONE launch and uses save / restore to step back to that point.  Also what slot
 * migration between engines / GPUs moves (1 120 B per stream). */
#define VAD_STREAM_SAVE_BYTES 1120
VAD_API int vad_stream_save(vad_engine *e, int64_t slot, void *buf, int64_t cap);
VAD_API int vad_stream_restore(vad_engine *e, int64_t slot, const void *buf, int64_t nbytes);

/* ---- the hot path ------------------------------------------------------------------- */

/*
 * SileroVADModel.predict for n streams at once (core/silero_model.py:403-447):
 *   frames  [n][512] in `frame_fmt`; short frames must be right-zero-padded by the caller
 *           exactly as _prepare_audio_input does (:464-468);
 *   denoise_thresh >= 0 applies AudioUtils.denoise_audio's gate  x if |x| > thresh else 0
 *           (utils/audio.py:117-118; VADProcessor._preprocess_audio_frame core/silero_model.py:782-783);
 *           a negative value disables it (VADConfig.enable_denoising = False);
 *   probs_out [n]: float(outputs[0][0][0]) per stream (:515); state is advanced in place (:533-537).
 * Steps on one slot are applied in call order.
 */
VAD_API int vad_step(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int frame_fmt,
             float denoise_thresh, float *probs_out);

/*
 * Same, plus VADProcessor._process_voice_state (core/silero_model.py:790-949) on the device:
 * events_out[n] receives VAD_EV_* bits per stream for this frame; seg_frames_out[n] (may be NULL)
 * receives, on VAD_EV_END, the finished segment's length in frames (pre-roll included), else 0.
 */
VAD_API int vad_step_events(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int frame_fmt,
                    float denoise_thresh, float *probs_out, uint8_t *events_out, int32_t *seg_frames_out);

/*
 * T consecutive frames per stream in one call (VADWrapper._process_audio_frames' inner loop,
 * core/vad_wrapper.py:632-644): frames [n][T][512], probs_out [n][T], events_out [n][T] or NULL.
 */
VAD_API int vad_step_multi(vad_engine *e, const int64_t *slots, int64_t n, int32_t T, const void *frames, int frame_fmt,
                   float denoise_thresh, float *probs_out, uint8_t *events_out);

/*
 * Device-resident variant of vad_step_events for callers that already hold audio in HBM
 * (bench.py, GPU decode pipelines): every d_* pointer is device memory on the engine's GPU;
 * d_slots may be NULL (= slots 0..n-1); d_events / d_seg_frames may be NULL; `stream` is a
 * hipStream_t (NULL = the engine's own stream).  Asynchronous: returns after enqueueing.
 */
VAD_API int vad_step_device(vad_engine *e, const int32_t *d_slots, int64_t n, const void *d_frames, int frame_fmt,
                    float denoise_thresh, float *d_probs, uint8_t *d_events, int32_t *d_seg_frames, void *stream);
/*
 * The same with T consecutive frames per stream (vad_step_multi on device pointers): d_frames [n][T][frame], d_probs [n][T],
 * d_events [n][T] or NULL, d_seg_frames [n] or NULL (length of the LAST segment that ended inside the call, else 0).
 * Rules for both device entry points:
 *   - d_slots is trusted (it lives on the GPU, the host cannot validate it): every entry must be an OPEN slot of this engine
 *     and must appear AT MOST ONCE per call - a duplicate makes two workgroups read-modify-write the same (h, c) and state
 *     machine, and the result is undefined (the host-pointer entry points check this and return VAD_ERR_BAD_SLOT);
 *   - calls that touch the same slot must be ordered by the caller (same HIP stream, or events between streams);
 *   - one call may address at most 2 GiB of frames (n * T * frame bytes), else VAD_ERR_INVALID_ARG.
 */
VAD_API int vad_step_multi_device(vad_engine *e, const int32_t *d_slots, int64_t n, int32_t T, const void *d_frames,
                                  int frame_fmt, float denoise_thresh, float *d_probs, uint8_t *d_events,
                                  int32_t *d_seg_frames, void *stream);

/*
 * Pipelined host ingest.  vad_step* on host pointers are copy -> kernel -> copy -> wait; at 8 192 streams the PCIe copy is
 * 5-8 x the kernel, so a serving loop should overlap the copy of tick t+1 with the kernel of tick t:
 *
 *     vad_step_submit(e, slots, n, T, frames_t1, fmt, thr, &ticket1);     // enqueues H2D -> kernel -> D2H, returns at once
 *     vad_step_collect(e, ticket0, probs, events, seg);                   // blocks until tick t's results are on the host
 *
 * Up to 2 tickets may be outstanding (a third submit returns VAD_ERR_BUSY); each ticket is collected exactly once, in any
 * order.  `frames` must stay valid and unchanged until its ticket is collected; allocate it
 * with vad_host_alloc for a true asynchronous DMA (a pageable buffer works, but the runtime then stages it synchronously).
 * Kernels of successive tickets run in submission order on the engine's stream, so a slot may appear in consecutive
 * tickets.  Results are identical, bit for bit, to vad_step_multi on the same inputs.
 * The reference has no counterpart: it calls session.run synchronously per frame (core/silero_model.py:433, 471-499).
 */
VAD_API int vad_step_submit(vad_engine *e, const int64_t *slots, int64_t n, int32_t T, const void *frames, int frame_fmt,
                            float denoise_thresh, int64_t *ticket);
VAD_API int vad_step_collect(vad_engine *e, int64_t ticket, float *probs_out /*[n][T]*/, uint8_t *events_out /*[n][T] or NULL*/,
                             int32_t *seg_frames_out /*[n] or NULL*/);

/*
 * AudioUtils.resample_audio (utils/audio.py:19-55 -> scipy.signal.resample, Fourier method)
 * for n streams: in [n][n_in] float32 at sr_in -> out [n][512] float32 at 16 kHz, one
 * 512-sample output chunk per call (n_in = 512 * sr_in / 16000: 256 / 768 / 1536).
 */
VAD_API int vad_resample(vad_engine *e, const float *in, int64_t n, int32_t n_in, int32_t sr_in, float *out);
VAD_API int vad_resample_device(vad_engine *e, const float *d_in, int64_t n, int32_t n_in, int32_t sr_in, float *d_out,
                        void *stream);
/* the same for up to 4 segments of different input rates in ONE launch (a tick's 8 / 24 / 48 kHz clients): segment k is
 * d_in[k] [n[k]][n_in[k]] at sr_in[k] -> d_out[k] [n[k]][512]; the tables are host arrays, the buffers device pointers */
#define VAD_RESAMPLE_MAX_SEGMENTS 4
VAD_API int vad_resample_multi_device(vad_engine *e, int32_t nseg, const float *const *d_in, const int64_t *n,
                                      const int32_t *n_in, const int32_t *sr_in, float *const *d_out, void *stream);

/*
 * AudioUtils.resample_audio for EVERY input the reference function accepts (utils/audio.py:39-49): the whole array goes through
 * scipy.signal.resample(x, int(len(x) * target_rate / original_rate)) - any length, any pair of rates.  The caller computes
 * n_out exactly as the reference does (Python float arithmetic, utils/audio.py:43-46); only the two lengths enter the maths.
 * in [rows][n_in] float32 (in_f64 = 0) or float64 (in_f64 = 1: scipy transforms float64 / integer arrays in double precision)
 * -> out [rows][n_out] float32 (the reference's .astype(np.float32), :49); rows = independent arrays of one length (the
 * columns of an [N, C] array: scipy resamples along axis 0).  Two kernels behind it, chosen by size: below 2^25 operator
 * entries (rows * n_in * n_out) every entry is evaluated where it is used, in float64, never stored (csrc/resample_generic.hip:
 * lowest latency, 70 us for 100 -> 50, 80 us for three 48 kHz chunks); from there the same function runs as two chirp-z
 * transforms on power-of-two float64 FFTs (csrc/resample_fft.hip: O(n log n) for any pair of lengths, up to 2^25 samples = 11
 * minutes of 48 kHz audio; host buffers in and out: 0.14 ms for one second of 48 kHz audio, 0.33 ms for ten, 18.5 ms for ten
 * minutes of 44.1 kHz - 2 x, 9 x and 22 x scipy on the box's host).  Beyond both (n > 2^25 and rows * n_in * n_out > 2^42): VAD_ERR_UNSUPPORTED - never cut into pieces,
 * the result of a cut array is NOT the reference's.
 * The _device form takes device pointers and is synchronous as well (the result is complete on return).
 */
VAD_API int vad_resample_generic(vad_engine *e, const void *in, int in_f64, int64_t rows, int64_t n_in, int64_t n_out, float *out);
VAD_API int vad_resample_generic_device(vad_engine *e, const void *d_in, int in_f64, int64_t rows, int64_t n_in, int64_t n_out,
                                        float *d_out);

/*
 * Tick assembler: the serving loop's side of vad_step_events, in C.  The reference runs the model inside every websocket's
 * receive loop, one client and one frame at a time (websocket_service/server/vad_websocket_server.py:326-382); a shared-pool
 * server instead collects the frames that arrived since the last tick and advances all those streams together:
 *
 *   vad_tick_push(e, slot, samples, nsamples, fmt, gate_on)   from any thread, as frames arrive: the frame is written straight
 *       into the page-locked staging row of the coming tick, right-zero-padded / truncated to the engine's frame length exactly
 *       as _prepare_audio_input does (core/silero_model.py:464-468).  One frame per slot and tick: a slot's further frames
 *       queue up and are fed in submission order, one per later tick (at most 256 waiting: VAD_ERR_BUSY).
 *   vad_tick_run(e, thr, &res)   advances every slot that has a frame by ONE frame - one launch per (frame format, gate)
 *       group, i.e. one launch when all clients speak one format - and returns compact arrays: res.slots[i], res.probs[i],
 *       res.events[i] (VAD_EV_* bits), res.seg_frames[i]; entries group_start[g] .. group_start[g+1]-1 belong to group
 *       g = frame_fmt * 2 + gate_on, in push order, and group_frames[g] is that group's staged audio [count][frame] in frame_fmt
 *       (what segment assembly keeps).  All pointers are engine-owned and stay valid until the next vad_tick_run.
 *       Pushes may continue while a tick runs (double-buffered staging).  `thr` is the gate threshold of the gate_on groups.
 *       Frames that waited are placed first, in the order their slots started waiting, then the frames pushed since, in push order.
 *       A tick that FAILS (a HIP error) has consumed its frames: res.n / res.slots / res.nsamples then list the streams that lost
 *       one (res.probs is NULL), everything queued behind them is intact and the next tick carries on.
 *   vad_tick_cancel(e, slot)   drops the slot's pending frames and segment audio.  vad_stream_close and vad_stream_open do the same
 *       for their slot, so a recycled slot never sees its predecessor's frames.
 *   vad_tick_pending(e, slot, &frames)   frames of `slot` that have not been stepped yet (staged + waiting).
 *   vad_tick_push_rate(e, slot, samples, nsamples, fmt, gate_on, sr_in)   the same for a client whose audio arrives at 8 / 24 /
 *       48 kHz (VADConfig.auto_convert_sample_rate; nsamples must be the chunk that yields one 16 kHz frame: 256 / 768 / 1536):
 *       the chunk is staged as float32 in group 6 + 3 * gate_on + {0, 1, 2}; vad_tick_run resamples those groups on the GPU and
 *       steps them like vad_step_rates (one fused launch when it fits); group_frames[g] then holds the chunks at their own rate
 *       [count][nsamples] float32 - what the segment keeps.  16 kHz engines only.
 */
#define VAD_TICK_GROUPS 12
typedef struct vad_tick_result {
    uint32_t struct_size;          /* sizeof(vad_tick_result) */
    int64_t n;
    const int64_t *slots;
    const float *probs;
    const uint8_t *events;
    const int32_t *seg_frames;
    int64_t group_start[VAD_TICK_GROUPS + 1];
    const void *group_frames[VAD_TICK_GROUPS];
    const int32_t *nsamples;       /* samples the caller pushed for entry i (before padding / truncation to the frame length) */
    float host_us[3];              /* where this tick's wall time went: buffer swap + queued frames | copies + launches + wait | segment assembly */
    int64_t dropped;               /* ABI 3: staged frames left out because their stream was closed (or closed and reopened) after the push */
    int64_t staged_next;           /* ABI 3: frames that had been waiting and are already staged for the NEXT tick - a ticker that sees
                                      > 0 runs again at once instead of sleeping (a client that sends faster than real time) */
} vad_tick_result;
VAD_API int vad_tick_push(vad_engine *e, int64_t slot, const void *samples, int32_t nsamples, int frame_fmt, int gate_on);
VAD_API int vad_tick_push_rate(vad_engine *e, int64_t slot, const void *samples, int32_t nsamples, int frame_fmt, int gate_on,
                               int32_t sr_in);
/* the same frame length / format / gate for n slots: frames [n][nsamples] (a front end that batches its sockets' frames) */
VAD_API int vad_tick_push_many(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int32_t nsamples,
                               int frame_fmt, int gate_on);
/* the same, but every frame is tried and status[i] receives its own result (VAD_OK, VAD_ERR_BAD_SLOT, VAD_ERR_BUSY ...): a
 * front end that coalesces the frames its sockets received during one tick window into ONE call learns which of them to
 * report to which client; returns the first failure (the last-error text belongs to the LAST one) */
VAD_API int vad_tick_push_status(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int32_t nsamples,
                                 int frame_fmt, int gate_on, int32_t *status);
/* the same with one pointer per frame (frames[i] -> nsamples samples): the sockets' receive buffers are copied straight into the
 * tick's staging rows, without being gathered into one array first */
VAD_API int vad_tick_push_gather(vad_engine *e, const int64_t *slots, int64_t n, const void *const *frames, int32_t nsamples,
                                 int frame_fmt, int gate_on, int32_t *status);
/* vad_tick_push_rate for n clients at ONE input rate, one pointer per chunk, a result per chunk: what vad_tick_push_gather is for
 * frames at the engine's rate (the reference declares this conversion and leaves it empty, vad_wrapper.py:621-624; its server
 * would call it once per message, vad_websocket_server.py:326-382).  int16 chunks are scaled to float32 on the way in. */
VAD_API int vad_tick_push_rate_gather(vad_engine *e, const int64_t *slots, int64_t n, const void *const *frames, int32_t nsamples,
                                      int frame_fmt, int gate_on, int32_t sr_in, int32_t *status);
VAD_API int vad_tick_cancel(vad_engine *e, int64_t slot);
VAD_API int vad_tick_pending(vad_engine *e, int64_t slot, int64_t *frames);
/*
 * Segment assembly inside the tick (off by default).  When on, vad_tick_run also does the host half of
 * VADProcessor._process_voice_state (core/silero_model.py:838-869, 891-895, 925-949) for every stepped stream, on the staged
 * audio converted to float32 and gated like the model input (utils/audio.py:117-118): frames at or above the slot's
 * vad_start_probability collect as pre-roll, START turns the pre-roll into the segment, frames of an open segment are appended
 * (whole frames, also beyond the model's 512 samples), END closes it.  vad_tick_take_segment then hands over the finished
 * segment's samples (out = NULL: size query; taking clears it) - the payload of voice_end_callback before WAV encoding.
 * A serving loop then touches a stream in its own language only on START / END.
 */
VAD_API int vad_tick_enable_segments(vad_engine *e, int on);
VAD_API int vad_tick_take_segment(vad_engine *e, int64_t slot, float *out, int64_t cap, int64_t *nsamples);
/* A stream's segment audio (pre-roll, open segment, finished segment not yet taken) as an opaque blob: with vad_stream_save /
 * vad_stream_restore this is everything a session needs to continue on ANOTHER engine (another GPU) in the middle of an
 * utterance.  buf = NULL: size query.  Restore replaces what the slot holds; blobs are checked before anything is touched. */
VAD_API int vad_tick_segment_save(vad_engine *e, int64_t slot, void *buf, int64_t cap, int64_t *nbytes);
VAD_API int vad_tick_segment_restore(vad_engine *e, int64_t slot, const void *buf, int64_t nbytes);
VAD_API int vad_tick_run(vad_engine *e, float denoise_thresh, vad_tick_result *out);

/*
 * ABI 4: vad_tick_run + the per-session bookkeeping a serving front end does with its result, in the same call - so that a
 * front end whose callbacks live in an interpreter touches only the sessions that HAVE something to hear (the reference does all
 * of this per frame and client in Python: VADProcessor.process_frame / VADWrapper._handle_callbacks,
 * core/silero_model.py:723-762, core/vad_wrapper.py:478-522).
 *   in  (caller-owned, one entry per slot, n_slots entries; updated for every stepped slot i = slots[k]):
 *         last_prob[i] = probs[k];  frames_done[i] += 1;  active[i] = (active[i] | START) & !END   ("inside a segment")
 *       continue_cb[i] / continue_payload[i]: the session registered a voice_continue callback / wants the frame's bytes with it
 *   out (engine-owned, valid until the next tick): the entries k of the tick's arrays the caller has work for, in order, with
 *       work_kind[j] = VAD_WORK_START | VAD_WORK_END (the tick's event bits) | VAD_WORK_CONTINUE (the session was inside a segment
 *       before this frame and has a voice_continue callback: vad_wrapper.py:513-519) | VAD_WORK_PAYLOAD (... which wants the bytes)
 *       | VAD_WORK_LONG (the pushed frame was longer than the model's frame: the caller kept the whole frame, vad_tick_push)
 * Entries without any of these (idle sessions, sessions talking without a continue callback) are not listed.
 */
#define VAD_WORK_START 1
#define VAD_WORK_END 2
#define VAD_WORK_CONTINUE 4
#define VAD_WORK_PAYLOAD 8
#define VAD_WORK_LONG 16
typedef struct vad_tick_work {
    uint32_t struct_size;          /* sizeof(vad_tick_work) */
    int64_t n_slots;               /* length of the five arrays below */
    float *last_prob;
    int64_t *frames_done;
    uint8_t *active;
    const uint8_t *continue_cb;
    const uint8_t *continue_payload;
    int64_t n_work;                /* out */
    const int32_t *work_index;     /* out: k into vad_tick_result's arrays */
    const uint8_t *work_kind;      /* out */
    const int64_t *work_samples;   /* out: VAD_WORK_END entries: samples of the finished segment (vad_tick_take_segment*), else 0 */
} vad_tick_work;
VAD_API int vad_tick_run_work(vad_engine *e, float denoise_thresh, vad_tick_result *out, vad_tick_work *work);

/*
 * ABI 4: vad_tick_take_segment as the finished payload of voice_end_callback: 44-byte RIFF/WAVE header + int16 PCM, byte for
 * byte what WAVWriter.write_wav_data makes of the segment (utils/wav_writer.py:40-136: clip(x * 32767, -32768, 32767) truncated
 * to int16, mono).  out == NULL: size query (*nbytes = 44 + 2 * samples; vad_tick_work.work_samples has the count already).
 * The segment is released when it has been written.
 */
VAD_API int vad_tick_take_segment_wav16(vad_engine *e, int64_t slot, int32_t sample_rate, void *out, int64_t cap, int64_t *nbytes);

/*
 * One tick for streams whose audio arrives at another rate: VADConfig.auto_convert_sample_rate.  The reference's hook for it
 * is core/vad_wrapper.py:621-624 - a `pass` - and the function it was meant to call is AudioUtils.resample_audio
 * (utils/audio.py:19-55); this entry point is that path, on the GPU: segment k holds n[k] chunks of one tick at sr_in[k]
 * (256 samples @ 8 kHz, 768 @ 24 kHz, 1536 @ 48 kHz - one 512-sample 16 kHz frame each; 512 @ 16 kHz passes through, as
 * resample_audio does at :39-40); all segments are resampled in ONE launch into engine-owned HBM and every stream advances one
 * frame in ONE model launch right behind it on the same HIP stream - the 16 kHz frames never travel.  A Silero V5 engine serves
 * ticks that fit one 16-stream tile per CU (each segment padded to whole tiles: at most 256 tiles, ~4 000 streams) with ONE
 * fused launch instead: every tile resamples its own chunks into LDS and steps the model from there (results equal the
 * two-launch form to rounding).  slots / probs / events / seg_frames are the
 * concatenation of the segments, in order.  16 kHz engines only (Silero V5, or V4's 16 kHz sub-model).
 * The device form is asynchronous like vad_step_device and follows its slot rules; calls on one engine must use one stream.
 */
VAD_API int vad_step_rates_device(vad_engine *e, int32_t nseg, const float *const *d_in, const int64_t *n, const int32_t *sr_in,
                                  const int32_t *d_slots, float denoise_thresh, float *d_probs, uint8_t *d_events,
                                  int32_t *d_seg_frames, void *stream);
VAD_API int vad_step_rates(vad_engine *e, int32_t nseg, const float *const *in, const int64_t *n, const int32_t *sr_in,
                           const int64_t *slots, float denoise_thresh, float *probs_out, uint8_t *events_out,
                           int32_t *seg_frames_out);

/*
 * Diagnostic (no GPU needed): run the host-side weight packer and return the per-wave MFMA
 * weight streams exactly as vad_engine_create uploads them.  out may be NULL to query the size.
 * sect_out receives [4 waves][16 sections] block offsets (1 block = 256 floats).  Used by the
 * CPU test-suite to check the packed layout against a NumPy model of the kernel's dataflow.
 */
VAD_API int vad_debug_pack_weights(int32_t model_version, const void *weights, size_t weights_len, float *out,
                                   size_t out_floats, size_t *n_floats, uint32_t *sect_out);

/*
 * Diagnostic (no GPU needed): the dense operator R[512][n_in] (row-major) the resampler kernel
 * applies for chunks of n_in samples; the CPU test-suite checks R @ x against scipy.signal.resample.
 */
VAD_API int vad_debug_resample_operator(int32_t n_in, float *R, size_t r_floats);
/* rows m0 .. m1-1 of the operator vad_resample_generic applies, R[(m - m0) * n_in + n] in float64, evaluated on the HOST with the
 * arithmetic of the kernel (same tables, same small-angle rule): the CPU test-suite checks R @ x against scipy for awkward shapes */
/* which kernel vad_resample_generic uses: 0 = chosen by size (default), 1 = the direct kernel, 2 = the chirp-z / FFT path (tests, benchmarks) */
VAD_API int vad_debug_resample_path(vad_engine *e, int mode);
VAD_API int vad_debug_resample_generic_entries(int64_t n_in, int64_t n_out, int64_t m0, int64_t m1, double *R, size_t r_doubles);

/*
 * Diagnostic (no GPU needed): the folded, MFMA-packed form of that operator exactly as the kernel
 * streams it (csrc/pack_weights.cpp: pack_resample_operator).  out may be NULL to query n_floats.
 * tile_blocks = 1 KiB blocks per 32-row tile, row128_block = first block of the VALU row.
 */
VAD_API int vad_debug_pack_resample(int32_t n_in, float *out, size_t out_floats, size_t *n_floats,
                                    uint32_t *tile_blocks, uint32_t *row128_block);

/* the same operator as the fused resample -> step kernel streams it (16 x 16 x 4 tiles: pack_resample_operator_t16);
 * wave_blocks = 1 KiB blocks per wave, which also tells the kernel the stream's shape: 4 vector blocks + 8 per k-iteration of 16
 * folded samples - over n_in / 4 samples per part (every sample contracted), or over n_in / 6 for n_in = 768 / 1536 (24 / 48 kHz:
 * every third input sample sits on an output instant and is copied) - or, for n_in = 256 (8 kHz: the even outputs are the input
 * samples), 2 vector blocks + 4 per k-iteration: only the odd output rows */
VAD_API int vad_debug_pack_resample_t16(int32_t n_in, float *out, size_t out_floats, size_t *n_floats,
                                        uint32_t *wave_blocks, uint32_t *row128_block);

/*
 * Diagnostic: replay a scripted probability sequence through ONE slot's device-side state
 * machine (the code path vad_step_events runs after the model).  probs [n] -> events_out [n],
 * seg_frames_out [n] (segment length in frames on END, else 0).  Lets the GPU test-suite check
 * the hysteresis logic against the reference's own traces without needing audio that produces
 * a given probability sequence.
 */
VAD_API int vad_debug_sm_replay(vad_engine *e, int64_t slot, const float *probs, int64_t n, uint8_t *events_out,
                                int32_t *seg_frames_out);

/*
 * Diagnostic: which kernel shape serves the step calls.  0 (default) = the engine's choice - Silero V5 (both sub-models): one-frame
 * calls run on 16-stream tiles whatever their size (the single-frame instantiation of that kernel fits two workgroups on a CU),
 * multi-frame calls on 16-stream tiles up to 4 096 streams and on 32-stream tiles above; Silero V4 (both sub-models): always
 * 16-stream tiles, two workgroups per CU.  16 / 32 force one shape (the test-suite checks that both give the same results to
 * rounding; tools/bench_configs.py times them).
 * -1 / -2: vad_step_rates as two launches (resample, then model) / as the fused launch (default), for the same comparison.
 */
VAD_API int vad_debug_set_tile(vad_engine *e, int32_t streams_per_tile);

/* block until everything enqueued on the engine's own stream has finished */
VAD_API int vad_engine_synchronize(vad_engine *e);

/*
 * Page-locked host memory for the caller's frame / result buffers.  The host-pointer entry points
 * (vad_step, vad_step_events, vad_step_multi, vad_resample) accept ANY host pointer; buffers from
 * this allocator are DMA'd directly (no runtime staging copy), which is what bounds a large batch:
 * 8 192 f32 frames are 16.8 MB per step.  The reference has no counterpart (numpy arrays handed to
 * onnxruntime, silero_model.py:471-499); the Python mirror exposes it as Engine.pinned_array().
 * Memory stays valid until vad_host_free or vad_engine_destroy.
 */
VAD_API int vad_host_alloc(vad_engine *e, size_t bytes, void **out);
VAD_API int vad_host_free(vad_engine *e, void *p);

#ifdef __cplusplus
}
#endif
#endif /* VAD_ENGINE_H */

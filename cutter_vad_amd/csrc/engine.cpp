// C ABI of the batched Silero-VAD engine (include/vad_engine.h): stream pool, device state,
// host<->device staging and kernel launches.  Plain C++ over the HIP runtime API; the kernels
// live in silero_v5.hip / silero_v4.hip / resample.hip and are reached through
// vadk_launch_* so that this file never needs device compilation.
//
// Reference behaviour mirrored here (paths under /root/reference/src/real_time_vad/):
//   core/silero_model.py:276-334  model load            -> vad_engine_create
//   core/silero_model.py:384-401  zero-initialised state -> vad_stream_open / vad_stream_reset
//   core/silero_model.py:403-447  predict               -> vad_step*
//   core/silero_model.py:548-566  get_model_info        -> vad_engine_info
// There is deliberately no CPU execution path in this file.
#include "../../include/vad_engine.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdlib>
#include <new>
#include <chrono>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "pack_weights.h"
#include "resample_generic.h"
#include "vad_layout.h"

extern "C" hipError_t vadk_launch_silero_v5(const vadk::StepParams *p, hipStream_t stream);
extern "C" hipError_t vadk_launch_silero_v4(const vadk::StepParams *p, hipStream_t stream);
extern "C" hipError_t vadk_launch_silero_v5_t16(const vadk::StepParams *p, hipStream_t stream);
extern "C" hipError_t vadk_launch_silero_v4_t16(const vadk::StepParams *p, int one_per_cu, hipStream_t stream);
extern "C" hipError_t vadk_launch_silero_v5_t16_rates(const vadk::StepParams *p, const vadk::RateParams *r, hipStream_t stream);
extern "C" hipError_t vadk_launch_resample(const vadk::ResampleParams *p, hipStream_t stream);
extern "C" hipError_t vadk_launch_slot_control(vadk::SmSlot *sm, float *state, const int32_t *d_slots, int n, int op,
                                               const vadk::SmSlot *def, const vad_thresholds *d_thr, int nthr, hipStream_t stream);
extern "C" hipError_t vadk_launch_sm_replay(vadk::SmSlot *sm, int slot, const float *probs, int n, uint8_t *events,
                                            int32_t *seg, hipStream_t stream);

namespace {

thread_local std::string g_create_error;
thread_local std::string g_error_copy;      // vad_last_error hands out a copy taken under the engine's mutex

const vadk::SmSlot kDefaultSm = [] {
    vadk::SmSlot s;
    std::memset(&s, 0, sizeof s);
    // VADConfig defaults, core/config.py:54-94
    s.start_prob = 0.7; s.end_prob = 0.7; s.start_ratio = 0.8; s.end_ratio = 0.95;
    s.start_count = 10; s.end_count = 50;
    s.seg_frames = -1;
    return s;
}();

}  // namespace

// Packed weight streams on a device, shared by every engine of the process that was created from the same bytes on that device
// (a sharded pool with several engines per GPU, two pools side by side: one copy in HBM - and in the L2s, which the engines' launches
// then share: two pools of 4 096 streams on one GPU step in 41.0 us with one copy, 41.9 with two).  Immutable after the upload;
// reference counted; freed with the last engine.
struct DeviceWeights {
    int device = 0;
    std::vector<float> host;          // the key: the packed stream itself
    float *dev = nullptr;
    int refs = 0;
};
static std::mutex g_weights_mu;
static std::vector<DeviceWeights *> g_weights;

// the device copy of `data` on `device` (the caller has made it the current device); +1 reference
static hipError_t weights_acquire(int device, const std::vector<float> &data, float **out) {
    std::lock_guard<std::mutex> lk(g_weights_mu);
    for (DeviceWeights *w : g_weights)
        if (w->device == device && w->host.size() == data.size() && std::memcmp(w->host.data(), data.data(), data.size() * sizeof(float)) == 0) {
            w->refs += 1;
            *out = w->dev;
            return hipSuccess;
        }
    DeviceWeights *w = new DeviceWeights;
    w->device = device;
    w->host = data;
    hipError_t r = hipMalloc((void **)&w->dev, data.size() * sizeof(float));
    if (r == hipSuccess) r = hipMemcpy(w->dev, data.data(), data.size() * sizeof(float), hipMemcpyHostToDevice);
    if (r != hipSuccess) {
        if (w->dev) (void)hipFree(w->dev);
        delete w;
        return r;
    }
    w->refs = 1;
    g_weights.push_back(w);
    *out = w->dev;
    return hipSuccess;
}

static void weights_release(float *dev) {
    if (!dev) return;
    std::lock_guard<std::mutex> lk(g_weights_mu);
    for (size_t i = 0; i < g_weights.size(); ++i)
        if (g_weights[i]->dev == dev) {
            if (--g_weights[i]->refs == 0) {
                (void)hipFree(dev);
                delete g_weights[i];
                g_weights.erase(g_weights.begin() + (long)i);
            }
            return;
        }
}

struct vad_engine {
    int version = 5;
    int device = 0;
    int max_streams = 0;
    hipStream_t stream = nullptr;
    float *d_wstream = nullptr;
    size_t wbytes = 0;
    float *d_state = nullptr;
    vadk::SmSlot *d_sm = nullptr;
    // staging for the host-pointer entry points (grown on demand)
    void *d_frames = nullptr;  size_t d_frames_cap = 0;
    float *d_probs = nullptr;  size_t d_probs_cap = 0;
    uint8_t *d_events = nullptr; size_t d_events_cap = 0;
    int32_t *d_seg = nullptr;  size_t d_seg_cap = 0;
    int32_t *d_slots = nullptr; size_t d_slots_cap = 0;
    // small calls (a few streams: the one-wrapper-per-client pattern): ONE pinned block in, ONE pinned block out
    static constexpr size_t SMALL_BYTES = 256u << 10;
    uint8_t *h_small_in = nullptr, *h_small_out = nullptr;   // hipHostMalloc
    std::vector<void *> host_blocks;                         // vad_host_alloc: freed with the engine
    uint8_t *d_small_in = nullptr, *d_small_out = nullptr;
    vadk::StepParams base{};
    // Silero V5: the same weights packed for the 16-stream tile kernel (csrc/silero_v5_t16.hip).  It serves
    //   - every ONE-frame call (T == 1: the serving tick, vad_step, the bench): its single-frame instantiation needs 220 registers
    //     and 80.6 KB of LDS, so TWO workgroups share a CU and fill each other's waits - 8 192 streams 44.1 us against 45.4 on
    //     32-stream tiles, 12 288 streams 65.9 against 87.9 (three tiles on a CU instead of two rounds of one);
    //   - multi-frame calls of at most T16_MAX_STREAMS streams: half-size tiles put a small batch on twice as many CUs (its
    //     multi-frame instantiation has 256 - 265 registers: one workgroup per CU, so a larger batch would run in two rounds).
    static constexpr int T16_MAX_STREAMS = 4096;
    float *d_wstream16 = nullptr;
    size_t wbytes16 = 0;
    uint32_t sect16[vadk::NWAVES][16] = {};
    int tile_policy = 0;                     // 0 = by batch size, 16 / 32 = forced (vad_debug_set_tile)
    bool rates_fused = true;                 // vad_debug_set_tile(-1 / -2): two-launch / fused form of vad_step_rates (benchmarks)
    bool shared_gpu = false;                 // VAD_ENGINE_SHARED_GPU: keep to 32-stream tiles (n / 32 CUs), leave the rest to the co-tenant
    int sample_rate = 16000;
    int frame_samples = VAD_FRAME_SAMPLES;   // samples per model step (512; Silero V5's 8 kHz sub-model: 256)
    std::vector<int32_t> work_index;         // vad_tick_run_work: the tick's entries the caller has work for
    std::vector<uint8_t> work_kind;
    std::vector<int64_t> work_samples;
    // batched slot control (open / reset / thresholds): one pinned block up, one kernel
    uint8_t *h_ctl = nullptr, *d_ctl = nullptr; size_t ctl_cap = 0;
    // pipelined host ingest (vad_step_submit / vad_step_collect): H2D of ticket t+1 on `copy_in` while the kernel of
    // ticket t runs on `stream`; results come back on `copy_out` into a pinned block
    static constexpr int PIPE_DEPTH = 2;
    struct PipeBuf {
        void *d_frames = nullptr; size_t d_frames_cap = 0;
        uint8_t *d_io = nullptr, *h_io = nullptr; size_t io_cap = 0;   // [slots i32 n | probs f32 n T | seg i32 n | events u8 n T]
        hipEvent_t copied = nullptr, done = nullptr, out = nullptr;
        int64_t ticket = -1; int64_t n = 0; int32_t T = 0; bool busy = false, collecting = false;
    } pipe[PIPE_DEPTH];
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    int64_t next_ticket = 0;
    // tick assembler (vad_tick_push / vad_tick_run): producers write a slot's next frame straight into the page-locked
    // staging row of the coming tick, grouped by (frame format, gate on/off) = the launches of that tick; double-buffered so
    // that frames keep arriving while a tick runs.  A slot's further frames wait in `tick_overflow` (one frame per slot and tick).
    // groups 0..5 = frame_fmt * 2 + gate_on: frames at the engine's own rate, staged in their wire format;
    // groups 6..11 = 6 + 3 * gate_on + {0: 8 kHz, 1: 24 kHz, 2: 48 kHz}: chunks at another input rate (vad_tick_push_rate), staged
    // as float32 [n_in] and resampled on the GPU inside the tick (vad_step_rates' path)
    static constexpr int TICK_GROUPS = VAD_TICK_GROUPS;
    static int tick_group_len(int group, int frame_samples) {
        static const int rate_len[3] = {256, 768, 1536};
        return group < 6 ? frame_samples : rate_len[(group - 6) % 3];
    }
    static int tick_group_rate(int group) {
        static const int rate[3] = {8000, 24000, 48000};
        return rate[(group - 6) % 3];
    }
    static size_t tick_sample_bytes(int group) { return (group >= 2 && group < 6) ? 2 : 4; }
    struct TickBuf {
        uint8_t *h = nullptr;                        // pinned: [cap] rows of frame bytes, then [cap] int32 slots, [cap] int32 lengths, [cap] uint32 epochs
        int64_t cap = 0, count = 0;
        size_t row_bytes = 0;
        uint8_t *row(int64_t r) const { return h + (size_t)r * row_bytes; }
        int32_t *slots() const { return reinterpret_cast<int32_t *>(h + (size_t)cap * row_bytes); }
        int32_t *lens() const { return slots() + cap; }   // samples the caller pushed (before padding / truncation)
        uint32_t *epochs() const { return reinterpret_cast<uint32_t *>(lens() + cap); }   // slot_epoch at push time: a row of a slot closed (and reopened) since is stale
        void drop_row(int64_t r) {                   // the last row fills the hole
            const int64_t last = count - 1;
            if (r != last) {
                std::memcpy(row(r), row(last), row_bytes);
                slots()[r] = slots()[last];
                lens()[r] = lens()[last];
                epochs()[r] = epochs()[last];
            }
            count = last;
        }
    };
    struct TickPending { std::vector<uint8_t> data; int32_t nsamples; int group; };   // the whole frame as pushed
    std::mutex tick_mu;                              // lock order: mu, then tick_mu
    int tick_cur = 0;
    TickBuf tick_buf[2][TICK_GROUPS];
    std::vector<uint32_t> tick_gen;                  // per slot: == tick_generation <=> the slot has a frame in the coming tick
    uint32_t tick_generation = 1;
    std::vector<uint32_t> slot_epoch;                // per slot: bumped by every open (written under mu + tick_mu)
    std::unordered_map<int64_t, std::deque<TickPending>> tick_overflow;
    std::vector<int64_t> tick_overflow_order;        // the slots of tick_overflow in the order they started waiting: their frames are placed in this order
    std::unordered_map<int64_t, std::deque<std::vector<uint8_t>>> tick_tails;   // samples past the model's frame of over-long frames, push order
    // segment assembly on the host side of the tick (vad_tick_enable_segments): what SegmentAssembler / VADProcessor keep per
    // stream (core/silero_model.py:838-869, 891-895, 925-949) - pre-roll, the open segment, the finished one until taken
    // The audio stays in its wire format (int16 stays int16: half the bytes, a memcpy per frame) as runs of (group, samples,
    // gate threshold); it becomes float32 - scaled with a true division and gated - when the finished segment is taken.
    // Storage: fixed 32 KB blocks from an arena (a free list over 8 MB chunks that were touched when they were allocated), so
    // that the tick never reallocates and never page-faults while 8 192 talking streams append a frame each; the tick's
    // assembly pass only PLANS the copies (destination block, offset, bytes) and a few threads then carry them out.
    struct SegArena {
        static constexpr size_t BLOCK = 32768, CHUNK_BLOCKS = 256;
        std::vector<uint8_t *> chunks, free_blocks;
        ~SegArena() { for (uint8_t *c : chunks) std::free(c); }
        void grow() {
            uint8_t *c = static_cast<uint8_t *>(std::malloc(BLOCK * CHUNK_BLOCKS));
            if (!c) throw std::bad_alloc();
            std::memset(c, 0, BLOCK * CHUNK_BLOCKS);          // fault the pages in now, not inside a tick
            chunks.push_back(c);
            for (size_t k = CHUNK_BLOCKS; k-- > 0;) free_blocks.push_back(c + k * BLOCK);
        }
        void reserve_blocks(size_t n) { while (free_blocks.size() < n) grow(); }
        uint8_t *get() {
            if (free_blocks.empty()) grow();
            uint8_t *b = free_blocks.back();
            free_blocks.pop_back();
            return b;
        }
        void put(uint8_t *b) { free_blocks.push_back(b); }
    };
    // one planned copy.  kind 0: `bytes` bytes as they are; 1 / 2: `bytes` bytes of int16 -> float32 / 32767 | / 32768 (dst holds
    // 2 x bytes): the resampled groups' chunks, converted by whoever carries the plan out
    struct SegCopy {
        uint8_t *dst; const uint8_t *src; uint32_t bytes; uint32_t kind = 0;
        void carry_out() const {
            if (kind == 0) { std::memcpy(dst, src, bytes); return; }
            const float sc = kind == 1 ? 32767.0f : 32768.0f;
            const int16_t *q = reinterpret_cast<const int16_t *>(src);
            float *o = reinterpret_cast<float *>(dst);
            for (uint32_t k = 0; k < bytes / 2; ++k) o[k] = (float)q[k] / sc;       // numpy's true division
        }
    };
    struct SegAudio {
        struct Run { int32_t group; float thr; int64_t samples; };      // group as in the tick: tells sample type, int16 scale, gate
        std::vector<uint8_t *> blocks;
        size_t tail = SegArena::BLOCK;                                  // bytes used in the last block (BLOCK: none / full)
        std::vector<Run> runs;
        int64_t samples = 0;
        static bool is_i16(int group) { return group >= 2 && group < 6; }
        static bool gated(int group) { return group < 6 ? (group & 1) : (group >= 9); }
        bool empty() const { return samples == 0; }
        void clear(SegArena &a) {
            for (uint8_t *b : blocks) a.put(b);
            blocks.clear();
            runs.clear();
            tail = SegArena::BLOCK;
            samples = 0;
        }
        void swap(SegAudio &o) { blocks.swap(o.blocks); runs.swap(o.runs); std::swap(samples, o.samples); std::swap(tail, o.tail); }
        // appends cnt samples of `group`: plans the byte copies into `out` (or does them, out == nullptr)
        void append(SegArena &a, int group, float thr, const uint8_t *src, size_t cnt, std::vector<SegCopy> *out) {
            if (!cnt) return;
            const size_t ss = is_i16(group) ? 2 : 4;
            const bool same = !runs.empty() && runs.back().group == group && runs.back().thr == thr;
            if (!same && ss == 4 && !blocks.empty()) tail = (tail + 3) & ~(size_t)3;     // a float32 run starts 4-aligned (the reader applies the same rule)
            size_t bytes = cnt * ss;
            while (bytes) {
                if (tail >= SegArena::BLOCK) {
                    blocks.push_back(a.get());
                    tail = 0;
                }
                const size_t k = std::min(bytes, SegArena::BLOCK - tail);
                if (out) out->push_back(SegCopy{blocks.back() + tail, src, (uint32_t)k});
                else std::memcpy(blocks.back() + tail, src, k);
                tail += k;
                src += k;
                bytes -= k;
            }
            if (same) runs.back().samples += (int64_t)cnt;
            else runs.push_back(Run{group, thr, (int64_t)cnt});
            samples += (int64_t)cnt;
        }
        // the stored bytes of every run in order: fn(run, pointer, samples in this piece); pieces never split a sample
        template <class F>
        void for_each_piece(F fn) const {
            size_t bi = 0, off = 0;
            for (const Run &r : runs) {
                const size_t ss = is_i16(r.group) ? 2 : 4;
                if (ss == 4) off = (off + 3) & ~(size_t)3;
                size_t left = (size_t)r.samples;
                while (left) {
                    if (off >= SegArena::BLOCK) { ++bi; off = 0; }
                    const size_t k = std::min(left, (SegArena::BLOCK - off) / ss);
                    fn(r, blocks[bi] + off, k);
                    off += k * ss;
                    left -= k;
                }
            }
        }
        void to_float(float *o) const {
            for_each_piece([&](const Run &r, const uint8_t *src, size_t cnt) {
                if (!is_i16(r.group)) std::memcpy(o, src, cnt * 4);
                else {
                    const float sc = r.group < 4 ? 32767.0f : 32768.0f;     // np.int16 -> float32 / 32767.0 (true division)
                    const int16_t *q = reinterpret_cast<const int16_t *>(src);
                    for (size_t k = 0; k < cnt; ++k) o[k] = (float)q[k] / sc;
                }
                if (gated(r.group))                                        // the group's gate: utils/audio.py:117-118
                    for (size_t k = 0; k < cnt; ++k) o[k] = std::fabs(o[k]) > r.thr ? o[k] : 0.f;
                o += cnt;
            });
        }
        size_t stored_bytes() const {
            size_t b = 0;
            for (const Run &r : runs) b += (size_t)r.samples * (is_i16(r.group) ? 2 : 4);
            return b;
        }
    };
    struct SegState {
        bool active = false;
        SegAudio pre, seg, done;
        void clear(SegArena &a) { active = false; pre.clear(a); seg.clear(a); done.clear(a); }
    };
    SegArena seg_arena;
    std::vector<SegCopy> seg_copies;                 // the copies one tick planned
    std::vector<SegCopy> push_copies;                // ... and the conversions a batched rate push planned (both under tick_mu)
    // the threads that carry out a tick's planned copies next to the calling one
    struct CopyCrew {
        std::vector<std::thread> th;
        std::mutex m;
        std::condition_variable cv_go, cv_done;
        const SegCopy *items = nullptr;
        size_t n = 0;
        std::atomic<size_t> next{0};
        uint64_t job = 0;
        int busy = 0;
        bool stop = false;
        static void work(const SegCopy *it, size_t n, std::atomic<size_t> &next) {
            for (;;) {
                const size_t a = next.fetch_add(128, std::memory_order_relaxed);
                if (a >= n) return;
                const size_t b = std::min(n, a + 128);
                for (size_t k = a; k < b; ++k) it[k].carry_out();
            }
        }
        void start(int threads) {
            for (int t = 0; t < threads; ++t)
                th.emplace_back([this] {
                    uint64_t seen = 0;
                    std::unique_lock<std::mutex> lk(m);
                    for (;;) {
                        cv_go.wait(lk, [&] { return stop || job != seen; });
                        if (stop) return;
                        seen = job;
                        const SegCopy *it = items;
                        const size_t cnt = n;
                        lk.unlock();
                        work(it, cnt, next);
                        lk.lock();
                        if (--busy == 0) cv_done.notify_one();
                    }
                });
        }
        void run(const SegCopy *it, size_t cnt) {
            size_t bytes = 0;
            for (size_t k = 0; k < cnt; ++k) bytes += it[k].bytes;
            if (th.empty() || bytes < (256u << 10)) {
                for (size_t k = 0; k < cnt; ++k) it[k].carry_out();
                return;
            }
            {
                std::lock_guard<std::mutex> lk(m);
                items = it;
                n = cnt;
                next.store(0, std::memory_order_relaxed);
                busy = (int)th.size();
                ++job;
            }
            cv_go.notify_all();
            work(it, cnt, next);
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return busy == 0; });
        }
        ~CopyCrew() {
            {
                std::lock_guard<std::mutex> lk(m);
                stop = true;
            }
            cv_go.notify_all();
            for (auto &t : th) t.join();
        }
    } copy_crew;
    bool tick_segments = false;
    std::vector<SegState> seg_state;
    std::vector<double> h_start_prob;                // host copy of each slot's vad_start_probability (pre-roll rule :832-839)
    uint8_t *h_tick_out = nullptr, *d_tick_out = nullptr; size_t tick_out_cap = 0;   // [slots i64 | probs f32 | seg i32 | events u8] x max n
    void *d_tick_frames = nullptr; size_t d_tick_frames_cap = 0;
    struct ResampleOp {
        int n_in = 0;
        float *d_w = nullptr;
        size_t bytes = 0;
        uint32_t tile_blocks = 0;
        uint32_t row128_block = 0;
    };
    std::vector<ResampleOp> resample_ops;   // built lazily, one per input rate
    std::vector<ResampleOp> resample_ops16; // the same operators packed for the fused resample -> step kernel (16-stream tiles)
    float *d_rs_in = nullptr;  size_t d_rs_in_cap = 0;
    float *d_rs_out = nullptr; size_t d_rs_out_cap = 0;
    // generic whole-array resampler (vad_resample_generic): the float64 tables of the last few (n_in, n_out) shapes, on the device
    struct RsgEntry {
        int64_t n_in = 0, n_out = 0, a = 0, b = 0, L = 0, P = 0;
        int32_t corrected = 0;
        double *d_tn = nullptr, *d_tm = nullptr;
        size_t bytes = 0;
        uint64_t used = 0;
    };
    static constexpr size_t RSG_CACHE_ENTRIES = 8;
    static constexpr size_t RSG_CACHE_BYTES = 512u << 20;
    std::vector<RsgEntry> rsg_cache;
    uint64_t rsg_clock = 0;
    // long arrays: chirp-z / FFT path (csrc/resample_fft.hip): tables of the last (n_in, n_out), work buffers
    struct RsfPlan {
        int64_t n_in = 0, n_out = 0, P1 = 0, P2 = 0;
        void *W1 = nullptr, *W2 = nullptr, *B1 = nullptr, *B2 = nullptr;
    } rsf_plan;
    void *d_rsf_a = nullptr, *d_rsf_b = nullptr; size_t d_rsf_cap = 0;
    int rsg_path = 0;                                // 0 = by size, 1 = direct kernel, 2 = FFT path (vad_debug_resample_path)
    double *d_rsg_partial = nullptr; size_t d_rsg_partial_cap = 0;
    void *d_rsg_in = nullptr;  size_t d_rsg_in_cap = 0;
    float *d_rsg_out = nullptr; size_t d_rsg_out_cap = 0;
    std::vector<uint8_t> open;
    std::vector<int64_t> free_list;
    std::vector<uint32_t> stamp;   // duplicate detection per step
    uint32_t stamp_gen = 0;
    int open_count = 0;
    int64_t steps = 0, frames = 0;
    hipDeviceProp_t prop{};
    mutable std::mutex mu;
    mutable std::mutex err_mu;        // `err` is written under `mu` by most entry points and under `tick_mu` by vad_tick_push
    mutable std::string err;

    int fail(int code, const char *fmt, ...) const {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        std::lock_guard<std::mutex> lk(err_mu);
        err = buf;
        return code;
    }
    int hip_fail(hipError_t e, const char *what) const {
        (void)hipGetLastError();      // reported here; do not leave it in HIP's process-wide last-error slot
        return fail(VAD_ERR_HIP, "Model prediction failed: %s: %s", what, hipGetErrorString(e));
    }
};

#define HIP_TRY(e, call)                                         \
    do {                                                         \
        hipError_t _r = (call);                                  \
        if (_r != hipSuccess) return (e)->hip_fail(_r, #call);   \
    } while (0)

namespace {

template <class T>
int ensure(vad_engine *e, T *&ptr, size_t &cap, size_t need) {
    if (need <= cap) return VAD_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    void *p = nullptr;
    hipError_t r = hipMalloc(&p, need);
    if (r != hipSuccess) return e->hip_fail(r, "hipMalloc(staging)");
    ptr = static_cast<T *>(p);
    cap = need;
    return VAD_OK;
}

size_t frame_bytes(const vad_engine *e, int fmt) { return (fmt == VAD_FMT_F32 ? 4u : 2u) * (size_t)e->frame_samples; }

// The kernels address frames through a 32-bit buffer descriptor with signed 32-bit offset arithmetic: one call may not
// span 2 GiB of frames (1 M float32 frames).  Rejected here instead of wrapping silently.
int check_call_size(vad_engine *e, int64_t n, int32_t T, int fmt) {
    if (n < 0 || T < 1 || n > e->max_streams)
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: bad stream or frame count (n = %lld, T = %d, max_streams = %d)",
                       (long long)n, T, e->max_streams);
    if ((uint64_t)n * (uint64_t)T * frame_bytes(e, fmt) >= (1ull << 31))
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: n * T * frame bytes = %llu exceeds the 2 GiB one call may address",
                       (unsigned long long)((uint64_t)n * (uint64_t)T * frame_bytes(e, fmt)));
    return VAD_OK;
}

int check_slots(vad_engine *e, const int64_t *slots, int64_t n) {
    if (++e->stamp_gen == 0) {
        std::fill(e->stamp.begin(), e->stamp.end(), 0u);
        e->stamp_gen = 1;
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t s = slots[i];
        if (s < 0 || s >= e->max_streams || !e->open[(size_t)s])
            return e->fail(VAD_ERR_BAD_SLOT, "Model prediction failed: slot %lld is not an open stream", (long long)s);
        if (e->stamp[(size_t)s] == e->stamp_gen)
            return e->fail(VAD_ERR_BAD_SLOT, "Model prediction failed: slot %lld appears twice in one step", (long long)s);
        e->stamp[(size_t)s] = e->stamp_gen;
    }
    return VAD_OK;
}

int launch(vad_engine *e, const vadk::StepParams &p, hipStream_t s) {
    hipError_t r = hipErrorInvalidValue;
    const bool t16 = e->d_wstream16 && (e->tile_policy == 16 || (e->tile_policy == 0 && !e->shared_gpu && (p.T == 1 || p.n <= vad_engine::T16_MAX_STREAMS)));
    if (e->version == 4 && e->d_wstream16 && e->tile_policy != 32) {
        // Silero V4: 16-stream tiles, two workgroups per CU (each fills the other's waits) - one per CU while the call has no
        // more tiles than the GPU has CUs, so that a small batch spreads out instead of pairing up
        vadk::StepParams p16 = p;
        p16.wstream = e->d_wstream16;
        p16.wstream_bytes = (uint32_t)e->wbytes16;
        std::memcpy(p16.sect, e->sect16, sizeof p16.sect);
        const int tiles16 = (p.n + 15) / 16;
        r = vadk_launch_silero_v4_t16(&p16, (!e->shared_gpu && tiles16 <= e->prop.multiProcessorCount) ? 1 : 0, s);
    } else if (e->version == 5 && t16) {
        vadk::StepParams p16 = p;
        p16.wstream = e->d_wstream16;
        p16.wstream_bytes = (uint32_t)e->wbytes16;
        std::memcpy(p16.sect, e->sect16, sizeof p16.sect);
        r = vadk_launch_silero_v5_t16(&p16, s);
    } else if (e->version == 5) r = vadk_launch_silero_v5(&p, s);
    else if (e->version == 4) r = vadk_launch_silero_v4(&p, s);
    if (r != hipSuccess) return e->hip_fail(r, "kernel launch");
    e->steps += 1;
    e->frames += (int64_t)p.n * p.T;
    return VAD_OK;
}

// shared body of vad_step / vad_step_events / vad_step_multi
int step_host(vad_engine *e, const int64_t *slots, int64_t n, int32_t T, const void *frames, int fmt, float thr,
              float *probs, uint8_t *events, int32_t *seg) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (n < 0 || T < 1 || (n > 0 && (!slots || !frames || !probs)))
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: null buffer or bad count");
    if (fmt < VAD_FMT_F32 || fmt > VAD_FMT_I16_32768)
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: unknown frame format %d", fmt);
    if (n == 0) return VAD_OK;
    if (int rc = check_call_size(e, n, T, fmt)) return rc;
    if (int rc = check_slots(e, slots, n)) return rc;
    HIP_TRY(e, hipSetDevice(e->device));
    const size_t fb = frame_bytes(e, fmt) * (size_t)n * T;
    // ---- small calls: frames + slots travel as one pinned block, probs + seg + events come back as one; one
    //      synchronisation.  (The general path below issues 2 pageable H2D copies, waits, launches, 3 D2H copies, waits.)
    {
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t o_slots = up16(fb), in_bytes = o_slots + sizeof(int32_t) * n;
        const size_t o_seg = up16(sizeof(float) * n * T), o_ev = o_seg + up16(sizeof(int32_t) * n), out_bytes = o_ev + (size_t)n * T;
        if (in_bytes <= vad_engine::SMALL_BYTES && out_bytes <= vad_engine::SMALL_BYTES) {
            // allocated on first use, each buffer on its own so that a failed attempt can be repeated
            if (!e->h_small_in) HIP_TRY(e, hipHostMalloc((void **)&e->h_small_in, vad_engine::SMALL_BYTES, hipHostMallocDefault));
            if (!e->h_small_out) HIP_TRY(e, hipHostMalloc((void **)&e->h_small_out, vad_engine::SMALL_BYTES, hipHostMallocDefault));
            if (!e->d_small_in) HIP_TRY(e, hipMalloc((void **)&e->d_small_in, vad_engine::SMALL_BYTES));
            if (!e->d_small_out) HIP_TRY(e, hipMalloc((void **)&e->d_small_out, vad_engine::SMALL_BYTES));
            std::memcpy(e->h_small_in, frames, fb);
            int32_t *hs = reinterpret_cast<int32_t *>(e->h_small_in + o_slots);
            for (int64_t i = 0; i < n; ++i) hs[i] = (int32_t)slots[i];
            HIP_TRY(e, hipMemcpyAsync(e->d_small_in, e->h_small_in, in_bytes, hipMemcpyHostToDevice, e->stream));
            vadk::StepParams p = e->base;
            p.slots = reinterpret_cast<const int32_t *>(e->d_small_in + o_slots);
            p.frames = e->d_small_in;
            p.probs = reinterpret_cast<float *>(e->d_small_out);
            p.seg_frames = reinterpret_cast<int32_t *>(e->d_small_out + o_seg);
            p.events = e->d_small_out + o_ev;
            p.n = (int32_t)n;
            p.T = T;
            p.fmt = fmt;
            p.thresh = thr;
            if (int rc = launch(e, p, e->stream)) return rc;
            HIP_TRY(e, hipMemcpyAsync(e->h_small_out, e->d_small_out, out_bytes, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(e, hipStreamSynchronize(e->stream));
            std::memcpy(probs, e->h_small_out, sizeof(float) * n * T);
            if (seg) std::memcpy(seg, e->h_small_out + o_seg, sizeof(int32_t) * n);
            if (events) std::memcpy(events, e->h_small_out + o_ev, (size_t)n * T);
            return VAD_OK;
        }
    }
    if (int rc = ensure(e, e->d_frames, e->d_frames_cap, fb)) return rc;
    if (int rc = ensure(e, e->d_probs, e->d_probs_cap, sizeof(float) * n * T)) return rc;
    if (int rc = ensure(e, e->d_events, e->d_events_cap, (size_t)n * T)) return rc;
    if (int rc = ensure(e, e->d_seg, e->d_seg_cap, sizeof(int32_t) * n)) return rc;
    if (int rc = ensure(e, e->d_slots, e->d_slots_cap, sizeof(int32_t) * n)) return rc;
    std::vector<int32_t> s32((size_t)n);
    for (int64_t i = 0; i < n; ++i) s32[(size_t)i] = (int32_t)slots[i];
    HIP_TRY(e, hipMemcpyAsync(e->d_slots, s32.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->d_frames, frames, fb, hipMemcpyHostToDevice, e->stream));
    vadk::StepParams p = e->base;
    p.slots = e->d_slots;
    p.frames = e->d_frames;
    p.probs = e->d_probs;
    p.events = e->d_events;
    p.seg_frames = e->d_seg;
    p.n = (int32_t)n;
    p.T = T;
    p.fmt = fmt;
    p.thresh = thr;
    // the pageable-memory H2D copies above must not be overwritten before they are consumed
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (int rc = launch(e, p, e->stream)) return rc;
    HIP_TRY(e, hipMemcpyAsync(probs, e->d_probs, sizeof(float) * n * T, hipMemcpyDeviceToHost, e->stream));
    if (events) HIP_TRY(e, hipMemcpyAsync(events, e->d_events, (size_t)n * T, hipMemcpyDeviceToHost, e->stream));
    if (seg) HIP_TRY(e, hipMemcpyAsync(seg, e->d_seg, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

}  // namespace

extern "C" {

const char *vad_last_create_error(void) { return g_create_error.c_str(); }
const char *vad_last_error(const vad_engine *e) {
    if (!e) return "null engine";
    std::lock_guard<std::mutex> lk(e->err_mu);  // other threads may be writing e->err: hand out a thread-local copy
    g_error_copy = e->err;
    return g_error_copy.c_str();
}

int vad_engine_create(const vad_engine_desc *desc, vad_engine **out) {
    g_create_error.clear();
    if (!desc || !out || desc->struct_size < sizeof(vad_engine_desc)) {
        g_create_error = "Failed to load model: bad vad_engine_desc";
        return VAD_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (desc->model_version != 4 && desc->model_version != 5) {
        g_create_error = "Failed to load model: model_version must be 4 or 5";
        return VAD_ERR_INVALID_ARG;
    }
    // `sample_rate` is the graph's `sr` input (core/silero_model.py:491): 16000 selects the 16 kHz sub-model, every other
    // value the graph's else-branch = the 8 kHz sub-model, which needs the blob that holds ITS weights (meta.variant = 8000).
    // V4's takes the same 512-sample frames at 8 / 24 / 48 kHz.  V5's is built for native 8 kHz audio in
    // 256-sample frames (with 512-sample frames a 3-D tensor reaches its LSTM and onnxruntime refuses): sample_rate 8000
    // creates that engine (vad_info.frame_samples = 256); 24 / 48 kHz audio has to be resampled first.
    const bool want_8k = desc->sample_rate != 16000;
    if (want_8k && desc->model_version == 5 && desc->sample_rate != 8000) {
        g_create_error = "Failed to load model: Silero V5 takes 16 kHz audio (512-sample frames) or native 8 kHz audio (256-sample frames); resample other rates first";
        return VAD_ERR_UNSUPPORTED;
    }
    if (desc->max_streams < 1) {
        g_create_error = "Failed to load model: max_streams must be >= 1";
        return VAD_ERR_INVALID_ARG;
    }
    vadk::PackedWeights pw;
    std::string perr;
    const bool ok = desc->model_version == 5 ? vadk::pack_silero_v5(desc->weights, desc->weights_len, pw, perr)
                                             : vadk::pack_silero_v4(desc->weights, desc->weights_len, pw, perr);
    if (!ok) {
        g_create_error = perr;
        return VAD_ERR_BAD_WEIGHTS;
    }
    if (want_8k != (pw.variant == 1)) {
        g_create_error = want_8k ? "Failed to load model: sample_rate selects the graph's 8 kHz sub-model but the weight blob holds the 16 kHz one"
                                 : "Failed to load model: sample_rate 16000 but the weight blob holds the graph's 8 kHz sub-model";
        return VAD_ERR_BAD_WEIGHTS;
    }
    int ndev = 0;
    hipError_t r = hipGetDeviceCount(&ndev);
    if (r != hipSuccess || ndev < 1) {
        g_create_error = "Failed to load model: no HIP device available (this engine has no CPU fallback)";
        return VAD_ERR_NO_DEVICE;
    }
    if (desc->device_id < 0 || desc->device_id >= ndev) {
        g_create_error = "Failed to load model: device_id out of range";
        return VAD_ERR_INVALID_ARG;
    }
    vad_engine *e = new vad_engine();
    e->version = desc->model_version;
    e->device = desc->device_id;
    e->max_streams = desc->max_streams;
    e->sample_rate = desc->sample_rate;
    e->shared_gpu = (desc->flags & VAD_ENGINE_SHARED_GPU) != 0;
    e->frame_samples = (desc->model_version == 5 && want_8k) ? vadk::v5::FRAME_8K : VAD_FRAME_SAMPLES;
    auto bail = [&](hipError_t hr, const char *what) {
        g_create_error = std::string("Failed to load model: ") + what + ": " + hipGetErrorString(hr);
        (void)hipGetLastError();      // consumed here: HIP keeps the failure in a process-wide slot otherwise
        vad_engine_destroy(e);
        return VAD_ERR_HIP;
    };
    if ((r = hipSetDevice(e->device)) != hipSuccess) return bail(r, "hipSetDevice");
    if ((r = hipGetDeviceProperties(&e->prop, e->device)) != hipSuccess) return bail(r, "hipGetDeviceProperties");
    if (std::strncmp(e->prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("Failed to load model: kernels are built for gfx950 only, device is ") + e->prop.gcnArchName;
        vad_engine_destroy(e);
        return VAD_ERR_NO_DEVICE;
    }
    if ((r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess) return bail(r, "hipStreamCreate");
    if ((r = hipStreamCreateWithFlags(&e->copy_in, hipStreamNonBlocking)) != hipSuccess) return bail(r, "hipStreamCreate");
    if ((r = hipStreamCreateWithFlags(&e->copy_out, hipStreamNonBlocking)) != hipSuccess) return bail(r, "hipStreamCreate");
    e->wbytes = pw.data.size() * sizeof(float);
    if ((r = weights_acquire(e->device, pw.data, &e->d_wstream)) != hipSuccess) return bail(r, "hipMalloc / hipMemcpy(weights)");
    {                                                    // every model has a 16-stream tile kernel for small calls
        vadk::PackedWeights pw16;
        const bool ok16 = desc->model_version == 4 ? vadk::pack_silero_v4_t16(desc->weights, desc->weights_len, pw16, perr)
                                                   : vadk::pack_silero_v5_t16(desc->weights, desc->weights_len, pw16, perr);
        if (!ok16) {
            g_create_error = perr;
            vad_engine_destroy(e);
            return VAD_ERR_BAD_WEIGHTS;
        }
        e->wbytes16 = pw16.data.size() * sizeof(float);
        if ((r = weights_acquire(e->device, pw16.data, &e->d_wstream16)) != hipSuccess)
            return bail(r, "hipMalloc / hipMemcpy(weights, 16-stream tiles)");
        std::memcpy(e->sect16, pw16.sect, sizeof pw16.sect);
    }
    const size_t sb = sizeof(float) * VAD_STATE_FLOATS * (size_t)e->max_streams;
    if ((r = hipMalloc((void **)&e->d_state, sb)) != hipSuccess) return bail(r, "hipMalloc(state)");
    if ((r = hipMemset(e->d_state, 0, sb)) != hipSuccess) return bail(r, "hipMemset(state)");
    if ((r = hipMalloc((void **)&e->d_sm, sizeof(vadk::SmSlot) * (size_t)e->max_streams)) != hipSuccess)
        return bail(r, "hipMalloc(sm)");
    {
        std::vector<vadk::SmSlot> init((size_t)e->max_streams, kDefaultSm);
        if ((r = hipMemcpy(e->d_sm, init.data(), sizeof(vadk::SmSlot) * init.size(), hipMemcpyHostToDevice)) != hipSuccess)
            return bail(r, "hipMemcpy(sm)");
    }
    e->base.wstream = e->d_wstream;
    e->base.wstream_bytes = (uint32_t)e->wbytes;
    std::memcpy(e->base.sect, pw.sect, sizeof pw.sect);
    e->base.variant = pw.variant;
    e->base.state = e->d_state;
    e->base.sm = e->d_sm;
    e->open.assign((size_t)e->max_streams, 0);
    e->stamp.assign((size_t)e->max_streams, 0);
    e->tick_gen.assign((size_t)e->max_streams, 0);
    e->slot_epoch.assign((size_t)e->max_streams, 0);
    e->h_start_prob.assign((size_t)e->max_streams, kDefaultSm.start_prob);
    e->free_list.reserve((size_t)e->max_streams);
    for (int64_t s = e->max_streams - 1; s >= 0; --s) e->free_list.push_back(s);
    *out = e;
    return VAD_OK;
}

void vad_engine_destroy(vad_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (hipStream_t st : {e->copy_in, e->stream, e->copy_out})
        if (st) (void)hipStreamSynchronize(st);
    weights_release(e->d_wstream);
    weights_release(e->d_wstream16);
    void *bufs[] = {e->d_state, e->d_sm, e->d_frames, e->d_probs, e->d_events, e->d_seg, e->d_slots,
                    e->d_rs_in, e->d_rs_out, e->d_small_in, e->d_small_out, e->d_ctl};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (auto &pb : e->pipe) {
        if (pb.d_frames) (void)hipFree(pb.d_frames);
        if (pb.d_io) (void)hipFree(pb.d_io);
        if (pb.h_io) (void)hipHostFree(pb.h_io);
        for (hipEvent_t ev : {pb.copied, pb.done, pb.out})
            if (ev) (void)hipEventDestroy(ev);
    }
    if (e->copy_in) (void)hipStreamDestroy(e->copy_in);
    if (e->copy_out) (void)hipStreamDestroy(e->copy_out);
    if (e->h_ctl) (void)hipHostFree(e->h_ctl);
    for (auto &bb : e->tick_buf)
        for (auto &tb : bb)
            if (tb.h) (void)hipHostFree(tb.h);
    if (e->h_tick_out) (void)hipHostFree(e->h_tick_out);
    if (e->d_tick_out) (void)hipFree(e->d_tick_out);
    if (e->d_tick_frames) (void)hipFree(e->d_tick_frames);
    if (e->h_small_in) (void)hipHostFree(e->h_small_in);
    if (e->h_small_out) (void)hipHostFree(e->h_small_out);
    for (void *b : e->host_blocks) (void)hipHostFree(b);
    for (auto &op : e->resample_ops)
        if (op.d_w) (void)hipFree(op.d_w);
    for (auto &op : e->resample_ops16)
        if (op.d_w) (void)hipFree(op.d_w);
    for (auto &c : e->rsg_cache) {
        if (c.d_tn) (void)hipFree(c.d_tn);
        if (c.d_tm) (void)hipFree(c.d_tm);
    }
    for (void *q : {e->rsf_plan.W1, e->rsf_plan.W2, e->rsf_plan.B1, e->rsf_plan.B2, e->d_rsf_a, e->d_rsf_b})
        if (q) (void)hipFree(q);
    if (e->d_rsg_partial) (void)hipFree(e->d_rsg_partial);
    if (e->d_rsg_in) (void)hipFree(e->d_rsg_in);
    if (e->d_rsg_out) (void)hipFree(e->d_rsg_out);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int vad_engine_info(const vad_engine *e, vad_info *info) {
    if (!e || !info || info->struct_size < sizeof(vad_info)) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    const uint32_t sz = info->struct_size;
    std::memset(info, 0, sizeof *info);
    info->struct_size = sz;
    info->abi_version = VAD_ABI_VERSION;
    info->model_version = e->version;
    info->device_id = e->device;
    info->max_streams = e->max_streams;
    info->open_streams = e->open_count;
    info->compute_units = e->prop.multiProcessorCount;
    info->streams_per_workgroup = vadk::MT;
    info->weight_bytes_device = (int64_t)(e->wbytes + e->wbytes16);
    info->state_bytes_device = (int64_t)(sizeof(float) * VAD_STATE_FLOATS + sizeof(vadk::SmSlot)) * e->max_streams;
    info->steps = e->steps;
    info->frames = e->frames;
    std::strncpy(info->device_name, e->prop.name, sizeof info->device_name - 1);
    std::strncpy(info->arch, e->prop.gcnArchName, sizeof info->arch - 1);
    info->frame_samples = e->frame_samples;
    info->sample_rate = e->sample_rate;
    return VAD_OK;
}

// One pinned block up (slots, optionally thresholds), ONE kernel for every listed slot, one synchronisation: open, reset and
// threshold updates cost the same for 1 slot and for 8 192 (the per-slot form was 2 round trips per slot).
static int slot_control(vad_engine *e, const int64_t *slots, int64_t n, int op, const vad_thresholds *thr, int64_t nthr) {
    if (n == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    const size_t o_thr = ((sizeof(int32_t) * (size_t)n) + 15) & ~(size_t)15;
    const size_t need = o_thr + sizeof(vad_thresholds) * (size_t)nthr;
    if (need > e->ctl_cap) {
        const size_t cap = std::max(need, (size_t)(sizeof(int32_t) + sizeof(vad_thresholds)) * (size_t)e->max_streams + 16);
        if (e->h_ctl) (void)hipHostFree(e->h_ctl);
        if (e->d_ctl) (void)hipFree(e->d_ctl);
        e->h_ctl = e->d_ctl = nullptr;
        e->ctl_cap = 0;
        HIP_TRY(e, hipHostMalloc((void **)&e->h_ctl, cap, hipHostMallocDefault));
        HIP_TRY(e, hipMalloc((void **)&e->d_ctl, cap));
        e->ctl_cap = cap;
    }
    int32_t *hs = reinterpret_cast<int32_t *>(e->h_ctl);
    for (int64_t i = 0; i < n; ++i) hs[i] = (int32_t)slots[i];
    if (nthr) std::memcpy(e->h_ctl + o_thr, thr, sizeof(vad_thresholds) * (size_t)nthr);
    {   // host mirrors used by the tick's segment assembly
        std::lock_guard<std::mutex> tl(e->tick_mu);
        for (int64_t i = 0; i < n; ++i) {
            const size_t sl = (size_t)slots[i];
            if (op & 2) e->h_start_prob[sl] = kDefaultSm.start_prob;
            if (op & 8) e->h_start_prob[sl] = thr[nthr == 1 ? 0 : i].start_probability;
            if ((op & (2 | 4)) && sl < e->seg_state.size()) e->seg_state[sl].clear(e->seg_arena);
        }
    }
    HIP_TRY(e, hipMemcpyAsync(e->d_ctl, e->h_ctl, need, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, vadk_launch_slot_control(e->d_sm, e->d_state, reinterpret_cast<const int32_t *>(e->d_ctl), (int)n, op, &kDefaultSm,
                                        reinterpret_cast<const vad_thresholds *>(e->d_ctl + o_thr), (int)nthr, e->stream));
    // callers may step on their own HIP stream next (vad_step_device): the slots must be ready when this returns
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}
enum { CTL_ZERO_STATE = 1, CTL_DEFAULT_SM = 2, CTL_RESET_DYNAMIC = 4, CTL_SET_THRESHOLDS = 8 };

int vad_stream_open(vad_engine *e, int64_t *slot) { return vad_stream_open_many(e, 1, slot); }

// everything the tick assembler holds for a slot: staged row, waiting frames, tails, segment audio (tick_mu held)
static void tick_forget(vad_engine *e, int64_t slot) {
    if (e->tick_overflow.erase(slot))
        e->tick_overflow_order.erase(std::remove(e->tick_overflow_order.begin(), e->tick_overflow_order.end(), slot), e->tick_overflow_order.end());
    e->tick_tails.erase(slot);
    if ((size_t)slot < e->seg_state.size()) e->seg_state[(size_t)slot].clear(e->seg_arena);
    if (e->tick_gen[(size_t)slot] == e->tick_generation) {
        for (auto &tb : e->tick_buf[e->tick_cur])
            for (int64_t r = 0; r < tb.count; ++r)
                if (tb.slots()[r] == (int32_t)slot) {
                    tb.drop_row(r);
                    break;
                }
        e->tick_gen[(size_t)slot] = 0;
    }
}

int vad_stream_open_many(vad_engine *e, int64_t n, int64_t *slots_out) {
    if (!e || n < 0 || (n > 0 && !slots_out)) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if ((int64_t)e->free_list.size() < n)
        return e->fail(VAD_ERR_NO_SLOT, "stream pool exhausted (%d slots, %d open, %lld requested)", e->max_streams,
                       e->open_count, (long long)n);
    for (int64_t i = 0; i < n; ++i) slots_out[i] = e->free_list[e->free_list.size() - 1 - (size_t)i];
    if (int rc = slot_control(e, slots_out, n, CTL_ZERO_STATE | CTL_DEFAULT_SM, nullptr, 0)) return rc;
    {   // `open` is read by the tick's producers under tick_mu alone: written under both locks.  A recycled slot starts with
        // nothing waiting, and a row it still has in a tick that is being run right now is recognised as stale by its epoch.
        std::lock_guard<std::mutex> tl(e->tick_mu);
        for (int64_t i = 0; i < n; ++i) {
            const int64_t sl = e->free_list.back();
            tick_forget(e, sl);
            e->slot_epoch[(size_t)sl] += 1;
            e->open[(size_t)sl] = 1;
            e->free_list.pop_back();
        }
    }
    e->open_count += (int)n;
    return VAD_OK;
}

int vad_stream_close(vad_engine *e, int64_t slot) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    {
        std::lock_guard<std::mutex> tl(e->tick_mu);
        e->open[(size_t)slot] = 0;
        tick_forget(e, slot);              // frames the client had waiting die with its stream
    }
    e->open_count -= 1;
    e->free_list.push_back(slot);
    return VAD_OK;
}

int vad_stream_reset(vad_engine *e, const int64_t *slots, int64_t n) {
    if (!e || (n > 0 && !slots) || n < 0) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (int rc = check_slots(e, slots, n)) return rc;
    // keep the slots' thresholds, reset the dynamic part + (h, c) (VADProcessor.reset, silero_model.py:951-968)
    return slot_control(e, slots, n, CTL_ZERO_STATE | CTL_RESET_DYNAMIC, nullptr, 0);
}

int vad_stream_get_state(vad_engine *e, int64_t slot, float *hc) {
    if (!e || !hc) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpyAsync(hc, e->d_state + (size_t)slot * VAD_STATE_FLOATS, sizeof(float) * VAD_STATE_FLOATS,
                              hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

int vad_stream_set_state(vad_engine *e, int64_t slot, const float *hc) {
    if (!e || !hc) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpyAsync(e->d_state + (size_t)slot * VAD_STATE_FLOATS, hc, sizeof(float) * VAD_STATE_FLOATS,
                              hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

static_assert(VAD_STREAM_SAVE_BYTES == sizeof(float) * VAD_STATE_FLOATS + sizeof(vadk::SmSlot), "save blob layout");

int vad_stream_save(vad_engine *e, int64_t slot, void *buf, int64_t cap) {
    if (!e || !buf) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    if (cap < VAD_STREAM_SAVE_BYTES) return e->fail(VAD_ERR_INVALID_ARG, "save buffer too small (%lld < %d)", (long long)cap, VAD_STREAM_SAVE_BYTES);
    HIP_TRY(e, hipSetDevice(e->device));
    char *b = static_cast<char *>(buf);
    HIP_TRY(e, hipMemcpyAsync(b, e->d_state + (size_t)slot * VAD_STATE_FLOATS, sizeof(float) * VAD_STATE_FLOATS,
                              hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(b + sizeof(float) * VAD_STATE_FLOATS, e->d_sm + slot, sizeof(vadk::SmSlot),
                              hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

int vad_stream_restore(vad_engine *e, int64_t slot, const void *buf, int64_t nbytes) {
    if (!e || !buf) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    if (nbytes != VAD_STREAM_SAVE_BYTES) return e->fail(VAD_ERR_INVALID_ARG, "not a stream save blob (%lld bytes)", (long long)nbytes);
    vadk::SmSlot s;
    memcpy(&s, static_cast<const char *>(buf) + sizeof(float) * VAD_STATE_FLOATS, sizeof s);
    if (s.start_count < 1 || s.end_count < 1 || s.start_len < 0 || s.start_len > 20 || s.end_len < 0 || s.end_len > 100)
        return e->fail(VAD_ERR_INVALID_ARG, "corrupt stream save blob");
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipMemcpyAsync(e->d_state + (size_t)slot * VAD_STATE_FLOATS, buf, sizeof(float) * VAD_STATE_FLOATS,
                              hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->d_sm + slot, &s, sizeof s, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    {   // the tick's pre-roll rule reads the host copy of vad_start_probability
        std::lock_guard<std::mutex> tl(e->tick_mu);
        e->h_start_prob[(size_t)slot] = s.start_prob;
    }
    return VAD_OK;
}

int vad_stream_set_thresholds(vad_engine *e, int64_t slot, const vad_thresholds *t) {
    return vad_stream_set_thresholds_many(e, &slot, 1, t, 1);
}

int vad_stream_set_thresholds_many(vad_engine *e, const int64_t *slots, int64_t n, const vad_thresholds *t, int64_t nt) {
    if (!e || n < 0 || (n > 0 && (!slots || !t))) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (n == 0) return VAD_OK;
    if (nt != 1 && nt != n) return e->fail(VAD_ERR_INVALID_ARG, "thresholds: pass 1 (shared) or n (one per slot) entries, got %lld for %lld slots", (long long)nt, (long long)n);
    if (int rc = check_slots(e, slots, n)) return rc;
    for (int64_t i = 0; i < nt; ++i)
        if (t[i].start_frame_count < 1 || t[i].end_frame_count < 1) return e->fail(VAD_ERR_INVALID_ARG, "frame counts must be >= 1");
    // values only: the dynamic part (counters, history) is untouched.  VADWrapper.set_thresholds resets the
    // processor afterwards (vad_wrapper.py:412-413) through vad_stream_reset.
    return slot_control(e, slots, n, CTL_SET_THRESHOLDS, t, nt);
}

int vad_step(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int frame_fmt, float denoise_thresh,
             float *probs_out) {
    return step_host(e, slots, n, 1, frames, frame_fmt, denoise_thresh, probs_out, nullptr, nullptr);
}

int vad_step_events(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int frame_fmt,
                    float denoise_thresh, float *probs_out, uint8_t *events_out, int32_t *seg_frames_out) {
    if (e && n > 0 && !events_out) {
        std::lock_guard<std::mutex> lk(e->mu);
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: events_out is null");
    }
    return step_host(e, slots, n, 1, frames, frame_fmt, denoise_thresh, probs_out, events_out, seg_frames_out);
}

int vad_step_multi(vad_engine *e, const int64_t *slots, int64_t n, int32_t T, const void *frames, int frame_fmt,
                   float denoise_thresh, float *probs_out, uint8_t *events_out) {
    return step_host(e, slots, n, T, frames, frame_fmt, denoise_thresh, probs_out, events_out, nullptr);
}

int vad_step_device(vad_engine *e, const int32_t *d_slots, int64_t n, const void *d_frames, int frame_fmt,
                    float denoise_thresh, float *d_probs, uint8_t *d_events, int32_t *d_seg_frames, void *stream) {
    return vad_step_multi_device(e, d_slots, n, 1, d_frames, frame_fmt, denoise_thresh, d_probs, d_events, d_seg_frames, stream);
}

int vad_step_multi_device(vad_engine *e, const int32_t *d_slots, int64_t n, int32_t T, const void *d_frames, int frame_fmt,
                          float denoise_thresh, float *d_probs, uint8_t *d_events, int32_t *d_seg_frames, void *stream) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (frame_fmt < VAD_FMT_F32 || frame_fmt > VAD_FMT_I16_32768)
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: unknown frame format %d", frame_fmt);
    if (int rc = check_call_size(e, n, T, frame_fmt)) return rc;
    if (n > 0 && (!d_frames || !d_probs)) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: null buffer");
    if (n == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    vadk::StepParams p = e->base;
    p.slots = d_slots;
    p.frames = d_frames;
    p.probs = d_probs;
    p.events = d_events;
    p.seg_frames = d_seg_frames;
    p.n = (int32_t)n;
    p.T = T;
    p.fmt = frame_fmt;
    p.thresh = denoise_thresh;
    return launch(e, p, stream ? static_cast<hipStream_t>(stream) : e->stream);
}

// ---- pipelined host ingest ---------------------------------------------------------------------------------------
int vad_step_submit(vad_engine *e, const int64_t *slots, int64_t n, int32_t T, const void *frames, int fmt, float thr,
                    int64_t *ticket) {
    if (!e || !ticket) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (fmt < VAD_FMT_F32 || fmt > VAD_FMT_I16_32768)
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: unknown frame format %d", fmt);
    if (int rc = check_call_size(e, n, T, fmt)) return rc;
    if (n < 1 || !slots || !frames) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: null buffer or bad count");
    vad_engine::PipeBuf &pb = e->pipe[e->next_ticket % vad_engine::PIPE_DEPTH];
    if (pb.busy)
        return e->fail(VAD_ERR_BUSY, "Model prediction failed: %d tickets are outstanding - collect ticket %lld first",
                       vad_engine::PIPE_DEPTH, (long long)pb.ticket);
    if (int rc = check_slots(e, slots, n)) return rc;
    HIP_TRY(e, hipSetDevice(e->device));
    if (!pb.copied) {
        HIP_TRY(e, hipEventCreateWithFlags(&pb.copied, hipEventDisableTiming));
        HIP_TRY(e, hipEventCreateWithFlags(&pb.done, hipEventDisableTiming));
        HIP_TRY(e, hipEventCreateWithFlags(&pb.out, hipEventDisableTiming));
    }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t fb = frame_bytes(e, fmt) * (size_t)n * T;
    const size_t o_probs = up16(sizeof(int32_t) * (size_t)n), o_seg = o_probs + up16(sizeof(float) * (size_t)n * T);
    const size_t o_ev = o_seg + up16(sizeof(int32_t) * (size_t)n), io_bytes = o_ev + up16((size_t)n * T);
    if (int rc = ensure(e, pb.d_frames, pb.d_frames_cap, fb)) return rc;
    if (io_bytes > pb.io_cap) {
        if (pb.d_io) (void)hipFree(pb.d_io);
        if (pb.h_io) (void)hipHostFree(pb.h_io);
        pb.d_io = pb.h_io = nullptr;
        pb.io_cap = 0;
        HIP_TRY(e, hipMalloc((void **)&pb.d_io, io_bytes));
        HIP_TRY(e, hipHostMalloc((void **)&pb.h_io, io_bytes, hipHostMallocDefault));
        pb.io_cap = io_bytes;
    }
    int32_t *hs = reinterpret_cast<int32_t *>(pb.h_io);
    for (int64_t i = 0; i < n; ++i) hs[i] = (int32_t)slots[i];
    // in: slots + frames on the copy-in stream (true DMA when `frames` is page-locked: vad_host_alloc)
    // from the first enqueued copy on, a failure must not leave `copy_in` reading the caller's frames (or this buffer) after
    // the call returned: the error path waits for the copy stream before it reports
    auto fail_after_copy = [&](int rc) {
        (void)hipStreamSynchronize(e->copy_in);
        (void)hipStreamSynchronize(e->stream);
        return rc;
    };
#define HIP_TRY_PIPE(call)                                                          \
    do {                                                                            \
        hipError_t _r = (call);                                                     \
        if (_r != hipSuccess) return fail_after_copy(e->hip_fail(_r, #call));       \
    } while (0)
    HIP_TRY_PIPE(hipMemcpyAsync(pb.d_io, pb.h_io, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, e->copy_in));
    HIP_TRY_PIPE(hipMemcpyAsync(pb.d_frames, frames, fb, hipMemcpyHostToDevice, e->copy_in));
    HIP_TRY_PIPE(hipEventRecord(pb.copied, e->copy_in));
    // compute: after the copy; kernels of successive tickets run in submission order on the engine's stream
    HIP_TRY_PIPE(hipStreamWaitEvent(e->stream, pb.copied, 0));
    vadk::StepParams p = e->base;
    p.slots = reinterpret_cast<const int32_t *>(pb.d_io);
    p.frames = pb.d_frames;
    p.probs = reinterpret_cast<float *>(pb.d_io + o_probs);
    p.seg_frames = reinterpret_cast<int32_t *>(pb.d_io + o_seg);
    p.events = pb.d_io + o_ev;
    p.n = (int32_t)n;
    p.T = T;
    p.fmt = fmt;
    p.thresh = thr;
    if (int rc = launch(e, p, e->stream)) return fail_after_copy(rc);
    HIP_TRY_PIPE(hipEventRecord(pb.done, e->stream));
    // out: probs | seg | events as one block on the copy-out stream
    HIP_TRY_PIPE(hipStreamWaitEvent(e->copy_out, pb.done, 0));
    HIP_TRY_PIPE(hipMemcpyAsync(pb.h_io + o_probs, pb.d_io + o_probs, io_bytes - o_probs, hipMemcpyDeviceToHost, e->copy_out));
    HIP_TRY_PIPE(hipEventRecord(pb.out, e->copy_out));
#undef HIP_TRY_PIPE
    pb.busy = true;
    pb.collecting = false;
    pb.ticket = e->next_ticket;
    pb.n = n;
    pb.T = T;
    *ticket = e->next_ticket++;
    return VAD_OK;
}

int vad_step_collect(vad_engine *e, int64_t ticket, float *probs_out, uint8_t *events_out, int32_t *seg_frames_out) {
    if (!e) return VAD_ERR_INVALID_ARG;
    hipEvent_t out_ev = nullptr;
    vad_engine::PipeBuf *pb = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (ticket < 0) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: bad ticket");
        pb = &e->pipe[ticket % vad_engine::PIPE_DEPTH];
        if (!pb->busy || pb->ticket != ticket || pb->collecting)
            return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: ticket %lld is not outstanding", (long long)ticket);
        if (!probs_out) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: probs_out is null");
        pb->collecting = true;         // a second collect of this ticket from another thread is refused, not raced
        out_ev = pb->out;
    }
    // wait WITHOUT the engine's mutex: another thread may submit the next ticket meanwhile (that is the overlap)
    hipError_t r = hipEventSynchronize(out_ev);
    std::lock_guard<std::mutex> lk(e->mu);
    pb->busy = false;
    pb->collecting = false;
    if (r != hipSuccess) return e->hip_fail(r, "hipEventSynchronize(ticket)");
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t n = (size_t)pb->n, T = (size_t)pb->T;
    const size_t o_probs = up16(sizeof(int32_t) * n), o_seg = o_probs + up16(sizeof(float) * n * T), o_ev = o_seg + up16(sizeof(int32_t) * n);
    std::memcpy(probs_out, pb->h_io + o_probs, sizeof(float) * n * T);
    if (seg_frames_out) std::memcpy(seg_frames_out, pb->h_io + o_seg, sizeof(int32_t) * n);
    if (events_out) std::memcpy(events_out, pb->h_io + o_ev, n * T);
    return VAD_OK;
}

namespace {

// chunk length that yields 512 samples at 16 kHz (AudioUtils.resample_audio: int(len * 16000 / sr), audio.py:46)
int resample_chunk_len(int sr_in) {
    switch (sr_in) {
        case 8000: return 256;
        case 24000: return 768;
        case 48000: return 1536;
        default: return 0;
    }
}

int get_resample_op(vad_engine *e, int n_in, vad_engine::ResampleOp **out, bool t16 = false) {
    auto &ops = t16 ? e->resample_ops16 : e->resample_ops;
    for (auto &op : ops)
        if (op.n_in == n_in) {
            *out = &op;
            return VAD_OK;
        }
    std::vector<float> packed;
    std::string perr;
    vad_engine::ResampleOp op;
    op.n_in = n_in;
    op.tile_blocks = t16 ? vadk::pack_resample_operator_t16(n_in, packed, &op.row128_block, perr)
                         : vadk::pack_resample_operator(n_in, packed, &op.row128_block, perr);
    if (op.tile_blocks == 0) return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: %s", perr.c_str());
    op.bytes = packed.size() * sizeof(float);
    hipError_t r = hipMalloc((void **)&op.d_w, op.bytes);
    if (r != hipSuccess) return e->hip_fail(r, "hipMalloc(resample operator)");
    r = hipMemcpy(op.d_w, packed.data(), op.bytes, hipMemcpyHostToDevice);
    if (r != hipSuccess) {
        (void)hipFree(op.d_w);
        return e->hip_fail(r, "hipMemcpy(resample operator)");
    }
    ops.push_back(op);
    *out = &ops.back();
    return VAD_OK;
}

// fills one segment descriptor (validates the chunk convention, builds / finds the operator)
int resample_segment(vad_engine *e, const float *d_in, int64_t n, int32_t n_in, int32_t sr_in, float *d_out, vadk::ResampleSeg &sg) {
    const int want = resample_chunk_len(sr_in);
    if (want == 0)
        return e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio from %dHz to 16000Hz: supported input rates are 8000, 24000, 48000", sr_in);
    if (n_in != want)
        return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio from %dHz to 16000Hz: a chunk must hold %d samples, got %d", sr_in, want, n_in);
    vad_engine::ResampleOp *op = nullptr;
    if (int rc = get_resample_op(e, n_in, &op)) return rc;
    sg.wstream = op->d_w;
    sg.wstream_bytes = (uint32_t)op->bytes;
    sg.tile_blocks = op->tile_blocks;
    sg.row128_block = op->row128_block;
    sg.in = d_in;
    sg.out = d_out;
    sg.n = (int32_t)n;
    sg.n_in = n_in;
    return VAD_OK;
}

int resample_launch(vad_engine *e, const float *d_in, int64_t n, int32_t n_in, int32_t sr_in, float *d_out, hipStream_t s) {
    vadk::ResampleParams p{};
    if (int rc = resample_segment(e, d_in, n, n_in, sr_in, d_out, p.seg[0])) return rc;
    p.nseg = 1;
    p.tile_start[0] = 0;
    p.tile_start[1] = (int32_t)((n + vadk::MT - 1) / vadk::MT);
    hipError_t r = vadk_launch_resample(&p, s);
    if (r != hipSuccess) return e->hip_fail(r, "resample kernel launch");
    return VAD_OK;
}

}  // namespace

int vad_resample(vad_engine *e, const float *in, int64_t n, int32_t n_in, int32_t sr_in, float *out) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (n < 0 || (n > 0 && (!in || !out))) return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: null buffer or bad count");
    if (n == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    const size_t ib = sizeof(float) * (size_t)n * n_in, ob = sizeof(float) * (size_t)n * VAD_FRAME_SAMPLES;
    if (resample_chunk_len(sr_in) != n_in)
        return resample_launch(e, nullptr, n, n_in, sr_in, nullptr, e->stream);   // reports the precise error
    if (int rc = ensure(e, e->d_rs_in, e->d_rs_in_cap, ib)) return rc;
    if (int rc = ensure(e, e->d_rs_out, e->d_rs_out_cap, ob)) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_rs_in, in, ib, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (int rc = resample_launch(e, e->d_rs_in, n, n_in, sr_in, e->d_rs_out, e->stream)) return rc;
    HIP_TRY(e, hipMemcpyAsync(out, e->d_rs_out, ob, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

int vad_resample_device(vad_engine *e, const float *d_in, int64_t n, int32_t n_in, int32_t sr_in, float *d_out, void *stream) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (n < 0 || (n > 0 && (!d_in || !d_out))) return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: null buffer or bad count");
    if (n == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    return resample_launch(e, d_in, n, n_in, sr_in, d_out, stream ? static_cast<hipStream_t>(stream) : e->stream);
}

int vad_resample_multi_device(vad_engine *e, int32_t nseg, const float *const *d_in, const int64_t *n, const int32_t *n_in,
                              const int32_t *sr_in, float *const *d_out, void *stream) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (nseg < 1 || nseg > vadk::RESAMPLE_MAX_SEGS || !d_in || !n || !n_in || !sr_in || !d_out)
        return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: 1..%d segments, non-null tables", vadk::RESAMPLE_MAX_SEGS);
    HIP_TRY(e, hipSetDevice(e->device));
    vadk::ResampleParams p{};
    int32_t tiles = 0;
    for (int k = 0; k < nseg; ++k) {
        if (n[k] < 0 || (n[k] > 0 && (!d_in[k] || !d_out[k])))
            return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: null buffer or bad count in segment %d", k);
        if (int rc = resample_segment(e, d_in[k], n[k], n_in[k], sr_in[k], d_out[k], p.seg[k])) return rc;
        p.tile_start[k] = tiles;
        tiles += (int32_t)((n[k] + vadk::MT - 1) / vadk::MT);
    }
    p.nseg = nseg;
    p.tile_start[nseg] = tiles;
    hipError_t r = vadk_launch_resample(&p, stream ? static_cast<hipStream_t>(stream) : e->stream);
    if (r != hipSuccess) return e->hip_fail(r, "resample kernel launch");
    return VAD_OK;
}

// ---- AudioUtils.resample_audio for any (length, rates): whole-array Fourier resampling, operator evaluated on the fly -----
namespace {

// which kernel serves a call: the direct one (every operator entry evaluated, O(n_in n_out), lowest latency) below 2^25 entries,
// the chirp-z / FFT one (O(n log n), ~100 launches: 0.2 ms at least) from there - when its lengths allow - which is where the two
// were measured to cross (profiles/r03_resample_generic.jsonl); vad_debug_resample_path pins one
bool rsf_fits(int64_t n_in, int64_t n_out) { return n_in <= vadk::RSF_MAX_LEN && n_out <= vadk::RSF_MAX_LEN; }
bool rsg_use_fft(const vad_engine *e, int64_t rows, int64_t n_in, int64_t n_out) {
    if (e->rsg_path == 1 || !rsf_fits(n_in, n_out)) return false;
    if (e->rsg_path == 2) return true;
    return (unsigned __int128)(rows ? rows : 1) * (unsigned __int128)n_in * (unsigned __int128)n_out >= ((unsigned __int128)1 << 25);
}

int rsg_check(vad_engine *e, int64_t rows, int64_t n_in, int64_t n_out) {
    if (rows < 0 || n_in < 1 || n_out < 1 || n_in > vadk::RSG_MAX_LEN || n_out > vadk::RSG_MAX_LEN)
        return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: rows >= 0 and 1 <= n_in, n_out < 2^31 (rows = %lld, n_in = %lld, n_out = %lld)",
                       (long long)rows, (long long)n_in, (long long)n_out);
    if (rsg_use_fft(e, rows, n_in, n_out)) return VAD_OK;
    const unsigned __int128 entries = (unsigned __int128)(rows ? rows : 1) * (unsigned __int128)n_in * (unsigned __int128)n_out;
    if (entries > (unsigned __int128)vadk::RSG_MAX_ENTRIES)
        return e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio: %lld arrays of %lld -> %lld samples: longer than the FFT path takes (2^25 samples) and "
                       "more than the 2^42 operator entries the direct kernel evaluates per call", (long long)rows, (long long)n_in, (long long)n_out);
    return VAD_OK;
}

// chirp-z / FFT path: tables of the last shape stay on the device; rows are processed in groups that keep a work buffer <= 1 GB
int rsf_run(vad_engine *e, const void *d_in, int in_f64, int64_t rows, int64_t n_in, int64_t n_out, float *d_out, hipStream_t s) {
    auto pow2 = [](int64_t v) { int64_t q = 1; while (q < v) q <<= 1; return q; };
    vad_engine::RsfPlan &pl = e->rsf_plan;
    const int64_t P1 = pow2(2 * n_in), P2 = pow2(2 * n_out), Pmax = std::max(P1, P2);
    const int64_t rows_per = std::max<int64_t>(1, std::min<int64_t>({rows, 65535, (1ll << 26) / Pmax}));
    const size_t need = (size_t)rows_per * (size_t)Pmax * 16;
    if (need > e->d_rsf_cap) {
        HIP_TRY(e, hipStreamSynchronize(s));
        if (e->d_rsf_a) (void)hipFree(e->d_rsf_a);
        if (e->d_rsf_b) (void)hipFree(e->d_rsf_b);
        e->d_rsf_a = e->d_rsf_b = nullptr;
        e->d_rsf_cap = 0;
        HIP_TRY(e, hipMalloc(&e->d_rsf_a, need));
        HIP_TRY(e, hipMalloc(&e->d_rsf_b, need));
        e->d_rsf_cap = need;
    }
    vadk::RsfParams p{};
    p.n_in = n_in; p.n_out = n_out; p.K = std::min(n_in, n_out) / 2;
    p.P1 = P1; p.P2 = P2; p.Pmax = Pmax;
    p.a = e->d_rsf_a; p.b = e->d_rsf_b;
    p.x_f64 = in_f64 ? 1 : 0;
    if (pl.n_in != n_in || pl.n_out != n_out) {
        HIP_TRY(e, hipStreamSynchronize(s));
        for (void **q : {&pl.W1, &pl.W2, &pl.B1, &pl.B2}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        pl.n_in = pl.n_out = 0;
        HIP_TRY(e, hipMalloc(&pl.W1, (size_t)P1 / 2 * 16));
        HIP_TRY(e, hipMalloc(&pl.W2, (size_t)P2 / 2 * 16));
        HIP_TRY(e, hipMalloc(&pl.B1, (size_t)P1 * 16));
        HIP_TRY(e, hipMalloc(&pl.B2, (size_t)P2 * 16));
        p.W1 = pl.W1; p.W2 = pl.W2; p.B1 = pl.B1; p.B2 = pl.B2;
        p.rows = 1;
        hipError_t r = vadk_rsf_build_tables(&p, s);
        if (r != hipSuccess) return e->hip_fail(r, "resample kernel launch (tables)");
        pl.n_in = n_in; pl.n_out = n_out; pl.P1 = P1; pl.P2 = P2;
    }
    p.W1 = pl.W1; p.W2 = pl.W2; p.B1 = pl.B1; p.B2 = pl.B2;
    const size_t esz = in_f64 ? 8 : 4;
    for (int64_t r0 = 0; r0 < rows; r0 += rows_per) {
        p.rows = (int32_t)std::min(rows_per, rows - r0);
        p.x = static_cast<const uint8_t *>(d_in) + (size_t)r0 * (size_t)n_in * esz;
        p.y = d_out + (size_t)r0 * (size_t)n_out;
        hipError_t r = vadk_rsf_run(&p, s);
        if (r != hipSuccess) return e->hip_fail(r, "resample kernel launch");
    }
    return VAD_OK;
}

int rsg_tables(vad_engine *e, int64_t n_in, int64_t n_out, vad_engine::RsgEntry **out) {
    for (auto &c : e->rsg_cache)
        if (c.n_in == n_in && c.n_out == n_out) {
            c.used = ++e->rsg_clock;
            *out = &c;
            return VAD_OK;
        }
    vadk::RsgTables t;
    std::string perr;
    if (!vadk::build_rsg_tables(n_in, n_out, t, perr)) return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: %s", perr.c_str());
    vad_engine::RsgEntry c;
    c.n_in = n_in; c.n_out = n_out; c.a = t.a; c.b = t.b; c.L = t.L; c.P = t.P; c.corrected = t.corrected;
    const size_t bn = t.tn.size() * sizeof(double), bm = t.tm.size() * sizeof(double);
    c.bytes = bn + bm;
    // make room: the least recently used shapes go first
    auto total = [&] { size_t b = 0; for (auto &x : e->rsg_cache) b += x.bytes; return b; };
    while (!e->rsg_cache.empty() && (e->rsg_cache.size() >= vad_engine::RSG_CACHE_ENTRIES || total() + c.bytes > vad_engine::RSG_CACHE_BYTES)) {
        auto lru = std::min_element(e->rsg_cache.begin(), e->rsg_cache.end(), [](const auto &x, const auto &y) { return x.used < y.used; });
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        (void)hipFree(lru->d_tn);
        (void)hipFree(lru->d_tm);
        e->rsg_cache.erase(lru);
    }
    hipError_t r = hipMalloc((void **)&c.d_tn, bn);
    if (r != hipSuccess) return e->hip_fail(r, "hipMalloc(resample tables)");
    r = hipMalloc((void **)&c.d_tm, bm);
    if (r == hipSuccess) r = hipMemcpy(c.d_tn, t.tn.data(), bn, hipMemcpyHostToDevice);
    if (r == hipSuccess) r = hipMemcpy(c.d_tm, t.tm.data(), bm, hipMemcpyHostToDevice);
    if (r != hipSuccess) {
        (void)hipFree(c.d_tn);
        if (c.d_tm) (void)hipFree(c.d_tm);
        return e->hip_fail(r, "hipMalloc / hipMemcpy(resample tables)");
    }
    c.used = ++e->rsg_clock;
    e->rsg_cache.push_back(c);
    *out = &e->rsg_cache.back();
    return VAD_OK;
}

// device buffers in, device buffer out; launches cover (rows chunk) x (output range) pieces of at most 2^34 entries each
int rsg_run(vad_engine *e, const void *d_in, int in_f64, int64_t rows, int64_t n_in, int64_t n_out, float *d_out, hipStream_t s) {
    if (rsg_use_fft(e, rows, n_in, n_out)) return rsf_run(e, d_in, in_f64, rows, n_in, n_out, d_out, s);
    vad_engine::RsgEntry *c = nullptr;
    if (int rc = rsg_tables(e, n_in, n_out, &c)) return rc;
    constexpr int64_t BUDGET = 1ll << 34;
    const int64_t tiles_all = (n_out + 63) / 64;
    const int64_t per_tile = n_in * 64;                              // entries of one 64-output tile of one array
    int64_t rows_per = std::max<int64_t>(1, std::min<int64_t>({rows, 65535, BUDGET / std::max<int64_t>(1, per_tile * tiles_all)}));
    int64_t tiles_per = rows_per > 1 ? tiles_all : std::max<int64_t>(1, std::min<int64_t>(tiles_all, BUDGET / per_tile));
    const size_t esz = in_f64 ? 8 : 4;
    for (int64_t r0 = 0; r0 < rows; r0 += rows_per) {
        const int64_t rc = std::min(rows_per, rows - r0);
        for (int64_t t0 = 0; t0 < tiles_all; t0 += tiles_per) {
            const int64_t tc = std::min(tiles_per, tiles_all - t0);
            vadk::RsgParams p{};
            p.x = static_cast<const uint8_t *>(d_in) + (size_t)r0 * (size_t)n_in * esz;
            p.y = d_out + (size_t)r0 * (size_t)n_out;
            p.tn = c->d_tn; p.tm = c->d_tm;
            p.n_in = n_in; p.n_out = n_out; p.a = c->a; p.b = c->b; p.L = c->L;
            p.m_begin = t0 * 64;
            p.m_end = std::min(n_out, (t0 + tc) * 64);
            p.rows = (int32_t)rc;
            // enough waves to fill the chip: slices of n per output tile, at least 64 samples each
            int64_t ns = std::max<int64_t>(1, std::min<int64_t>((4096 + tc * rc - 1) / (tc * rc), std::max<int64_t>(1, n_in / 64)));
            const int64_t sl = (n_in + ns - 1) / ns;
            ns = (n_in + sl - 1) / sl;
            p.nslice = (int32_t)ns;
            p.slice_len = (int32_t)sl;
            p.x_f64 = in_f64 ? 1 : 0;
            p.corrected = c->corrected;
            p.peak = (double)(c->P - c->corrected);
            p.inv_n_in = 1.0 / (double)n_in;
            if (int rc2 = ensure(e, e->d_rsg_partial, e->d_rsg_partial_cap, sizeof(double) * (size_t)ns * (size_t)rc * (size_t)(p.m_end - p.m_begin))) {
                return rc2;
            }
            p.partial = e->d_rsg_partial;
            hipError_t r = vadk_launch_rsg_partial(&p, s);
            if (r == hipSuccess) r = vadk_launch_rsg_finish(&p, s);
            if (r != hipSuccess) return e->hip_fail(r, "resample kernel launch");
        }
    }
    return VAD_OK;
}

}  // namespace

int vad_resample_generic(vad_engine *e, const void *in, int in_f64, int64_t rows, int64_t n_in, int64_t n_out, float *out) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (int rc = rsg_check(e, rows, n_in, n_out)) return rc;
    if (rows > 0 && (!in || !out)) return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: null buffer");
    if (rows == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    const size_t ib = (in_f64 ? 8u : 4u) * (size_t)rows * (size_t)n_in, ob = sizeof(float) * (size_t)rows * (size_t)n_out;
    HIP_TRY(e, hipStreamSynchronize(e->stream));          // the shared partial / staging buffers may still be in use by an earlier call
    if (int rc = ensure(e, e->d_rsg_in, e->d_rsg_in_cap, ib)) return rc;
    if (int rc = ensure(e, e->d_rsg_out, e->d_rsg_out_cap, ob)) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_rsg_in, in, ib, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (int rc = rsg_run(e, e->d_rsg_in, in_f64, rows, n_in, n_out, e->d_rsg_out, e->stream)) return rc;
    HIP_TRY(e, hipMemcpyAsync(out, e->d_rsg_out, ob, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

int vad_resample_generic_device(vad_engine *e, const void *d_in, int in_f64, int64_t rows, int64_t n_in, int64_t n_out, float *d_out) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (int rc = rsg_check(e, rows, n_in, n_out)) return rc;
    if (rows > 0 && (!d_in || !d_out)) return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio: null buffer");
    if (rows == 0) return VAD_OK;
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (int rc = rsg_run(e, d_in, in_f64, rows, n_in, n_out, d_out, e->stream)) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));          // synchronous: d_out is complete on return
    return VAD_OK;
}

int vad_debug_resample_path(vad_engine *e, int mode) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (mode < 0 || mode > 2) return e->fail(VAD_ERR_INVALID_ARG, "vad_debug_resample_path: 0 = by size, 1 = direct kernel, 2 = FFT path");
    e->rsg_path = mode;
    return VAD_OK;
}

int vad_debug_resample_generic_entries(int64_t n_in, int64_t n_out, int64_t m0, int64_t m1, double *R, size_t r_doubles) {
    g_create_error.clear();
    if (m0 < 0 || m1 < m0 || m1 > n_out || !R || r_doubles < (size_t)(m1 - m0) * (size_t)n_in) {
        g_create_error = "vad_debug_resample_generic_entries: 0 <= m0 <= m1 <= n_out and R must hold (m1 - m0) * n_in doubles";
        return VAD_ERR_INVALID_ARG;
    }
    vadk::RsgTables t;
    std::string perr;
    if (!vadk::build_rsg_tables(n_in, n_out, t, perr)) {
        g_create_error = "vad_debug_resample_generic_entries: " + perr;
        return VAD_ERR_INVALID_ARG;
    }
    for (int64_t m = m0; m < m1; ++m)
        for (int64_t n = 0; n < n_in; ++n) R[(size_t)(m - m0) * (size_t)n_in + (size_t)n] = vadk::rsg_entry(t, m, n);
    return VAD_OK;
}

// ---- one tick for streams at other input rates: resample on the GPU -> model step, chained on the device -----------------
namespace {
int step_rates_enqueue(vad_engine *e, int32_t nseg, const float *const *d_in, const int64_t *n, const int32_t *sr_in,
                       const int32_t *d_slots, float thr, float *d_probs, uint8_t *d_events, int32_t *d_seg, hipStream_t s) {
    if (e->frame_samples != VAD_FRAME_SAMPLES)
        return e->fail(VAD_ERR_UNSUPPORTED, "Model prediction failed: this engine runs an 8 kHz sub-model on %d-sample frames; "
                       "resampled streams need the 16 kHz one", e->frame_samples);
    int64_t total = 0;
    for (int k = 0; k < nseg; ++k) {
        if (n[k] < 0 || (n[k] > 0 && !d_in[k])) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: null buffer or bad count in segment %d", k);
        total += n[k];
    }
    if (int rc = check_call_size(e, total, 1, VAD_FMT_F32)) return rc;
    if (total == 0) return VAD_OK;
    if (!d_probs) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: null buffer");
    // Silero V5, at most 4 096 streams: ONE launch - every 16-stream tile resamples its own chunks into LDS and steps the
    // model from there (silero_v5_t16.hip, RS instantiation); the 16 kHz frames never exist in HBM
    // The tiles walk the segments laid end to end, so a tile may hold the tail of one input rate and the head of the next
    // (it resamples the two parts one after the other): no padding per segment, 4 096 streams in uneven thirds are 256 tiles.
    // The walk order pairs the most expensive prologue (48 kHz, ~20 us) with the cheapest (8 kHz, ~2 us) in the tiles that
    // straddle a boundary: 48 k | 8 k | 24 k | 16 k.
    const int64_t tiles16 = (total + 15) / 16;
    if (e->version == 5 && e->d_wstream16 && !e->shared_gpu && e->tile_policy != 32 && nseg <= vadk::RATE_MAX_SEGS && e->rates_fused &&
        tiles16 <= e->prop.multiProcessorCount) {      // at most one tile per CU: a second round of tiles would cost a whole tile time
        vadk::RateParams rp{};
        int32_t stream0[vadk::RATE_MAX_SEGS], order[vadk::RATE_MAX_SEGS];
        int32_t at = 0, ns = 0;
        for (int k = 0; k < nseg; ++k) {
            stream0[k] = at;
            at += (int32_t)n[k];
            if (n[k] > 0) order[ns++] = k;
        }
        auto rank = [&](int k) { return sr_in[k] == 48000 ? 0 : sr_in[k] == 8000 ? 1 : sr_in[k] == 24000 ? 2 : 3; };
        std::stable_sort(order, order + ns, [&](int a, int b) { return rank(a) < rank(b); });
        int32_t vstart = 0;
        for (int i = 0; i < ns; ++i) {
            const int k = order[i];
            vadk::RateSeg &sg = rp.seg[i];
            if (sr_in[k] == 16000) {
                sg.wstream = nullptr;
                sg.n_in = VAD_FRAME_SAMPLES;
            } else {
                const int want = resample_chunk_len(sr_in[k]);
                if (want == 0)
                    return e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio from %dHz to 16000Hz: supported input rates are 8000, 16000, 24000, 48000", sr_in[k]);
                vad_engine::ResampleOp *op = nullptr;
                if (int rc = get_resample_op(e, want, &op, true)) return rc;
                sg.wstream = op->d_w;
                sg.wstream_bytes = (uint32_t)op->bytes;
                sg.wave_blocks = op->tile_blocks;
                sg.row128_block = op->row128_block;
                sg.n_in = want;
            }
            sg.in = d_in[k];
            sg.n = (int32_t)n[k];
            sg.stream0 = stream0[k];
            sg.vstart = vstart;
            vstart += (int32_t)n[k];
        }
        rp.nseg = ns;
        rp.total = vstart;
        vadk::StepParams p = e->base;
        p.wstream = e->d_wstream16;
        p.wstream_bytes = (uint32_t)e->wbytes16;
        std::memcpy(p.sect, e->sect16, sizeof p.sect);
        p.slots = d_slots;
        p.frames = nullptr;
        p.probs = d_probs;
        p.events = d_events;
        p.seg_frames = d_seg;
        p.n = (int32_t)total;
        p.T = 1;
        p.fmt = VAD_FMT_F32;
        p.thresh = thr;
        hipError_t r = vadk_launch_silero_v5_t16_rates(&p, &rp, s);
        if (r != hipSuccess) return e->hip_fail(r, "kernel launch");
        e->steps += 1;
        e->frames += total;
        return VAD_OK;
    }
    // the 16 kHz frames of the tick: [total][512] f32, engine-owned, written by the resampler and read by the model launch
    // right behind it on the same HIP stream - they never leave the GPU
    if (int rc = ensure(e, e->d_rs_out, e->d_rs_out_cap, sizeof(float) * VAD_FRAME_SAMPLES * (size_t)total)) return rc;
    vadk::ResampleParams rp{};
    int32_t tiles = 0, nrs = 0;
    int64_t row = 0;
    for (int k = 0; k < nseg; ++k) {
        float *dst = e->d_rs_out + (size_t)row * VAD_FRAME_SAMPLES;
        row += n[k];
        if (n[k] == 0) continue;
        if (sr_in[k] == 16000) {      // already 16 kHz (AudioUtils.resample_audio returns its input: utils/audio.py:39-40)
            HIP_TRY(e, hipMemcpyAsync(dst, d_in[k], sizeof(float) * VAD_FRAME_SAMPLES * (size_t)n[k], hipMemcpyDeviceToDevice, s));
            continue;
        }
        if (nrs == vadk::RESAMPLE_MAX_SEGS) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: at most %d resampled segments per call", vadk::RESAMPLE_MAX_SEGS);
        if (int rc = resample_segment(e, d_in[k], n[k], resample_chunk_len(sr_in[k]), sr_in[k], dst, rp.seg[nrs])) return rc;
        rp.tile_start[nrs++] = tiles;
        tiles += (int32_t)((n[k] + vadk::MT - 1) / vadk::MT);
    }
    if (nrs) {
        rp.nseg = nrs;
        rp.tile_start[nrs] = tiles;
        hipError_t r = vadk_launch_resample(&rp, s);
        if (r != hipSuccess) return e->hip_fail(r, "resample kernel launch");
    }
    vadk::StepParams p = e->base;
    p.slots = d_slots;
    p.frames = e->d_rs_out;
    p.probs = d_probs;
    p.events = d_events;
    p.seg_frames = d_seg;
    p.n = (int32_t)total;
    p.T = 1;
    p.fmt = VAD_FMT_F32;
    p.thresh = thr;
    return launch(e, p, s);
}
}  // namespace

int vad_step_rates_device(vad_engine *e, int32_t nseg, const float *const *d_in, const int64_t *n, const int32_t *sr_in,
                          const int32_t *d_slots, float denoise_thresh, float *d_probs, uint8_t *d_events, int32_t *d_seg_frames,
                          void *stream) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (nseg < 1 || nseg > 8 || !d_in || !n || !sr_in) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: 1..8 segments, non-null tables");
    HIP_TRY(e, hipSetDevice(e->device));
    return step_rates_enqueue(e, nseg, d_in, n, sr_in, d_slots, denoise_thresh, d_probs, d_events, d_seg_frames,
                              stream ? static_cast<hipStream_t>(stream) : e->stream);
}

int vad_step_rates(vad_engine *e, int32_t nseg, const float *const *in, const int64_t *n, const int32_t *sr_in, const int64_t *slots,
                   float denoise_thresh, float *probs_out, uint8_t *events_out, int32_t *seg_frames_out) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (nseg < 1 || nseg > 8 || !in || !n || !sr_in || !slots || !probs_out)
        return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: 1..8 segments, non-null tables");
    int64_t total = 0;
    size_t in_floats = 0;
    for (int k = 0; k < nseg; ++k) {
        const int len = sr_in[k] == 16000 ? VAD_FRAME_SAMPLES : resample_chunk_len(sr_in[k]);
        if (len == 0)
            return e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio from %dHz to 16000Hz: supported input rates are 8000, 16000, 24000, 48000", sr_in[k]);
        if (n[k] < 0 || (n[k] > 0 && !in[k])) return e->fail(VAD_ERR_INVALID_ARG, "Model prediction failed: null buffer or bad count in segment %d", k);
        total += n[k];
        in_floats += (size_t)n[k] * len;
    }
    if (int rc = check_call_size(e, total, 1, VAD_FMT_F32)) return rc;
    if (total == 0) return VAD_OK;
    if (int rc = check_slots(e, slots, total)) return rc;
    HIP_TRY(e, hipSetDevice(e->device));
    if (int rc = ensure(e, e->d_rs_in, e->d_rs_in_cap, sizeof(float) * in_floats)) return rc;
    if (int rc = ensure(e, e->d_probs, e->d_probs_cap, sizeof(float) * total)) return rc;
    if (int rc = ensure(e, e->d_events, e->d_events_cap, (size_t)total)) return rc;
    if (int rc = ensure(e, e->d_seg, e->d_seg_cap, sizeof(int32_t) * total)) return rc;
    if (int rc = ensure(e, e->d_slots, e->d_slots_cap, sizeof(int32_t) * total)) return rc;
    std::vector<int32_t> s32((size_t)total);
    for (int64_t i = 0; i < total; ++i) s32[(size_t)i] = (int32_t)slots[i];
    HIP_TRY(e, hipMemcpyAsync(e->d_slots, s32.data(), sizeof(int32_t) * total, hipMemcpyHostToDevice, e->stream));
    const float *d_in[8];
    size_t off = 0;
    for (int k = 0; k < nseg; ++k) {
        const size_t len = sr_in[k] == 16000 ? VAD_FRAME_SAMPLES : (size_t)resample_chunk_len(sr_in[k]);
        d_in[k] = e->d_rs_in + off;
        if (n[k]) HIP_TRY(e, hipMemcpyAsync(e->d_rs_in + off, in[k], sizeof(float) * len * (size_t)n[k], hipMemcpyHostToDevice, e->stream));
        off += len * (size_t)n[k];
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));       // pageable sources must not change under the copies
    if (int rc = step_rates_enqueue(e, nseg, d_in, n, sr_in, e->d_slots, denoise_thresh, e->d_probs, e->d_events, e->d_seg, e->stream)) return rc;
    HIP_TRY(e, hipMemcpyAsync(probs_out, e->d_probs, sizeof(float) * total, hipMemcpyDeviceToHost, e->stream));
    if (events_out) HIP_TRY(e, hipMemcpyAsync(events_out, e->d_events, (size_t)total, hipMemcpyDeviceToHost, e->stream));
    if (seg_frames_out) HIP_TRY(e, hipMemcpyAsync(seg_frames_out, e->d_seg, sizeof(int32_t) * total, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

// ---- tick assembler: the multi-stream caller's side of vad_step_events, in C ---------------------------------------------
namespace {
// one frame of `slot` into the staging of the coming tick (tick_mu held).  With `defer` the row is assigned and its address handed
// back in defer->dst, the samples are NOT copied: the caller carries that out later (a batch converting int16 chunks on the copy
// crew); `pending` is then the caller's plan so far - it points into this buffer and is carried out before the buffer moves.
int tick_place(vad_engine *e, int64_t slot, const void *samples, int32_t nsamples, int group, vad_engine::SegCopy *defer = nullptr,
               std::vector<vad_engine::SegCopy> *pending = nullptr) {
    vad_engine::TickBuf &tb = e->tick_buf[e->tick_cur][group];
    const size_t ss = vad_engine::tick_sample_bytes(group);
    const int flen = vad_engine::tick_group_len(group, e->frame_samples);
    const size_t rb = ss * (size_t)flen;
    if (tb.count == tb.cap) {
        const int64_t cap = std::min<int64_t>(e->max_streams, std::max<int64_t>(256, 2 * tb.cap));
        if (cap <= tb.cap) return e->fail(VAD_ERR_INVALID_ARG, "tick: more pending frames than slots");
        if (pending && !pending->empty()) {
            e->copy_crew.run(pending->data(), pending->size());
            pending->clear();
        }
        uint8_t *nh = nullptr;
        hipError_t r = hipSetDevice(e->device);
        if (r == hipSuccess) r = hipHostMalloc((void **)&nh, (size_t)cap * (rb + 3 * sizeof(int32_t)), hipHostMallocDefault);
        if (r != hipSuccess) return e->hip_fail(r, "hipHostMalloc(tick staging)");
        if (tb.count) {
            std::memcpy(nh, tb.h, (size_t)tb.count * rb);
            std::memcpy(nh + (size_t)cap * rb, tb.slots(), sizeof(int32_t) * (size_t)tb.count);
            std::memcpy(nh + (size_t)cap * (rb + sizeof(int32_t)), tb.lens(), sizeof(int32_t) * (size_t)tb.count);
            std::memcpy(nh + (size_t)cap * (rb + 2 * sizeof(int32_t)), tb.epochs(), sizeof(int32_t) * (size_t)tb.count);
        }
        if (tb.h) (void)hipHostFree(tb.h);
        tb.h = nh;
        tb.cap = cap;
        tb.row_bytes = rb;
    }
    // SileroVADModel._prepare_audio_input (core/silero_model.py:464-468): right-zero-pad short frames, truncate long ones
    const size_t take = std::min<size_t>((size_t)nsamples, (size_t)flen) * ss;
    uint8_t *dst = tb.row(tb.count);
    if (defer) defer->dst = dst;
    else std::memcpy(dst, samples, take);
    if (take < rb) std::memset(dst + take, 0, rb - take);
    if (e->tick_segments && nsamples > flen) {                  // the model sees the head; a segment keeps the whole frame
        const uint8_t *src = static_cast<const uint8_t *>(samples);
        e->tick_tails[slot].emplace_back(src + take, src + ss * (size_t)nsamples);
    }
    tb.slots()[tb.count] = (int32_t)slot;
    tb.lens()[tb.count] = nsamples;
    tb.epochs()[tb.count] = e->slot_epoch[(size_t)slot];
    tb.count += 1;
    e->tick_gen[(size_t)slot] = e->tick_generation;
    return VAD_OK;
}

// a frame at the engine's own rate (tick_mu held, arguments checked)
int tick_push_locked(vad_engine *e, int64_t slot, const void *samples, int32_t nsamples, int frame_fmt, int group) {
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    if (e->tick_gen[(size_t)slot] != e->tick_generation) return tick_place(e, slot, samples, nsamples, group);
    // the slot already has its frame of the coming tick: later frames wait their turn (one per tick, submission order)
    auto it = e->tick_overflow.find(slot);
    if (it == e->tick_overflow.end()) {
        it = e->tick_overflow.emplace(slot, std::deque<vad_engine::TickPending>()).first;
        e->tick_overflow_order.push_back(slot);
    }
    auto &q = it->second;
    if (q.size() >= 256) return e->fail(VAD_ERR_BUSY, "tick: slot %lld has 256 frames waiting - is vad_tick_run being called?", (long long)slot);
    const size_t ss = vad_engine::tick_sample_bytes(group);
    const int flen = vad_engine::tick_group_len(group, e->frame_samples);
    const int32_t keep = e->tick_segments ? nsamples : std::min<int32_t>(nsamples, flen);
    const uint8_t *src = static_cast<const uint8_t *>(samples);
    q.push_back(vad_engine::TickPending{std::vector<uint8_t>(src, src + ss * (size_t)keep), nsamples, group});
    return VAD_OK;
}
}  // namespace

int vad_tick_push(vad_engine *e, int64_t slot, const void *samples, int32_t nsamples, int frame_fmt, int gate_on) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (!samples || nsamples < 1 || frame_fmt < VAD_FMT_F32 || frame_fmt > VAD_FMT_I16_32768)
        return e->fail(VAD_ERR_INVALID_ARG, "tick: null frame, empty frame or unknown format");
    return tick_push_locked(e, slot, samples, nsamples, frame_fmt, frame_fmt * 2 + (gate_on ? 1 : 0));
}

int vad_tick_push_rate(vad_engine *e, int64_t slot, const void *samples, int32_t nsamples, int frame_fmt, int gate_on, int32_t sr_in) {
    if (!e) return VAD_ERR_INVALID_ARG;
    if (sr_in == e->sample_rate) return vad_tick_push(e, slot, samples, nsamples, frame_fmt, gate_on);
    if (!samples || nsamples < 1 || frame_fmt < VAD_FMT_F32 || frame_fmt > VAD_FMT_I16_32768) {
        std::lock_guard<std::mutex> lk(e->tick_mu);
        return e->fail(VAD_ERR_INVALID_ARG, "tick: null frame, empty frame or unknown format");
    }
    if (e->frame_samples != VAD_FRAME_SAMPLES || e->sample_rate != 16000) {
        std::lock_guard<std::mutex> lk(e->tick_mu);
        return e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio: resampled streams need a 16 kHz engine");
    }
    const int ri = sr_in == 8000 ? 0 : sr_in == 24000 ? 1 : sr_in == 48000 ? 2 : -1;
    const int group = 6 + 3 * (gate_on ? 1 : 0) + (ri < 0 ? 0 : ri);
    const int want = vad_engine::tick_group_len(group, e->frame_samples);
    if (ri < 0 || nsamples != want) {
        std::lock_guard<std::mutex> lk(e->tick_mu);
        if (ri < 0)
            return e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio from %dHz to 16000Hz: supported input rates are 8000, 24000, 48000", sr_in);
        return e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio from %dHz to 16000Hz: a chunk must hold %d samples, got %d", sr_in, want, nsamples);
    }
    // staged as float32 (the resampler's input type): int16 wire frames are scaled here - before the lock - with numpy's true division
    thread_local std::vector<float> cvt;
    const float *src = static_cast<const float *>(samples);
    if (frame_fmt != VAD_FMT_F32) {
        cvt.resize((size_t)nsamples);
        const float sc = frame_fmt == VAD_FMT_I16_32767 ? 32767.0f : 32768.0f;
        const int16_t *q = static_cast<const int16_t *>(samples);
        for (int32_t k = 0; k < nsamples; ++k) cvt[(size_t)k] = (float)q[k] / sc;
        src = cvt.data();
    }
    std::lock_guard<std::mutex> lk(e->tick_mu);
    return tick_push_locked(e, slot, src, nsamples, VAD_FMT_F32, group);
}

int vad_tick_enable_segments(vad_engine *e, int on) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    e->tick_segments = on != 0;
    if (e->tick_segments) {
        if (e->seg_state.size() < (size_t)e->max_streams) e->seg_state.resize((size_t)e->max_streams);
        // one block per stream up to 64 MB, touched now; the arena grows by 8 MB chunks after that
        try {
            e->seg_arena.reserve_blocks(std::min<size_t>((size_t)e->max_streams, 2048));
        } catch (const std::bad_alloc &) {
            return e->fail(VAD_ERR_INVALID_ARG, "tick: out of host memory for the segment arena");
        }
        if (e->copy_crew.th.empty()) {
            int want = 3;
            if (const char *v = std::getenv("VAD_TICK_COPY_THREADS")) want = std::max(0, std::min(16, std::atoi(v)));
            const unsigned hw = std::thread::hardware_concurrency();
            if (hw && (int)hw / 2 < want + 1) want = std::max(0, (int)hw / 2 - 1);
            e->copy_crew.start(want);
        }
    }
    return VAD_OK;
}

int vad_tick_take_segment(vad_engine *e, int64_t slot, float *out, int64_t cap, int64_t *nsamples) {
    if (!e || !nsamples) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (slot < 0 || (size_t)slot >= e->seg_state.size()) return e->fail(VAD_ERR_BAD_SLOT, "slot %lld has no segment state", (long long)slot);
    vad_engine::SegAudio &d = e->seg_state[(size_t)slot].done;
    *nsamples = d.samples;
    if (!out) return VAD_OK;                       // size query
    if (cap < d.samples) return e->fail(VAD_ERR_INVALID_ARG, "segment buffer too small (%lld < %lld samples)", (long long)cap, (long long)d.samples);
    d.to_float(out);
    d.clear(e->seg_arena);
    return VAD_OK;
}

int vad_tick_push_many(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int32_t nsamples, int frame_fmt, int gate_on) {
    if (!e || n < 0 || (n > 0 && (!slots || !frames))) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);             // one lock for the batch
    if (n > 0 && (nsamples < 1 || frame_fmt < VAD_FMT_F32 || frame_fmt > VAD_FMT_I16_32768))
        return e->fail(VAD_ERR_INVALID_ARG, "tick: empty frame or unknown format");
    const size_t stride = (frame_fmt == VAD_FMT_F32 ? 4 : 2) * (size_t)(nsamples > 0 ? nsamples : 0);
    const int group = frame_fmt * 2 + (gate_on ? 1 : 0);
    for (int64_t i = 0; i < n; ++i)
        if (int rc = tick_push_locked(e, slots[i], static_cast<const uint8_t *>(frames) + (size_t)i * stride, nsamples, frame_fmt, group)) return rc;
    return VAD_OK;
}

int vad_tick_push_status(vad_engine *e, const int64_t *slots, int64_t n, const void *frames, int32_t nsamples, int frame_fmt, int gate_on,
                         int32_t *status) {
    if (!e || n < 0 || (n > 0 && (!slots || !frames || !status))) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (n > 0 && (nsamples < 1 || frame_fmt < VAD_FMT_F32 || frame_fmt > VAD_FMT_I16_32768))
        return e->fail(VAD_ERR_INVALID_ARG, "tick: empty frame or unknown format");
    const size_t stride = (frame_fmt == VAD_FMT_F32 ? 4 : 2) * (size_t)(nsamples > 0 ? nsamples : 0);
    const int group = frame_fmt * 2 + (gate_on ? 1 : 0);
    int first = VAD_OK;
    for (int64_t i = 0; i < n; ++i) {           // every frame is tried: one full queue or one closed stream does not hold the others back
        status[i] = tick_push_locked(e, slots[i], static_cast<const uint8_t *>(frames) + (size_t)i * stride, nsamples, frame_fmt, group);
        if (status[i] != VAD_OK && first == VAD_OK) first = status[i];
    }
    return first;
}

int vad_tick_push_gather(vad_engine *e, const int64_t *slots, int64_t n, const void *const *frames, int32_t nsamples, int frame_fmt,
                         int gate_on, int32_t *status) {
    if (!e || n < 0 || (n > 0 && (!slots || !frames || !status))) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (n > 0 && (nsamples < 1 || frame_fmt < VAD_FMT_F32 || frame_fmt > VAD_FMT_I16_32768))
        return e->fail(VAD_ERR_INVALID_ARG, "tick: empty frame or unknown format");
    const int group = frame_fmt * 2 + (gate_on ? 1 : 0);
    int first = VAD_OK;
    for (int64_t i = 0; i < n; ++i) {
        status[i] = frames[i] ? tick_push_locked(e, slots[i], frames[i], nsamples, frame_fmt, group)
                              : e->fail(VAD_ERR_INVALID_ARG, "tick: null frame");
        if (status[i] != VAD_OK && first == VAD_OK) first = status[i];
    }
    return first;
}

int vad_tick_push_rate_gather(vad_engine *e, const int64_t *slots, int64_t n, const void *const *frames, int32_t nsamples, int frame_fmt,
                              int gate_on, int32_t sr_in, int32_t *status) {
    if (!e || n < 0 || (n > 0 && (!slots || !frames || !status))) return VAD_ERR_INVALID_ARG;
    if (sr_in == e->sample_rate) return vad_tick_push_gather(e, slots, n, frames, nsamples, frame_fmt, gate_on, status);
    std::lock_guard<std::mutex> lk(e->tick_mu);             // one lock for the batch
    auto all = [&](int rc) {
        for (int64_t i = 0; i < n; ++i) status[i] = rc;
        return rc;
    };
    if (n == 0) return VAD_OK;
    if (nsamples < 1 || frame_fmt < VAD_FMT_F32 || frame_fmt > VAD_FMT_I16_32768)
        return all(e->fail(VAD_ERR_INVALID_ARG, "tick: empty frame or unknown format"));
    if (e->frame_samples != VAD_FRAME_SAMPLES || e->sample_rate != 16000)
        return all(e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio: resampled streams need a 16 kHz engine"));
    const int ri = sr_in == 8000 ? 0 : sr_in == 24000 ? 1 : sr_in == 48000 ? 2 : -1;
    if (ri < 0)
        return all(e->fail(VAD_ERR_UNSUPPORTED, "Failed to resample audio from %dHz to 16000Hz: supported input rates are 8000, 24000, 48000", sr_in));
    const int group = 6 + 3 * (gate_on ? 1 : 0) + ri;
    const int want = vad_engine::tick_group_len(group, e->frame_samples);
    if (nsamples != want)
        return all(e->fail(VAD_ERR_INVALID_ARG, "Failed to resample audio from %dHz to 16000Hz: a chunk must hold %d samples, got %d", sr_in, want, nsamples));
    // rows are assigned here, under the lock; the chunks themselves - int16 -> float32 is most of this call's time - are converted
    // into their rows by the copy crew once the batch is placed.  A chunk that has to WAIT (its stream already has one in this
    // tick) is converted on the spot into the stream's queue.
    std::vector<float> cvt;
    std::vector<vad_engine::SegCopy> &plan = e->push_copies;
    plan.clear();
    const uint32_t kind = frame_fmt == VAD_FMT_F32 ? 0u : frame_fmt == VAD_FMT_I16_32767 ? 1u : 2u;
    const uint32_t src_bytes = (uint32_t)nsamples * (kind ? 2u : 4u);
    int first = VAD_OK;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t slot = slots[i];
        if (!frames[i]) {
            status[i] = e->fail(VAD_ERR_INVALID_ARG, "tick: null frame");
        } else if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot]) {
            status[i] = e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
        } else if (e->tick_gen[(size_t)slot] != e->tick_generation) {
            vad_engine::SegCopy cp{nullptr, static_cast<const uint8_t *>(frames[i]), src_bytes, kind};
            status[i] = tick_place(e, slot, frames[i], nsamples, group, &cp, &plan);
            if (status[i] == VAD_OK) plan.push_back(cp);
        } else {
            const float *src = static_cast<const float *>(frames[i]);
            if (kind) {
                cvt.resize((size_t)nsamples);
                vad_engine::SegCopy{reinterpret_cast<uint8_t *>(cvt.data()), static_cast<const uint8_t *>(frames[i]), src_bytes, kind}.carry_out();
                src = cvt.data();
            }
            status[i] = tick_push_locked(e, slot, src, nsamples, VAD_FMT_F32, group);
        }
        if (status[i] != VAD_OK && first == VAD_OK) first = status[i];
    }
    e->copy_crew.run(plan.data(), plan.size());
    plan.clear();
    return first;
}

int vad_tick_cancel(vad_engine *e, int64_t slot) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (slot < 0 || slot >= e->max_streams) return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is out of range", (long long)slot);
    tick_forget(e, slot);
    return VAD_OK;
}

int vad_tick_pending(vad_engine *e, int64_t slot, int64_t *frames) {
    if (!e || !frames) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (slot < 0 || slot >= e->max_streams) return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is out of range", (long long)slot);
    auto it = e->tick_overflow.find(slot);
    *frames = (e->tick_gen[(size_t)slot] == e->tick_generation ? 1 : 0) + (it == e->tick_overflow.end() ? 0 : (int64_t)it->second.size());
    return VAD_OK;
}

// ---- a stream's segment audio as bytes: what moves with vad_stream_save when a session changes engines ------------------------
namespace {
struct SegBlobHeader { uint32_t magic, active; uint32_t nruns[3]; uint32_t pad; uint64_t bytes[3]; };
constexpr uint32_t SEG_BLOB_MAGIC = 0x31474553u;       // "SEG1"
}  // namespace

int vad_tick_segment_save(vad_engine *e, int64_t slot, void *buf, int64_t cap, int64_t *nbytes) {
    if (!e || !nbytes) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (slot < 0 || slot >= e->max_streams) return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is out of range", (long long)slot);
    static const vad_engine::SegState none;
    const vad_engine::SegState &st = (size_t)slot < e->seg_state.size() ? e->seg_state[(size_t)slot] : none;
    const vad_engine::SegAudio *parts[3] = {&st.pre, &st.seg, &st.done};
    SegBlobHeader h{};
    h.magic = SEG_BLOB_MAGIC;
    h.active = st.active ? 1u : 0u;
    size_t total = sizeof h;
    for (int k = 0; k < 3; ++k) {
        h.nruns[k] = (uint32_t)parts[k]->runs.size();
        h.bytes[k] = parts[k]->stored_bytes();
        total += parts[k]->runs.size() * sizeof(vad_engine::SegAudio::Run) + (size_t)h.bytes[k];
    }
    *nbytes = (int64_t)total;
    if (!buf) return VAD_OK;                       // size query
    if (cap < (int64_t)total) return e->fail(VAD_ERR_INVALID_ARG, "segment save buffer too small (%lld < %lld bytes)", (long long)cap, (long long)total);
    uint8_t *o = static_cast<uint8_t *>(buf);
    std::memcpy(o, &h, sizeof h);
    o += sizeof h;
    for (int k = 0; k < 3; ++k) {
        if (!parts[k]->runs.empty()) std::memcpy(o, parts[k]->runs.data(), parts[k]->runs.size() * sizeof(vad_engine::SegAudio::Run));
        o += parts[k]->runs.size() * sizeof(vad_engine::SegAudio::Run);
        parts[k]->for_each_piece([&](const vad_engine::SegAudio::Run &r, const uint8_t *src, size_t cnt) {
            const size_t b = cnt * (vad_engine::SegAudio::is_i16(r.group) ? 2 : 4);
            std::memcpy(o, src, b);
            o += b;
        });
    }
    return VAD_OK;
}

int vad_tick_segment_restore(vad_engine *e, int64_t slot, const void *buf, int64_t nbytes) {
    if (!e || !buf) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    SegBlobHeader h;
    if (nbytes < (int64_t)sizeof h) return e->fail(VAD_ERR_INVALID_ARG, "not a segment save blob (%lld bytes)", (long long)nbytes);
    std::memcpy(&h, buf, sizeof h);
    size_t total = sizeof h;
    for (int k = 0; k < 3; ++k) total += (size_t)h.nruns[k] * sizeof(vad_engine::SegAudio::Run) + (size_t)h.bytes[k];
    if (h.magic != SEG_BLOB_MAGIC || h.active > 1 || (int64_t)total != nbytes) return e->fail(VAD_ERR_INVALID_ARG, "corrupt segment save blob");
    // check the runs against the byte counts before touching the slot
    const uint8_t *p = static_cast<const uint8_t *>(buf) + sizeof h;
    for (int k = 0; k < 3; ++k) {
        size_t want = 0;
        for (uint32_t r = 0; r < h.nruns[k]; ++r) {
            vad_engine::SegAudio::Run run;
            std::memcpy(&run, p + (size_t)r * sizeof run, sizeof run);
            if (run.group < 0 || run.group >= vad_engine::TICK_GROUPS || run.samples < 1) return e->fail(VAD_ERR_INVALID_ARG, "corrupt segment save blob");
            want += (size_t)run.samples * (vad_engine::SegAudio::is_i16(run.group) ? 2 : 4);
        }
        if (want != (size_t)h.bytes[k]) return e->fail(VAD_ERR_INVALID_ARG, "corrupt segment save blob");
        p += (size_t)h.nruns[k] * sizeof(vad_engine::SegAudio::Run) + (size_t)h.bytes[k];
    }
    if (e->seg_state.size() < (size_t)e->max_streams) e->seg_state.resize((size_t)e->max_streams);
    vad_engine::SegState &st = e->seg_state[(size_t)slot];
    st.clear(e->seg_arena);
    st.active = h.active != 0;
    vad_engine::SegAudio *parts[3] = {&st.pre, &st.seg, &st.done};
    p = static_cast<const uint8_t *>(buf) + sizeof h;
    try {
        for (int k = 0; k < 3; ++k) {
            const uint8_t *runs = p, *data = p + (size_t)h.nruns[k] * sizeof(vad_engine::SegAudio::Run);
            for (uint32_t r = 0; r < h.nruns[k]; ++r) {
                vad_engine::SegAudio::Run run;
                std::memcpy(&run, runs + (size_t)r * sizeof run, sizeof run);
                parts[k]->append(e->seg_arena, run.group, run.thr, data, (size_t)run.samples, nullptr);
                data += (size_t)run.samples * (vad_engine::SegAudio::is_i16(run.group) ? 2 : 4);
            }
            p = data;
        }
    } catch (const std::bad_alloc &) {
        st.clear(e->seg_arena);
        return e->fail(VAD_ERR_INVALID_ARG, "tick: out of host memory for the segment arena");
    }
    return VAD_OK;
}

namespace {
// the HIP half of a tick: copies in, launches, results back (mu held).  Fills the result pointers on success.
int tick_execute(vad_engine *e, vad_engine::TickBuf *tbs, float denoise_thresh, vad_tick_result *out, int64_t total, size_t frame_total,
                 size_t o_probs, size_t o_seg, size_t o_ev, size_t o_s32) {
    if (int rc = ensure(e, e->d_tick_frames, e->d_tick_frames_cap, frame_total)) return rc;
    const int32_t *h_s32 = reinterpret_cast<const int32_t *>(e->h_tick_out + o_s32);
    HIP_TRY(e, hipMemcpyAsync(e->d_tick_out + o_s32, h_s32, sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice, e->stream));
    size_t foff = 0;
    const float *d_rate_in[vad_engine::TICK_GROUPS] = {};
    for (int g = 0; g < vad_engine::TICK_GROUPS; ++g) {        // one launch per (format, gate) group: one in practice
        const int64_t cnt = tbs[g].count;
        if (!cnt) continue;
        const size_t fbytes = (size_t)cnt * tbs[g].row_bytes;
        uint8_t *d_fr = static_cast<uint8_t *>(e->d_tick_frames) + foff;
        HIP_TRY(e, hipMemcpyAsync(d_fr, tbs[g].h, fbytes, hipMemcpyHostToDevice, e->stream));
        foff += (fbytes + 255) & ~(size_t)255;
        if (g >= 6) {               // chunks at another rate: stepped below, all rates of a gate value in one vad_step_rates tick
            d_rate_in[g] = reinterpret_cast<const float *>(d_fr);
            continue;
        }
        vadk::StepParams p = e->base;
        p.slots = reinterpret_cast<const int32_t *>(e->d_tick_out + o_s32) + out->group_start[g];
        p.frames = d_fr;
        p.probs = reinterpret_cast<float *>(e->d_tick_out + o_probs) + out->group_start[g];
        p.seg_frames = reinterpret_cast<int32_t *>(e->d_tick_out + o_seg) + out->group_start[g];
        p.events = e->d_tick_out + o_ev + out->group_start[g];
        p.n = (int32_t)cnt;
        p.T = 1;
        p.fmt = g / 2;
        p.thresh = (g & 1) ? denoise_thresh : -1.0f;
        if (int rc = launch(e, p, e->stream)) return rc;
    }
    for (int gate = 0; gate < 2; ++gate) {
        const int g0 = 6 + 3 * gate;
        const float *seg_in[3];
        int64_t seg_n[3];
        int32_t seg_sr[3];
        int ns = 0;
        for (int ri = 0; ri < 3; ++ri)
            if (tbs[g0 + ri].count) {
                seg_in[ns] = d_rate_in[g0 + ri];
                seg_n[ns] = tbs[g0 + ri].count;
                seg_sr[ns++] = vad_engine::tick_group_rate(g0 + ri);
            }
        if (!ns) continue;
        const int64_t at = out->group_start[g0];               // the gate's rate groups are adjacent in the result arrays
        if (int rc = step_rates_enqueue(e, ns, seg_in, seg_n, seg_sr, reinterpret_cast<const int32_t *>(e->d_tick_out + o_s32) + at,
                                        gate ? denoise_thresh : -1.0f, reinterpret_cast<float *>(e->d_tick_out + o_probs) + at,
                                        e->d_tick_out + o_ev + at, reinterpret_cast<int32_t *>(e->d_tick_out + o_seg) + at, e->stream))
            return rc;
    }
    HIP_TRY(e, hipMemcpyAsync(e->h_tick_out + o_probs, e->d_tick_out + o_probs, o_s32 - o_probs, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}
}  // namespace

int vad_tick_run(vad_engine *e, float denoise_thresh, vad_tick_result *out) {
    if (!e || !out || out->struct_size < sizeof(vad_tick_result)) return VAD_ERR_INVALID_ARG;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
        return (float)std::chrono::duration<double, std::micro>(b2 - a).count();
    };
    const auto t0 = now();
    out->host_us[0] = out->host_us[1] = out->host_us[2] = 0.f;
    out->n = 0;
    out->dropped = 0;
    out->staged_next = 0;
    out->slots = nullptr; out->probs = nullptr; out->events = nullptr; out->seg_frames = nullptr; out->nsamples = nullptr;
    for (int g = 0; g < vad_engine::TICK_GROUPS; ++g) { out->group_start[g] = 0; out->group_frames[g] = nullptr; }
    out->group_start[vad_engine::TICK_GROUPS] = 0;
    // `mu` first (lock order), for the whole tick: slots cannot be opened or closed under it
    std::lock_guard<std::mutex> lk(e->mu);
    int b;
    vad_engine::TickBuf *tbs;
    {   // swap the staging buffers; every slot that has more frames waiting gets its next one into the new buffer, in the order
        // the slots started waiting; rows whose stream was closed (or closed and reopened) since the push are dropped
        std::lock_guard<std::mutex> tl(e->tick_mu);
        b = e->tick_cur;
        tbs = e->tick_buf[b];
        for (int g = 0; g < vad_engine::TICK_GROUPS; ++g)
            for (int64_t r = 0; r < tbs[g].count;) {
                const int32_t sl = tbs[g].slots()[r];
                if (sl >= 0 && sl < e->max_streams && e->open[(size_t)sl] && tbs[g].epochs()[r] == e->slot_epoch[(size_t)sl]) { ++r; continue; }
                tbs[g].drop_row(r);                            // (close and open forget the slot's tick state, so this is belt and braces)
                out->dropped += 1;
            }
        e->tick_cur ^= 1;
        for (auto &tb : e->tick_buf[e->tick_cur]) tb.count = 0;
        if (++e->tick_generation == 0) {
            std::fill(e->tick_gen.begin(), e->tick_gen.end(), 0u);
            e->tick_generation = 1;
        }
        size_t keep = 0;
        for (size_t k = 0; k < e->tick_overflow_order.size(); ++k) {
            const int64_t sl = e->tick_overflow_order[k];
            auto it = e->tick_overflow.find(sl);
            if (it == e->tick_overflow.end() || it->second.empty()) {
                if (it != e->tick_overflow.end()) e->tick_overflow.erase(it);
                continue;
            }
            vad_engine::TickPending &p = it->second.front();
            if (tick_place(e, sl, p.data.data(), (int32_t)(p.data.size() / vad_engine::tick_sample_bytes(p.group)), p.group) == VAD_OK) {
                vad_engine::TickBuf &nb = e->tick_buf[e->tick_cur][p.group];
                nb.lens()[nb.count - 1] = p.nsamples;
                it->second.pop_front();
                out->staged_next += 1;
            }                                                   // (a failed placement - pinned memory exhausted - leaves the frame waiting)
            if (it->second.empty()) e->tick_overflow.erase(it);
            else e->tick_overflow_order[keep++] = sl;
        }
        e->tick_overflow_order.resize(keep);
    }
    const auto t1 = now();
    out->host_us[0] = us(t0, t1);
    int64_t total = 0;
    size_t frame_total = 0;
    for (int g = 0; g < vad_engine::TICK_GROUPS; ++g) {
        out->group_start[g] = total;
        out->group_frames[g] = tbs[g].count ? tbs[g].h : nullptr;
        total += tbs[g].count;
        frame_total += ((size_t)tbs[g].count * tbs[g].row_bytes + 255) & ~(size_t)255;
    }
    out->group_start[vad_engine::TICK_GROUPS] = total;
    if (total == 0) return VAD_OK;
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_probs = up16(sizeof(int64_t) * (size_t)total), o_seg = o_probs + up16(sizeof(float) * (size_t)total);
    const size_t o_ev = o_seg + up16(sizeof(int32_t) * (size_t)total), o_s32 = o_ev + up16((size_t)total);
    const size_t o_len = o_s32 + up16(sizeof(int32_t) * (size_t)total);
    const size_t out_bytes = o_len + up16(sizeof(int32_t) * (size_t)total);
    // a tick that fails from here on has consumed its frames: the tails that belong to them go too, so that the slots' queues
    // stay aligned, and the caller is told which streams lost a frame (slots / nsamples / n are valid on failure, probs is NULL)
    auto lost = [&](int rc) {
        std::lock_guard<std::mutex> tl(e->tick_mu);
        for (int g = 0; g < vad_engine::TICK_GROUPS; ++g) {
            const int flen = vad_engine::tick_group_len(g, e->frame_samples);
            for (int64_t r = 0; r < tbs[g].count; ++r)
                if (tbs[g].lens()[r] > flen) {
                    auto it = e->tick_tails.find(tbs[g].slots()[r]);
                    if (it != e->tick_tails.end() && !it->second.empty()) it->second.pop_front();
                }
        }
        out->probs = nullptr; out->events = nullptr; out->seg_frames = nullptr;
        return rc;
    };
    hipError_t hr = hipSetDevice(e->device);
    if (hr == hipSuccess && out_bytes > e->tick_out_cap) {
        const size_t cap = std::max(out_bytes, (size_t)e->max_streams * 28 + 128);
        if (e->h_tick_out) (void)hipHostFree(e->h_tick_out);
        if (e->d_tick_out) (void)hipFree(e->d_tick_out);
        e->h_tick_out = e->d_tick_out = nullptr;
        e->tick_out_cap = 0;
        hr = hipHostMalloc((void **)&e->h_tick_out, cap, hipHostMallocDefault);
        if (hr == hipSuccess) hr = hipMalloc((void **)&e->d_tick_out, cap);
        if (hr == hipSuccess) e->tick_out_cap = cap;
    }
    if (hr != hipSuccess) return lost(e->hip_fail(hr, "tick: result buffers"));
    int64_t *h_slots = reinterpret_cast<int64_t *>(e->h_tick_out);
    int32_t *h_s32 = reinterpret_cast<int32_t *>(e->h_tick_out + o_s32);
    int32_t *h_len = reinterpret_cast<int32_t *>(e->h_tick_out + o_len);
    for (int g = 0; g < vad_engine::TICK_GROUPS; ++g)
        for (int64_t r = 0; r < tbs[g].count; ++r) {
            const int32_t sl = tbs[g].slots()[r];
            h_slots[out->group_start[g] + r] = sl;
            h_s32[out->group_start[g] + r] = sl;
            h_len[out->group_start[g] + r] = tbs[g].lens()[r];
        }
    out->n = total;
    out->slots = h_slots;
    out->nsamples = h_len;
    if (int rc = tick_execute(e, tbs, denoise_thresh, out, total, frame_total, o_probs, o_seg, o_ev, o_s32)) return lost(rc);
    out->probs = reinterpret_cast<const float *>(e->h_tick_out + o_probs);
    out->seg_frames = reinterpret_cast<const int32_t *>(e->h_tick_out + o_seg);
    out->events = e->h_tick_out + o_ev;
    const auto t2 = now();
    out->host_us[1] = us(t1, t2);
    if (e->tick_segments) {
        // the host half of _process_voice_state, per stepped stream, on the staged audio (kept in its wire format; float32 and
        // the gate of utils/audio.py:117-118 are applied when the finished segment is taken).  This pass decides and PLANS -
        // which buffer a frame goes to, which block and offset - and the copies themselves then run on the copy crew.
        std::lock_guard<std::mutex> tl(e->tick_mu);
        if (e->seg_state.size() < (size_t)e->max_streams) e->seg_state.resize((size_t)e->max_streams);
        std::vector<vad_engine::SegCopy> &plan = e->seg_copies;
        plan.clear();
        std::vector<std::vector<uint8_t>> spent;               // tails whose bytes the plan still points into
        try {
            for (int g = 0; g < vad_engine::TICK_GROUPS; ++g) {
                const int flen = vad_engine::tick_group_len(g, e->frame_samples);
                const size_t ss = vad_engine::tick_sample_bytes(g);
                for (int64_t r = 0; r < tbs[g].count; ++r) {
                    const int64_t i = out->group_start[g] + r;
                    const size_t sl = (size_t)h_slots[i];
                    vad_engine::SegState &st = e->seg_state[sl];
                    const int ev = out->events[i];
                    const bool above = (double)out->probs[i] >= e->h_start_prob[sl];
                    const int32_t L = tbs[g].lens()[r];
                    std::deque<std::vector<uint8_t>> *tails = nullptr;
                    if (L > flen) {
                        auto it = e->tick_tails.find((int64_t)sl);
                        if (it != e->tick_tails.end() && !it->second.empty()) tails = &it->second;
                    }
                    if (!st.active && !above) {                                            // idle stream: nothing is kept (:873-874)
                        if (!st.pre.empty()) st.pre.clear(e->seg_arena);
                        if (tails) tails->pop_front();
                        continue;
                    }
                    vad_engine::SegAudio &dst = st.active ? st.seg : st.pre;               // :891 / :838-839
                    dst.append(e->seg_arena, g, denoise_thresh, tbs[g].row(r), (size_t)std::min<int32_t>(L, flen), &plan);
                    if (tails) {
                        spent.push_back(std::move(tails->front()));
                        tails->pop_front();
                        dst.append(e->seg_arena, g, denoise_thresh, spent.back().data(), spent.back().size() / ss, &plan);
                    }
                    if (!st.active) {
                        if (ev & VAD_EV_START) {                                           // :860-869
                            st.active = true;
                            st.seg.swap(st.pre);
                            st.pre.clear(e->seg_arena);
                        }
                    } else if (ev & VAD_EV_END) {                                          // :932-949
                        st.done.clear(e->seg_arena);
                        st.done.swap(st.seg);
                        st.active = false;
                    }
                }
            }
        } catch (const std::bad_alloc &) {
            e->copy_crew.run(plan.data(), plan.size());        // what was planned is consistent; the rest of this tick's audio is lost
            return e->fail(VAD_ERR_INVALID_ARG, "tick: out of host memory for the segment arena");
        }
        e->copy_crew.run(plan.data(), plan.size());
        out->host_us[2] = us(t2, now());
    }
    return VAD_OK;
}

int vad_tick_run_work(vad_engine *e, float denoise_thresh, vad_tick_result *out, vad_tick_work *work) {
    if (!e || !work || work->struct_size < sizeof(vad_tick_work)) return VAD_ERR_INVALID_ARG;
    work->n_work = 0;
    work->work_index = nullptr;
    work->work_kind = nullptr;
    if (work->n_slots < 0 || (work->n_slots > 0 && (!work->last_prob || !work->frames_done || !work->active || !work->continue_cb ||
                                                    !work->continue_payload)))
        return e->fail(VAD_ERR_INVALID_ARG, "vad_tick_run_work: the per-slot arrays are missing");
    if (const int rc = vad_tick_run(e, denoise_thresh, out)) return rc;
    const int64_t n = out->n;
    // one tick at a time per engine (the caller's rule for vad_tick_run): the two vectors belong to this call until the next one
    e->work_index.clear();
    e->work_kind.clear();
    e->work_samples.clear();
    const int64_t first_rate_entry = out->group_start[6];       // chunks at another rate always have their exact length
    for (int64_t k = 0; k < n; ++k) {
        const int64_t sl = out->slots[k];
        if (sl < 0 || sl >= work->n_slots) return e->fail(VAD_ERR_INVALID_ARG, "vad_tick_run_work: slot %lld beyond the caller's arrays (%lld)", (long long)sl, (long long)work->n_slots);
        const uint8_t ev = out->events[k];
        const bool was = work->active[sl] != 0, started = (ev & VAD_EV_START) != 0, ended = (ev & VAD_EV_END) != 0;
        work->last_prob[sl] = out->probs[k];
        work->frames_done[sl] += 1;
        work->active[sl] = (uint8_t)((was || started) && !ended);
        uint8_t kind = (uint8_t)((started ? VAD_WORK_START : 0) | (ended ? VAD_WORK_END : 0));
        if (was && work->continue_cb[sl]) kind |= (uint8_t)(VAD_WORK_CONTINUE | (work->continue_payload[sl] ? VAD_WORK_PAYLOAD : 0));
        if (k < first_rate_entry && out->nsamples[k] > e->frame_samples) kind |= VAD_WORK_LONG;
        if (kind) {
            e->work_index.push_back((int32_t)k);
            e->work_kind.push_back(kind);
            e->work_samples.push_back(0);
        }
    }
    {   // the finished segments' lengths (segment assembly on): the caller sizes its payload buffers without a second call
        std::lock_guard<std::mutex> tl(e->tick_mu);
        for (size_t j = 0; j < e->work_kind.size(); ++j)
            if (e->work_kind[j] & VAD_WORK_END) {
                const size_t sl = (size_t)out->slots[e->work_index[j]];
                e->work_samples[j] = sl < e->seg_state.size() ? e->seg_state[sl].done.samples : 0;
            }
    }
    work->n_work = (int64_t)e->work_index.size();
    work->work_index = e->work_index.data();
    work->work_kind = e->work_kind.data();
    work->work_samples = e->work_samples.data();
    return VAD_OK;
}

int vad_tick_take_segment_wav16(vad_engine *e, int64_t slot, int32_t sample_rate, void *out, int64_t cap, int64_t *nbytes) {
    if (!e || !nbytes || sample_rate < 1) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->tick_mu);
    if (slot < 0 || (size_t)slot >= e->seg_state.size()) return e->fail(VAD_ERR_BAD_SLOT, "slot %lld has no segment state", (long long)slot);
    vad_engine::SegAudio &d = e->seg_state[(size_t)slot].done;
    const int64_t ns = d.samples, total = 44 + 2 * ns;
    *nbytes = total;
    if (!out) return VAD_OK;                       // size query
    if (cap < total) return e->fail(VAD_ERR_INVALID_ARG, "WAV buffer too small (%lld < %lld bytes)", (long long)cap, (long long)total);
    uint8_t *o = static_cast<uint8_t *>(out);
    auto put32 = [&](size_t at, uint32_t v) { std::memcpy(o + at, &v, 4); };       // (little-endian host, as everywhere in this file)
    auto put16 = [&](size_t at, uint16_t v) { std::memcpy(o + at, &v, 2); };
    std::memcpy(o, "RIFF", 4); put32(4, (uint32_t)(36 + 2 * ns)); std::memcpy(o + 8, "WAVEfmt ", 8);
    put32(16, 16); put16(20, 1); put16(22, 1); put32(24, (uint32_t)sample_rate); put32(28, (uint32_t)sample_rate * 2u); put16(32, 2); put16(34, 16);
    std::memcpy(o + 36, "data", 4); put32(40, (uint32_t)(2 * ns));
    // every stored run as vad_tick_take_segment hands it over (int16 / its scale as a float32 division, the run's gate), then
    // WAVWriter's conversion - float32 product, clip, truncation toward zero - in ONE pass, no float copy of the segment in between
    int16_t *q = reinterpret_cast<int16_t *>(o + 44);
    auto wav = [](float x) -> int16_t {
        float y = x * 32767.0f;
        y = y < -32768.0f ? -32768.0f : (y > 32767.0f ? 32767.0f : y);
        return (int16_t)y;
    };
    d.for_each_piece([&](const vad_engine::SegAudio::Run &r, const uint8_t *src, size_t cnt) {
        const bool g = vad_engine::SegAudio::gated(r.group);
        const float thr = r.thr;
        if (!vad_engine::SegAudio::is_i16(r.group)) {
            const float *x = reinterpret_cast<const float *>(src);
            if (g) for (size_t k = 0; k < cnt; ++k) q[k] = wav(std::fabs(x[k]) > thr ? x[k] : 0.f);
            else for (size_t k = 0; k < cnt; ++k) q[k] = wav(x[k]);
        } else {
            const float sc = r.group < 4 ? 32767.0f : 32768.0f;
            const int16_t *p16 = reinterpret_cast<const int16_t *>(src);
            if (g) for (size_t k = 0; k < cnt; ++k) { const float x = (float)p16[k] / sc; q[k] = wav(std::fabs(x) > thr ? x : 0.f); }
            else for (size_t k = 0; k < cnt; ++k) q[k] = wav((float)p16[k] / sc);
        }
        q += cnt;
    });
    d.clear(e->seg_arena);
    return VAD_OK;
}

int vad_debug_resample_operator(int32_t n_in, float *R, size_t r_floats) {
    g_create_error.clear();
    if (n_in < 8 || n_in % 8 || !R || r_floats < (size_t)n_in * VAD_FRAME_SAMPLES) {
        g_create_error = "vad_debug_resample_operator: n_in must be a positive multiple of 8 and R must hold 512*n_in floats";
        return VAD_ERR_INVALID_ARG;
    }
    std::vector<float> op;
    vadk::build_resample_operator(n_in, op);
    std::memcpy(R, op.data(), op.size() * sizeof(float));
    return VAD_OK;
}

int vad_debug_pack_resample(int32_t n_in, float *out, size_t out_floats, size_t *n_floats, uint32_t *tile_blocks,
                            uint32_t *row128_block) {
    g_create_error.clear();
    if (n_in < 256 || n_in % 256 || !n_floats || !tile_blocks || !row128_block) {
        g_create_error = "vad_debug_pack_resample: n_in must be a positive multiple of 256";
        return VAD_ERR_INVALID_ARG;
    }
    std::vector<float> packed;
    std::string perr;
    *tile_blocks = vadk::pack_resample_operator(n_in, packed, row128_block, perr);
    if (*tile_blocks == 0) {
        g_create_error = perr;
        return VAD_ERR_INVALID_ARG;
    }
    *n_floats = packed.size();
    if (out) {
        if (out_floats < packed.size()) {
            g_create_error = "vad_debug_pack_resample: output buffer too small";
            return VAD_ERR_INVALID_ARG;
        }
        std::memcpy(out, packed.data(), packed.size() * sizeof(float));
    }
    return VAD_OK;
}

int vad_debug_pack_resample_t16(int32_t n_in, float *out, size_t out_floats, size_t *n_floats, uint32_t *wave_blocks,
                                uint32_t *row128_block) {
    g_create_error.clear();
    if (n_in < 256 || n_in % 256 || !n_floats || !wave_blocks || !row128_block) {
        g_create_error = "vad_debug_pack_resample_t16: n_in must be a positive multiple of 256";
        return VAD_ERR_INVALID_ARG;
    }
    std::vector<float> packed;
    std::string perr;
    *wave_blocks = vadk::pack_resample_operator_t16(n_in, packed, row128_block, perr);
    if (*wave_blocks == 0) {
        g_create_error = perr;
        return VAD_ERR_INVALID_ARG;
    }
    *n_floats = packed.size();
    if (out) {
        if (out_floats < packed.size()) {
            g_create_error = "vad_debug_pack_resample_t16: output buffer too small";
            return VAD_ERR_INVALID_ARG;
        }
        std::memcpy(out, packed.data(), packed.size() * sizeof(float));
    }
    return VAD_OK;
}

int vad_debug_pack_weights(int32_t model_version, const void *weights, size_t weights_len, float *out,
                           size_t out_floats, size_t *n_floats, uint32_t *sect_out) {
    g_create_error.clear();
    vadk::PackedWeights pw;
    std::string perr;
    // model_version 516 / 416: Silero V5 / V4 packed for the 16-stream tile kernels (tests/kernel_model.py models both packings)
    const bool ok = model_version == 5     ? vadk::pack_silero_v5(weights, weights_len, pw, perr)
                    : model_version == 516 ? vadk::pack_silero_v5_t16(weights, weights_len, pw, perr)
                    : model_version == 4   ? vadk::pack_silero_v4(weights, weights_len, pw, perr)
                    : model_version == 416 ? vadk::pack_silero_v4_t16(weights, weights_len, pw, perr)
                                           : false;
    if (!ok) {
        g_create_error = perr.empty() ? "Failed to load model: model_version must be 4 or 5" : perr;
        return VAD_ERR_BAD_WEIGHTS;
    }
    if (n_floats) *n_floats = pw.data.size();
    if (sect_out) std::memcpy(sect_out, pw.sect, sizeof pw.sect);
    if (out) {
        if (out_floats < pw.data.size()) {
            g_create_error = "vad_debug_pack_weights: output buffer too small";
            return VAD_ERR_INVALID_ARG;
        }
        std::memcpy(out, pw.data.data(), pw.data.size() * sizeof(float));
    }
    return VAD_OK;
}

int vad_debug_set_tile(vad_engine *e, int32_t streams_per_tile) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (streams_per_tile == -1 || streams_per_tile == -2) {       // vad_step_rates: -1 = resample launch + model launch, -2 = fused (default)
        e->rates_fused = streams_per_tile == -2;
        return VAD_OK;
    }
    if (streams_per_tile != 0 && streams_per_tile != 16 && streams_per_tile != 32)
        return e->fail(VAD_ERR_INVALID_ARG, "tile: 0 (by batch size), 16 or 32");
    if (streams_per_tile == 16 && !e->d_wstream16)
        return e->fail(VAD_ERR_UNSUPPORTED, "Silero V5's 8 kHz sub-model has no 16-stream tile kernel");
    e->tile_policy = streams_per_tile;
    return VAD_OK;
}

int vad_debug_sm_replay(vad_engine *e, int64_t slot, const float *probs, int64_t n, uint8_t *events_out,
                        int32_t *seg_frames_out) {
    if (!e || !probs || !events_out || !seg_frames_out || n < 1) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (slot < 0 || slot >= e->max_streams || !e->open[(size_t)slot])
        return e->fail(VAD_ERR_BAD_SLOT, "slot %lld is not an open stream", (long long)slot);
    HIP_TRY(e, hipSetDevice(e->device));
    if (int rc = ensure(e, e->d_probs, e->d_probs_cap, sizeof(float) * n)) return rc;
    if (int rc = ensure(e, e->d_events, e->d_events_cap, (size_t)n)) return rc;
    if (int rc = ensure(e, e->d_seg, e->d_seg_cap, sizeof(int32_t) * n)) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_probs, probs, sizeof(float) * n, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, vadk_launch_sm_replay(e->d_sm, (int)slot, e->d_probs, (int)n, e->d_events, e->d_seg, e->stream));
    HIP_TRY(e, hipMemcpyAsync(events_out, e->d_events, (size_t)n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(seg_frames_out, e->d_seg, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

int vad_host_alloc(vad_engine *e, size_t bytes, void **out) {
    if (!e || !out || bytes == 0) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(e, hipSetDevice(e->device));
    void *p = nullptr;
    HIP_TRY(e, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    e->host_blocks.push_back(p);
    *out = p;
    return VAD_OK;
}

int vad_host_free(vad_engine *e, void *p) {
    if (!e || !p) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    auto it = std::find(e->host_blocks.begin(), e->host_blocks.end(), p);
    if (it == e->host_blocks.end()) return e->fail(VAD_ERR_INVALID_ARG, "vad_host_free: not a block of this engine");
    e->host_blocks.erase(it);
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));      // no copy of ours may still be reading it
    HIP_TRY(e, hipHostFree(p));
    return VAD_OK;
}

int vad_engine_synchronize(vad_engine *e) {
    if (!e) return VAD_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return VAD_OK;
}

}  // extern "C"

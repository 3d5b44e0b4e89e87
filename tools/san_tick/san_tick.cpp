// Sanitizer harness for the tick assembler (csrc/engine.cpp, host logic only; see hip/hip_runtime_api.h for the stand-in):
// producers push / push_rate / cancel from 4 threads WHILE vad_tick_run loops on the main thread and takes finished segments,
// a fifth thread opens and closes streams with frames still queued.  Checked: every stream's frames were stepped exactly once
// and in push order (the stand-in model returns |first sample|, so the audio scripts the probabilities), events equal a serial
// replay of the real state machine, segments hold exactly the frames the reference would keep; then, single-threaded: a failing
// tick keeps the queues aligned, and a stream saved mid-segment continues bit-identically on another slot.
// Built three times by tools/san_tick/run.sh: -fsanitize=thread, -fsanitize=address,undefined, and plain.
#include "../../include/vad_engine.h"
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

#include "sm_device.h"

extern int fake_hip_fail_after;

#define CHECK(c)                                                                 \
    do {                                                                         \
        if (!(c)) {                                                              \
            std::fprintf(stderr, "san_tick: CHECK failed at line %d: %s\n", __LINE__, #c); \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)
#define OK(call)                                                                                                      \
    do {                                                                                                              \
        int _rc = (call);                                                                                             \
        if (_rc != VAD_OK) {                                                                                          \
            std::fprintf(stderr, "san_tick: %s -> %d (%s) at line %d\n", #call, _rc, vad_last_error(eng), __LINE__);  \
            std::exit(1);                                                                                             \
        }                                                                                                             \
    } while (0)

static vad_engine *eng;
static const int NPROD = 4, PER = 160, K = 60;
static const vad_thresholds THR = {0.5, 0.5, 0.5, 0.5, 2, 2};

struct Plan { int fmt; bool gate; int rate; int len; };        // how a stream sends its frames
static Plan plan_of(int id) {
    Plan p;
    p.gate = id & 1;
    switch (id % 5) {
        case 0: p = {VAD_FMT_F32, p.gate, 16000, 512}; break;
        case 1: p = {VAD_FMT_I16_32767, p.gate, 16000, 512}; break;
        case 2: p = {VAD_FMT_F32, p.gate, 48000, 1536}; break;        // resampled inside the tick
        case 3: p = {VAD_FMT_I16_32768, p.gate, 16000, 600}; break;   // over-long frames: the segment keeps the tail
        default: p = {VAD_FMT_F32, p.gate, 16000, 400}; break;        // short frames: zero-padded for the model
    }
    p.gate = id & 1;
    return p;
}
static float script(int id, int k) { return ((k + id) / 7) % 2 ? 0.875f : 0.0f; }   // exactly representable in int16 / 32768 too

static void make_frame(int id, int k, const Plan &p, std::vector<uint8_t> &buf) {
    const float v = script(id, k);
    if (p.fmt == VAD_FMT_F32) {
        buf.resize((size_t)p.len * 4);
        float *f = reinterpret_cast<float *>(buf.data());
        for (int i = 0; i < p.len; ++i) f[i] = v;
    } else {
        buf.resize((size_t)p.len * 2);
        int16_t *q = reinterpret_cast<int16_t *>(buf.data());
        const float sc = p.fmt == VAD_FMT_I16_32767 ? 32767.0f : 32768.0f;
        for (int i = 0; i < p.len; ++i) q[i] = (int16_t)std::lround(v * sc);
    }
}

struct Seen { std::vector<float> probs; std::vector<uint8_t> events; std::vector<int64_t> seg_samples; std::vector<float> seg_first; };

static void take_all(vad_tick_result &res, std::vector<Seen> &seen_by_slot) {
    for (int64_t i = 0; i < res.n; ++i) {
        Seen &s = seen_by_slot[(size_t)res.slots[i]];
        s.probs.push_back(res.probs[i]);
        s.events.push_back(res.events[i]);
        if (res.events[i] & VAD_EV_END) {
            int64_t n = 0;
            OK(vad_tick_take_segment(eng, res.slots[i], nullptr, 0, &n));
            std::vector<float> seg((size_t)n + 1);
            OK(vad_tick_take_segment(eng, res.slots[i], seg.data(), n, &n));
            s.seg_samples.push_back(n);
            s.seg_first.push_back(n ? seg[0] : -1.f);
            for (int64_t j = 0; j < n; ++j) CHECK(seg[(size_t)j] == seg[0] || seg[(size_t)j] == 0.f || seg[0] == 0.f);
        }
    }
}

// what a serial run of the same frames yields: probabilities, events (the real state machine), segment lengths
static void expect(int id, const Plan &p, int frames, Seen &out) {
    vadk::SmSlot s{};
    s.start_prob = THR.start_probability; s.end_prob = THR.end_probability; s.start_ratio = THR.start_ratio; s.end_ratio = THR.end_ratio;
    s.start_count = THR.start_frame_count; s.end_count = THR.end_frame_count; s.seg_frames = -1;
    int64_t pre = 0, seg = 0;
    bool active = false;
    for (int k = 0; k < frames; ++k) {
        float v = script(id, k);
        if (p.fmt == VAD_FMT_I16_32767) v = (float)(int16_t)std::lround(v * 32767.0f) / 32767.0f;
        const float prob = std::fmin(1.f, std::fabs(v));         // 0.875 > the 0.01 gate, 0 stays 0
        int sg = 0;
        const int ev = vadk::sm_step(s, prob, &sg);
        out.probs.push_back(prob);
        out.events.push_back((uint8_t)ev);
        const bool above = (double)prob >= THR.start_probability;
        if (!active && !above) { pre = 0; continue; }
        (active ? seg : pre) += p.len;
        if (!active && (ev & VAD_EV_START)) { active = true; seg = pre; pre = 0; }
        else if (active && (ev & VAD_EV_END)) { out.seg_samples.push_back(seg); seg = 0; active = false; }
    }
}

int main(int argc, char **argv) {
    CHECK(argc == 2);
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), {});
    vad_engine_desc d{};
    d.struct_size = sizeof d;
    d.model_version = 5;
    d.weights = blob.data();
    d.weights_len = blob.size();
    d.max_streams = 1024;
    d.sample_rate = 16000;
    if (vad_engine_create(&d, &eng) != VAD_OK) {
        std::fprintf(stderr, "san_tick: create failed: %s\n", vad_last_create_error());
        return 1;
    }
    OK(vad_tick_enable_segments(eng, 1));
    const int N = NPROD * PER;
    std::vector<int64_t> slots((size_t)N);
    OK(vad_stream_open_many(eng, N, slots.data()));
    OK(vad_stream_set_thresholds_many(eng, slots.data(), N, &THR, 1));
    std::vector<Seen> seen(1024);
    std::atomic<int> producers_left{NPROD + 1};
    std::atomic<long> pushed{0};

    std::vector<std::thread> th;
    for (int t = 0; t < NPROD; ++t)
        th.emplace_back([&, t] {
            std::vector<uint8_t> buf;
            for (int k = 0; k < K; ++k)
                for (int j = 0; j < PER; ++j) {
                    const int id = t * PER + j;
                    const Plan p = plan_of(id);
                    make_frame(id, k, p, buf);
                    for (;;) {
                        // the three ways a frame at the engine's rate gets in: one by one, as a batch with per-frame status, gathered
                        int rc;
                        int32_t st = 0;
                        const void *one = buf.data();
                        if (p.rate != 16000 && (k + j) % 2 == 0) rc = vad_tick_push_rate(eng, slots[(size_t)id], buf.data(), p.len, p.fmt, p.gate, p.rate);
                        else if (p.rate != 16000) { rc = vad_tick_push_rate_gather(eng, &slots[(size_t)id], 1, &one, p.len, p.fmt, p.gate, p.rate, &st); CHECK(rc == st); }
                        else if ((k + j) % 3 == 0) rc = vad_tick_push(eng, slots[(size_t)id], buf.data(), p.len, p.fmt, p.gate);
                        else if ((k + j) % 3 == 1) { rc = vad_tick_push_status(eng, &slots[(size_t)id], 1, buf.data(), p.len, p.fmt, p.gate, &st); CHECK(rc == st); }
                        else { rc = vad_tick_push_gather(eng, &slots[(size_t)id], 1, &one, p.len, p.fmt, p.gate, &st); CHECK(rc == st); }
                        if (rc == VAD_OK) break;
                        CHECK(rc == VAD_ERR_BUSY);                 // 256 frames waiting: the ticker is behind, try again
                        std::this_thread::yield();
                    }
                    pushed.fetch_add(1);
                    if (j % 40 == 39) std::this_thread::sleep_for(std::chrono::microseconds(150));   // spread the pushes over many ticks
                }
            producers_left.fetch_sub(1);
        });
    // streams that come and go with frames still queued: their rows must never reach the slot's next owner
    th.emplace_back([&] {
        std::vector<uint8_t> buf;
        const Plan p = {VAD_FMT_F32, true, 16000, 512};
        for (int round = 0; round < 300; ++round) {
            int64_t s[8];
            if (vad_stream_open_many(eng, 8, s) != VAD_OK) { std::this_thread::yield(); continue; }
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 8; ++j) {
                    make_frame(1000 + j, 1, p, buf);
                    (void)vad_tick_push(eng, s[j], buf.data(), p.len, p.fmt, 1);
                }
            if (round % 3 == 0) for (int j = 0; j < 8; ++j) (void)vad_tick_cancel(eng, s[j]);
            for (int j = 0; j < 8; ++j) OK(vad_stream_close(eng, s[j]));
        }
        producers_left.fetch_sub(1);
    });

    vad_tick_result res{};
    res.struct_size = sizeof res;
    long ticks = 0, frames = 0, dropped = 0;
    std::vector<uint8_t> mine(1024, 0);
    for (int id = 0; id < N; ++id) mine[(size_t)slots[(size_t)id]] = 1;
    for (;;) {
        const bool done = producers_left.load() == 0;
        OK(vad_tick_run(eng, 0.01f, &res));
        ++ticks;
        dropped += (long)res.dropped;
        // only the producers' streams are tallied; the churn thread's slots are different slots by construction
        vad_tick_result mineonly = res;
        std::vector<int64_t> sl; std::vector<float> pr; std::vector<uint8_t> ev;
        for (int64_t i = 0; i < res.n; ++i)
            if (mine[(size_t)res.slots[i]]) { sl.push_back(res.slots[i]); pr.push_back(res.probs[i]); ev.push_back(res.events[i]); }
        mineonly.n = (int64_t)sl.size(); mineonly.slots = sl.data(); mineonly.probs = pr.data(); mineonly.events = ev.data();
        take_all(mineonly, seen);
        frames += (long)sl.size();
        CHECK(res.staged_next >= 0 && res.staged_next <= 1024);
        if (done && res.n == 0) break;
    }
    for (auto &t : th) t.join();
    CHECK(frames == (long)N * K);
    for (int id = 0; id < N; ++id) {
        Seen want;
        expect(id, plan_of(id), K, want);
        const Seen &got = seen[(size_t)slots[(size_t)id]];
        CHECK(got.probs == want.probs);                              // every frame once, in push order
        CHECK(got.events == want.events);
        CHECK(got.seg_samples == want.seg_samples);
        CHECK(!want.seg_samples.empty());
    }
    std::printf("san_tick: concurrent phase ok: %d streams x %d frames in %ld ticks, %ld stale rows dropped\n", N, K, ticks, dropped);

    // ---- a failing tick: its frames are consumed, the tails that belonged to them too; what was queued behind carries on
    {
        const int id = 3;                                            // 600-sample frames: tails
        const Plan p = plan_of(id);
        int64_t s;
        OK(vad_stream_open(eng, &s));
        OK(vad_stream_set_thresholds(eng, s, &THR));
        std::vector<uint8_t> buf;
        for (int k = 0; k < 30; ++k) { make_frame(id, k, p, buf); OK(vad_tick_push(eng, s, buf.data(), p.len, p.fmt, p.gate)); }
        Seen got;
        std::vector<Seen> by(1024);
        for (int k = 0; k < 30; ++k) {
            if (k == 11) fake_hip_fail_after = 2;                    // the frame copy of this tick fails
            const int rc = vad_tick_run(eng, 0.01f, &res);
            if (k == 11) {
                CHECK(rc == VAD_ERR_HIP && res.n == 1 && res.slots && res.slots[0] == s && res.nsamples[0] == 600 && !res.probs);
                continue;
            }
            CHECK(rc == VAD_OK && res.n == 1);
            take_all(res, by);
        }
        // the reference: the same script with frame 11 missing
        vadk::SmSlot sm{};
        sm.start_prob = sm.end_prob = sm.start_ratio = sm.end_ratio = 0.5; sm.start_count = sm.end_count = 2; sm.seg_frames = -1;
        std::vector<int64_t> want_seg;
        int64_t pre = 0, seg = 0; bool active = false;
        for (int k = 0; k < 30; ++k) {
            if (k == 11) continue;
            const float prob = std::fabs((float)(int16_t)std::lround(script(id, k) * 32768.0f) / 32768.0f);
            int sg = 0;
            const int ev = vadk::sm_step(sm, prob, &sg);
            const bool above = prob >= 0.5;
            if (!active && !above) { pre = 0; continue; }
            (active ? seg : pre) += 600;
            if (!active && (ev & 1)) { active = true; seg = pre; pre = 0; }
            else if (active && (ev & 2)) { want_seg.push_back(seg); seg = 0; active = false; }
        }
        CHECK(by[(size_t)s].seg_samples == want_seg && !want_seg.empty());
        OK(vad_stream_close(eng, s));
        std::printf("san_tick: failing tick ok (%zu segments, lengths aligned)\n", want_seg.size());
    }
    // ---- migration mid-segment: save (h, c, state machine) + segment audio, restore on another slot, carry on
    {
        const int id = 1;
        const Plan p = plan_of(id);
        int64_t a, b, c;
        OK(vad_stream_open(eng, &a)); OK(vad_stream_open(eng, &c));
        OK(vad_stream_set_thresholds(eng, a, &THR)); OK(vad_stream_set_thresholds(eng, c, &THR));
        std::vector<uint8_t> buf;
        std::vector<Seen> by(1024);
        int64_t cur = a;
        for (int k = 0; k < 40; ++k) {
            make_frame(id, k, p, buf);
            OK(vad_tick_push(eng, cur, buf.data(), p.len, p.fmt, p.gate));
            OK(vad_tick_push(eng, c, buf.data(), p.len, p.fmt, p.gate));
            OK(vad_tick_run(eng, 0.01f, &res));
            take_all(res, by);
            if (k == 10) {                                           // inside the first utterance
                int64_t pend = -1, nb = 0;
                OK(vad_tick_pending(eng, cur, &pend));
                CHECK(pend == 0);
                std::vector<uint8_t> st(VAD_STREAM_SAVE_BYTES);
                OK(vad_stream_save(eng, cur, st.data(), (int64_t)st.size()));
                OK(vad_tick_segment_save(eng, cur, nullptr, 0, &nb));
                std::vector<uint8_t> sg((size_t)nb);
                OK(vad_tick_segment_save(eng, cur, sg.data(), nb, &nb));
                CHECK(nb > 64);
                OK(vad_stream_open(eng, &b));
                OK(vad_stream_restore(eng, b, st.data(), (int64_t)st.size()));
                OK(vad_tick_segment_restore(eng, b, sg.data(), nb));
                sg[sg.size() / 2 + 1] ^= 0x40;                       // a damaged blob is refused before anything is touched ...
                sg[4] = 7;
                CHECK(vad_tick_segment_restore(eng, b, sg.data(), nb) == VAD_ERR_INVALID_ARG);
                OK(vad_stream_close(eng, cur));
                by[(size_t)b] = by[(size_t)cur];
                cur = b;
            }
        }
        const Seen &m = by[(size_t)cur], &r = by[(size_t)c];
        CHECK(m.probs == r.probs && m.events == r.events && m.seg_samples == r.seg_samples && m.seg_first == r.seg_first && !r.seg_samples.empty());
        std::printf("san_tick: migration mid-segment ok (%zu segments identical)\n", r.seg_samples.size());
    }
    vad_engine_destroy(eng);
    // ---- whole batches (many / status / gather), on a fresh engine so that the staging grows in the middle of a batch
    //      (256 rows -> 512); a second thread pushes single frames meanwhile; the sources are overwritten right after each call
    {
        OK(vad_engine_create(&d, &eng));
        OK(vad_tick_enable_segments(eng, 1));
        const int B = 300, KB = 24, RB = 100;                        // RB streams at 48 kHz: their chunks go in as ONE rate batch per tick
        std::vector<int64_t> bs((size_t)B + 20 + RB);
        OK(vad_stream_open_many(eng, B + 20 + RB, bs.data()));
        OK(vad_stream_set_thresholds_many(eng, bs.data(), B + 20 + RB, &THR, 1));
        const Plan p = {VAD_FMT_I16_32767, true, 16000, 480};
        const Plan pr = {VAD_FMT_I16_32767, true, 48000, 1536};
        std::vector<std::vector<uint8_t>> rfr((size_t)RB);
        std::vector<const void *> rptrs((size_t)RB);
        std::vector<int32_t> rst((size_t)RB);
        std::atomic<bool> side_done{false};
        std::thread side([&] {
            std::vector<uint8_t> buf;
            for (int k = 0; k < KB; ++k)
                for (int j = 0; j < 20; ++j) {
                    make_frame(B + j, k, p, buf);
                    while (vad_tick_push(eng, bs[(size_t)(B + j)], buf.data(), p.len, p.fmt, p.gate) != VAD_OK) std::this_thread::yield();
                }
            side_done.store(true);
        });
        std::vector<Seen> by(1024);
        std::vector<std::vector<uint8_t>> fr((size_t)B);
        std::vector<const void *> ptrs((size_t)B);
        std::vector<uint8_t> flat;
        std::vector<int32_t> st((size_t)B);
        long got = 0;
        for (int k = 0; k < KB; ++k) {
            for (int j = 0; j < B; ++j) { make_frame(j, k, p, fr[(size_t)j]); ptrs[(size_t)j] = fr[(size_t)j].data(); }
            if (k % 3 == 0) {
                OK(vad_tick_push_gather(eng, bs.data(), B, ptrs.data(), p.len, p.fmt, p.gate, st.data()));
            } else {
                flat.clear();
                for (int j = 0; j < B; ++j) flat.insert(flat.end(), fr[(size_t)j].begin(), fr[(size_t)j].end());
                if (k % 3 == 1) OK(vad_tick_push_status(eng, bs.data(), B, flat.data(), p.len, p.fmt, p.gate, st.data()));
                else OK(vad_tick_push_many(eng, bs.data(), B, flat.data(), p.len, p.fmt, p.gate));
            }
            // 100 x 3 072 bytes: above the copy crew's threshold, so the int16 -> float32 conversions run on its threads
            for (int j = 0; j < RB; ++j) { make_frame(B + 20 + j, k, pr, rfr[(size_t)j]); rptrs[(size_t)j] = rfr[(size_t)j].data(); }
            OK(vad_tick_push_rate_gather(eng, bs.data() + B + 20, RB, rptrs.data(), pr.len, pr.fmt, pr.gate, pr.rate, rst.data()));
            for (auto &f2 : rfr) std::fill(f2.begin(), f2.end(), 0x5a);
            for (auto &f2 : fr) std::fill(f2.begin(), f2.end(), 0x5a);       // the sources may go once the call has returned
            std::fill(flat.begin(), flat.end(), 0x5a);
            if (k % 4 != 3) {                                               // some batches queue up behind the previous one
                OK(vad_tick_run(eng, 0.01f, &res));
                take_all(res, by);
                got += (long)res.n;
            }
        }
        for (;;) {
            const bool done = side_done.load();
            OK(vad_tick_run(eng, 0.01f, &res));
            take_all(res, by);
            got += (long)res.n;
            if (done && res.n == 0) break;
        }
        side.join();
        CHECK(got == (long)(B + 20 + RB) * KB);
        for (int j = 0; j < B + 20 + RB; ++j) {
            Seen want;
            expect(j, j < B + 20 ? p : pr, KB, want);
            const Seen &g = by[(size_t)bs[(size_t)j]];
            CHECK(g.probs == want.probs && g.events == want.events && g.seg_samples == want.seg_samples);
        }
        vad_engine_destroy(eng);
        std::printf("san_tick: batched pushes ok (%d streams x %d frames, %d of them rate batches)\n", B + 20 + RB, KB, RB);
    }
    std::printf("san_tick: all ok\n");
    return 0;
}

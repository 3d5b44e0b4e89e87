// TEST-ONLY host stand-ins for the vadk_launch_* entry points of the .hip files (see hip/hip_runtime_api.h): the "model" is
// p = clamp(|first sample of the frame|, 0, 1) - the harness scripts a stream's probabilities through its audio - and the events
// come from the REAL state machine (csrc/sm_device.h) on the slot's SmSlot, exactly as the kernels apply it.
#include "../../include/vad_engine.h"
#include <hip/hip_runtime.h>

#include <cmath>

#include "resample_generic.h"
#include "sm_device.h"
#include "vad_layout.h"

int fake_hip_fail_after = 0;

hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    if (fake_hip_fail_after > 0 && --fake_hip_fail_after == 0) return hipErrorInvalidValue;
    std::memcpy(d, s, n);
    return hipSuccess;
}

using namespace vadk;

static float first_sample(const void *frames, size_t index, int fmt, int frame_samples) {
    if (fmt == VAD_FMT_F32) return static_cast<const float *>(frames)[index * (size_t)frame_samples];
    const int16_t q = static_cast<const int16_t *>(frames)[index * (size_t)frame_samples];
    return (float)q / (fmt == VAD_FMT_I16_32767 ? 32767.0f : 32768.0f);
}

static hipError_t fake_step(const StepParams *p, int frame_samples) {
    for (int i = 0; i < p->n; ++i) {
        const int slot = p->slots ? p->slots[i] : i;
        SmSlot &s = p->sm[slot];
        int seg_last = 0;
        for (int t = 0; t < p->T; ++t) {
            float x = first_sample(p->frames, (size_t)i * p->T + t, p->fmt, frame_samples);
            if (p->thresh >= 0.f && !(std::fabs(x) > p->thresh)) x = 0.f;
            const float prob = std::fmin(1.0f, std::fabs(x));
            int sg = 0;
            const int ev = sm_step(s, prob, &sg);
            if (ev & 2) seg_last = sg;
            p->probs[(size_t)i * p->T + t] = prob;
            if (p->events) p->events[(size_t)i * p->T + t] = (uint8_t)ev;
            p->state[(size_t)slot * 256] += 1.0f;              // "h" counts the frames this stream has seen
        }
        if (p->seg_frames) p->seg_frames[i] = seg_last;
    }
    return hipSuccess;
}

extern "C" hipError_t vadk_launch_silero_v5(const StepParams *p, hipStream_t) { return fake_step(p, p->variant ? 256 : 512); }
extern "C" hipError_t vadk_launch_silero_v4(const StepParams *p, hipStream_t) { return fake_step(p, 512); }
extern "C" hipError_t vadk_launch_silero_v5_t16(const StepParams *p, hipStream_t) { return fake_step(p, 512); }
extern "C" hipError_t vadk_launch_silero_v4_t16(const StepParams *p, int, hipStream_t) { return fake_step(p, 512); }

extern "C" hipError_t vadk_launch_silero_v5_t16_rates(const StepParams *p, const RateParams *r, hipStream_t) {
    for (int k = 0; k < r->nseg; ++k) {
        StepParams q = *p;
        q.n = r->seg[k].n;
        q.T = 1;
        q.fmt = VAD_FMT_F32;
        q.frames = r->seg[k].in;
        q.slots = p->slots ? p->slots + r->seg[k].stream0 : nullptr;
        q.probs = p->probs + r->seg[k].stream0;
        q.events = p->events ? p->events + r->seg[k].stream0 : nullptr;
        q.seg_frames = p->seg_frames ? p->seg_frames + r->seg[k].stream0 : nullptr;
        fake_step(&q, r->seg[k].n_in);                         // a chunk's first sample stands for its resampled frame's
    }
    return hipSuccess;
}

extern "C" hipError_t vadk_launch_resample(const ResampleParams *p, hipStream_t) {
    for (int k = 0; k < p->nseg; ++k)
        for (int i = 0; i < p->seg[k].n; ++i) {
            float *o = p->seg[k].out + (size_t)i * 512;
            std::memset(o, 0, 512 * sizeof(float));
            o[0] = p->seg[k].in[(size_t)i * p->seg[k].n_in];
        }
    return hipSuccess;
}

extern "C" hipError_t vadk_launch_rsg_partial(const RsgParams *, hipStream_t) { return hipErrorInvalidValue; }
extern "C" hipError_t vadk_launch_rsg_finish(const RsgParams *, hipStream_t) { return hipErrorInvalidValue; }
extern "C" hipError_t vadk_rsf_build_tables(const RsfParams *, hipStream_t) { return hipErrorInvalidValue; }
extern "C" hipError_t vadk_rsf_run(const RsfParams *, hipStream_t) { return hipErrorInvalidValue; }

extern "C" hipError_t vadk_launch_slot_control(SmSlot *sm, float *state, const int32_t *slots, int n, int op, const SmSlot *def,
                                               const vad_thresholds *thr, int nthr, hipStream_t) {
    for (int i = 0; i < n; ++i) {                              // csrc/vad_util.hip: vadk_slot_control, statement for statement
        const int s = slots[i];
        if (op & 1) std::memset(state + (size_t)s * 256, 0, 256 * sizeof(float));
        SmSlot cur = sm[s];
        if (op & 2) cur = *def;
        if (op & 4) {
            SmSlot fresh = *def;
            fresh.start_prob = cur.start_prob; fresh.end_prob = cur.end_prob;
            fresh.start_ratio = cur.start_ratio; fresh.end_ratio = cur.end_ratio;
            fresh.start_count = cur.start_count; fresh.end_count = cur.end_count;
            cur = fresh;
        }
        if (op & 8) {
            const vad_thresholds t = thr[nthr == 1 ? 0 : i];
            cur.start_prob = t.start_probability; cur.end_prob = t.end_probability;
            cur.start_ratio = t.start_ratio; cur.end_ratio = t.end_ratio;
            cur.start_count = t.start_frame_count; cur.end_count = t.end_frame_count;
        }
        sm[s] = cur;
    }
    return hipSuccess;
}

extern "C" hipError_t vadk_launch_sm_replay(SmSlot *sm, int slot, const float *probs, int n, uint8_t *events, int32_t *seg, hipStream_t) {
    SmSlot s = sm[slot];
    for (int i = 0; i < n; ++i) {
        int sg = 0;
        const int ev = sm_step(s, probs[i], &sg);
        events[i] = (uint8_t)ev;
        seg[i] = (ev & 2) ? sg : 0;
    }
    sm[slot] = s;
    return hipSuccess;
}

// Small utility kernels of the engine: batched slot control (open / reset / thresholds for many slots in ONE launch) and the
// state-machine replay (diagnostic).
#include <hip/hip_runtime.h>
#include "../../include/vad_engine.h"
#include "sm_device.h"
#include "vad_layout.h"

using namespace vadk;

// One 64-thread block per listed slot.  op bits: 1 = zero (h, c); 2 = state machine := `def` (stream open); 4 = reset the
// dynamic part of the state machine, thresholds kept (VADProcessor.reset, core/silero_model.py:951-968); 8 = thresholds :=
// thr[nthr == 1 ? 0 : i] (VADWrapper.set_thresholds, core/vad_wrapper.py:367-419: values only).
extern "C" __global__ void vadk_slot_control(SmSlot *sm, float *state, const int32_t *slots, int n, int op, SmSlot def,
                                             const vad_thresholds *thr, int nthr) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const int s = slots[i];
    if (op & 1) reinterpret_cast<float4 *>(state + (size_t)s * 256)[threadIdx.x] = float4{0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x != 0) return;
    SmSlot cur = sm[s];
    if (op & 2) cur = def;
    if (op & 4) {
        SmSlot fresh = def;
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

extern "C" hipError_t vadk_launch_slot_control(SmSlot *sm, float *state, const int32_t *d_slots, int n, int op, const SmSlot *def,
                                               const vad_thresholds *d_thr, int nthr, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(vadk_slot_control, dim3(n), dim3(64), 0, stream, sm, state, d_slots, n, op, *def, d_thr, nthr);
    return hipGetLastError();
}

// one thread replays a scripted probability sequence through one slot's state machine
extern "C" __global__ void vadk_sm_replay(SmSlot *sm, int slot, const float *probs, int n, uint8_t *events, int32_t *seg) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    SmSlot s = sm[slot];
    for (int i = 0; i < n; ++i) {
        int sg = 0;
        const int ev = sm_step(s, probs[i], &sg);
        events[i] = (uint8_t)ev;
        seg[i] = (ev & 2) ? sg : 0;
    }
    sm[slot] = s;
}

extern "C" hipError_t vadk_launch_sm_replay(SmSlot *sm, int slot, const float *probs, int n, uint8_t *events, int32_t *seg,
                                            hipStream_t stream) {
    (void)hipGetLastError();   // HIP's last-error slot is sticky and process-wide: a stale failure from anywhere else must not become ours
    hipLaunchKernelGGL(vadk_sm_replay, dim3(1), dim3(64), 0, stream, sm, slot, probs, n, events, seg);
    return hipGetLastError();
}

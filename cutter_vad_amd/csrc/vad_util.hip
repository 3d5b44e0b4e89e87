// Small utility kernels of the engine: state-machine replay (diagnostic).
#include <hip/hip_runtime.h>
#include "sm_device.h"
#include "vad_layout.h"

using namespace vadk;

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

/* A serving loop in C99 over the drop-in boundary (include/vad_engine.h, ABI v2): what a native websocket front end would do in
 * place of the reference's per-client receive loop (websocket_service/server/vad_websocket_server.py:326-382).
 *   gcc -std=c99 -Iinclude examples/c_abi_serve.c -Lcutter_vad_amd -lvad_engine -Wl,-rpath,$PWD/cutter_vad_amd -lm -o /tmp/c_abi_serve
 *   /tmp/c_abi_serve cutter_vad_amd/weights/silero_v5_16k.svw
 * N clients send 30 ms int16 PCM frames (480 samples); every tick one vad_tick_push_many hands the frames that arrived to the
 * engine and one vad_tick_run advances all clients (one launch) and keeps their segments' audio; on END the finished segment is
 * taken.  Then the same frames once more through the pipelined host API (vad_step_submit / vad_step_collect) to show that the two
 * paths agree.  Needs an MI355X: there is no CPU path. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vad_engine.h"

#define N 48
#define TICKS 60
#define WIRE 480

static void synth(int16_t *dst, int client, int tick) {      /* a tone burst between ticks 10 and 34, quiet hum elsewhere */
    const double loud = (tick >= 10 && tick < 34) ? 0.35 : 0.002;
    for (int k = 0; k < WIRE; ++k) {
        const double t = (double)(tick * WIRE + k) / 16000.0;
        const double v = loud * (sin(6.283185307179586 * (140.0 + 7.0 * client) * t) + 0.5 * sin(6.283185307179586 * (420.0 + 3.0 * client) * t));
        dst[k] = (int16_t)lrint(32767.0 * (v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v)));
    }
}

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != VAD_OK) { fprintf(stderr, "%s: %d: %s\n", #call, rc_, vad_last_error(e)); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <weights.svw>\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *blob = malloc((size_t)len);
    if (fread(blob, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);

    vad_engine_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.model_version = 5;
    d.weights = blob;
    d.weights_len = (size_t)len;
    d.max_streams = 2 * N;
    d.sample_rate = 16000;
    vad_engine *e = NULL;
    if (vad_engine_create(&d, &e) != VAD_OK) { fprintf(stderr, "vad_engine_create: %s\n", vad_last_create_error()); return 1; }

    int64_t slots[N], again[N];
    vad_thresholds thr = {0.5, 0.35, 0.8, 0.95, 3, 6};
    CHECK(vad_stream_open_many(e, N, slots));
    CHECK(vad_stream_open_many(e, N, again));
    CHECK(vad_stream_set_thresholds_many(e, slots, N, &thr, 1));
    CHECK(vad_stream_set_thresholds_many(e, again, N, &thr, 1));
    CHECK(vad_tick_enable_segments(e, 1));

    /* ---- the tick loop ---- */
    static int16_t wire[N][WIRE];
    static float p_tick[TICKS][N];
    int starts = 0, ends = 0;
    long seg_samples = 0;
    for (int t = 0; t < TICKS; ++t) {
        for (int i = 0; i < N; ++i) synth(wire[i], i, t);
        CHECK(vad_tick_push_many(e, slots, N, wire, WIRE, VAD_FMT_I16_32767, 1));
        vad_tick_result r;
        memset(&r, 0, sizeof r);
        r.struct_size = sizeof r;
        CHECK(vad_tick_run(e, 0.01f, &r));
        if (r.n != N) { fprintf(stderr, "tick %d stepped %lld streams\n", t, (long long)r.n); return 1; }
        for (int64_t i = 0; i < r.n; ++i) {
            p_tick[t][r.slots[i] - slots[0]] = r.probs[i];          /* slots of one open_many call are consecutive */
            if (r.events[i] & VAD_EV_START) ++starts;
            if (r.events[i] & VAD_EV_END) {
                int64_t ns = 0;
                CHECK(vad_tick_take_segment(e, r.slots[i], NULL, 0, &ns));
                float *seg = (float *)malloc(sizeof(float) * (size_t)(ns > 0 ? ns : 1));
                CHECK(vad_tick_take_segment(e, r.slots[i], seg, ns, &ns));
                if (ns != (int64_t)WIRE * r.seg_frames[i]) { fprintf(stderr, "segment length mismatch\n"); return 1; }
                seg_samples += (long)ns;
                free(seg);
                ++ends;
            }
        }
    }

    /* ---- the same audio through the pipelined host API: copy of tick t+1 under the kernel of tick t ---- */
    int16_t *pin[2];
    CHECK(vad_host_alloc(e, sizeof(int16_t) * N * VAD_FRAME_SAMPLES, (void **)&pin[0]));
    CHECK(vad_host_alloc(e, sizeof(int16_t) * N * VAD_FRAME_SAMPLES, (void **)&pin[1]));
    double worst = 0.0;
    int64_t ticket[2] = {-1, -1};
    float probs[N];
    for (int t = 0; t <= TICKS; ++t) {
        if (t < TICKS) {
            int16_t *dst = pin[t & 1];
            memset(dst, 0, sizeof(int16_t) * N * VAD_FRAME_SAMPLES);       /* right zero-pad 480 -> 512, as the tick does */
            for (int i = 0; i < N; ++i) synth(dst + (size_t)i * VAD_FRAME_SAMPLES, i, t);
            CHECK(vad_step_submit(e, again, N, 1, dst, VAD_FMT_I16_32767, 0.01f, &ticket[t & 1]));
        }
        if (t > 0) {
            CHECK(vad_step_collect(e, ticket[(t - 1) & 1], probs, NULL, NULL));
            for (int i = 0; i < N; ++i) {
                const double dp = fabs((double)probs[i] - (double)p_tick[t - 1][i]);
                if (dp > worst) worst = dp;
            }
        }
    }
    printf("clients %d ticks %d starts %d ends %d segment_samples %ld max_dp_tick_vs_pipelined %.3g p_last %.6f\n", N, TICKS, starts,
           ends, seg_samples, worst, p_tick[TICKS - 1][0]);
    vad_engine_destroy(e);
    free(blob);
    /* (which clients the model takes for voice depends on their tone; every started segment must have ended by the last tick) */
    return (starts > 0 && ends == starts && worst == 0.0) ? 0 : 1;
}

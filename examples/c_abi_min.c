/* Minimal C99 caller of the drop-in boundary (include/vad_engine.h): one engine, a few streams, one frame each.
 *   gcc -std=c99 -Iinclude examples/c_abi_min.c -Lcutter_vad_amd -lvad_engine -Wl,-rpath,$PWD/cutter_vad_amd -o /tmp/c_abi_min
 *   /tmp/c_abi_min cutter_vad_amd/weights/silero_v5_16k.svw
 * Needs an MI355X: there is no CPU path, vad_engine_create fails with VAD_ERR_NO_DEVICE elsewhere (and says so). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vad_engine.h"

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
    d.device_id = 0;
    d.max_streams = 64;
    d.sample_rate = 16000;
    vad_engine *e = NULL;
    int rc = vad_engine_create(&d, &e);
    if (rc != VAD_OK) { fprintf(stderr, "vad_engine_create: %d: %s\n", rc, vad_last_create_error()); return 1; }

    enum { N = 3 };
    int64_t slots[N];
    float frames[N][VAD_FRAME_SAMPLES], probs[N];
    for (int i = 0; i < N; ++i) {
        if (vad_stream_open(e, &slots[i]) != VAD_OK) { fprintf(stderr, "%s\n", vad_last_error(e)); return 1; }
        for (int k = 0; k < VAD_FRAME_SAMPLES; ++k) frames[i][k] = 0.1f * (float)((k * (i + 3)) % 17 - 8) / 8.0f;
    }
    rc = vad_step(e, slots, N, frames, VAD_FMT_F32, 0.01f, probs);
    if (rc != VAD_OK) { fprintf(stderr, "vad_step: %d: %s\n", rc, vad_last_error(e)); return 1; }
    for (int i = 0; i < N; ++i) printf("stream %d: p = %.6f\n", i, probs[i]);
    vad_engine_destroy(e);
    free(blob);
    return 0;
}

/*
 * ORACLE / TEST INFRASTRUCTURE — CPU restatement of the reference's per-frame hot path.
 *
 * NOT product code: only tests/, tools/make_goldens.py, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may link or load this file.  The product (the HIP engine
 * behind include/vad_engine.h) never calls into it and has no CPU fallback.
 *
 * What it restates (plain C, one stream at a time, no SIMD intrinsics):
 *   - svo_step_v5  : `session.run` on silero_vad_v5.onnx, 16 kHz branch (If_0.then_branch),
 *                    as called from /root/reference/src/real_time_vad/core/silero_model.py:433
 *                    with the feeds built at :488-492.  Node-by-node source: SURVEY.md §8 a7.
 *   - svo_step_v4  : the same call on silero_vad.onnx, 16 kHz branch (If_25.then_branch),
 *                    feeds :494-499.  Node-by-node source: SURVEY.md §8 a8.
 *   - svo_denoise  : AudioUtils.denoise_audio, /root/reference/src/real_time_vad/utils/audio.py:117-118
 *   - svo_pad_frame: SileroVADModel._prepare_audio_input, silero_model.py:464-468
 *   - svo_num_frames / svo_split_frames: AudioUtils.split_into_frames, audio.py:183-188
 *   - svo_sm_*     : VADProcessor._process_voice_state and helpers, silero_model.py:790-949
 *   - svo_resample : scipy.signal.resample (Fourier method) as called by
 *                    AudioUtils.resample_audio, audio.py:46-49 (scipy is a third-party
 *                    dependency, pinned scipy>=1.7.0 in /root/reference/pyproject.toml:33;
 *                    algorithm restated from its published source, 1.15.3 installed here)
 *   - svo_wav_size / svo_write_wav16: WAVWriter.write_wav_data, utils/wav_writer.py:40-136
 *
 * The arithmetic itself belongs to onnxruntime (third-party, `onnxruntime>=1.10.0`,
 * /root/reference/pyproject.toml:32), absent from /root/reference and from this image.
 * PARITY STATUS: "parity unpinned" against onnxruntime — the reference's tests mock
 * `InferenceSession.run` and hold no numeric vectors for the model.  This restatement is
 * pinned instead (tests/test_oracle.py) against (a) oracle/onnx_interp.py, an independent
 * node-by-node ONNX-spec interpreter run on the reference's own .onnx files (golden vectors
 * in tests/golden/), (b) live scipy for the resampler, (c) the reference's own state-machine
 * code and its test expectations for svo_sm_*.
 *
 * Build: see oracle/Makefile.  -DSVO_ACC_FLOAT selects float accumulators (used for the CPU
 * timing baseline, same arithmetic width as ORT's CPU EP); the default accumulates in double.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#ifdef SVO_ACC_FLOAT
typedef float acc_t;
#define SVO_EXP expf
#define SVO_TANH tanhf
#define SVO_SQRT sqrtf
#define SVO_LOG logf
#else
typedef double acc_t;
#define SVO_EXP exp
#define SVO_TANH tanh
#define SVO_SQRT sqrt
#define SVO_LOG log
#endif

#define SVO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ weight blob (SVW) */

typedef struct {
    char name[48];
    uint32_t ndim, dims[4], reserved;
    uint64_t offset, nelem;
} svw_entry; /* 88 bytes, see cutter_vad_amd/weights_io.py */

typedef struct svo_model {
    int version;
    uint8_t *blob;
    size_t blob_len;
    uint32_t n;
    const svw_entry *tab;
    /* V5 */
    const float *stft, *enc_w[4], *enc_b[4], *w_ih, *w_hh, *b_ih, *b_hh, *head_w, *head_b;
    /* V4 */
    const float *filt;
    const float *dw_w[4], *dw_b[4], *pw_w[4], *pw_b[4], *pj_w[4], *pj_b[4], *s_w[4], *s_b[4];
    const float *l_wih[2], *l_whh[2], *l_bih[2], *l_bhh[2];
    int v5_8k;   /* V5's else-branch (sr != 16000): window 128, hop 64, 65 bins, encoder.0 65 -> 128, 256-sample frames */
    int v4_8k;   /* the graph's else-branch (sr != 16000): third stride conv has stride 1 -> 2 time steps reach the LSTMs */
} svo_model;

static const float *svw_find(const svo_model *m, const char *name, uint64_t expect) {
    for (uint32_t i = 0; i < m->n; ++i) {
        if (strncmp(m->tab[i].name, name, sizeof m->tab[i].name) == 0) {
            if (m->tab[i].nelem != expect) return NULL;
            if (m->tab[i].offset + 4 * m->tab[i].nelem > m->blob_len) return NULL;
            return (const float *)(m->blob + m->tab[i].offset);
        }
    }
    return NULL;
}

#define NEED(ptr, name, cnt)                                                        \
    do {                                                                            \
        (ptr) = svw_find(m, (name), (cnt));                                         \
        if (!(ptr)) {                                                               \
            snprintf(err, errlen, "weight blob: tensor %s missing or wrong size", (name)); \
            goto fail;                                                              \
        }                                                                           \
    } while (0)

SVO_API svo_model *svo_load(const void *blob, size_t len, char *err, size_t errlen) {
    svo_model *m = (svo_model *)calloc(1, sizeof *m);
    char nm[64];
    if (!m) return NULL;
    if (len < 16 || memcmp(blob, "SVADW001", 8) != 0) {
        snprintf(err, errlen, "weight blob: bad magic");
        free(m);
        return NULL;
    }
    m->blob = (uint8_t *)malloc(len);
    memcpy(m->blob, blob, len);
    m->blob_len = len;
    m->version = (int)((const uint32_t *)m->blob)[2];
    m->n = ((const uint32_t *)m->blob)[3];
    m->tab = (const svw_entry *)(m->blob + 16);
    if (16 + (size_t)m->n * sizeof(svw_entry) > len) {
        snprintf(err, errlen, "weight blob: truncated table");
        goto fail;
    }
    {
        const float *var = svw_find(m, "meta.variant", 1);
        m->v5_8k = m->version == 5 && var && var[0] == 8000.0f;
    }
    if (m->v5_8k) NEED(m->stft, "stft.basis", 130 * 128);
    else NEED(m->stft, "stft.basis", 258 * 256);
    if (m->version == 5) {
        static const int co[4] = {128, 64, 64, 128};
        const int ci[4] = {m->v5_8k ? 65 : 129, 128, 64, 64};
        for (int i = 0; i < 4; ++i) {
            snprintf(nm, sizeof nm, "enc%d.w", i);
            NEED(m->enc_w[i], nm, (uint64_t)co[i] * ci[i] * 3);
            snprintf(nm, sizeof nm, "enc%d.b", i);
            NEED(m->enc_b[i], nm, co[i]);
        }
        NEED(m->w_ih, "lstm.w_ih", 512 * 128);
        NEED(m->w_hh, "lstm.w_hh", 512 * 128);
        NEED(m->b_ih, "lstm.b_ih", 512);
        NEED(m->b_hh, "lstm.b_hh", 512);
        NEED(m->head_w, "head.w", 128);
        NEED(m->head_b, "head.b", 1);
    } else if (m->version == 4) {
        static const int ci[4] = {258, 16, 32, 32}, co[4] = {16, 32, 32, 64}, sc[4] = {16, 32, 32, 64};
        NEED(m->filt, "norm.filter", 7);
        for (int i = 0; i < 4; ++i) {
            snprintf(nm, sizeof nm, "l%d.dw.w", i); NEED(m->dw_w[i], nm, (uint64_t)ci[i] * 5);
            snprintf(nm, sizeof nm, "l%d.dw.b", i); NEED(m->dw_b[i], nm, ci[i]);
            snprintf(nm, sizeof nm, "l%d.pw.w", i); NEED(m->pw_w[i], nm, (uint64_t)co[i] * ci[i]);
            snprintf(nm, sizeof nm, "l%d.pw.b", i); NEED(m->pw_b[i], nm, co[i]);
            if (i != 2) {
                snprintf(nm, sizeof nm, "l%d.proj.w", i); NEED(m->pj_w[i], nm, (uint64_t)co[i] * ci[i]);
                snprintf(nm, sizeof nm, "l%d.proj.b", i); NEED(m->pj_b[i], nm, co[i]);
            }
            snprintf(nm, sizeof nm, "s%d.w", i); NEED(m->s_w[i], nm, (uint64_t)sc[i] * sc[i]);
            snprintf(nm, sizeof nm, "s%d.b", i); NEED(m->s_b[i], nm, sc[i]);
        }
        for (int l = 0; l < 2; ++l) {
            snprintf(nm, sizeof nm, "lstm%d.w_ih", l); NEED(m->l_wih[l], nm, 256 * 64);
            snprintf(nm, sizeof nm, "lstm%d.w_hh", l); NEED(m->l_whh[l], nm, 256 * 64);
            snprintf(nm, sizeof nm, "lstm%d.b_ih", l); NEED(m->l_bih[l], nm, 256);
            snprintf(nm, sizeof nm, "lstm%d.b_hh", l); NEED(m->l_bhh[l], nm, 256);
        }
        NEED(m->head_w, "head.w", 64);
        NEED(m->head_b, "head.b", 1);
        {
            const float *var = svw_find(m, "meta.variant", 1);
            m->v4_8k = var && var[0] == 8000.0f;
        }
    } else {
        snprintf(err, errlen, "weight blob: unknown model version %d", m->version);
        goto fail;
    }
    return m;
fail:
    free(m->blob);
    free(m);
    return NULL;
}

SVO_API void svo_free(svo_model *m) {
    if (m) {
        free(m->blob);
        free(m);
    }
}

SVO_API int svo_version(const svo_model *m) { return m->version; }
/* samples one model step consumes: 512, or 256 for V5's 8 kHz sub-model */
SVO_API int svo_frame_samples(const svo_model *m) { return m->v5_8k ? 256 : 512; }

/* ------------------------------------------------------------------ small helpers */

static inline acc_t sigmoid_a(acc_t v) { return (acc_t)1 / ((acc_t)1 + SVO_EXP(-v)); }

static inline acc_t dot_f(const float *a, const float *b, int n) {
    acc_t s = 0;
#pragma omp simd reduction(+ : s)
    for (int i = 0; i < n; ++i) s += (acc_t)a[i] * (acc_t)b[i];
    return s;
}

/* windowed-DFT conv: out[f][t] = sum_n basis[f][n] * xin[t*hop + n]; magnitude of (re,im)=(f,f+nbin); nfft = 2 (nbin - 1) */
static void stft_mag_n(const float *basis, const float *xin, int nfft, int hop, int T, float *mag /*[nbin][T]*/) {
    const int nbin = nfft / 2 + 1;
    for (int t = 0; t < T; ++t) {
        const float *seg = xin + t * hop;
        for (int c = 0; c < nbin; ++c) {
            acc_t re = dot_f(basis + (size_t)c * nfft, seg, nfft);
            acc_t im = dot_f(basis + (size_t)(nbin + c) * nfft, seg, nfft);
            mag[c * T + t] = (float)SVO_SQRT(re * re + im * im);
        }
    }
}
static void stft_mag(const float *basis, const float *xin, int hop, int T, float *mag /*[129][T]*/) {
    stft_mag_n(basis, xin, 256, hop, T, mag);
}

/* ONNX Conv (cross-correlation), 1-D, group 1, zero padding `pad` both sides, + bias, optional relu.
 * in [Cin][Tin], w [Cout][Cin][K], out [Cout][Tout]. */
static void conv1d(const float *in, int Cin, int Tin, const float *w, const float *b, int Cout, int K,
                   int stride, int pad, int relu, float *out, int Tout) {
    for (int o = 0; o < Cout; ++o) {
        for (int t = 0; t < Tout; ++t) {
            acc_t s = b ? (acc_t)b[o] : 0;
            for (int c = 0; c < Cin; ++c) {
                const float *wr = w + ((size_t)o * Cin + c) * K;
                for (int k = 0; k < K; ++k) {
                    int ti = t * stride + k - pad;
                    if (ti >= 0 && ti < Tin) s += (acc_t)wr[k] * (acc_t)in[c * Tin + ti];
                }
            }
            float v = (float)s;
            out[o * Tout + t] = (relu && v < 0.f) ? 0.f : v;
        }
    }
}

/* LSTM cell, PyTorch gate order i,f,g,o; w_ih [4H][In], w_hh [4H][H]. h,c updated in place. */
static void lstm_cell(const float *x, int In, int H, const float *w_ih, const float *w_hh,
                      const float *b_ih, const float *b_hh, float *h, float *c) {
    float hn[128], cn[128];
    for (int j = 0; j < H; ++j) {
        acc_t g[4];
        for (int q = 0; q < 4; ++q) {
            int r = q * H + j;
            g[q] = dot_f(w_ih + (size_t)r * In, x, In) + (acc_t)b_ih[r] + dot_f(w_hh + (size_t)r * H, h, H) +
                   (acc_t)b_hh[r];
        }
        acc_t cc = sigmoid_a(g[1]) * (acc_t)c[j] + sigmoid_a(g[0]) * SVO_TANH(g[2]);
        cn[j] = (float)cc;
        hn[j] = (float)(sigmoid_a(g[3]) * SVO_TANH((acc_t)cn[j]));
    }
    memcpy(h, hn, sizeof(float) * H);
    memcpy(c, cn, sizeof(float) * H);
}

/* ------------------------------------------------------------------ V5, 16 kHz */

static void step_v5(const svo_model *m, const float *x /*[512]*/, float *state /*[h128|c128]*/, float *prob) {
    float mag[129 * 3], e0[128 * 3], e1[64 * 2], e2[64], e3[128];
    if (m->v5_8k) {
        /* else-branch on a 256-sample frame: right reflect pad 32 (never read), window 128, hop 64 -> columns at 0, 64, 128 */
        stft_mag_n(m->stft, x, 128, 64, 3, mag);
        conv1d(mag, 65, 3, m->enc_w[0], m->enc_b[0], 128, 3, 1, 1, 1, e0, 3);
    } else {
        /* a7 step 1-3: the right reflect pad (64) is never read when L=512: columns start at 0,128,256 */
        stft_mag(m->stft, x, 128, 3, mag);
        conv1d(mag, 129, 3, m->enc_w[0], m->enc_b[0], 128, 3, 1, 1, 1, e0, 3); /* step 4 */
    }
    conv1d(e0, 128, 3, m->enc_w[1], m->enc_b[1], 64, 3, 2, 1, 1, e1, 2);   /* step 5 */
    conv1d(e1, 64, 2, m->enc_w[2], m->enc_b[2], 64, 3, 2, 1, 1, e2, 1);    /* step 6 */
    conv1d(e2, 64, 1, m->enc_w[3], m->enc_b[3], 128, 3, 1, 1, 1, e3, 1);   /* step 7 */
    lstm_cell(e3, 128, 128, m->w_ih, m->w_hh, m->b_ih, m->b_hh, state, state + 128); /* step 8 */
    acc_t s = (acc_t)m->head_b[0];                                          /* step 9 */
    for (int j = 0; j < 128; ++j)
        if (state[j] > 0.f) s += (acc_t)m->head_w[j] * (acc_t)state[j];
    *prob = (float)sigmoid_a(s);
}

/* ------------------------------------------------------------------ V4, 16 kHz */

/* depthwise-separable block: dw k5 p2 (+relu) -> pw 1x1 ; residual = proj(in) or in ; relu(sum) */
static void v4_block(const svo_model *m, int i, const float *in, int Cin, int Cout, int T, float *out) {
    float d[258 * 8];
    for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < T; ++t) {
            acc_t s = (acc_t)m->dw_b[i][c];
            for (int k = 0; k < 5; ++k) {
                int ti = t + k - 2;
                if (ti >= 0 && ti < T) s += (acc_t)m->dw_w[i][c * 5 + k] * (acc_t)in[c * T + ti];
            }
            float v = (float)s;
            d[c * T + t] = v < 0.f ? 0.f : v;
        }
    for (int o = 0; o < Cout; ++o)
        for (int t = 0; t < T; ++t) {
            acc_t s = (acc_t)m->pw_b[i][o];
            for (int c = 0; c < Cin; ++c) s += (acc_t)m->pw_w[i][o * Cin + c] * (acc_t)d[c * T + t];
            float x0 = (float)s, r;
            if (m->pj_w[i]) {
                acc_t q = (acc_t)m->pj_b[i][o];
                for (int c = 0; c < Cin; ++c) q += (acc_t)m->pj_w[i][o * Cin + c] * (acc_t)in[c * T + t];
                r = (float)q;
            } else {
                r = in[o * T + t];
            }
            float v = x0 + r;
            out[o * T + t] = v < 0.f ? 0.f : v;
        }
}

/* 1x1 conv with stride + relu */
static void v4_stride(const svo_model *m, int i, const float *in, int C, int Tin, int stride, float *out, int Tout) {
    for (int o = 0; o < C; ++o)
        for (int t = 0; t < Tout; ++t) {
            acc_t s = (acc_t)m->s_b[i][o];
            for (int c = 0; c < C; ++c) s += (acc_t)m->s_w[i][o * C + c] * (acc_t)in[c * Tin + t * stride];
            float v = (float)s;
            out[o * Tout + t] = v < 0.f ? 0.f : v;
        }
}

static void step_v4(const svo_model *m, const float *x /*[512]*/, float *state /*[h0|h1|c0|c1] x64*/, float *prob) {
    float xp[704], x1[258 * 8], a[16 * 8], b[32 * 4], c2[32 * 2], c3[64 * 2], t0[64 * 4], xt[64];
    float *mag = x1, *norm = x1 + 129 * 8;
    /* 1. reflect pad 96 each side (numpy 'reflect': edge sample not repeated) */
    for (int i = 0; i < 96; ++i) xp[i] = x[96 - i];
    memcpy(xp + 96, x, 512 * sizeof(float));
    for (int i = 0; i < 96; ++i) xp[608 + i] = x[510 - i];
    stft_mag(m->stft, xp, 64, 8, mag);
    /* 2. spect = log(1 + mag * 2^20)   (Mul, Add, Log: intermediates are fp32 tensors) */
    acc_t mean[14], mean1, mm = 0;
    for (int t = 0; t < 8; ++t) {
        acc_t s = 0;
        for (int c = 0; c < 129; ++c) {
            float v = (float)SVO_LOG((acc_t)(1.0f + mag[c * 8 + t] * 1048576.0f));
            norm[c * 8 + t] = v;
            s += (acc_t)v;
        }
        mean[3 + t] = (acc_t)(float)(s / 129);
    }
    /* 3. adaptive normalisation: reflect-pad the per-frame mean by 3, 7-tap smoothing, mean over time */
    mean[0] = mean[3 + 3]; mean[1] = mean[3 + 2]; mean[2] = mean[3 + 1];
    mean[11] = mean[3 + 6]; mean[12] = mean[3 + 5]; mean[13] = mean[3 + 4];
    for (int t = 0; t < 8; ++t) {
        mean1 = 0;
        for (int k = 0; k < 7; ++k) mean1 += (acc_t)m->filt[k] * mean[t + k];
        mm += (acc_t)(float)mean1;
    }
    float mean_mean = (float)(mm / 8);
    for (int i = 0; i < 129 * 8; ++i) norm[i] = norm[i] - mean_mean;
    /* 4-12. encoder */
    v4_block(m, 0, x1, 258, 16, 8, a);
    v4_stride(m, 0, a, 16, 8, 2, t0, 4);
    v4_block(m, 1, t0, 16, 32, 4, b);
    v4_stride(m, 1, b, 32, 4, 2, t0, 2);
    v4_block(m, 2, t0, 32, 32, 2, c2);
    /* 16 kHz branch: stride 2 -> one time step.  8 kHz branch (torch_jit10, taken for every sr != 16000): the third
     * stride conv has stride 1 (Conv_632), so T = 2 time steps go through block 3, the last 1x1 conv, both LSTMs
     * (sequentially) and the head; the output is the mean of the two sigmoids (ReduceMean over T). */
    const int T3 = m->v4_8k ? 2 : 1;
    v4_stride(m, 2, c2, 32, 2, m->v4_8k ? 1 : 2, t0, T3);
    v4_block(m, 3, t0, 32, 64, T3, c3);
    v4_stride(m, 3, c3, 64, T3, 1, t0, T3);
    acc_t psum = 0;
    for (int t = 0; t < T3; ++t) {
        for (int j = 0; j < 64; ++j) xt[j] = t0[j * T3 + t];
        /* 13. two stacked LSTM(64): state = h[2][64] then c[2][64] (ONNX inputs h, c) */
        lstm_cell(xt, 64, 64, m->l_wih[0], m->l_whh[0], m->l_bih[0], m->l_bhh[0], state, state + 128);
        lstm_cell(state, 64, 64, m->l_wih[1], m->l_whh[1], m->l_bih[1], m->l_bhh[1], state + 64, state + 192);
        /* 14. head */
        acc_t s = (acc_t)m->head_b[0];
        for (int j = 0; j < 64; ++j)
            if (state[64 + j] > 0.f) s += (acc_t)m->head_w[j] * (acc_t)state[64 + j];
        psum += (acc_t)(float)sigmoid_a(s);
    }
    *prob = (float)(psum / T3);
}

/* One frame, one stream.  state: 256 floats, ONNX tensor order
 * (V5: state[2][1][128] = h,c ; V4: h[2][1][64] then c[2][1][64]). */
SVO_API void svo_step(const svo_model *m, const float *frame512, float *state256, float *prob) {
    if (m->version == 5)
        step_v5(m, frame512, state256, prob);
    else
        step_v4(m, frame512, state256, prob);
}

typedef struct {
    const svo_model *m;
    const float *frames;
    float *states, *probs;
    int begin, end;
} svo_job;

static void *svo_worker(void *p) {
    svo_job *j = (svo_job *)p;
    for (int i = j->begin; i < j->end; ++i)
        svo_step(j->m, j->frames + (size_t)i * svo_frame_samples(j->m), j->states + (size_t)i * 256, j->probs + i);
    return NULL;
}

/* n independent streams, one frame each, split over nthreads host threads. */
SVO_API void svo_step_batch(const svo_model *m, const float *frames, int n, float *states, float *probs, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (nthreads > n) nthreads = n > 0 ? n : 1;
    pthread_t th[256];
    svo_job jobs[256];
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (svo_job){m, frames, states, probs, (int)((long long)n * t / nthreads),
                            (int)((long long)n * (t + 1) / nthreads)};
        if (t > 0) pthread_create(&th[t], NULL, svo_worker, &jobs[t]);
    }
    svo_worker(&jobs[0]);
    for (int t = 1; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------ pre-steps */

/* audio.py:117-118: np.where(np.abs(x) > thr, x, 0.0)  — strict '>' */
SVO_API void svo_denoise(float *x, int n, float thr) {
    for (int i = 0; i < n; ++i)
        if (!(fabsf(x[i]) > thr)) x[i] = 0.f;
}

/* silero_model.py:464-468: right-zero-pad or truncate to exactly 512 */
SVO_API void svo_pad_frame(const float *in, int n, float *out512) {
    int k = n < 512 ? n : 512;
    memcpy(out512, in, (size_t)k * sizeof(float));
    for (int i = k; i < 512; ++i) out512[i] = 0.f;
}

/* audio.py:183: num_frames = (len - frame) // hop + 1 with Python floor division */
SVO_API int svo_num_frames(int len, int frame, int hop) {
    int d = len - frame;
    int q = d / hop;
    if ((d % hop != 0) && ((d < 0) != (hop < 0))) --q;
    return q + 1;
}

SVO_API int svo_split_frames(const float *audio, int len, int frame, int hop, float *out) {
    int n = svo_num_frames(len, frame, hop);
    for (int i = 0; i < n; ++i) memcpy(out + (size_t)i * frame, audio + (size_t)i * hop, (size_t)frame * sizeof(float));
    return n;
}

/* ------------------------------------------------------------------ resampler (a11) */

/* scipy.signal.resample(x, num) for real x, window=None, followed by astype(float32). */
SVO_API void svo_resample(const float *in, int n_in, float *out, int n_out) {
    int N = n_in < n_out ? n_in : n_out;
    int nyq = N / 2 + 1;
    int nb = n_out / 2 + 1;
    double *re = (double *)calloc((size_t)nb, sizeof(double));
    double *im = (double *)calloc((size_t)nb, sizeof(double));
    const double tau = 6.283185307179586476925286766559;
    for (int k = 0; k < nyq && k < nb; ++k) {
        double sr = 0, si = 0;
        for (int n = 0; n < n_in; ++n) {
            /* exact phase reduction keeps the O(n^2) DFT accurate */
            long long ph = ((long long)k * n) % n_in;
            double a = tau * (double)ph / (double)n_in;
            sr += (double)in[n] * cos(a);
            si -= (double)in[n] * sin(a);
        }
        re[k] = sr;
        im[k] = si;
    }
    if (N % 2 == 0) {
        if (n_out < n_in) { re[N / 2] *= 2.0; im[N / 2] *= 2.0; }
        else if (n_in < n_out) { re[N / 2] *= 0.5; im[N / 2] *= 0.5; }
    }
    double scale = (double)n_out / (double)n_in;
    for (int n = 0; n < n_out; ++n) {
        double s = re[0];
        for (int k = 1; k < nb; ++k) {
            long long ph = ((long long)k * n) % n_out;
            double a = tau * (double)ph / (double)n_out;
            if ((n_out % 2 == 0) && k == n_out / 2)
                s += re[k] * cos(a); /* irfft ignores the imaginary part of the Nyquist bin */
            else
                s += 2.0 * (re[k] * cos(a) - im[k] * sin(a));
        }
        out[n] = (float)(s / (double)n_out * scale);
    }
    free(re);
    free(im);
}

/* ------------------------------------------------------------------ state machine (a10) */

typedef struct svo_sm {
    /* config (config.py:54-94) */
    double start_prob, end_prob, start_ratio, end_ratio; /* Python floats */
    int start_count, end_count;
    /* silero_model.py:596-639 */
    int active;
    int n_start, n_end;           /* voice_start_frame_count / voice_end_frame_count */
    uint8_t start_hist[20];       /* recent_start_frames, deque(maxlen=20)  :620-623 */
    int start_len, start_head;
    uint8_t end_hist[100];        /* recent_end_frames,   deque(maxlen=100) :625-628 */
    int end_len, end_head;
    int buffered;                 /* len(voice_buffer) in frames            :631-634 */
    long long seg_samples;        /* len(current_voice_data) or -1 if None  :636-639 */
} svo_sm;

enum { SVO_EV_START = 1, SVO_EV_END = 2, SVO_EV_CONTINUE = 4 };

SVO_API size_t svo_sm_sizeof(void) { return sizeof(svo_sm); }

/* silero_model.py:951-968 */
SVO_API void svo_sm_reset(svo_sm *s) {
    s->active = 0;
    s->n_start = s->n_end = 0;
    s->start_len = s->start_head = 0;
    s->end_len = s->end_head = 0;
    s->buffered = 0;
    s->seg_samples = -1;
}

SVO_API void svo_sm_init(svo_sm *s, double start_prob, double end_prob, double start_ratio, double end_ratio,
                         int start_count, int end_count) {
    memset(s, 0, sizeof *s);
    s->start_prob = start_prob; s->end_prob = end_prob;
    s->start_ratio = start_ratio; s->end_ratio = end_ratio;
    s->start_count = start_count; s->end_count = end_count;
    svo_sm_reset(s);
}

static void dq_push(uint8_t *buf, int cap, int *len, int *head, uint8_t v) {
    /* deque(maxlen=cap).append: drop the oldest when full */
    if (*len < cap) {
        buf[(*head + *len) % cap] = v;
        ++*len;
    } else {
        buf[*head] = v;
        *head = (*head + 1) % cap;
    }
}

static int dq_sum_last(const uint8_t *buf, int cap, int len, int head, int k) {
    int s = 0;
    for (int i = len - k; i < len; ++i) s += buf[(head + i) % cap];
    return s;
}

/* One frame of `frame_len` samples with probability p.  Returns event bits; on END,
 * *seg_samples_out = length of the finished segment in samples (what the WAV holds). */
SVO_API int svo_sm_step(svo_sm *s, double p, int frame_len, long long *seg_samples_out) {
    int ev = 0;
    if (seg_samples_out) *seg_samples_out = 0;
    if (!s->active) {
        /* _handle_voice_start_detection :818-858 */
        int above = p >= s->start_prob;
        dq_push(s->start_hist, 20, &s->start_len, &s->start_head, (uint8_t)above);
        if (above) {
            s->n_start += 1;
            s->buffered += 1; /* voice_buffer.append(frame) */
            if (s->n_start >= s->start_count && s->start_len >= s->start_count) {
                int k = s->start_count;
                double ratio = (double)dq_sum_last(s->start_hist, 20, s->start_len, s->start_head, k) / (double)k;
                if (ratio >= s->start_ratio) {
                    /* _confirm_voice_start :860-869 */
                    s->active = 1;
                    s->n_start = 0;
                    s->n_end = 0;
                    if (s->buffered > 0) s->seg_samples = (long long)s->buffered * frame_len;
                    s->buffered = 0;
                    ev |= SVO_EV_START;
                }
            }
        } else {
            /* _reset_voice_start_detection :871-875 */
            s->n_start = 0;
            s->buffered = 0;
        }
    } else {
        /* _handle_ongoing_voice_activity :877-923 */
        s->seg_samples = (s->seg_samples < 0 ? 0 : s->seg_samples) + frame_len; /* _accumulate_voice_data */
        ev |= SVO_EV_CONTINUE;
        int below = p < s->end_prob;
        dq_push(s->end_hist, 100, &s->end_len, &s->end_head, (uint8_t)below);
        if (below) {
            s->n_end += 1;
            if (s->n_end >= s->end_count && s->end_len >= s->end_count) {
                int k = s->end_count;
                double ratio = (double)dq_sum_last(s->end_hist, 100, s->end_len, s->end_head, k) / (double)k;
                if (ratio >= s->end_ratio) {
                    /* _finalize_voice_segment :932-949 */
                    if (seg_samples_out) *seg_samples_out = s->seg_samples;
                    s->active = 0;
                    s->n_end = 0;
                    s->seg_samples = -1;
                    ev |= SVO_EV_END;
                }
            }
        } else {
            s->n_end = 0;
        }
    }
    return ev;
}

SVO_API int svo_sm_active(const svo_sm *s) { return s->active; }
SVO_API int svo_sm_counts(const svo_sm *s, int *n_start, int *n_end, int *buffered, long long *seg) {
    *n_start = s->n_start; *n_end = s->n_end; *buffered = s->buffered; *seg = s->seg_samples < 0 ? 0 : s->seg_samples;
    return s->active;
}

/* ------------------------------------------------------------------ WAV payload (f1) */

/* wav_writer.py:40-136, 16-bit mono: 44-byte RIFF header + clip(x*32767) -> int16 */
SVO_API size_t svo_wav_size(long long n_samples) { return 44 + 2 * (size_t)n_samples; }

static void put32(uint8_t *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = v >> 24; }
static void put16(uint8_t *p, uint16_t v) { p[0] = v & 255; p[1] = v >> 8; }

SVO_API size_t svo_write_wav16(const float *x, long long n, int sample_rate, uint8_t *out) {
    uint32_t data = (uint32_t)(2 * n);
    memcpy(out, "RIFF", 4); put32(out + 4, 36 + data); memcpy(out + 8, "WAVE", 4);
    memcpy(out + 12, "fmt ", 4); put32(out + 16, 16); put16(out + 20, 1); put16(out + 22, 1);
    put32(out + 24, (uint32_t)sample_rate); put32(out + 28, (uint32_t)sample_rate * 2); put16(out + 32, 2);
    put16(out + 34, 16); memcpy(out + 36, "data", 4); put32(out + 40, data);
    for (long long i = 0; i < n; ++i) {
        float v = x[i] * 32767.0f; /* np.clip(x*32767, -32768, 32767).astype(int16): truncation toward zero */
        v = v > 32767.f ? 32767.f : (v < -32768.f ? -32768.f : v);
        int16_t q = (int16_t)v;
        put16(out + 44 + 2 * i, (uint16_t)q);
    }
    return 44 + (size_t)data;
}

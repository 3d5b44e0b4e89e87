#!/usr/bin/env python3
"""The other BASELINE.json configs on ONE GPU (bench.py measures the headline, configs[2]):

  configs[1]  batch=1024 streams, V5
  configs[3]  batch=4096 streams mixed 8/24/48 kHz -> on-GPU resample -> V5
  configs[4]  per-GPU share of the 65 536-stream job: 4096 V4 + 4096 V5 streams (two engines, two HIP streams)

Prints one JSON object per config (frames/s, us per step).  Device-resident inputs, HIP-event timing.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cutter_vad_amd import weights_io  # noqa: E402
from cutter_vad_amd.engine import Engine  # noqa: E402

K, WU = int(os.environ.get("VAD_BENCH_K", "400")), int(os.environ.get("VAD_BENCH_WU", "50"))     # (tools/pmc_rs.sh shortens the runs under --pmc)


def blob(v):
    return open(weights_io.packaged_blob_path(v), "rb").read()


def timed(fn, streams):
    for i in range(WU):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(streams[0])
    for i in range(K):
        fn(WU + i)
    for s in streams[1:]:
        streams[0].wait_stream(s)
    e1.record(streams[0])
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / K


def config1():
    B = 1024
    eng = Engine(blob(5), max_streams=B)
    eng.open_streams(B)
    ring = (0.1 * torch.randn(32, B, 512, device="cuda")).contiguous()
    probs = torch.empty(B, device="cuda")
    ts = torch.cuda.Stream()
    dt = timed(lambda i: eng.step_device(B, ring[i % 32].data_ptr(), probs.data_ptr(), stream=ts.cuda_stream), [ts])
    eng.close()
    return {"config": "configs[1]: batch=1024, V5, 16 kHz", "us_per_step": dt * 1e6, "frames_per_s": B / dt}


def tile_shapes():
    """V5, device-resident, both tile shapes at the batch sizes where the choice matters (vad_debug_set_tile)."""
    out = []
    for B in (256, 1024, 2048, 4096, 8192):
        eng = Engine(blob(5), max_streams=B)
        eng.open_streams(B)
        ring = (0.1 * torch.randn(16, B, 512, device="cuda")).contiguous()
        probs = torch.empty(B, device="cuda")
        ts = torch.cuda.Stream()
        row = {"config": f"batch={B}, V5, device-resident", "streams": B}
        for tile in (32, 16):
            eng.set_tile(tile)
            dt = timed(lambda i: eng.step_device(B, ring[i % 16].data_ptr(), probs.data_ptr(), stream=ts.cuda_stream), [ts])
            row[f"us_per_step_tile{tile}"] = dt * 1e6
            row[f"frames_per_s_tile{tile}"] = B / dt
        eng.close()
        out.append(row)
    return out


def config3(per=None):
    """configs[3]: 4 096 streams, a third each at 8 / 24 / 48 kHz (1 365 + 1 365 + 1 366) unless `per` names another split"""
    B = 4096
    per = list(per) if per is not None else [1365, 1365, 1366]
    eng = Engine(blob(5), max_streams=B)
    eng.open_streams(B)
    rates = ((8000, 256), (24000, 768), (48000, 1536))
    rings = [(0.1 * torch.randn(8, per[k], n_in, device="cuda")).contiguous() for k, (_, n_in) in enumerate(rates)]
    total = sum(per)
    probs = torch.empty(total, device="cuda")
    ts = torch.cuda.Stream()

    def step(i):
        # ONE product call (vad_step_rates_device)
        eng.step_rates_device([(rings[k][i % 8].data_ptr(), per[k], sr) for k, (sr, _n) in enumerate(rates)], probs.data_ptr(),
                              stream=ts.cuda_stream)

    dt = timed(step, [ts])
    eng.set_tile(-1)                  # the two-launch form of the same call: resample kernel, then the model kernel
    dt2 = timed(step, [ts])
    eng.close()
    tiles = (total + 15) // 16
    return {"config": f"configs[3]: batch={total} ({' + '.join(map(str, per))}) mixed 8/24/48 kHz -> vad_step_rates_device; {tiles} 16-stream tiles "
                      "(the tiles walk the segments end to end; a tile at a rate boundary resamples its two parts in turn): "
                      + ("ONE fused launch (every tile resamples its chunks into LDS and steps from there)" if tiles <= 256 else
                         "more tiles than CUs -> resample launch + model launch"),
            "us_per_step": dt * 1e6, "frames_per_s": total / dt, "us_per_step_two_launches_forced": dt2 * 1e6}


def rates_one(which):
    """every tile at ONE input rate (4 096 streams): the fused launch's time = model tile + that rate's resample prologue"""
    per = [0, 0, 0]
    per[which] = 4096
    r = config3(per)
    r["config"] = f"4 096 streams, all at {(8000, 24000, 48000)[which]} Hz, fused resample -> step"
    return r


def rates8():
    return rates_one(0)


def rates24():
    return rates_one(1)


def rates48():
    return rates_one(2)


def config3_255_tiles():
    """configs[3] with the three thirds rounded to whole tiles (3 x 1 360 = 4 080 streams = 255 tiles): the fused launch"""
    return config3([1360, 1360, 1360])


def config3_pipelined():
    """configs[3] with the caller's software pipeline: the resample launch of tick t+1 runs on a second HIP stream beside
    the model step of tick t (4 095 streams leave half the CUs free, and the resample workgroups fit beside nothing else of
    the model kernel's).  Every tick is still resampled and then stepped in order; two 16 kHz buffers alternate."""
    B = 4096
    per = B // 3
    eng = Engine(blob(5), max_streams=B)
    eng.open_streams(B)
    rates = ((8000, 256), (24000, 768), (48000, 1536))
    rings = [(0.1 * torch.randn(8, per, n_in, device="cuda")).contiguous() for _, n_in in rates]
    f16 = [torch.empty(3 * per, 512, device="cuda") for _ in range(2)]
    probs = torch.empty(3 * per, device="cuda")
    sr_, sm_ = torch.cuda.Stream(), torch.cuda.Stream()
    resampled = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    for ev in consumed:
        ev.record(sm_)

    def step(i):
        b = i & 1
        sr_.wait_event(consumed[b])                    # the model step two ticks back has read this buffer
        eng.resample_multi_device([(rings[k][i % 8].data_ptr(), per, n_in, sr, f16[b][k * per:(k + 1) * per].data_ptr())
                                   for k, (sr, n_in) in enumerate(rates)], stream=sr_.cuda_stream)
        resampled[b].record(sr_)
        sm_.wait_event(resampled[b])
        eng.step_device(3 * per, f16[b].data_ptr(), probs.data_ptr(), stream=sm_.cuda_stream)
        consumed[b].record(sm_)

    dt = timed(step, [sm_, sr_])
    eng.close()
    return {"config": "configs[3], resample of tick t+1 on a second HIP stream beside the model step of tick t",
            "us_per_step": dt * 1e6, "frames_per_s": 3 * per / dt}


def config4_per_gpu():
    B = 4096
    # two engines share the GPU: VAD_ENGINE_SHARED_GPU keeps V5 on 32-stream tiles (128 CUs), V4's 128 tiles run beside it
    e5, e4 = Engine(blob(5), max_streams=B, shared_gpu=True), Engine(blob(4), model_version=4, max_streams=B, shared_gpu=True)
    e5.open_streams(B)
    e4.open_streams(B)
    ring = (0.1 * torch.randn(16, 2 * B, 512, device="cuda")).contiguous()
    p5, p4 = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    s5, s4 = torch.cuda.Stream(), torch.cuda.Stream()

    def step(i):
        e5.step_device(B, ring[i % 16, :B].data_ptr(), p5.data_ptr(), stream=s5.cuda_stream)
        e4.step_device(B, ring[i % 16, B:].data_ptr(), p4.data_ptr(), stream=s4.cuda_stream)

    dt = timed(step, [s5, s4])
    e5.close()
    e4.close()
    return {"config": "configs[4] per GPU: 4096 V4 + 4096 V5 streams, two engines on two HIP streams",
            "us_per_step": dt * 1e6, "frames_per_s": 2 * B / dt, "x8_gpus_frames_per_s": 16 * B / dt}


def two_pools_one_gpu():
    """configs[2]'s 8 192 streams as TWO independent pools of 4 096 on one GPU (two engines, a HIP stream each; what
    ShardedStreamPool(devices=[0, 0]) gives a serving process): the pools' launches are not ordered against each other, so the
    start of one pool's launch - cold L2, state and first weights on their way - runs under the other pool's tiles."""
    B = 4096
    R = int(os.environ.get("VAD_BENCH_RING", "16"))
    ea, eb = Engine(blob(5), max_streams=B), Engine(blob(5), max_streams=B)
    ea.open_streams(B)
    eb.open_streams(B)
    ring = (0.1 * torch.randn(R, 2 * B, 512, device="cuda")).contiguous()
    pa, pb = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    fa = [ring[k, :B].data_ptr() for k in range(R)]        # pointers taken once: two tensor views per step would cost the host
    fb = [ring[k, B:].data_ptr() for k in range(R)]        # more than the launches it has to stay ahead of
    qa, qb, ha, hb = pa.data_ptr(), pb.data_ptr(), sa.cuda_stream, sb.cuda_stream

    def step(i):
        ea.step_device(B, fa[i % R], qa, stream=ha)
        eb.step_device(B, fb[i % R], qb, stream=hb)

    dt = timed(step, [sa, sb])
    one = Engine(blob(5), max_streams=2 * B)
    one.open_streams(2 * B)
    p1 = torch.empty(2 * B, device="cuda")
    q1 = p1.data_ptr()
    dt1 = timed(lambda i: one.step_device(2 * B, fa[i % R], q1, stream=ha), [sa])
    for e in (ea, eb, one):
        e.close()
    return {"config": "8192 V5 streams as two independent pools of 4096 on one GPU (two engines, two HIP streams)",
            "us_per_step": dt * 1e6, "frames_per_s": 2 * B / dt, "us_per_step_one_pool_of_8192": dt1 * 1e6,
            "frames_per_s_one_pool_of_8192": 2 * B / dt1}


def v4_alone():
    B = 8192
    eng = Engine(blob(4), model_version=4, max_streams=B)
    eng.open_streams(B)
    ring = (0.1 * torch.randn(16, B, 512, device="cuda")).contiguous()
    probs = torch.empty(B, device="cuda")
    ts = torch.cuda.Stream()
    dt = timed(lambda i: eng.step_device(B, ring[i % 16].data_ptr(), probs.data_ptr(), stream=ts.cuda_stream), [ts])
    eng.close()
    return {"config": "batch=8192, V4, 16 kHz (device-resident)", "us_per_step": dt * 1e6, "frames_per_s": B / dt,
            "frac_of_fp32_peak_at_1.38_MFLOP_per_frame": 1.38e6 * B / dt / 157.3e12}


def v4_tile_shapes():
    """V4 (both sub-models), device-resident, both tile shapes (vad_debug_set_tile): the 32-stream kernel with one wave per SIMD
    against 16-stream tiles, two workgroups per CU."""
    out = []
    for sr in (16000, 8000):
        for B in (256, 1024, 2048, 4096, 8192):
            eng = Engine(open(weights_io.packaged_blob_path(4, sr), "rb").read(), model_version=4, max_streams=B, sample_rate=sr)
            eng.open_streams(B)
            ring = (0.1 * torch.randn(16, B, 512, device="cuda")).contiguous()
            probs = torch.empty(B, device="cuda")
            ts = torch.cuda.Stream()
            row = {"config": f"batch={B}, V4 {sr} Hz sub-model, device-resident", "streams": B}
            for tile in (32, 16):
                eng.set_tile(tile)
                dt = timed(lambda i: eng.step_device(B, ring[i % 16].data_ptr(), probs.data_ptr(), stream=ts.cuda_stream), [ts])
                row[f"us_per_step_tile{tile}"] = dt * 1e6
                row[f"frames_per_s_tile{tile}"] = B / dt
            eng.close()
            out.append(row)
    return out


def resampler_alone():
    """vadk_resample_512 alone: 4096 chunks per launch, device-resident, per input rate."""
    B = 4096
    eng = Engine(blob(5), max_streams=64)
    out = []
    ts = torch.cuda.Stream()
    y = torch.empty(B, 512, device="cuda")
    for sr, n_in in ((8000, 256), (24000, 768), (48000, 1536)):
        x = (0.1 * torch.randn(4, B, n_in, device="cuda")).contiguous()
        lib = eng._lib

        def step(i):
            assert lib.vad_resample_device(eng.handle, x[i % 4].data_ptr(), B, n_in, sr, y.data_ptr(), ts.cuda_stream) == 0

        dt = timed(step, [ts])
        flop = 2.0 * 512 * n_in * B
        out.append({"config": f"resampler alone: {B} chunks of {n_in} samples ({sr} Hz) -> 512", "us_per_launch": dt * 1e6,
                    "chunks_per_s": B / dt, "TFLOP_s_dense_operator": flop / dt / 1e12,
                    "frac_of_fp32_peak": flop / dt / 157.3e12})
    eng.close()
    return out


def v4_8k():
    B = 8192
    eng = Engine(open(weights_io.packaged_blob_path(4, 8000), "rb").read(), model_version=4, max_streams=B,
                 sample_rate=8000)
    eng.open_streams(B)
    ring = (0.1 * torch.randn(16, B, 512, device="cuda")).contiguous()
    probs = torch.empty(B, device="cuda")
    ts = torch.cuda.Stream()
    dt = timed(lambda i: eng.step_device(B, ring[i % 16].data_ptr(), probs.data_ptr(), stream=ts.cuda_stream), [ts])
    eng.close()
    return {"config": "batch=8192, V4 8 kHz sub-model (a9; two LSTM steps per frame)", "us_per_step": dt * 1e6, "frames_per_s": B / dt}


def v5_8k():
    """Silero V5's 8 kHz sub-model: native 8 kHz audio in 256-sample frames (32 ms as well), 32-stream tiles."""
    B = 8192
    eng = Engine(open(weights_io.packaged_blob_path(5, 8000), "rb").read(), model_version=5, max_streams=B, sample_rate=8000)
    eng.open_streams(B)
    ring = (0.1 * torch.randn(16, B, 256, device="cuda")).contiguous()
    probs = torch.empty(B, device="cuda")
    ts = torch.cuda.Stream()
    dt = timed(lambda i: eng.step_device(B, ring[i % 16].data_ptr(), probs.data_ptr(), stream=ts.cuda_stream), [ts])
    eng.close()
    return {"config": "batch=8192, V5 8 kHz sub-model (256-sample frames)", "us_per_step": dt * 1e6, "frames_per_s": B / dt}


def v5_8k_tile_shapes():
    """Silero V5's 8 kHz sub-model on 16- against 32-stream tiles, 256 ... 8 192 streams (the engine picks 16-stream tiles up to 4 096)."""
    out = []
    eng = Engine(open(weights_io.packaged_blob_path(5, 8000), "rb").read(), model_version=5, max_streams=8192, sample_rate=8000)
    eng.open_streams(8192)
    ring = (0.1 * torch.randn(16, 8192, 256, device="cuda")).contiguous()
    probs = torch.empty(8192, device="cuda")
    ts = torch.cuda.Stream()
    for B in (256, 1024, 4096, 8192):
        row = {"config": f"batch={B}, V5 8 kHz sub-model, device-resident", "streams": B}
        for tile in (32, 16):
            eng.set_tile(tile)
            dt = timed(lambda i: eng.step_device(B, ring[i % 16].data_ptr(), probs.data_ptr(), stream=ts.cuda_stream), [ts])
            row[f"us_per_step_tile{tile}"] = dt * 1e6
            row[f"frames_per_s_tile{tile}"] = B / dt
        out.append(row)
    eng.close()
    return out


def int16_ingest():
    """Device-resident int16 frames (the wire format of the serving path) against float32 ones, V5 and V4, 8 192 streams."""
    from cutter_vad_amd import _ffi
    out = []
    B = 8192
    for v in (5, 4):
        eng = Engine(blob(v), model_version=v, max_streams=B)
        eng.open_streams(B)
        f32 = (0.1 * torch.randn(16, B, 512, device="cuda")).contiguous()
        i16 = (f32 * 32767.0).round().clamp(-32768, 32767).to(torch.int16).contiguous()
        probs = torch.empty(B, device="cuda")
        ts = torch.cuda.Stream()
        row = {"config": f"batch={B}, V{v}, device-resident frames, float32 vs int16 (/32767) ingest"}
        for name, ring, fmt in (("f32", f32, _ffi.VAD_FMT_F32), ("i16", i16, _ffi.VAD_FMT_I16_32767)):
            dt = timed(lambda i: eng.step_device(B, ring[i % 16].data_ptr(), probs.data_ptr(), fmt=fmt, stream=ts.cuda_stream), [ts])
            row[f"us_per_step_{name}"] = dt * 1e6
        eng.close()
        out.append(row)
    return out


def single_stream_wrapper():
    """configs[0]: ONE stream through the drop-in VADWrapper (host framing + one launch + sync + callbacks per chunk)."""
    import time
    import numpy as np
    from cutter_vad_amd import VADConfig, VADWrapper
    pcm = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                               "speech16k_i16.npz"))["pcm"]
    x = pcm.astype(np.float32) / 32767.0
    out = []
    for chunk in (512, 480, 1024):
        n = min(x.size // chunk, 600)
        seg = [0]
        with VADWrapper(VADConfig(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=6,
                                  voice_end_frame_count=12, buffer_size=chunk if chunk < 512 else 512)) as w:
            w.set_callbacks(None, lambda wav: seg.__setitem__(0, seg[0] + 1), None)
            for i in range(20):
                w.process_audio_data(x[i * chunk:(i + 1) * chunk])
            w.reset()
            t0 = time.perf_counter()
            for i in range(n):
                w.process_audio_data(x[i * chunk:(i + 1) * chunk])
            dt = (time.perf_counter() - t0) / n
            frames = w.get_statistics()["total_frames_processed"]
        out.append({"config": f"configs[0]: single stream, VADWrapper.process_audio_data, {chunk}-sample chunks",
                    "us_per_chunk": dt * 1e6, "frames_per_chunk": frames / n, "real_time_factor": (chunk / 16000) / dt,
                    "segments": seg[0]})
    return out


def host_api():
    """PCIe-inclusive: vad_step with HOST pointers (pageable numpy arrays): H2D frames, kernel, D2H probs, sync."""
    import time
    import numpy as np
    B = 8192
    out = []
    eng = Engine(blob(5), max_streams=B)
    slots = eng.open_streams(B)
    rng = np.random.default_rng(0)
    x32 = (0.1 * rng.standard_normal((4, B, 512))).astype(np.float32)
    x16 = np.clip(x32 * 32767.0, -32768, 32767).astype(np.int16)
    p32, p16 = eng.pinned_array(x32.shape, np.float32), eng.pinned_array(x16.shape, np.int16)
    p32[:], p16[:] = x32, x16
    for name, x in (("f32", x32), ("int16", x16), ("f32 (page-locked: Engine.pinned_array)", p32),
                    ("int16 (page-locked: Engine.pinned_array)", p16)):
        for i in range(5):
            eng.step(slots, x[i % 4])
        t0 = time.perf_counter()
        n = 40
        for i in range(n):
            eng.step(slots, x[i % 4])
        dt = (time.perf_counter() - t0) / n
        where = "" if "page-locked" in name else " in pageable host memory"
        out.append({"config": f"batch=8192, V5, host-pointer API (vad_step, {name} frames{where}, "
                              "H2D + kernel + D2H + sync per step)", "us_per_step": dt * 1e6, "frames_per_s": B / dt,
                    "h2d_GBps": x[0].nbytes / dt / 1e9})
    eng.close()
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:                      # e.g. `bench_configs.py config3 config3_pipelined`
        for name in sys.argv[1:]:
            r = globals()[name]()
            for row in (r if isinstance(r, list) else [r]):
                print(json.dumps(row), flush=True)
        sys.exit(0)
    for r in host_api():
        print(json.dumps(r), flush=True)
    for r in single_stream_wrapper():
        print(json.dumps(r), flush=True)
    for r in resampler_alone():
        print(json.dumps(r), flush=True)
    for fn in (config1, config3, config3_255_tiles, config3_pipelined, config4_per_gpu, two_pools_one_gpu, v4_alone, v4_8k):
        print(json.dumps(fn()), flush=True)

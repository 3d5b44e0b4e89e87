#!/usr/bin/env python3
"""Headline benchmark: 512-sample frames/sec, Silero V5 16 kHz, batch = 8192 streams per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: every one of the 8192 resident streams
of this GPU advances by one 512-sample frame (denoise gate -> V5 -> probability -> state
machine), inputs already resident in HBM (ring [32][B][512] f32).  Streams are independent
(SURVEY §8 e): ranks shard them with NO data-path collective; torch.distributed is used only for
the barrier around the timed region and the max-over-ranks of the elapsed time ("weak" scaling:
per-GPU work is fixed).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (torchrun, before this process
touches a GPU) and refuses to run if fewer than N devices are visible: it never reports n_gpus smaller than asked.
`--mix v4v5` is BASELINE.json's configs[4] per-GPU share: 4096 Silero V4 + 4096 Silero V5 streams, two engines stepping
concurrently on two HIP streams (default: configs[2], all V5).

The JSON line carries, besides the driver's contract:
  roofline     - dominant kernel (silero_v5_step) against the fp32 MFMA peak, from HIP events
                 recorded on the launch stream around the timed region;
  cpu_baseline - the oracle's C port (oracle/silero_oracle.c, float accumulators) timed on this
                 host's cores on a bounded sample of the same workload (rank 0, N=1 only), which
                 also yields the in-run parity figure; `single_thread` inside it is the reference's own
                 configuration (one stream, one thread: silero_model.py:316-317).
Before the W warm-up steps an untimed clock-ramp preamble runs the same step for ~0.4 s (`preamble_steps` in the line):
with W = 5 and K = 20 the whole measurement is 1.4 ms long and would otherwise be taken while the GPU clocks are still rising.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_PER_GPU = 8192
RING = 32
FLOP_PER_FRAME = {5: 988160, 4: 1380000}   # SURVEY §8 d: V5 494 080 valid-tap MAC; V4 ~0.69 M MAC
BYTES_PER_FRAME = 4100                     # SURVEY §8 d: 2048 in + 1024 state R + 1024 state W + 4 prob
# What the V5 kernel EXECUTES per frame.  One-frame calls run on 16-stream tiles (silero_v5_step16, two workgroups per CU; engine.cpp
# launch()): 1 178 v_mfma_f32_16x16x4_f32 (1 024 MAC each) per wave - recurrent half 256, folded DFT 144 (the even bins fold three
# times), Toom-3 enc0 330, enc1 128, enc2 32, enc3 32, LSTM input half 256 - x 4 waves / 16 streams (the SQ counter
# SQ_INSTS_VALU_MFMA_F32 says the same: profiles/).  The 32-stream kernel (multi-frame calls of more than 4 096 streams, the mix)
# executes 1 173 units of 2 048 MAC per wave and 32 streams: the same count to 0.4 %.  Fewer than the algorithmic count because
# the folds and the Toom-3 product are exact algebraic reductions of the sums.
EXECUTED_MFMA_16X16X4_PER_WAVE = 1178
EXECUTED_FLOP_PER_FRAME = EXECUTED_MFMA_16X16X4_PER_WAVE * 4 * 1024 * 2 // 16
PEAK_FP32_MFMA = 157.3e12        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM = 8.0e12                # same guide, HBM3E spec
PARITY_STEPS = 4
CPU_STREAMS = 2048
PREAMBLE_S = 0.4


def synth_ring(first_stream: int, n: int, threads: int = 1) -> np.ndarray:
    """BASELINE.md §4 generator -> [RING][n][512] float32 (one contiguous [n,512] batch per step).  Every stream has its own
    seeded generator, so the streams are split over `threads` host threads (numpy's generators release the GIL): 8 ranks
    starting together on one node each take their share of the cores instead of ~8 s of one core each."""
    from tests.signals import make_streams
    threads = max(1, min(int(threads), n // 256 or 1))
    if threads == 1:
        return np.ascontiguousarray(make_streams(n, RING, seed=1234, first_stream=first_stream).transpose(1, 0, 2))
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty((RING, n, 512), np.float32)
    edges = [n * k // threads for k in range(threads + 1)]

    def part(k: int) -> None:
        lo, hi = edges[k], edges[k + 1]
        out[:, lo:hi] = make_streams(hi - lo, RING, seed=1234, first_stream=first_stream + lo).transpose(1, 0, 2)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(part, range(threads)))
    return out


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pe = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = max(1, min(cores, int(q / pe + 0.5)))
        except Exception:
            pass
    return cores


def cpu_leg(ring: np.ndarray, gpu_probs: np.ndarray, budget_s: float = 12.0) -> tuple:
    """Time the oracle port on host cores (same frames, same gate) and compare with the GPU."""
    from cutter_vad_amd import weights_io
    from oracle import oracle
    with open(weights_io.packaged_blob_path(5), "rb") as f:
        blob = f.read()
    threads = min(usable_cores(), 256)
    om = oracle.OracleModel(blob, "f32")
    n = min(CPU_STREAMS, ring.shape[1])
    st = np.zeros((n, 256), np.float32)
    worst, total_dp, cnt = 0.0, 0.0, 0
    frames_done, elapsed, step = 0, 0.0, 0
    while True:
        x = oracle.denoise(ring[step % RING, :n]).reshape(n, 512)
        t0 = time.perf_counter()
        p = om.step_batch(x, st, nthreads=threads)
        elapsed += time.perf_counter() - t0
        frames_done += n
        if step < gpu_probs.shape[0]:
            d = np.abs(p - gpu_probs[step, :n])
            worst = max(worst, float(d.max()))
            total_dp += float(d.sum())
            cnt += d.size
        step += 1
        if (elapsed >= budget_s and step >= gpu_probs.shape[0]) or step >= 400:
            break
    # the reference's own configuration: ONE stream per session, intra_op = inter_op = 1 (silero_model.py:316-317)
    st1 = np.zeros(256, np.float32)
    x1 = [oracle.denoise(ring[k % RING, 1]).reshape(512) for k in range(RING)]
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(256):
            om.step(x1[k % RING], st1)
            k += 1
    single = k / (time.perf_counter() - t0)
    base = {
        "value": frames_done / elapsed,
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n} streams x {step} steps of the same workload (oracle/silero_oracle.c, float accumulators, "
                  f"{threads} pthreads; onnxruntime is not installed on this box)",
        "single_thread": {"value": single, "unit": "frames/s", "cores": 1,
                          "sample": f"1 stream x {k} sequential frames, 1 thread: the reference's session options "
                                    "(batch 1, intra_op = inter_op = 1, silero_model.py:316-317), same C port"},
    }
    parity = {"max_abs_dp": worst, "mean_abs_dp": total_dp / max(cnt, 1), "frames": cnt,
              "against": "oracle C port (f32) on identical inputs", "bar": 1e-4}
    return base, parity


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """`--gpus N` without a launcher: start the N ranks ourselves.  Runs BEFORE any GPU call in this process (counting
    devices does not initialise the GPU); the parent only relays the children's output and exit code."""
    fake = os.environ.get("VAD_BENCH_FAKE") == "1"
    if not fake and os.environ.get("VAD_BENCH_SINGLE_DEVICE") != "1":
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to run a smaller job "
                             "under the requested name\n")
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


class FakeStepper:
    """CPU rehearsal of the launcher / timing / aggregation plumbing (tests/test_sharding.py, VAD_BENCH_FAKE=1): a "step"
    sleeps 1 ms.  Never used on a GPU box; its JSON line says so."""
    kernel = "fake (CPU rehearsal of the launch path)"

    def __init__(self, B):
        self.B = B

    def step(self, i):
        time.sleep(0.001)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--streams", type=int, default=B_PER_GPU, help="streams per GPU (headline: 8192)")
    ap.add_argument("--mix", choices=["v5", "v4v5"], default="v5", help="v4v5 = configs[4] per-GPU share (4096 V4 + 4096 V5)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--pools", type=int, default=1, choices=[1, 2],
                    help="2 = the GPU's streams as two independent pools (an engine and a HIP stream each; what "
                         "ShardedStreamPool(devices=[0, 0]) gives a serving process): the pools' launches are not ordered against "
                         "each other.  NOT the headline: the default is one pool, one launch per step")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    # stdout carries the ONE JSON line and nothing else: gloo and RCCL announce themselves on fd 1 from C ("[Gloo] Rank 0 is
    # connected ...", "RCCL version : ...") - from here on fd 1 is stderr, the line goes to the kept copy of the real stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from cutter_vad_amd import sharding
    info = sharding.RankInfo.from_env()
    rank, local_rank, world = info.rank, info.local_rank, info.world
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    fake = os.environ.get("VAD_BENCH_FAKE") == "1"
    if not fake and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # rehearsal knobs for a 1-GPU box (never set by the driver): all ranks on device 0, gloo for the two
    # control-plane collectives.  The real multi-GPU run is one rank per GPU over RCCL.
    if os.environ.get("VAD_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    prefer = "gloo" if fake and "VAD_BENCH_BACKEND" not in os.environ else os.environ.get("VAD_BENCH_BACKEND", "nccl")   # "nccl" == RCCL on ROCm
    device = None
    if not fake:
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    # control plane: gloo always, RCCL on top when every rank's probe all-reduce works (sharding.ControlPlane); a failing RCCL
    # never ends the job - the line says which one carried the barrier and the MAX
    cp = sharding.ControlPlane(info, prefer, device)
    # one rank per GPU, really: two ranks on one device would report n_gpus = N over fewer GPUs
    # identity = what this rank asked for (device ordinal under its visibility masks), not a property the runtime reports:
    # a runtime that reported one uuid for all GPUs must not be able to stop a legitimate run
    masks = "|".join(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    pci = None                                         # "dddd:bb:dd" of the device this rank steps, when the runtime tells
    if not fake:
        pr = torch.cuda.get_device_properties(local_rank)
        ident = (socket.gethostname(), masks, local_rank)
        bus = getattr(pr, "pci_bus_id", None)
        if isinstance(bus, int):
            pci = f"{int(getattr(pr, 'pci_domain_id', 0) or 0):04x}:{bus:02x}:{int(getattr(pr, 'pci_device_id', 0) or 0):02x}"
        shown = f"cuda:{local_rank} pci {bus:02x}" if isinstance(bus, int) else f"cuda:{local_rank}"
    else:
        ident = (socket.gethostname(), masks, int(os.environ.get("VAD_BENCH_FAKE_DEVICE", local_rank)))
        shown = f"fake:{ident[2]}"
        if os.environ.get("VAD_BENCH_FAKE_PCI"):       # "a,b,..." per rank (tests: two ranks that report one PCI id)
            pci = os.environ["VAD_BENCH_FAKE_PCI"].split(",")[rank]
    single = os.environ.get("VAD_BENCH_SINGLE_DEVICE") == "1"
    dup = sharding.duplicate_devices(cp.gather(ident))
    if dup and not single:
        if rank == 0:
            sys.stderr.write(f"bench.py: ranks {dup[0][0]} and {dup[0][1]} map to the same GPU {dup[0][2]}; refusing to report "
                             f"n_gpus = {world} for fewer devices (LOCAL_RANK -> device map)\n")
        cp.close()
        return 2
    # ... and the hardware's own word for it: with more than one rank the PCI addresses must be pairwise distinct (the identity
    # above is what the ranks ASKED for; two ordinals that resolve to one physical device would pass it)
    pcis = cp.gather((socket.gethostname(), pci))
    if world > 1 and not single and all(p[1] is not None for p in pcis):
        dup = sharding.duplicate_devices(pcis)
        if dup:
            if rank == 0:
                sys.stderr.write(f"bench.py: ranks {dup[0][0]} and {dup[0][1]} step the same physical GPU (PCI {dup[0][2][1]}); refusing "
                                 f"to report n_gpus = {world} for fewer devices\n")
            cp.close()
            return 2

    B = args.streams
    first_stream, _ = sharding.stream_shard(world * B, world, rank)   # weak scaling: B streams per rank
    host_enqueue = [0.0]
    if fake:
        stepper, sync, kernel_s = FakeStepper(B), (lambda: None), [None]
        for i in range(args.warmup):
            stepper.step(i)

        def run():
            for i in range(args.steps):
                stepper.step(i)
        elapsed, own = sharding.timed_region_detail(cp, run, sync)
        preamble, ring_h, gpu_probs, versions = 0, None, None, ([5] * args.pools if args.mix == "v5" else [5, 4])
    else:
        from cutter_vad_amd import weights_io
        from cutter_vad_amd.engine import Engine
        versions = [5] * args.pools if args.mix == "v5" else [5, 4]
        nb = B // len(versions)                       # streams per engine
        engines, streams, probs, events = [], [], [], []
        for v in versions:
            with open(weights_io.packaged_blob_path(v), "rb") as f:
                e = Engine(f.read(), model_version=v, device_id=local_rank, max_streams=nb, shared_gpu=args.mix != "v5")
            e.open_streams(nb)                        # slots 0..nb-1, zero state, default thresholds
            engines.append(e)
            streams.append(torch.cuda.Stream())
            probs.append(torch.empty(nb, device="cuda"))
            events.append(torch.empty(nb, dtype=torch.uint8, device="cuda"))
        ring_h = synth_ring(first_stream, B, threads=max(1, usable_cores() // world))
        ring = torch.from_numpy(ring_h).cuda()
        torch.cuda.synchronize()
        ptrs = [[ring[i, k * nb:(k + 1) * nb].data_ptr() for i in range(RING)] for k in range(len(versions))]
        ts = streams[0]

        def step(i: int) -> None:
            for k, e in enumerate(engines):
                e.step_device(nb, ptrs[k][i % RING], probs[k].data_ptr(), d_events=events[k].data_ptr(), denoise=0.01,
                              stream=streams[k].cuda_stream)

        def join() -> None:                           # the other engines' streams meet on the first one (mix only)
            for s in streams[1:]:
                ts.wait_stream(s)

        gpu_probs = torch.empty(PARITY_STEPS, B, device="cuda")
        for i in range(PARITY_STEPS):                 # from zero state: these frames are what the CPU leg replays
            step(i)
            for k in range(len(versions)):
                with torch.cuda.stream(streams[k]):
                    gpu_probs[i, k * nb:(k + 1) * nb].copy_(probs[k])
        torch.cuda.synchronize()
        # clock ramp (untimed, not part of W): the same step until PREAMBLE_S of wall time have passed
        t0, preamble = time.perf_counter(), 0
        while time.perf_counter() - t0 < PREAMBLE_S:
            for _ in range(64):
                step(PARITY_STEPS + preamble)
                preamble += 1
            torch.cuda.synchronize()
        base = PARITY_STEPS + preamble
        for i in range(args.warmup):
            step(base + i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def run() -> None:
            for s in streams[1:]:
                s.wait_stream(ts)
            e0.record(ts)
            h0 = time.perf_counter()
            for i in range(args.steps):
                step(base + args.warmup + i)
            host_enqueue[0] = time.perf_counter() - h0
            join()
            e1.record(ts)

        elapsed, own = sharding.timed_region_detail(cp, run, torch.cuda.synchronize)
        kernel_s = [e0.elapsed_time(e1) * 1e-3 / args.steps]     # avg launch duration on the launch stream(s)
        for p in probs:
            assert bool(torch.isfinite(p).all()) and float(p.min()) >= 0.0 and float(p.max()) <= 1.0

    # every rank's own figures, so that a straggler is visible behind the MAX
    per_rank = cp.gather({"rank": rank, "device": shown, "pci": pci,
                          "kernel_us": None if kernel_s[0] is None else kernel_s[0] * 1e6,
                          "ms_per_step": own / args.steps * 1e3, "frames_per_s": B * args.steps / own})
    if rank == 0:
        value = sharding.aggregate_rate(B, args.steps, world, elapsed)
        mixed = args.mix != "v5"
        workload = ("configs[2]: batch=8192 concurrent streams per GPU, Silero V5, 16 kHz, one 512-sample frame per stream "
                    "per step, denoise gate 0.01, state machine on") if not mixed else (
                    f"configs[4]: batch={world * B} streams sharded {B}/GPU across {world} x MI355X, Silero V4 + V5 mixed - per GPU "
                    f"{B // 2} Silero V4 + {B // 2} Silero V5 streams (two engines, two HIP streams), 16 kHz, one 512-sample frame per "
                    "stream per step, denoise gate 0.01, state machines on" + ("" if world == 8 and B == B_PER_GPU else
                    f" [configs[4] as stated is 8 GPUs x {B_PER_GPU}: this run is its {world}-GPU share]"))
        if not mixed and args.pools > 1:
            workload += (f" - as {args.pools} independent pools of {B // args.pools} streams per GPU (an engine and a HIP stream each; the "
                         "pools' launches are not ordered against each other, every stream's own frames stay in order)")
        # how evenly the ranks ran: behind the MAX the slowest rank sets the line's time
        kus = [r["kernel_us"] for r in per_rank if r["kernel_us"] is not None]
        best = max(r["frames_per_s"] for r in per_rank)
        ranks = {"kernel_us": ({"min": min(kus), "max": max(kus), "mean": sum(kus) / len(kus)} if kus else None),
                 "best_rank_frames_per_s": best,
                 # value / (N x the best single rank's own rate): 1.0 = every rank as fast as the fastest and no time lost at the
                 # barrier.  NOT the scaling efficiency across N - the driver computes that from the per-N lines.
                 "scaling_efficiency": value / (world * best)}
        out = {
            "metric": "512-sample frames/sec, Silero V5 16kHz, batch=8192 streams per GPU" if not mixed else
                      "512-sample frames/sec, Silero V4 + V5 mixed, 8192 streams per GPU",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "streams_per_gpu": B,
                "streams_total": world * B,
                "frames_per_step_per_gpu": B,
                "ring_frames": RING,
                "sharding": f"{world} independent per-GPU stream pools, no collective",
            },
            "preamble_steps": preamble,
            "control_plane": cp.backend or "none (single process)",
            "per_rank": per_rank,
            "ranks": ranks,
        }
        if cp.fallback_reason:
            out["control_plane_fallback"] = cp.fallback_reason
        if fake:
            out["data"] = "none (VAD_BENCH_FAKE=1: CPU rehearsal of the launch path, not a measurement)"
            out["roofline"] = None
        else:
            ks = kernel_s[0]
            flop = sum(FLOP_PER_FRAME[v] for v in versions) * (B // len(versions))
            achieved = flop / ks
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic_mix.json" if mixed else "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    traffic = rec.get("hbm_bytes_per_step") if mixed else rec.get("hbm_bytes_per_launch")
                    traffic_src = (f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                   f"build {rec.get('build', 'see profiles/README.md')}; not re-measured in this run)")
                except Exception:
                    traffic = None
            out["roofline"] = {
                "bound": "mfma",
                "kernel": ("silero_v5_step16 (16-stream tiles, two workgroups per CU)" if not mixed and args.pools == 1
                           else f"silero_v5_step16 x {args.pools} pools (concurrent; kernel_us = the step, not one launch)" if not mixed
                           else "silero_v5_step + silero_v4_step16 (concurrent)"),
                "achieved": achieved / 1e12,
                "peak": PEAK_FP32_MFMA / 1e12,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_us": ks * 1e6,
                "host_enqueue_us_per_launch": host_enqueue[0] / args.steps / len(versions) * 1e6,
                "algorithmic_flop_per_launch": flop,
                "note": "frac = ALGORITHMIC FLOPs (the graph's dense sums, SURVEY 8d) / launch time / peak, as the bench "
                        "contract prescribes; the V5 kernel executes 61 % of them (folded DFT, Toom-3 enc0), so frac can "
                        "exceed 1 while the MFMA pipe is busy executed_frac_of_peak of the time: read executed_frac_of_peak "
                        "as the utilisation",
                "hbm_algorithmic_GBps": BYTES_PER_FRAME * B / ks / 1e9,
                "hbm_frac": BYTES_PER_FRAME * B / ks / PEAK_HBM,
            }
            if not mixed:
                out["roofline"]["executed_mfma_flop_per_launch"] = EXECUTED_FLOP_PER_FRAME * B
                out["roofline"]["executed_frac_of_peak"] = EXECUTED_FLOP_PER_FRAME * B / ks / PEAK_FP32_MFMA
            if world == 1 and not args.no_cpu and not mixed:
                base_, parity = cpu_leg(ring_h, gpu_probs.cpu().numpy())
                out["cpu_baseline"] = base_
                out["parity"] = parity
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    cp.close()
    if not fake:
        for e in engines:
            e.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

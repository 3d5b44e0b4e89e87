"""N > 1 path on CPU: two gloo ranks run the same plumbing bench.py uses on RCCL — static stream
shards with no data-path collective, barrier + MAX-reduced timing, whole-job aggregation."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from cutter_vad_amd import sharding


def test_stream_shards_partition_the_streams():
    for total, world in ((65536, 8), (8192, 1), (10, 3), (7, 8), (0, 2)):
        seen = []
        for r in range(world):
            lo, hi = sharding.stream_shard(total, world, r)
            assert 0 <= lo <= hi <= total
            seen.extend(range(lo, hi))
            for s in (lo, hi - 1):
                if lo < hi:
                    assert sharding.owner_of(s, total, world) == r
        assert seen == list(range(total))
    assert sharding.stream_shard(65536, 8, 3) == (3 * 8192, 4 * 8192)     # config 5: 8192 per GPU
    with pytest.raises(ValueError):
        sharding.stream_shard(10, 2, 2)
    assert sharding.aggregate_rate(8192, 200, 8, 0.02) == pytest.approx(8 * 8192 * 200 / 0.02)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    info = sharding.RankInfo.from_env()
    dist = sharding.init_process_group(info, "gloo")
    per_rank = 64
    lo, hi = sharding.stream_shard(world * per_rank, world, info.rank)
    # a rank only ever touches its own streams: a fake "step" that advances per-stream counters
    state = np.zeros(hi - lo, np.int64)
    calls = {"n": 0}

    def run():
        import time
        for _ in range(5):
            state[:] += 1
            calls["n"] += 1
        time.sleep(0.05 * (rank + 1))          # uneven ranks: the MAX must win

    elapsed = sharding.timed_region(dist, run, lambda: None)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([lo, hi, elapsed, calls["n"], state.sum()], np.float64))
    dist.destroy_process_group()


def test_two_gloo_ranks_share_nothing_but_the_clock(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npy") for k in range(world)]
    assert (r[0][0], r[0][1], r[1][0], r[1][1]) == (0, 64, 64, 128)           # disjoint, covering
    assert r[0][2] == r[1][2] and r[0][2] >= 0.1                              # both see the slowest rank's time
    assert r[0][3] == r[1][3] == 5 and r[0][4] == r[1][4] == 5 * 64
    assert sharding.aggregate_rate(64, 5, world, r[0][2]) == pytest.approx(2 * 64 * 5 / r[0][2])


def _cp_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    info = sharding.RankInfo.from_env()
    cp = sharding.ControlPlane(info, "nccl", None)        # no GPU here: the RCCL probe fails on every rank -> gloo, same process
    import time
    total, own = sharding.timed_region_detail(cp, lambda: time.sleep(0.05 * (rank + 1)), lambda: None)
    recs = cp.gather({"rank": rank, "own": own})
    with open(os.path.join(out_dir, f"cp{rank}.txt"), "w") as f:
        f.write(repr((cp.backend, cp.fallback_reason, total, own, [r["rank"] for r in recs], cp.max(float(rank)))))
    cp.close()


def test_control_plane_falls_back_to_gloo_in_process_when_rccl_cannot_start(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_cp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [eval(open(tmp_path / f"cp{k}.txt").read()) for k in range(world)]
    for k, (backend, why, total, own, ranks, mx) in enumerate(got):
        assert backend == "gloo" and "rank 0:" in why and len(why) > 12           # every rank knows why, and agrees
        assert ranks == [0, 1] and mx == 1.0
        assert total >= own and total == got[0][2] and total >= 0.1             # MAX over ranks, identical everywhere
    assert got[0][3] < got[1][3]                                                 # each rank also keeps its own time
    one = sharding.ControlPlane(sharding.RankInfo(0, 0, 1), "nccl", None)        # a single process never opens a group
    assert one.backend is None and not one.active and one.gather(5) == [5] and one.max(2.5) == 2.5
    one.barrier()
    one.close()
    assert sharding.duplicate_devices([("h", "", 0), ("h", "", 1), ("h", "", 0)]) == [(0, 2, ("h", "", 0))]
    assert sharding.duplicate_devices([("h", "0", 0), ("h", "1", 0)]) == []      # one visible GPU each: different devices


def test_single_process_needs_no_process_group():
    assert sharding.init_process_group(sharding.RankInfo(0, 0, 1), "gloo") is None
    t = sharding.timed_region(None, lambda: None, lambda: None)
    assert 0 <= t < 0.1


# ------------------------------------------------------------------ bench.py's own launcher (`--gpus N` without torchrun)
def _bench(*argv, **env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, env=e,
                          timeout=300)


def test_bench_gpus_n_launches_n_ranks_itself_and_reports_n():
    """VAD_BENCH_FAKE=1: the launch / barrier / MAX-of-ranks / aggregation path of bench.py with a sleeping step (gloo)."""
    import json
    r = _bench("--gpus", "2", "--steps", "20", "--warmup", "2", VAD_BENCH_FAKE="1")
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1                                    # rank 0 prints, once
    # ... and stdout is that line alone: gloo / RCCL announce themselves on fd 1 from C, bench.py points fd 1 at stderr
    assert r.stdout.strip().splitlines() == [x for x in r.stdout.splitlines() if x.startswith("{")] and "[Gloo]" not in r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] == pytest.approx(2 * 8192 * 20 / (out["ms_per_step"] * 1e-3 * 20))
    assert out["ms_per_step"] >= 1.0 and "FAKE" in out["data"] and out["roofline"] is None
    # a straggler is visible: one record per rank, the MAX is the line's time
    assert out["control_plane"] == "gloo" and [r["rank"] for r in out["per_rank"]] == [0, 1]
    assert max(r["ms_per_step"] for r in out["per_rank"]) == pytest.approx(out["ms_per_step"], rel=1e-6)
    assert {r["device"] for r in out["per_rank"]} == {"fake:0", "fake:1"}


def test_bench_survives_a_failing_rccl_and_refuses_two_ranks_on_one_device():
    import json
    r = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", VAD_BENCH_FAKE="1", VAD_BENCH_BACKEND="nccl")
    assert r.returncode == 0, r.stdout + r.stderr
    out = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][0]
    assert out["n_gpus"] == 2 and out["control_plane"] == "gloo" and out["control_plane_fallback"].startswith("rank 0:")
    r = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", VAD_BENCH_FAKE="1", VAD_BENCH_FAKE_DEVICE="0")
    assert r.returncode != 0 and "map to the same GPU" in r.stderr and not [x for x in r.stdout.splitlines() if x.startswith("{")]


def test_bench_line_checks_itself_before_the_first_8_gpu_run():
    """pairwise distinct PCI addresses when world > 1, min / max / mean kernel time and the within-run
    efficiency over the ranks, and `--mix v4v5 --gpus N` naming configs[4] with its total stream count."""
    import json
    # two ranks whose ordinals differ but whose devices answer with ONE PCI address: refused
    r = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", VAD_BENCH_FAKE="1", VAD_BENCH_FAKE_PCI="0000:05:00,0000:05:00")
    assert r.returncode != 0 and "same physical GPU (PCI 0000:05:00)" in r.stderr and not [x for x in r.stdout.splitlines() if x.startswith("{")]
    # distinct addresses: the line carries them, the spread over the ranks and the efficiency against the best rank
    r = _bench("--gpus", "2", "--steps", "10", "--warmup", "1", "--mix", "v4v5", VAD_BENCH_FAKE="1", VAD_BENCH_FAKE_PCI="0000:05:00,0000:15:00")
    assert r.returncode == 0, r.stdout + r.stderr
    out = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][0]
    assert [x["pci"] for x in out["per_rank"]] == ["0000:05:00", "0000:15:00"]
    best = max(x["frames_per_s"] for x in out["per_rank"])
    assert out["ranks"]["best_rank_frames_per_s"] == pytest.approx(best)
    assert out["ranks"]["scaling_efficiency"] == pytest.approx(out["value"] / (2 * best)) and 0.5 < out["ranks"]["scaling_efficiency"] <= 1.0
    assert out["ranks"]["kernel_us"] is None                                     # the fake step has no kernel
    assert out["config"]["streams_total"] == 2 * 8192 and out["config"]["streams_per_gpu"] == 8192
    assert out["config"]["workload"].startswith("configs[4]: batch=16384 streams sharded 8192/GPU across 2 x MI355X, Silero V4 + V5 mixed")
    assert "its 2-GPU share" in out["config"]["workload"] and "V4 + V5 mixed" in out["metric"]
    # at the north-star size the workload IS configs[4] (string only: eight fake ranks are eight processes of one core each)
    r = _bench("--gpus", "8", "--steps", "3", "--warmup", "0", "--mix", "v4v5", VAD_BENCH_FAKE="1")
    assert r.returncode == 0, r.stdout + r.stderr
    out = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][0]
    assert out["n_gpus"] == 8 and out["config"]["streams_total"] == 65536 and len(out["per_rank"]) == 8
    assert out["config"]["workload"].startswith("configs[4]: batch=65536 streams sharded 8192/GPU across 8 x MI355X") and "share]" not in out["config"]["workload"]


def test_the_two_pool_line_says_what_it_is_and_the_default_line_does_not_change():
    """`bench.py --pools 2` (two independent pools per GPU) is an extra measurement: its workload string says so, and the default
    line - one pool, one launch per step - is what it was."""
    import json
    one = _bench("--steps", "3", "--warmup", "0", "--no-cpu", VAD_BENCH_FAKE="1")
    two = _bench("--steps", "3", "--warmup", "0", "--no-cpu", "--pools", "2", VAD_BENCH_FAKE="1")
    assert one.returncode == 0 and two.returncode == 0, one.stderr + two.stderr
    a = [json.loads(x) for x in one.stdout.splitlines() if x.startswith("{")][0]
    b = [json.loads(x) for x in two.stdout.splitlines() if x.startswith("{")][0]
    assert a["metric"] == b["metric"] and a["config"]["streams_per_gpu"] == b["config"]["streams_per_gpu"] == 8192
    assert "independent pools" not in a["config"]["workload"]
    assert b["config"]["workload"].startswith(a["config"]["workload"]) and "as 2 independent pools of 4096 streams per GPU" in b["config"]["workload"]


def test_bench_refuses_to_run_fewer_gpus_than_asked():
    r = _bench("--gpus", "8")                                 # no GPU in the CPU suite's container; 1 on a gpurun box
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()
    r = _bench("--gpus", "2", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr

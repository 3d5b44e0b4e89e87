"""The serving pool's C inbox for wire frames (cutter_vad_amd/csrc/wirebox.c): queueing, the hand-over to a
``vad_tick_push_gather``-shaped function, reference counts, invalidation, and that a pool behaves the same with and without it."""

import ctypes as C
import sys
import threading

import numpy as np
import pytest

from cutter_vad_amd import VADConfig, _build
from cutter_vad_amd.core.exceptions import AudioProcessingError
from cutter_vad_amd.server import shared_pool
from cutter_vad_amd.server.shared_pool import SharedStreamPool
from tests.fakes import FakeEngine, FakePool

_build.build_wirebox()
from cutter_vad_amd import _wirebox  # noqa: E402

GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_void_p), C.c_int32, C.c_int, C.c_int,
                     C.POINTER(C.c_int32))


def test_push_takes_only_short_even_bytes_and_drain_returns_them_in_order():
    box = _wirebox.Inbox(1024)
    a, b = box.pusher(3, True), box.pusher(9, False)
    f512, f240 = b"\x01\x02" * 512, b"\x03\x04" * 240
    c = box.pusher(4, True)
    assert a(f512) and b(f512) and c(f240) and a(f512)
    assert not a(f240)                           # another length from the same session within one flush: it could overtake -> not taken
    assert not a(b"") and not a(b"\0") and not a(b"\0" * 3) and not a(b"\0" * 1026)           # empty / odd / too long
    assert not a(bytearray(f512)) and not a(np.zeros(4, np.int16)) and not a(None)            # bytes objects only
    with pytest.raises(TypeError):
        a()
    with pytest.raises(TypeError):
        a(f512, f512)
    assert len(box) == 4
    got = box.drain()
    assert got == [(1024, True, 0, [(3, f512), (3, f512)]), (1024, False, 0, [(9, f512)]), (480, True, 0, [(4, f240)])]
    assert len(box) == 0 and box.drain() == []
    assert a(f240) and not a(f512) and len(box) == 1               # ... after a flush it is
    box.drain()
    a.invalidate()
    assert not a.valid and b.valid and not a(f512) and b(f512) and len(box) == 1
    with pytest.raises(ValueError):
        _wirebox.Inbox(1)
    with pytest.raises(ValueError):
        box.flush(0, 0)
    # a session at another rate: exactly its chunk length (which may exceed max_bytes), its own group
    r48 = box.pusher(11, True, 48000, 3072)
    r8 = box.pusher(12, True, 8000, 512)
    c48, c8 = bytes(3072), bytes(512)
    assert r48(c48) and r8(c8) and not r48(c8) and not r8(c48) and not r48(bytes(3070))
    box.drain()
    assert r48(c48) and r8(c8) and b(c8)
    assert sorted(box.drain()) == sorted([(512, False, 0, [(9, c8)]), (3072, True, 48000, [(11, c48)]), (512, True, 8000, [(12, c8)])])
    with pytest.raises(ValueError):
        box.pusher(1, True, 48000)                       # a rate needs its chunk length
    assert r48(c48)
    with pytest.raises(ValueError, match="rate entry point"):
        box.flush(1, 1)                                  # chunks at another rate waiting, no entry point for them
    box.drain()


def test_flush_hands_slots_and_frame_pointers_to_the_engine_entry_and_reports_refusals():
    box = _wirebox.Inbox(1024)
    frames = {slot: bytes([slot]) * (1024 if slot % 3 else 600) for slot in range(1, 40)}
    before = {slot: sys.getrefcount(frames[slot]) for slot in frames}
    for slot in frames:
        assert box.pusher(slot, bool(slot & 1))(frames[slot])
    assert all(sys.getrefcount(frames[s]) == before[s] + 1 for s in frames)                   # the inbox keeps them alive
    seen = []

    def entry(eng, slots, n, ptrs, nsamples, fmt, gate, status):
        assert eng == 0xABCD and fmt == 1
        for i in range(n):
            data = C.string_at(ptrs[i], 2 * nsamples)
            seen.append((int(slots[i]), nsamples, gate, data))
            if slots[i] in (7, 21):
                status[i] = -6
        return 0

    cb = GATHER(entry)
    refused = box.flush(C.cast(cb, C.c_void_p).value, 0xABCD)
    assert sorted(refused) == [(7, -6), (21, -6)]
    assert len(box) == 0 and len(seen) == len(frames)
    for slot, nsamples, gate, data in seen:
        assert data == frames[slot] and nsamples == len(frames[slot]) // 2 and gate == (slot & 1)
    assert [s for s, *_ in seen if len(frames[s]) == 1024 and s & 1] == [s for s in frames if len(frames[s]) == 1024 and s & 1]
    assert all(sys.getrefcount(frames[s]) == before[s] for s in frames)                       # ... and lets go of them
    # a second round reuses the arrays
    for slot in frames:
        box.pusher(slot, True)(frames[slot])
    seen.clear()
    assert box.flush(C.cast(cb, C.c_void_p).value, 0xABCD) == [(7, -6), (21, -6)] and len(seen) == len(frames)


RATE_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_void_p), C.c_int32, C.c_int, C.c_int,
                          C.c_int32, C.POINTER(C.c_int32))


def test_flush_sends_chunks_at_another_rate_through_the_rate_entry_point():
    box = _wirebox.Inbox(1024)
    seen = []

    def entry(eng, slots, n, ptrs, nsamples, fmt, gate, status):
        seen.append(("rate0", [int(slots[i]) for i in range(n)], nsamples, gate))
        return 0

    def rate_entry(eng, slots, n, ptrs, nsamples, fmt, gate, sr, status):
        seen.append((sr, [int(slots[i]) for i in range(n)], nsamples, gate, C.string_at(ptrs[0], 4)))
        if n > 1:
            status[1] = -8
        return 0

    cb, rcb = GATHER(entry), RATE_GATHER(rate_entry)
    a, r1, r2, r3 = box.pusher(1, True), box.pusher(2, True, 8000, 512), box.pusher(3, True, 8000, 512), box.pusher(4, False, 48000, 3072)
    assert a(bytes(960)) and r1(b"\x01\x02\x03\x04" * 128) and r2(bytes(512)) and r3(b"\x09" * 3072)
    refused = box.flush(C.cast(cb, C.c_void_p).value, 0x77, C.cast(rcb, C.c_void_p).value)
    assert refused == [(3, -8)] and len(box) == 0
    assert seen == [("rate0", [1], 480, 1), (8000, [2, 3], 256, 1, b"\x01\x02\x03\x04"), (48000, [4], 1536, 0, b"\x09" * 4)]


def test_frames_pushed_while_a_flush_is_in_the_engine_are_kept_for_the_next_one():
    box = _wirebox.Inbox(64)
    push = box.pusher(1, True)
    counted = []
    inside, go = threading.Event(), threading.Event()

    def entry(eng, slots, n, ptrs, nsamples, fmt, gate, status):
        counted.append(int(n))
        inside.set()
        go.wait(5)
        return 0

    cb = GATHER(entry)
    addr = C.cast(cb, C.c_void_p).value
    for _ in range(1000):
        push(b"ab" * 16)
    t = threading.Thread(target=lambda: box.flush(addr, 1))
    t.start()
    assert inside.wait(5)
    for _ in range(300):                         # the flush has released the GIL: these start a fresh array
        assert push(b"cd" * 16)
    go.set()
    t.join()
    assert counted == [1000] and len(box) == 300
    box.flush(addr, 1)
    assert counted == [1000, 300] and len(box) == 0


def test_a_frame_pushed_during_a_flush_never_overtakes_its_sessions_earlier_frame():
    """flush used to walk the live boxes with the GIL released per box, so a frame pushed in that window (the epoch
    guard lets it through: its session's last frame carries the previous epoch) could leave in the SAME flush from a box not yet
    visited, ahead of the session's earlier frame waiting in a box visited later.  Three boxes in creation order 512 B, 640 B,
    1024 B; slot 2 has 1024 B queued and pushes 640 B from inside the first group's engine call."""
    box = _wirebox.Inbox(2048)
    other1, other2, two = box.pusher(1, True), box.pusher(3, True), box.pusher(2, True)
    assert other1(bytes(512)) and other2(bytes(640))            # boxes 0 (512 B) and 1 (640 B)
    assert two(b"\x01\x00" * 512)                                # box 2 (1024 B): slot 2's EARLIER frame
    seen = []
    pushed_inside = []

    def entry(eng, slots, n, ptrs, nsamples, fmt, gate, status):
        if not pushed_inside:                                    # first group in the engine: slot 2's LATER frame arrives now
            pushed_inside.append(two(b"\x02\x00" * 320))
        seen.extend((int(slots[i]), int(nsamples)) for i in range(n))
        return 0

    cb = GATHER(entry)
    addr = C.cast(cb, C.c_void_p).value
    assert box.flush(addr, 1) == []
    # whether the push was taken or refused, slot 2's frames reach the engine in the order they were sent
    if pushed_inside == [True]:
        assert len(box) == 1
        assert box.flush(addr, 1) == []
    else:
        assert len(box) == 0
    mine = [ns for slot, ns in seen if slot == 2]
    assert mine == ([512, 320] if pushed_inside == [True] else [512]), (seen, pushed_inside)
    assert sorted(seen[:3]) == [(1, 256), (2, 512), (3, 320)]    # the first flush carried exactly the three frames queued before it


def test_a_pusher_with_a_fallback_is_a_complete_submit_method():
    """What PooledSession.submit_pcm16 is while a session is live: one C call per wire frame; a frame the inbox does not take is
    handed to the fallback from inside that call, and its result (or exception) is the call's."""
    import gc
    import weakref
    box = _wirebox.Inbox(1024)
    got = []

    class Owner:
        def general(self, data):
            got.append(data)
            if data == b"boom":
                raise ValueError("general path failed")
            return "general"
    o = Owner()
    push = box.pusher(5, True, 0, 0, o.general)
    o.push = push                                   # owner <-> pusher <-> bound method: a cycle through a C object
    assert push(bytes(512)) is True and len(box) == 1 and got == []
    assert push(bytes(3)) == "general" and push(bytearray(8)) == "general" and len(box) == 1      # odd length / not bytes
    assert push(bytes(480)) == "general"            # another length within one flush: the general path flushes first
    with pytest.raises(ValueError, match="general path failed"):
        push(b"boom")
    push.invalidate()
    assert push(bytes(512)) == "general" and len(box) == 1 and len(got) == 5
    with pytest.raises(TypeError):
        box.pusher(5, True, 0, 0, 42)
    assert box.pusher(5, True, 0, 0, None)(bytes(3)) is False          # no fallback: the old contract
    w = weakref.ref(o)
    del o, push
    gc.collect()
    assert w() is None                              # the pusher takes part in garbage collection: the cycle does not leak


def test_many_producers_and_a_flusher_lose_nothing():
    box = _wirebox.Inbox(64)
    total = []

    def entry(eng, slots, n, ptrs, nsamples, fmt, gate, status):
        total.extend(int(slots[i]) * 1000000 + int.from_bytes(C.string_at(ptrs[i], 4), "little") for i in range(n))
        return 0

    cb = GATHER(entry)
    addr = C.cast(cb, C.c_void_p).value
    stop = threading.Event()

    def producer(slot):
        push = box.pusher(slot, bool(slot & 1))
        for k in range(5000):
            assert push(k.to_bytes(4, "little") + b"\0" * (4 * (slot % 3)))

    def flusher():
        while not stop.is_set():
            box.flush(addr, 1)

    f = threading.Thread(target=flusher)
    f.start()
    ps = [threading.Thread(target=producer, args=(slot,)) for slot in range(6)]
    [p.start() for p in ps]
    [p.join() for p in ps]
    stop.set()
    f.join()
    box.flush(addr, 1)
    assert len(total) == 6 * 5000 and len(set(total)) == len(total)
    for slot in range(6):                        # per producer, in the order pushed
        mine = [v % 1000000 for v in total if v // 1000000 == slot]
        assert mine == list(range(5000))


def _scenario(monkeypatch, use_wirebox: bool):
    monkeypatch.setattr(shared_pool, "_wirebox", _wirebox if use_wirebox else None)
    eng = FakeEngine(fn=lambda fr: 0.9 if np.abs(fr).max() > 0.3 else 0.05)
    pool = SharedStreamPool(pool=FakePool(eng))
    assert (pool._wire is not None) == use_wirebox
    cfg = dict(vad_start_probability=0.4, vad_end_probability=0.3, voice_start_frame_count=2, voice_end_frame_count=3, buffer_size=480)
    loud, quiet = (np.full(480, 0.5) * 32767).astype("<i2").tobytes(), bytes(960)
    long16 = (np.full(700, 0.5) * 32767).astype("<i2").tobytes()
    logs, errs, sessions = [], [], []
    for k in range(6):
        s = pool.open_session(VADConfig(**cfg, enable_denoising=bool(k & 1)))
        log = []
        s.set_callbacks(lambda log=log: log.append("S"), lambda wav, log=log: log.append(("E", len(wav))),
                        lambda pcm, log=log: log.append(("C", len(pcm))), lambda e, k=k: errs.append((k, str(e))))
        assert (s._push is not None) == use_wirebox
        sessions.append(s)
        logs.append(log)
    for k, s in enumerate(sessions):
        for f in [quiet] * k + [loud, bytearray(loud), long16, loud[:600]] + [quiet] * 4:      # bytes, bytearray, over-long, ragged
            s.submit_pcm16(f)
    pool.drain()
    # reconfigure flips the gate of session 1 and keeps its callbacks; close drops what session 0 had waiting
    sessions[0].submit_pcm16(loud)
    sessions[0].close()
    with pytest.raises(AudioProcessingError, match="closed"):
        sessions[0].submit_pcm16(loud)
    sessions[1].submit_pcm16(loud)
    pool.reconfigure(sessions[1], VADConfig(**cfg, enable_denoising=False))
    for f in [loud] * 3 + [quiet] * 4:
        sessions[1].submit_pcm16(f)
    # 258 frames ahead of the ticker (one staged + 256 waiting + 1): the last is refused and only its own session hears about it
    for _ in range(258):
        sessions[2].submit_pcm16(quiet)
    pool.drain()
    seen = [fr.copy() for fr in eng.frames_seen]
    pool.close()
    return logs, errs, seen


def test_a_pool_behaves_the_same_with_and_without_the_c_inbox(monkeypatch):
    with_c = _scenario(monkeypatch, True)
    without = _scenario(monkeypatch, False)
    assert with_c[0] == without[0] and with_c[1] == without[1]
    assert len(with_c[1]) == 1 and with_c[1][0][0] == 2 and "frame not queued: 256 frames are waiting for this stream" in with_c[1][0][1]
    assert len(with_c[2]) == len(without[2]) and all(np.array_equal(a, b) for a, b in zip(with_c[2], without[2]))
    assert any(log and log[0] == "S" for log in with_c[0])


def test_closing_or_moving_a_session_takes_its_pusher_away_first(monkeypatch):
    monkeypatch.setattr(shared_pool, "_wirebox", _wirebox)
    pool = SharedStreamPool(pool=FakePool(FakeEngine(fn=lambda fr: 0.0)))
    s = pool.open_session(VADConfig(buffer_size=480))
    push = s._push
    assert push is not None and push.valid
    s.submit_pcm16(bytes(960))
    assert len(pool._wire) == 1
    s.close()
    assert not push.valid and s._push is None and len(pool._wire) == 0
    with pytest.raises(AudioProcessingError, match="closed"):       # a stale reference to the pusher cannot reach a recycled slot:
        push(bytes(960))                                            # not taken -> its fallback, the general path, which says why
    assert len(pool._wire) == 0 and s.submit_pcm16 == s._submit_pcm16_general
    t = pool.open_session(VADConfig(buffer_size=480, sample_rate=16000))
    assert t._push is not push and t._push.valid
    pool.close()


@pytest.mark.parametrize("use_wirebox", [True, False])
def test_a_session_that_changes_its_frame_length_keeps_its_order(monkeypatch, use_wirebox):
    """Frames wait grouped by length and the groups are pushed one after the other: the 480-sample frame that follows a
    300-sample one (a file's last chunk, then the next file) must not reach the stream first."""
    monkeypatch.setattr(shared_pool, "_wirebox", _wirebox if use_wirebox else None)
    heard = []
    eng = FakeEngine(fn=lambda fr: heard.append(int(round(float(fr[0]) * 32767))) or 0.0)
    pool = SharedStreamPool(pool=FakePool(eng))
    a, b = pool.open_session(VADConfig(buffer_size=480, enable_denoising=False)), pool.open_session(VADConfig(buffer_size=480, enable_denoising=False))

    def pcm(v, n):
        return np.full(n, v, "<i2").tobytes()

    a.submit_pcm16(pcm(100, 480))                 # the 960-byte group exists before b's short frame arrives
    b.submit_pcm16(pcm(1, 300))
    b.submit_pcm16(pcm(2, 480))
    b.submit_pcm16(pcm(3, 300))
    b.submit_pcm16(pcm(4, 480))
    pool.drain()
    assert [v for v in heard if v != 100] == [1, 2, 3, 4] and heard.count(100) == 1
    pool.close()

"""Test doubles (CPU): a scripted engine so that host logic can be exercised without a GPU.

Plays the role the reference's tests give to ``Mock(spec=ort.InferenceSession)``
(/root/reference/tests/test_silero_model.py:277-292): probabilities are canned, the device-side
state machine is played by the oracle's restatement of it.
"""

from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional

import numpy as np

from oracle import oracle


class FakeEngine:
    def __init__(self, probs: Optional[Iterable[float]] = None, fn: Optional[Callable[[np.ndarray], float]] = None):
        self._it = iter(probs) if probs is not None else None
        self._fn = fn
        self.handle = True
        self.sm: Dict[int, oracle.StateMachine] = {}
        self.thr: Dict[int, tuple] = {}
        self.state: Dict[int, np.ndarray] = {}
        self.frames_seen = []
        self.denoise_seen = []
        self._next = 0
        self.closed = []
        self.model_version = 5

    def script(self, probs):
        self._it = iter(probs)

    def _p(self, frame):
        if self._fn is not None:
            return float(self._fn(frame))
        return float(next(self._it))

    def open_stream(self):
        s = self._next
        self._next += 1
        self.thr[s] = (0.7, 0.7, 0.8, 0.95, 10, 50)
        self.sm[s] = oracle.StateMachine(*self.thr[s])
        self.state[s] = np.zeros(256, np.float32)
        return s

    def open_streams(self, n):
        return np.array([self.open_stream() for _ in range(n)], np.int64)

    def close_stream(self, s):
        self.closed.append(int(s))
        self.sm.pop(int(s), None)

    def reset(self, slots):
        for s in slots:
            self.sm[int(s)] = oracle.StateMachine(*self.thr[int(s)])
            self.state[int(s)] = np.zeros(256, np.float32)

    def set_thresholds(self, s, *t):
        # values only, like vad_stream_set_thresholds: unchanged values leave the machine (counters, history) alone
        if self.thr.get(int(s)) == tuple(t) and int(s) in self.sm:
            return
        self.thr[int(s)] = tuple(t)
        self.sm[int(s)] = oracle.StateMachine(*t)

    def set_thresholds_many(self, slots, thresholds):
        rows = [thresholds] * len(slots) if np.isscalar(thresholds[0]) else list(thresholds)
        for s, t in zip(slots, rows):
            self.set_thresholds(int(s), *t)

    def get_state(self, s):
        return self.state[int(s)].copy()

    def set_state(self, s, hc):
        self.state[int(s)] = np.asarray(hc, np.float32).copy()

    def step(self, slots, frames, denoise=0.01, i16_scale=32767):
        frames = np.asarray(frames)
        if frames.dtype == np.int16:                       # the engine's int16 ingest: true division by the scale
            frames = frames.astype(np.float32) / np.float32(i16_scale)
        frames = np.asarray(frames, np.float32).reshape(len(slots), -1)
        assert frames.shape[1] in (512, 256), "callers must pad to the model's frame length (512; 256 on V5's 8 kHz sub-model)"
        self.frames_seen.append(frames.copy())
        self.denoise_seen.append(denoise)
        return np.array([self._p(f) for f in frames], np.float64)  # exact scripted values

    def step_events(self, slots, frames, denoise=0.01, i16_scale=32767):
        p = self.step(slots, frames, denoise)
        ev = np.zeros(len(slots), np.uint8)
        seg = np.zeros(len(slots), np.int32)
        for i, s in enumerate(slots):
            e, sg = self.sm[int(s)].step(float(p[i]), 1)
            ev[i], seg[i] = e, sg
        return p, ev, seg

    def step_multi(self, slots, frames, denoise=0.01, i16_scale=32767):
        frames = np.asarray(frames, np.float32)
        n, T = frames.shape[0], frames.shape[1]
        p = np.empty((n, T), np.float64)
        ev = np.zeros((n, T), np.uint8)
        for t in range(T):
            p[:, t], ev[:, t], _ = self.step_events(slots, frames[:, t], denoise)
        self.multi_calls = getattr(self, "multi_calls", 0) + 1
        return p, ev

    # ---- the C tick assembler (vad_tick_push / vad_tick_run), same semantics: one frame per slot and tick, submission
    #      order, one step_events call per (format, gate) group
    frame_samples = 512

    sample_rate = 16000

    def tick_push(self, slot, frame, gate_on=True, i16_scale=32767, sample_rate=None):
        q = self.__dict__.setdefault("_tickq", {})
        if isinstance(frame, (bytes, bytearray, memoryview)):
            x = np.frombuffer(bytes(frame), dtype="<i2")
            fmt = 2 if i16_scale == 32768 else 1
        else:
            x = np.ascontiguousarray(frame, np.float32)
            fmt = 0
        if len(q.setdefault(int(slot), [])) >= 257:
            raise RuntimeError("tick: 256 frames waiting")
        g = fmt * 2 + int(bool(gate_on))
        if sample_rate is not None and int(sample_rate) != self.sample_rate:      # vad_tick_push_rate
            ri = (8000, 24000, 48000).index(int(sample_rate))
            if x.size != (256, 768, 1536)[ri]:
                raise RuntimeError(f"Failed to resample audio from {sample_rate}Hz to 16000Hz: a chunk must hold "
                                   f"{(256, 768, 1536)[ri]} samples, got {x.size}")
            if fmt:
                x = x.astype(np.float32) / np.float32(32768.0 if fmt == 2 else 32767.0)
            g = 6 + 3 * int(bool(gate_on)) + ri
        q[int(slot)].append((g, x))

    def tick_push_status(self, slots, frames, nsamples, gate_on=True, i16_scale=32767):
        """vad_tick_push_status: every frame is tried, one status each (0 = queued, -6 = not an open stream, -8 = busy)"""
        if isinstance(frames, (bytes, bytearray, memoryview)):
            rows = [bytes(frames)[2 * nsamples * i:2 * nsamples * (i + 1)] for i in range(len(slots))]
        else:
            rows = list(np.asarray(frames).reshape(len(slots), nsamples))
        self.push_status_calls = getattr(self, "push_status_calls", 0) + 1
        st = np.zeros(len(slots), np.int32)
        for i, (s, r) in enumerate(zip(slots, rows)):
            if int(s) not in self.sm:
                st[i] = -6
                continue
            try:
                self.tick_push(int(s), r, gate_on, i16_scale)
            except RuntimeError:
                st[i] = -8
        return st

    def tick_push_gather(self, slots, frames, nsamples, gate_on=True, i16_scale=32767):
        assert all(len(f) == 2 * nsamples for f in frames)
        return self.tick_push_status(slots, b"".join(frames), nsamples, gate_on, i16_scale)

    def tick_pending(self, slot):
        return len(self.__dict__.setdefault("_tickq", {}).get(int(slot), []))

    def save_segment(self, slot):
        import copy
        return copy.deepcopy(self.__dict__.setdefault("_seg", {}).get(int(slot)))

    def restore_segment(self, slot, blob):
        import copy
        if blob is not None:
            self.__dict__.setdefault("_seg", {})[int(slot)] = copy.deepcopy(blob)

    def tick_cancel(self, slot):
        self.__dict__.setdefault("_tickq", {}).pop(int(slot), None)
        self.__dict__.setdefault("_seg", {}).pop(int(slot), None)

    def tick_enable_segments(self, on=True):
        self._segments = bool(on)

    def tick_take_segment(self, slot):
        st = self.__dict__.setdefault("_seg", {}).get(int(slot))
        if not st or st["done"] is None:
            return np.empty(0, np.float32)
        out, st["done"] = st["done"], None
        return out

    def tick_take_segment_wav16(self, slot, sample_rate):
        """vad_tick_take_segment_wav16: the finished segment as WAVWriter(sample_rate, 16, 1) writes it"""
        from cutter_vad_amd.utils.wav_writer import WAVWriter
        return WAVWriter(sample_rate=int(sample_rate), bit_depth=16, channels=1).write_wav_data(self.tick_take_segment(slot))

    def tick_run_work(self, denoise, last_prob, frames_done, active, continue_cb, continue_payload):
        """vad_tick_run_work (include/vad_engine.h, ABI 4), statement for statement: the tick, the bookkeeping on the caller's
        per-slot arrays, and the entries the caller has work for with their VAD_WORK_* bits"""
        slots, p, ev, _seg, gs, frames, nsamp = self.tick_run(denoise)
        widx, wkind, wsamp = [], [], []
        first_rate_entry = int(gs[6])
        for k in range(int(slots.size)):
            sl, e = int(slots[k]), int(ev[k])
            was, started, ended = bool(active[sl]), bool(e & 1), bool(e & 2)
            last_prob[sl] = p[k]
            frames_done[sl] += 1
            active[sl] = (was or started) and not ended
            kind = (1 if started else 0) | (2 if ended else 0)
            if was and continue_cb[sl]:
                kind |= 4 | (8 if continue_payload[sl] else 0)
            if k < first_rate_entry and int(nsamp[k]) > self.frame_samples:
                kind |= 16
            if kind:
                widx.append(k)
                wkind.append(kind)
                st = self.__dict__.get("_seg", {}).get(sl)
                wsamp.append(int(st["done"].size) if (kind & 2) and st and st["done"] is not None else 0)
        return slots, gs, frames, nsamp, np.array(widx, np.int32), np.array(wkind, np.uint8), np.array(wsamp, np.int64)

    def _assemble(self, slot, g, x, p, ev, denoise):
        """the host half of _process_voice_state on the frame as pushed (vad_tick_enable_segments)"""
        st = self.__dict__.setdefault("_seg", {}).setdefault(int(slot), dict(active=False, pre=[], seg=[], done=None))
        k = x.astype(np.float32) / np.float32(32768.0 if g >= 4 else 32767.0) if 2 <= g < 6 else x.astype(np.float32)
        if (g & 1) if g < 6 else g >= 9:
            k = np.where(np.abs(k) > np.float32(denoise), k, np.float32(0.0)).astype(np.float32)
        if not st["active"]:
            st["pre"] = st["pre"] + [k] if float(p) >= self.thr[int(slot)][0] else []
            if ev & 1:
                st["active"], st["seg"], st["pre"] = True, st["pre"], []
        else:
            st["seg"].append(k)
            if ev & 2:
                st["done"], st["seg"], st["active"] = np.concatenate(st["seg"]), [], False

    def tick_run(self, denoise=0.01):
        q = self.__dict__.setdefault("_tickq", {})
        F = self.frame_samples
        groups = {g: [] for g in range(12)}
        for slot in list(q):
            g, x = q[slot].pop(0)
            if not q[slot]:
                del q[slot]
            groups[g].append((slot, x))
        slots, probs, events, segs, ns, frames, gs = [], [], [], [], [], [None] * 12, [0]
        for g in range(12):
            rows = groups[g]
            if rows:
                Fg = F if g < 6 else (256, 768, 1536)[(g - 6) % 3]
                arr = np.zeros((len(rows), Fg), np.int16 if 2 <= g < 6 else np.float32)
                for i, (_, x) in enumerate(rows):
                    m = min(x.size, Fg)
                    arr[i, :m] = x[:m]
                sl = [r[0] for r in rows]
                if g < 6:
                    p, ev, sg = self.step_events(sl, arr, denoise if g & 1 else None, 32768 if g >= 4 else 32767)
                else:           # resample (scipy's Fourier method, as the oracle restates it) -> gate -> model
                    x16 = np.stack([oracle.resample(r, 512) for r in arr])
                    p, ev, sg = self.step_events(sl, x16, denoise if g >= 9 else None)
                slots += sl
                probs += list(p)
                events += list(ev)
                segs += list(sg)
                ns += [r[1].size for r in rows]
                frames[g] = arr
                if getattr(self, "_segments", False):
                    for (slot, x), pi, ei in zip(rows, p, ev):
                        self._assemble(slot, g, x, np.float32(pi), int(ei), denoise)
            gs.append(len(slots))
        self.last_tick_staged_next = len(q)                 # slots that still have a frame waiting
        return (np.array(slots, np.int64), np.array(probs, np.float32), np.array(events, np.uint8), np.array(segs, np.int32),
                np.array(gs, np.int64), frames, np.array(ns, np.int32))

    def save_stream(self, s):
        import ctypes
        sm = self.sm[int(s)]
        return (self.state[int(s)].copy(), ctypes.string_at(sm._buf, len(sm._buf)), self.thr[int(s)])

    def restore_stream(self, s, blob):
        import ctypes
        st, raw, thr = blob
        self.state[int(s)] = st.copy()
        self.thr[int(s)] = thr
        sm = oracle.StateMachine(*thr)
        ctypes.memmove(sm._buf, raw, len(raw))
        self.sm[int(s)] = sm
        self.restores = getattr(self, "restores", 0) + 1

    def close(self):
        self.handle = False


class FakePool:
    def __init__(self, engine: FakeEngine):
        self.engine = engine

    def engine_for(self, model_path, version, device_id=None, max_streams=None, sample_rate=16000):
        self.requested = getattr(self, "requested", []) + [(str(model_path), int(sample_rate))]
        return self.engine

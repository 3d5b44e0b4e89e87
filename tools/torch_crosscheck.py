#!/usr/bin/env python3
"""Third, independent evaluation of the reference's two ONNX graphs - TEST INFRASTRUCTURE, container only.

Why: onnxruntime is not importable here, so the model arithmetic is pinned by this repository's own readings of the
`.onnx` files (oracle/silero_oracle.c and oracle/onnx_interp.py, which also shares its protobuf reader with the product's
weight loader).  This script breaks that common mode twice:

  1. it reads the files with its OWN protobuf wire-format walker (below; nothing imported from cutter_vad_amd.onnx_lite)
     and checks every tensor both readers see byte for byte;
  2. it evaluates the 16 kHz branches (and V4's 8 kHz branch) with PyTorch's operators - F.pad(mode="reflect"), F.conv1d
     (stride / padding / groups), torch's LSTM cell (`torch._VF.lstm_cell`, the kernel nn.LSTMCell runs; V4 was exported
     from PyTorch, so PyTorch is the authority on what Conv / LSTM meant) - from the initialisers found by name, following
     SURVEY.md §8 a7 / a8, and asserts agreement with tests/golden/model_*.npz to <= 2e-6 in float64.

It reads /root/reference/src/real_time_vad/models/*.onnx (silero_model.py:321-325 hands the same files to onnxruntime), so it
runs in the build container only; tests/test_torch_crosscheck.py calls it and skips on the GPU box.

    python3 tools/torch_crosscheck.py            -> one JSON line per model, exit code 1 on disagreement
"""
from __future__ import annotations

import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODELS = "/root/reference/src/real_time_vad/models"
TOL = 2e-6


# ---------------------------------------------------------------------------------------------------------------------
# protobuf wire format, just enough for ModelProto -> tensors and a flat node list (field numbers: onnx.proto3)
# ---------------------------------------------------------------------------------------------------------------------
def _fields(buf: memoryview):
    """yield (field number, wire type, value) over one message; value = int (varint / fixed) or memoryview (bytes)"""
    i, n = 0, len(buf)
    while i < n:
        key = 0
        shift = 0
        while True:
            b = buf[i]
            i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if b < 0x80:
                break
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v = 0
            shift = 0
            while True:
                b = buf[i]
                i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if b < 0x80:
                    break
            yield fno, wt, v
        elif wt == 1:
            yield fno, wt, bytes(buf[i:i + 8])
            i += 8
        elif wt == 2:
            ln = 0
            shift = 0
            while True:
                b = buf[i]
                i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if b < 0x80:
                    break
            yield fno, wt, buf[i:i + ln]
            i += ln
        elif wt == 5:
            yield fno, wt, bytes(buf[i:i + 4])
            i += 4
        else:
            raise ValueError(f"wire type {wt}")


def _varints(mv: memoryview):
    out, v, shift = [], 0, 0
    for b in bytes(mv):
        v |= (b & 0x7F) << shift
        shift += 7
        if b < 0x80:
            out.append(v)
            v, shift = 0, 0
    return out


def _tensor(mv: memoryview):
    """TensorProto -> (name, ndarray)"""
    dims, dtype, name, raw, f32, i64, i32 = [], 0, "", None, [], [], []
    for fno, wt, v in _fields(mv):
        if fno == 1:
            dims += _varints(v) if wt == 2 else [v]
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 4:
            f32 += list(struct.unpack(f"<{len(v) // 4}f", bytes(v))) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fno == 7:
            i64 += _varints(v) if wt == 2 else [v]
        elif fno == 5:
            i32 += _varints(v) if wt == 2 else [v]
    np_t = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}.get(dtype)
    if np_t is None:
        return name, None
    if raw is not None:
        a = np.frombuffer(raw, dtype=np.dtype(np_t).newbyteorder("<")).astype(np_t)
    elif dtype == 1:
        a = np.array(f32, np.float32)
    elif dtype == 7:
        a = np.array([x - (1 << 64) if x >= (1 << 63) else x for x in i64], np.int64)
    else:
        a = np.array(i32, np_t)
    return name, a.reshape(dims) if dims else a.reshape(())


def walk(path: str):
    """-> (tensors {name: array} over ALL graphs incl. Constant outputs, nodes [(graph path, op, inputs, outputs, attrs)])"""
    tensors, nodes = {}, []

    def graph(mv, gpath):
        for fno, wt, v in _fields(mv):
            if fno == 5:                                   # initializer
                name, a = _tensor(v)
                if a is not None:
                    tensors[name] = a
            elif fno == 1:                                 # node
                ins, outs, op, nname, attrs, subs = [], [], "", "", {}, []
                for f2, w2, v2 in _fields(v):
                    if f2 == 1:
                        ins.append(bytes(v2).decode())
                    elif f2 == 2:
                        outs.append(bytes(v2).decode())
                    elif f2 == 3:
                        nname = bytes(v2).decode()
                    elif f2 == 4:
                        op = bytes(v2).decode()
                    elif f2 == 5:                          # attribute
                        an, val = "", None
                        for f3, w3, v3 in _fields(v2):
                            if f3 == 1:
                                an = bytes(v3).decode()
                            elif f3 == 3:
                                val = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                            elif f3 == 2:
                                val = struct.unpack("<f", v3)[0]
                            elif f3 == 8:
                                val = (val or []) + (_varints(v3) if w3 == 2 else [v3])
                            elif f3 == 5:
                                val = ("tensor", v3)
                            elif f3 == 6:
                                val = ("graph", v3)
                        attrs[an] = val
                        if isinstance(val, tuple) and val[0] == "graph":
                            subs.append((an, val[1]))
                for an, val in list(attrs.items()):
                    if isinstance(val, tuple) and val[0] == "tensor":
                        _, a = _tensor(val[1])
                        attrs[an] = a
                        if op == "Constant" and outs and a is not None:
                            tensors[outs[0]] = a
                nodes.append((gpath, op, ins, outs, {k: v for k, v in attrs.items() if not isinstance(v, tuple)}))
                for an, sub in subs:
                    graph(sub, gpath + (f"{nname or op}.{an}",))

    with open(path, "rb") as f:
        data = memoryview(f.read())
    for fno, wt, v in _fields(data):
        if fno == 7:
            graph(v, ())
    return tensors, nodes


# ---------------------------------------------------------------------------------------------------------------------
# the two graphs with torch operators (float64), from tensors found BY NAME
# ---------------------------------------------------------------------------------------------------------------------
def _t(a, dt):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt)


def _suffix(tensors, suffix, within=None):
    hits = [k for k in tensors if k.endswith(suffix) and (within is None or within in k)]
    if len(hits) != 1:
        raise KeyError(f"{suffix!r} (within {within!r}): {len(hits)} tensors")
    return tensors[hits[0]]


class TorchV5:
    """silero_vad_v5.onnx, If_0 then-branch (sr == 16000): SURVEY §8 a7; k8: its else-branch on 256-sample frames (a9)"""

    def __init__(self, tensors, dt, k8=False):
        br = "else_branch" if k8 else "then_branch"
        self.k8 = k8
        g = lambda s: _t(_suffix(tensors, s, br), dt)
        self.basis = g("stft.forward_basis_buffer")
        self.enc = [(g(f"encoder.{i}.reparam_conv.weight"), g(f"encoder.{i}.reparam_conv.bias")) for i in range(4)]
        self.w_ih, self.w_hh = g("decoder.rnn.weight_ih"), g("decoder.rnn.weight_hh")       # PyTorch rows i, f, g, o
        self.b_ih, self.b_hh = g("decoder.rnn.bias_ih"), g("decoder.rnn.bias_hh")
        self.hw, self.hb = g("decoder.decoder.2.weight"), g("decoder.decoder.2.bias")
        self.dt = dt

    def features(self, x):
        import torch
        import torch.nn.functional as F
        nb = 65 if self.k8 else 129
        x = F.pad(x[:, None, :], (0, 32 if self.k8 else 64), mode="reflect")
        s = F.conv1d(x, self.basis, stride=64 if self.k8 else 128)
        y = torch.sqrt(s[:, :nb] ** 2 + s[:, nb:] ** 2)
        for (w, b), stride in zip(self.enc, (1, 2, 2, 1)):
            y = F.relu(F.conv1d(y, w, b, stride=stride, padding=1))
        return y[:, :, 0]

    def run(self, frames):
        """frames [T,512] of one stream or [B,T,512] -> probs [.,T], final state [.,256]"""
        import torch
        import torch.nn.functional as F
        fr = _t(frames, self.dt)
        single = fr.dim() == 2
        if single:
            fr = fr[None]
        B, T, _ = fr.shape
        feats = self.features(fr.reshape(B * T, fr.shape[-1])).reshape(B, T, 128)     # everything before the LSTM is per frame
        h = torch.zeros(B, 128, dtype=self.dt)
        c = torch.zeros(B, 128, dtype=self.dt)
        probs = []
        for t in range(T):
            h, c = torch._VF.lstm_cell(feats[:, t], (h, c), self.w_ih, self.w_hh, self.b_ih, self.b_hh)
            h, c = h.float().to(self.dt), c.float().to(self.dt)      # the graph's state output is a float32 tensor (silero_model.py:533-537)
            probs.append(torch.sigmoid(F.conv1d(F.relu(h)[:, :, None], self.hw, self.hb))[:, 0, 0])
        p = torch.stack(probs, 1).numpy()
        st = torch.cat([h, c], 1).numpy()
        return (p[0], st[0]) if single else (p, st)


class TorchV4:
    """silero_vad.onnx, If_25 then-branch (16 kHz) or else-branch (8 kHz sub-model): SURVEY §8 a8 / a9"""

    def __init__(self, tensors, nodes, dt, k8=False):
        pre = "model_8k." if k8 else "model."
        g = lambda s: _t(tensors[pre + s], dt)
        self.basis = g("feature_extractor.forward_basis_buffer")
        self.filt = g("adaptive_normalization.filter_")
        self.blocks = []
        for name, proj in (("first_layer.0", True), ("encoder.3.0", True), ("encoder.7.0", False), ("encoder.11.0", True)):
            self.blocks.append(dict(dw=(g(name + ".dw_conv.0.weight"), g(name + ".dw_conv.0.bias")),
                                    pw=(g(name + ".pw_conv.0.weight"), g(name + ".pw_conv.0.bias")),
                                    proj=(g(name + ".proj.weight"), g(name + ".proj.bias")) if proj else None))
        self.hw, self.hb = g("decoder.decoder.1.weight"), g("decoder.decoder.1.bias")
        # the sub-model's own graph = the top-level If's then / else branch and everything nested in it
        branch = "else_branch" if k8 else "then_branch"
        mine = [n for n in nodes if n[0] and n[0][0].endswith(branch)]
        # the four anonymous 1x1 convs between the blocks (initialisers with numeric names), in graph order
        convs = [n for n in mine if len(n[0]) == 1 and n[1] == "Conv" and n[2][1].isdigit()]
        assert len(convs) == 4, len(convs)
        self.sconv = [(_t(tensors[n[2][1]], dt), _t(tensors[n[2][2]], dt), int(n[4]["strides"][0])) for n in convs]
        assert [s for _, _, s in self.sconv] == ([2, 2, 1, 1] if k8 else [2, 2, 2, 1])
        # the two stacked LSTMs of the "state supplied" branch: W [1,256,64], R [1,256,64], B [1,512], ONNX gate order i, o, f, c
        lstm = [n for n in mine if n[1] == "LSTM" and n[0][-1].endswith("then_branch")]
        assert len(lstm) == 2, len(lstm)

        def ifgo(a):      # ONNX i, o, f, c  ->  PyTorch i, f, g, o
            i, o, f, c = a[0:64], a[64:128], a[128:192], a[192:256]
            return np.concatenate([i, f, c, o], 0)

        self.lstm = []
        for n in lstm:
            W, R, Bb = tensors[n[2][1]][0], tensors[n[2][2]][0], tensors[n[2][3]][0]
            assert n[4]["hidden_size"] == 64 and W.shape == (256, 64) and Bb.shape == (512,)
            self.lstm.append(tuple(_t(ifgo(a), dt) for a in (W, R, Bb[:256], Bb[256:])))
        self.dt, self.k8 = dt, k8

    def features(self, x):
        import torch
        import torch.nn.functional as F
        s = F.conv1d(F.pad(x[:, None, :], (96, 96), mode="reflect"), self.basis, stride=64)       # [N,258,8]
        mag = torch.sqrt(s[:, :129] ** 2 + s[:, 129:] ** 2)
        spect = torch.log(1 + mag * 1048576.0)
        mean = spect.mean(1, keepdim=True)                                                        # [N,1,8]
        sm = F.conv1d(F.pad(mean, (3, 3), mode="reflect"), self.filt.reshape(1, 1, 7))
        y = torch.cat([mag, spect - sm.mean(-1, keepdim=True)], 1)
        for blk, (sw, sb, stride) in zip(self.blocks, self.sconv):
            dw, db = blk["dw"]
            z = F.conv1d(F.relu(F.conv1d(y, dw, db, padding=2, groups=y.shape[1])), *blk["pw"])
            res = y if blk["proj"] is None else F.conv1d(y, *blk["proj"])
            y = F.relu(F.conv1d(F.relu(z + res), sw, sb, stride=stride))
        return y                                                                                  # [N,64,T3]

    def run(self, frames):
        import torch
        import torch.nn.functional as F
        fr = _t(frames, self.dt)
        single = fr.dim() == 2
        if single:
            fr = fr[None]
        B, T, _ = fr.shape
        feats = self.features(fr.reshape(B * T, 512))
        T3 = feats.shape[-1]
        feats = feats.reshape(B, T, 64, T3)
        h = [torch.zeros(B, 64, dtype=self.dt) for _ in range(2)]
        c = [torch.zeros(B, 64, dtype=self.dt) for _ in range(2)]
        probs = []
        for t in range(T):
            ps = []
            for k in range(T3):
                x = feats[:, t, :, k]
                for layer in range(2):
                    h[layer], c[layer] = torch._VF.lstm_cell(x, (h[layer], c[layer]), *self.lstm[layer])
                    x = h[layer]
                ps.append(torch.sigmoid(F.conv1d(F.relu(x)[:, :, None], self.hw, self.hb))[:, 0, 0])
            probs.append(torch.stack(ps, 0).mean(0))
            h = [v.float().to(self.dt) for v in h]                   # hn / cn leave the graph as float32 tensors
            c = [v.float().to(self.dt) for v in c]
        p = torch.stack(probs, 1).numpy()
        st = torch.cat(h + c, 1).numpy()
        return (p[0], st[0]) if single else (p, st)


# ---------------------------------------------------------------------------------------------------------------------
def compare_readers(path, tensors):
    """every tensor this walker and cutter_vad_amd.onnx_lite both see must be identical, bit for bit"""
    from cutter_vad_amd import onnx_lite
    model = onnx_lite.load_model(path)
    seen, same = 0, 0

    def graph(g):
        nonlocal seen, same
        for k, v in g.initializers.items():
            if k in tensors:
                seen += 1
                same += int(np.asarray(v).tobytes() == tensors[k].astype(np.asarray(v).dtype).tobytes())
        for n in g.nodes:
            if n.op_type == "Constant" and n.outputs and n.outputs[0] in tensors and isinstance(n.attrs.get("value"), np.ndarray):
                seen += 1
                same += int(n.attrs["value"].tobytes() == tensors[n.outputs[0]].tobytes())
            for a in n.attrs.values():
                if isinstance(a, onnx_lite.Graph):
                    graph(a)

    graph(model.graph)
    return seen, same


def main():
    import torch
    from tests.signals import gate, make_streams, model_cases, model_cases_8k
    torch.set_num_threads(4)
    pcm = np.load(os.path.join(ROOT, "tests/golden/speech16k_i16.npz"))["pcm"]
    cases = model_cases(pcm)
    cases8 = model_cases_8k(pcm)
    ok = True
    for label, fname, gold, build in (
            ("v5_16k", "silero_vad_v5.onnx", "model_v5.npz", lambda t, n, dt: TorchV5(t, dt)),
            ("v5_8k", "silero_vad_v5.onnx", "model_v5_8k.npz", lambda t, n, dt: TorchV5(t, dt, k8=True)),
            ("v4_16k", "silero_vad.onnx", "model_v4.npz", lambda t, n, dt: TorchV4(t, n, dt)),
            ("v4_8k", "silero_vad.onnx", "model_v4_8k.npz", lambda t, n, dt: TorchV4(t, n, dt, k8=True))):
        path = os.path.join(MODELS, fname)
        tensors, nodes = walk(path)
        seen, same = compare_readers(path, tensors)
        g = np.load(os.path.join(ROOT, "tests/golden", gold))
        m64, m32 = build(tensors, nodes, torch.float64), build(tensors, nodes, torch.float32)
        worst_p = worst_s = worst32 = 0.0
        frames_checked = 0
        for name in sorted({k.split(".probs")[0] for k in g.files if k.endswith(".probs")}):
            if name == "batch7" and label == "v5_8k":
                x = gate(make_streams(7, 5, seed=78).reshape(7, 10, 256))
            elif label == "v5_8k":
                x = cases8[name]
            elif name == "batch7":
                x = gate(make_streams(7, 10, seed=77))                # tools/make_goldens.py: the batched-call golden
            else:
                x = cases[name]["frames"]
                if gold == "model_v4_8k.npz" and name == "speech_gate":
                    x = x[:240]
            p, s = m64.run(x)
            p32, _ = m32.run(x)
            worst_p = max(worst_p, float(np.abs(p - g[name + ".probs"]).max()))
            worst_s = max(worst_s, float(np.abs(s - g[name + ".state"]).max()))
            if name != "square_fullscale":        # exact spectral nulls under V4's log: float32 is order dependent there
                worst32 = max(worst32, float(np.abs(p32 - g[name + ".probs"]).max()))
            frames_checked += int(np.prod(p.shape))
        row = {"model": label, "tensors_compared_with_onnx_lite": seen, "identical": same, "frames": frames_checked,
               "torch_f64_vs_golden_max_dp": worst_p, "torch_f64_vs_golden_max_dstate": worst_s,
               "torch_f32_vs_golden_max_dp": worst32, "tol": TOL}
        print(json.dumps(row), flush=True)
        ok &= seen > 0 and seen == same and worst_p <= TOL and worst_s <= 2e-6
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

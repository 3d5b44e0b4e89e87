"""Silero VAD weight extraction and the engine's weight-blob container ("SVW").

The reference hands the ``.onnx`` file to onnxruntime
(/root/reference/src/real_time_vad/core/silero_model.py:321-325, file chosen by
/root/reference/src/real_time_vad/core/config.py:242-249).  The HIP engine only needs
the tensors of the 16 kHz branch, under canonical names, in one flat blob that the
C ABI (`include/vad_engine.h`, ``vad_engine_desc.weights``) can consume without any
protobuf knowledge.

Canonical tensors (all float32):

V5 / 16 kHz (``If_0.then_branch`` of silero_vad_v5.onnx)
    stft.basis [258,256]
    enc{0..3}.w [Cout,Cin,3]  enc{0..3}.b [Cout]
    lstm.w_ih [512,128]  lstm.w_hh [512,128]  lstm.b_ih [512]  lstm.b_hh [512]   (gate rows i,f,g,o)
    head.w [128]  head.b [1]

V4 / 16 kHz (``If_25.then_branch`` of silero_vad.onnx)
    stft.basis [258,256]  norm.filter [7]
    l0.dw.w [258,5] l0.dw.b  l0.pw.w [16,258] l0.pw.b  l0.proj.w [16,258] l0.proj.b
    s0.w [16,16] s0.b
    l1.dw.w [16,5] l1.dw.b   l1.pw.w [32,16] l1.pw.b   l1.proj.w [32,16] l1.proj.b
    s1.w [32,32] s1.b
    l2.dw.w [32,5] l2.dw.b   l2.pw.w [32,32] l2.pw.b   (identity residual)
    s2.w [32,32] s2.b
    l3.dw.w [32,5] l3.dw.b   l3.pw.w [64,32] l3.pw.b   l3.proj.w [64,32] l3.proj.b
    s3.w [64,64] s3.b
    lstm{0,1}.w_ih [256,64] lstm{0,1}.w_hh [256,64] lstm{0,1}.b_ih [256] lstm{0,1}.b_hh [256]
                                                      (gate rows re-ordered to i,f,g,o)
    head.w [64]  head.b [1]
"""

from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np

from . import onnx_lite

SVW_MAGIC = b"SVADW001"
_NAME_LEN = 48
_ENTRY = struct.Struct(f"<{_NAME_LEN}sI4IIQQ")  # name, ndim, dims[4], reserved, offset, nelem (88 B, C-natural)
_HEADER = struct.Struct("<8sII")                 # magic, model version (4|5), n tensors


class WeightFormatError(ValueError):
    pass


# ----------------------------------------------------------------------------------
# ONNX -> canonical
# ----------------------------------------------------------------------------------

def _find_branch(model: onnx_lite.Model, which: str) -> Tuple[str, onnx_lite.Graph]:
    """Locate the top-level ``If(sr == 16000)`` and return its then-branch (16 kHz)."""
    for node in model.graph.nodes:
        if node.op_type == "If" and which in node.attrs:
            return node.name, node.attrs[which]
    raise WeightFormatError("top-level If(sr==16000) node not found")


def _iofc_to_ifgo(a: np.ndarray, hidden: int) -> np.ndarray:
    """ONNX LSTM packs gate blocks as i,o,f,c; canonical order here is PyTorch's i,f,g,o."""
    i, o, f, c = (a[k * hidden:(k + 1) * hidden] for k in range(4))
    return np.concatenate([i, f, c, o], axis=0)


def _extract_v5(model: onnx_lite.Model, sample_rate: int = 16000) -> Dict[str, np.ndarray]:
    """16 kHz sub-model (``If_0`` then-branch) or the 8 kHz one (else-branch: window 128, hop 64, 65 bins,
    ``encoder.0`` 65 -> 128, its own weights throughout; it runs on 256-sample frames - SURVEY a9 / f3).  Same tensor
    names; the 8 kHz blob carries ``meta.variant = 8000``."""
    k8 = sample_rate != 16000
    nfft, nbin = (128, 65) if k8 else (256, 129)
    _, g = _find_branch(model, "else_branch" if k8 else "then_branch")
    c = {}
    for node in g.nodes:
        if node.op_type == "Constant" and node.outputs:
            # spox prefixes inlined names with "If_0_then_branch__Inline_0__"
            c[node.outputs[0].split("__Inline_0__")[-1]] = node.attrs["value"]
    try:
        t = {
            "stft.basis": c["stft.forward_basis_buffer"].reshape(2 * nbin, nfft),
            "lstm.w_ih": c["decoder.rnn.weight_ih"],
            "lstm.w_hh": c["decoder.rnn.weight_hh"],
            "lstm.b_ih": c["decoder.rnn.bias_ih"],
            "lstm.b_hh": c["decoder.rnn.bias_hh"],
            "head.w": c["decoder.decoder.2.weight"].reshape(128),
            "head.b": c["decoder.decoder.2.bias"].reshape(1),
        }
        for i in range(4):
            t[f"enc{i}.w"] = c[f"encoder.{i}.reparam_conv.weight"]
            t[f"enc{i}.b"] = c[f"encoder.{i}.reparam_conv.bias"]
        if t["enc0.w"].shape != (128, nbin, 3):
            raise WeightFormatError(f"not a Silero V5 graph: encoder.0 has shape {t['enc0.w'].shape}")
        if k8:
            t["meta.variant"] = np.array([8000.0], np.float32)
    except KeyError as e:  # pragma: no cover - wrong file
        raise WeightFormatError(f"not a Silero V5 graph: missing tensor {e}") from e
    return t


def _extract_v4(model: onnx_lite.Model, sample_rate: int = 16000) -> Dict[str, np.ndarray]:
    """16 kHz sub-model (then-branch, ``model.*``) or the 8 kHz one (else-branch, ``model_8k.*``) that the graph
    takes for EVERY ``sr != 16000``.  Same tensor names either way; the 8 kHz blob carries
    ``meta.variant = 8000`` (its third stride conv has stride 1, which the engines must know)."""
    k8 = sample_rate != 16000
    pre = "model_8k" if k8 else "model"
    top = {(("model" + k[len(pre):]) if k.startswith(pre + ".") else k): v for k, v in model.graph.initializers.items()
           if not (k.startswith("model_8k.") and not k8)}
    _, g = _find_branch(model, "else_branch" if k8 else "then_branch")
    # the two stacked LSTMs live in the "state supplied" sub-branch of the inner If
    lstm_inits = None
    for node in g.nodes:
        if node.op_type == "If" and len(node.outputs) == 3:
            lstm_inits = node.attrs["then_branch"]
    if lstm_inits is None:
        raise WeightFormatError("not a Silero V4 graph: LSTM branch not found")
    lstm_nodes = [n for n in lstm_inits.nodes if n.op_type == "LSTM"]
    if len(lstm_nodes) != 2:
        raise WeightFormatError("not a Silero V4 graph: expected two LSTM nodes")
    sq = lambda a: np.squeeze(a, axis=-1)
    try:
        t = {
            "stft.basis": top["model.feature_extractor.forward_basis_buffer"].reshape(258, 256),
            "norm.filter": top["model.adaptive_normalization.filter_"].reshape(7),
            "head.w": top["model.decoder.decoder.1.weight"].reshape(64),
            "head.b": top["model.decoder.decoder.1.bias"].reshape(1),
        }
        blocks = [("l0", "model.first_layer.0", True), ("l1", "model.encoder.3.0", True),
                  ("l2", "model.encoder.7.0", False), ("l3", "model.encoder.11.0", True)]
        for short, long, has_proj in blocks:
            t[f"{short}.dw.w"] = top[f"{long}.dw_conv.0.weight"].reshape(-1, 5)
            t[f"{short}.dw.b"] = top[f"{long}.dw_conv.0.bias"]
            t[f"{short}.pw.w"] = sq(top[f"{long}.pw_conv.0.weight"])
            t[f"{short}.pw.b"] = top[f"{long}.pw_conv.0.bias"]
            if has_proj:
                t[f"{short}.proj.w"] = sq(top[f"{long}.proj.weight"])
                t[f"{short}.proj.b"] = top[f"{long}.proj.bias"]
        # the stride convs between blocks are anonymous initializers; find them by consumer
        convs = [n for n in g.nodes if n.op_type == "Conv" and n.inputs[1].isdigit()]
        if len(convs) != 4:
            raise WeightFormatError("not a Silero V4 graph: expected 4 anonymous 1x1 convs")
        for i, n in enumerate(convs):
            t[f"s{i}.w"] = sq(top[n.inputs[1]])
            t[f"s{i}.b"] = top[n.inputs[2]]
        for li, n in enumerate(lstm_nodes):
            W = lstm_inits.initializers[n.inputs[1]][0]
            R = lstm_inits.initializers[n.inputs[2]][0]
            B = lstm_inits.initializers[n.inputs[3]][0]
            H = n.attrs["hidden_size"]
            t[f"lstm{li}.w_ih"] = _iofc_to_ifgo(W, H)
            t[f"lstm{li}.w_hh"] = _iofc_to_ifgo(R, H)
            t[f"lstm{li}.b_ih"] = _iofc_to_ifgo(B[:4 * H], H)
            t[f"lstm{li}.b_hh"] = _iofc_to_ifgo(B[4 * H:], H)
        if k8:
            t["meta.variant"] = np.array([8000.0], np.float32)
    except KeyError as e:  # pragma: no cover - wrong file
        raise WeightFormatError(f"not a Silero V4 graph: missing tensor {e}") from e
    return t


def extract_from_onnx(path: str, version: int, sample_rate: int = 16000) -> Dict[str, np.ndarray]:
    model = onnx_lite.load_model(path)
    n_in, n_out = len(model.graph.inputs), len(model.graph.outputs)
    # same arity rule as the reference's _validate_model_signature (silero_model.py:369-376)
    exp_in, exp_out = (3, 2) if version == 5 else (4, 3)
    if n_in != exp_in:
        raise WeightFormatError(f"Expected {exp_in} inputs, got {n_in}")
    if n_out != exp_out:
        raise WeightFormatError(f"Expected {exp_out} outputs, got {n_out}")
    t = _extract_v5(model, sample_rate) if version == 5 else _extract_v4(model, sample_rate)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in t.items()}


# ----------------------------------------------------------------------------------
# SVW container
# ----------------------------------------------------------------------------------

def pack_svw(version: int, tensors: Dict[str, np.ndarray]) -> bytes:
    names = sorted(tensors)
    table_end = _HEADER.size + _ENTRY.size * len(names)
    off = (table_end + 63) // 64 * 64
    entries = []
    chunks = []
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype="<f4")
        if a.ndim > 4 or len(n.encode()) >= _NAME_LEN:
            raise WeightFormatError(f"tensor {n}: unsupported rank/name")
        dims = list(a.shape) + [1] * (4 - a.ndim)
        entries.append(_ENTRY.pack(n.encode(), a.ndim, *dims, 0, off, a.size))
        chunks.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64
    buf = bytearray(off)
    _HEADER.pack_into(buf, 0, SVW_MAGIC, version, len(names))
    pos = _HEADER.size
    for e in entries:
        buf[pos:pos + len(e)] = e
        pos += len(e)
    for o, b in chunks:
        buf[o:o + len(b)] = b
    return bytes(buf)


def unpack_svw(blob: bytes) -> Tuple[int, Dict[str, np.ndarray]]:
    if len(blob) < _HEADER.size or blob[:8] != SVW_MAGIC:
        raise WeightFormatError("not an SVW weight blob (bad magic)")
    _, version, n = _HEADER.unpack_from(blob, 0)
    out = {}
    pos = _HEADER.size
    for _ in range(n):
        name, ndim, d0, d1, d2, d3, _rsv, off, nelem = _ENTRY.unpack_from(blob, pos)
        pos += _ENTRY.size
        shape = (d0, d1, d2, d3)[:ndim]
        a = np.frombuffer(blob, dtype="<f4", count=nelem, offset=off).reshape(shape)
        out[name.rstrip(b"\0").decode()] = a.astype(np.float32)
    return version, out


def is_8k_variant(version: int, sample_rate: int) -> bool:
    """Both graphs run their 8 kHz sub-model for every rate but 16 000 (``Equal(sr, 16000)``).  V4's takes the
    reference's 512-sample frames at any of those rates; V5's only makes sense on native 8 kHz audio in 256-sample
    frames (:func:`frame_samples`) - with 512-sample frames a 3-D tensor reaches its LSTM and onnxruntime refuses
    (SURVEY a9), which the host mirror reproduces for every other combination."""
    return int(sample_rate) != 16000


def frame_samples(version: int, sample_rate: int) -> int:
    """Samples one model step consumes: 512 (core/silero_model.py:464-468), except Silero V5's 8 kHz sub-model: 256."""
    return 256 if version == 5 and is_8k_variant(version, sample_rate) else 512


def packaged_blob_path(version: int, sample_rate: int = 16000) -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    tag = "8k" if is_8k_variant(version, sample_rate) else "16k"
    return os.path.join(here, "weights", f"silero_v{version}_{tag}.svw")


def load_weight_blob(model_path: str, version: int, sample_rate: int = 16000) -> bytes:
    """Return the SVW blob for ``model_path`` — an ``.onnx`` file (converted on the fly,
    like the reference's ``_load_model``) or an ``.svw`` blob."""
    if model_path.endswith(".svw"):
        with open(model_path, "rb") as f:
            blob = f.read()
        v, t = unpack_svw(blob)
        if v != version:
            raise WeightFormatError(f"weight blob is for Silero v{v}, requested v{version}")
        if ("meta.variant" in t) != is_8k_variant(version, sample_rate):
            raise WeightFormatError(f"weight blob is the {'8' if 'meta.variant' in t else '16'} kHz sub-model, "
                                    f"requested sample rate {sample_rate}")
        return blob
    return pack_svw(version, extract_from_onnx(model_path, version, sample_rate))

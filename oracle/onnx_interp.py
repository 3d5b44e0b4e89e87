"""ORACLE / TEST INFRASTRUCTURE — generic ONNX opset-16 graph interpreter in NumPy.

This is *not* product code.  Only ``tests/``, ``tools/make_goldens.py``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use anything
under ``oracle/``.

Purpose: the reference's arithmetic for this path lives in third-party ``onnxruntime``
(pinned ``>=1.10.0`` in /root/reference/pyproject.toml:32; call sites
/root/reference/src/real_time_vad/core/silero_model.py:321-325 and :433), which is not
installed here.  What *is* in-tree is the graph + weights
(/root/reference/src/real_time_vad/models/silero_vad{,_v5}.onnx).  This file executes
those graphs node by node following the published ONNX operator specification
(opset 16): every op below is implemented from the spec text, independently of the fused
restatement in ``oracle/silero_oracle.c`` — the two must agree (tests/test_oracle.py),
which is how the golden vectors in tests/golden/ were produced.

PARITY STATUS: "parity unpinned" against onnxruntime itself — the reference's own tests
mock ``InferenceSession.run`` (/root/reference/tests/test_silero_model.py:362-365) and
hold no numeric fixtures for the model; see DESIGN.md §Oracle.

``compute_dtype`` selects float64 (reference-quality "exact" evaluation of the fp32
weights) or float32 (same arithmetic width as ORT's CPU EP).
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from cutter_vad_amd import onnx_lite as ol

_ONNX_TO_NP = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}


class _Scope:
    def __init__(self, parent: Optional["_Scope"] = None):
        self.vals: Dict[str, Any] = {}
        self.parent = parent

    def get(self, name: str) -> Any:
        s: Optional[_Scope] = self
        while s is not None:
            if name in s.vals:
                return s.vals[name]
            s = s.parent
        raise KeyError(f"ONNX value {name!r} not found in any enclosing scope")

    def set(self, name: str, v: Any) -> None:
        self.vals[name] = v


class OnnxInterpreter:
    def __init__(self, path: str, compute_dtype=np.float64):
        self.model = ol.load_model(path)
        self.fdt = np.dtype(compute_dtype)
        self.trace: Dict[str, np.ndarray] = {}
        self.keep_trace = False

    # ------------------------------------------------------------------ public
    def run(self, feeds: Dict[str, np.ndarray]) -> List[np.ndarray]:
        top = _Scope()
        for k, v in feeds.items():
            v = np.asarray(v)
            top.set(k, v.astype(self.fdt) if v.dtype.kind == "f" else v)
        return self._run_graph(self.model.graph, top)

    # ----------------------------------------------------------------- helpers
    def _f(self, a: np.ndarray) -> np.ndarray:
        a = np.asarray(a)
        return a.astype(self.fdt) if a.dtype.kind == "f" else a

    def _run_graph(self, g: ol.Graph, scope: _Scope) -> List[np.ndarray]:
        for name, arr in g.initializers.items():
            scope.set(name, self._f(arr))
        for node in g.nodes:
            ins = [scope.get(i) if i else None for i in node.inputs]
            outs = self._exec(node, ins, scope)
            for oname, o in zip(node.outputs, outs):
                if oname:
                    scope.set(oname, o)
                    if self.keep_trace and isinstance(o, np.ndarray):
                        self.trace[oname] = o
        return [scope.get(o) for o in g.outputs]

    def _exec(self, node: ol.Node, x: List[Any], scope: _Scope) -> Sequence[Any]:
        fn = getattr(self, "op_" + node.op_type, None)
        if fn is None:
            raise NotImplementedError(f"ONNX op {node.op_type} ({node.name})")
        r = fn(node, x, scope)
        return r if isinstance(r, (list, tuple)) else [r]

    # --------------------------------------------------------------------- ops
    def op_Constant(self, n, x, s):
        return self._f(n.attrs["value"])

    def op_Identity(self, n, x, s):
        return x[0]

    def op_Equal(self, n, x, s):
        return np.equal(x[0], x[1])

    def op_Not(self, n, x, s):
        return np.logical_not(x[0])

    def op_If(self, n, x, s):
        cond = bool(np.asarray(x[0]).reshape(-1)[0])
        branch = n.attrs["then_branch"] if cond else n.attrs["else_branch"]
        return self._run_graph(branch, _Scope(s))

    def op_ConstantOfShape(self, n, x, s):
        val = n.attrs.get("value", np.zeros(1, np.float32))
        val = np.asarray(val).reshape(-1)[0]
        out = np.full(tuple(int(d) for d in x[0]), val)
        return self._f(out)

    def op_Concat(self, n, x, s):
        return np.concatenate([np.asarray(a) for a in x], axis=n.attrs["axis"])

    def op_Reshape(self, n, x, s):
        data, shape = x[0], [int(d) for d in x[1]]
        if not n.attrs.get("allowzero", 0):
            shape = [data.shape[i] if d == 0 else d for i, d in enumerate(shape)]
        return np.reshape(data, shape)

    def op_Slice(self, n, x, s):
        data = x[0]
        starts = [int(v) for v in x[1]]
        ends = [int(v) for v in x[2]]
        axes = [int(v) for v in x[3]] if len(x) > 3 and x[3] is not None else list(range(len(starts)))
        steps = [int(v) for v in x[4]] if len(x) > 4 and x[4] is not None else [1] * len(starts)
        sl = [slice(None)] * data.ndim
        for st, en, ax, sp in zip(starts, ends, axes, steps):
            ax = ax % data.ndim
            dim = data.shape[ax]
            # spec: clamp starts/ends; INT64_MAX / INT64_MIN mean "to the end"
            if sp > 0:
                st = min(max(st + dim if st < 0 else st, 0), dim)
                en = min(max(en + dim if en < 0 else en, 0), dim)
                sl[ax] = slice(st, en, sp)
            else:
                st = min(max(st + dim if st < 0 else st, 0), dim - 1)
                en_c = en + dim if en < 0 else en
                en_c = min(max(en_c, -1), dim - 1)
                sl[ax] = slice(st, None if en_c < 0 else en_c, sp)
        return data[tuple(sl)]

    def op_Transpose(self, n, x, s):
        return np.transpose(x[0], n.attrs.get("perm"))

    def op_Cast(self, n, x, s):
        to = n.attrs["to"]
        if to in (1, 11):
            return np.asarray(x[0]).astype(self.fdt)
        return np.asarray(x[0]).astype(_ONNX_TO_NP[to])

    def op_Pad(self, n, x, s):
        data = x[0]
        pads = [int(p) for p in x[1]]
        nd = data.ndim
        width = [(pads[i], pads[i + nd]) for i in range(nd)]
        mode = n.attrs.get("mode", b"constant").decode()
        if mode == "constant":
            cv = 0 if len(x) < 3 or x[2] is None else np.asarray(x[2]).reshape(-1)[0]
            return np.pad(data, width, mode="constant", constant_values=cv)
        return np.pad(data, width, mode={"reflect": "reflect", "edge": "edge"}[mode])

    def op_Unsqueeze(self, n, x, s):
        out = np.asarray(x[0])
        axes = sorted(int(a) % (out.ndim + len(x[1])) for a in np.asarray(x[1]).reshape(-1))
        for a in axes:
            out = np.expand_dims(out, a)
        return out

    def op_Squeeze(self, n, x, s):
        a = np.asarray(x[0])
        if len(x) > 1 and x[1] is not None:
            return np.squeeze(a, axis=tuple(int(v) for v in np.asarray(x[1]).reshape(-1)))
        return np.squeeze(a)

    def op_Shape(self, n, x, s):
        return np.asarray(np.asarray(x[0]).shape, dtype=np.int64)

    def op_Size(self, n, x, s):
        return np.asarray(np.asarray(x[0]).size, dtype=np.int64)

    def op_Gather(self, n, x, s):
        return np.take(x[0], np.asarray(x[1]), axis=n.attrs.get("axis", 0))

    def op_Conv(self, n, x, s):
        X, W = x[0], x[1]
        B = x[2] if len(x) > 2 else None
        assert X.ndim == 3 and W.ndim == 3, "only 1-D convolutions occur in the Silero graphs"
        group = n.attrs.get("group", 1)
        stride = n.attrs.get("strides", [1])[0]
        dil = n.attrs.get("dilations", [1])[0]
        pl, pr = n.attrs.get("pads", [0, 0])
        if pl or pr:
            X = np.pad(X, ((0, 0), (0, 0), (pl, pr)))
        Bn, C, L = X.shape
        M, Cg, K = W.shape
        assert C == Cg * group
        Lout = (L - dil * (K - 1) - 1) // stride + 1
        # cross-correlation (ONNX Conv does not flip the kernel)
        idx = (np.arange(Lout)[:, None] * stride + np.arange(K)[None, :] * dil)  # [Lout,K]
        cols = X[:, :, idx]  # [B,C,Lout,K]
        Mg = M // group
        out = np.empty((Bn, M, Lout), dtype=self.fdt)
        for g in range(group):
            xc = cols[:, g * Cg:(g + 1) * Cg]            # [B,Cg,Lout,K]
            wc = W[g * Mg:(g + 1) * Mg]                  # [Mg,Cg,K]
            out[:, g * Mg:(g + 1) * Mg] = np.einsum("bclk,mck->bml", xc, wc)
        if B is not None:
            out += B[None, :, None]
        return out

    def op_Pow(self, n, x, s):
        return np.power(x[0], x[1].astype(x[0].dtype) if hasattr(x[1], "astype") else x[1])

    def op_Add(self, n, x, s):
        return x[0] + x[1]

    def op_Mul(self, n, x, s):
        return x[0] * x[1]

    def op_Neg(self, n, x, s):
        return -x[0]

    def op_Log(self, n, x, s):
        return np.log(x[0])

    def op_Sqrt(self, n, x, s):
        return np.sqrt(x[0])

    def op_Relu(self, n, x, s):
        return np.maximum(x[0], 0)

    def op_Sigmoid(self, n, x, s):
        return 1.0 / (1.0 + np.exp(-x[0]))

    def op_ReduceMean(self, n, x, s):
        axes = n.attrs.get("axes")
        kd = bool(n.attrs.get("keepdims", 1))
        return np.mean(x[0], axis=tuple(axes) if axes is not None else None, keepdims=kd)

    def op_LSTM(self, n, x, s):
        """ONNX LSTM, forward direction, default activations (sigmoid, tanh, tanh),
        gate order i, o, f, c in W/R/B, no peepholes, no clip (spec for opset 14)."""
        X, W, R = x[0], x[1], x[2]
        B = x[3] if len(x) > 3 else None
        h0 = x[5] if len(x) > 5 else None
        c0 = x[6] if len(x) > 6 else None
        assert n.attrs.get("direction", b"forward") == b"forward"
        assert (len(x) <= 7 or x[7] is None), "peepholes not used by Silero"
        H = n.attrs["hidden_size"]
        T, Bn, _ = X.shape
        Wd, Rd = W[0], R[0]
        bias = np.zeros(8 * H, dtype=self.fdt) if B is None else B[0]
        Wb, Rb = bias[:4 * H], bias[4 * H:]
        h = np.zeros((Bn, H), dtype=self.fdt) if h0 is None else h0[0]
        c = np.zeros((Bn, H), dtype=self.fdt) if c0 is None else c0[0]
        sig = lambda v: 1.0 / (1.0 + np.exp(-v))
        Y = np.empty((T, 1, Bn, H), dtype=self.fdt)
        for t in range(T):
            g = X[t] @ Wd.T + h @ Rd.T + Wb + Rb
            i, o, f, cc = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = sig(f) * c + sig(i) * np.tanh(cc)
            h = sig(o) * np.tanh(c)
            Y[t, 0] = h
        return [Y, h[None], c[None]]


class SileroOnnxSession:
    """Drop-in for the three ``InferenceSession`` calls the reference makes
    (silero_model.py:365-366 ``get_inputs/get_outputs``, :433 ``run``)."""

    def __init__(self, path: str, compute_dtype=np.float64):
        self.interp = OnnxInterpreter(path, compute_dtype)

    def get_inputs(self):
        return list(self.interp.model.graph.inputs)

    def get_outputs(self):
        return list(self.interp.model.graph.outputs)

    def run(self, _names, feeds: Dict[str, np.ndarray]) -> List[np.ndarray]:
        outs = self.interp.run(feeds)
        return [np.asarray(o, dtype=np.float32) for o in outs]

"""Minimal ONNX protobuf wire-format reader (no `onnx` / `onnxruntime` dependency).

The reference loads ``silero_vad{,_v5}.onnx`` through ``onnxruntime.InferenceSession``
(/root/reference/src/real_time_vad/core/silero_model.py:321-325).  The MI355X engine does
not execute the ONNX graph; it only needs the *tensors* stored in the file, so this module
decodes just enough of the protobuf wire format (ModelProto / GraphProto / NodeProto /
AttributeProto / TensorProto, including nested ``If`` sub-graphs and ``Constant`` nodes) to
hand back plain Python objects and numpy arrays.

Field numbers follow the public ``onnx.proto3`` schema (IR version 8, opset 16 files).
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# protobuf wire primitives
# --------------------------------------------------------------------------------------


def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _fields(buf: memoryview) -> Iterator[Tuple[int, int, Any]]:
    """Yield (field_number, wire_type, value) for every field of one message."""
    pos = 0
    end = len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
            yield fno, wt, v
        elif wt == 1:
            yield fno, wt, bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            yield fno, wt, buf[pos:pos + n]
            pos += n
        elif wt == 5:
            yield fno, wt, bytes(buf[pos:pos + 4])
            pos += 4
        else:  # groups (3/4) never appear in ONNX files
            raise ValueError(f"unsupported protobuf wire type {wt}")


def _packed_varints(v: Any, wt: int) -> List[int]:
    if wt == 0:
        return [_signed64(v)]
    out = []
    pos = 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed64(x))
    return out


# --------------------------------------------------------------------------------------
# ONNX messages
# --------------------------------------------------------------------------------------

_DTYPES = {
    1: np.float32, 2: np.uint8, 3: np.int8, 4: np.uint16, 5: np.int16, 6: np.int32,
    7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64,
}


def parse_tensor(buf: memoryview) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype = 0
    name = ""
    raw: Optional[bytes] = None
    floats: List[float] = []
    int32s: List[int] = []
    int64s: List[int] = []
    doubles: List[float] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v, wt)
        elif fno == 2:
            dtype = v
        elif fno == 4:
            if wt == 5:
                floats.append(struct.unpack("<f", v)[0])
            else:
                floats += list(np.frombuffer(bytes(v), dtype="<f4"))
        elif fno == 5:
            int32s += _packed_varints(v, wt)
        elif fno == 7:
            int64s += _packed_varints(v, wt)
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 10:
            if wt == 1:
                doubles.append(struct.unpack("<d", v)[0])
            else:
                doubles += list(np.frombuffer(bytes(v), dtype="<f8"))
    if dtype not in _DTYPES:
        raise ValueError(f"tensor {name!r}: unsupported ONNX data_type {dtype}")
    np_dtype = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_dtype.newbyteorder("<")).astype(np_dtype)
    elif floats:
        arr = np.asarray(floats, dtype=np_dtype)
    elif int64s:
        arr = np.asarray(int64s, dtype=np_dtype)
    elif int32s:
        arr = np.asarray(int32s, dtype=np.int64).astype(np_dtype)
    elif doubles:
        arr = np.asarray(doubles, dtype=np_dtype)
    else:
        arr = np.zeros(0, dtype=np_dtype)
    return name, arr.reshape(dims) if dims or arr.size == 1 else arr


@dataclass
class Node:
    op_type: str = ""
    name: str = ""
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    attrs: Dict[str, Any] = field(default_factory=dict)


@dataclass
class Graph:
    name: str = ""
    nodes: List[Node] = field(default_factory=list)
    initializers: Dict[str, np.ndarray] = field(default_factory=dict)
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)


def _parse_value_info_name(buf: memoryview) -> str:
    for fno, _wt, v in _fields(buf):
        if fno == 1:
            return bytes(v).decode()
    return ""


def _parse_attribute(buf: memoryview) -> Tuple[str, Any]:
    name = ""
    val: Any = None
    ints: List[int] = []
    floats: List[float] = []
    strings: List[bytes] = []
    atype = 0
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _signed64(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = parse_tensor(v)[1]
        elif fno == 6:
            val = parse_graph(v)
        elif fno == 7:
            if wt == 5:
                floats.append(struct.unpack("<f", v)[0])
            else:
                floats += list(np.frombuffer(bytes(v), dtype="<f4"))
        elif fno == 8:
            ints += _packed_varints(v, wt)
        elif fno == 9:
            strings.append(bytes(v))
        elif fno == 20:
            atype = v
    if atype == 7 or (val is None and ints):
        val = ints
    elif atype == 6 or (val is None and floats):
        val = floats
    elif atype == 8 or (val is None and strings):
        val = strings
    return name, val


def _parse_node(buf: memoryview) -> Node:
    n = Node()
    for fno, _wt, v in _fields(buf):
        if fno == 1:
            n.inputs.append(bytes(v).decode())
        elif fno == 2:
            n.outputs.append(bytes(v).decode())
        elif fno == 3:
            n.name = bytes(v).decode()
        elif fno == 4:
            n.op_type = bytes(v).decode()
        elif fno == 5:
            k, a = _parse_attribute(v)
            n.attrs[k] = a
    return n


def parse_graph(buf: memoryview) -> Graph:
    g = Graph()
    for fno, _wt, v in _fields(buf):
        if fno == 1:
            g.nodes.append(_parse_node(v))
        elif fno == 2:
            g.name = bytes(v).decode()
        elif fno == 5:
            name, arr = parse_tensor(v)
            g.initializers[name] = arr
        elif fno == 11:
            g.inputs.append(_parse_value_info_name(v))
        elif fno == 12:
            g.outputs.append(_parse_value_info_name(v))
    return g


@dataclass
class Model:
    graph: Graph
    producer: str = ""
    opset: int = 0
    ir_version: int = 0


def load_model(path: str) -> Model:
    with open(path, "rb") as f:
        data = memoryview(f.read())
    graph: Optional[Graph] = None
    producer = ""
    opset = 0
    ir = 0
    for fno, _wt, v in _fields(data):
        if fno == 1:
            ir = v
        elif fno == 2:
            producer = bytes(v).decode()
        elif fno == 7:
            graph = parse_graph(v)
        elif fno == 8:
            for f2, _w2, v2 in _fields(v):
                if f2 == 2:
                    opset = max(opset, v2)
    if graph is None:
        raise ValueError(f"{path}: no GraphProto found (not an ONNX model?)")
    return Model(graph=graph, producer=producer, opset=opset, ir_version=ir)


def walk_graphs(g: Graph, prefix: str = "") -> Iterator[Tuple[str, Graph]]:
    """Yield (path, graph) for ``g`` and every sub-graph nested in node attributes."""
    yield prefix or g.name, g
    for node in g.nodes:
        for aname, a in node.attrs.items():
            if isinstance(a, Graph):
                yield from walk_graphs(a, f"{prefix or g.name}/{node.name or node.op_type}.{aname}")


def collect_tensors(g: Graph) -> Dict[str, Dict[str, np.ndarray]]:
    """Per (sub)graph path: initializers plus the payload of every ``Constant`` node."""
    out: Dict[str, Dict[str, np.ndarray]] = {}
    for path, sub in walk_graphs(g):
        d = dict(sub.initializers)
        for node in sub.nodes:
            if node.op_type == "Constant" and "value" in node.attrs and node.outputs:
                d[node.outputs[0]] = node.attrs["value"]
        out[path] = d
    return out

"""Minimal reader / writer for the ONNX protobuf wire format (no `onnx` package in this image; executes nothing from the file).

Only what the Silero VAD graph needs: ModelProto.graph -> nodes (op_type, inputs, outputs, attributes incl. sub-graphs and
tensors) and initializers (name, dims, data).  Field numbers are those of onnx.proto3 (public schema):
ModelProto.graph = 7; GraphProto.node = 1, name = 2, initializer = 5, input = 11, output = 12;
NodeProto.input = 1, output = 2, name = 3, op_type = 4, attribute = 5;
AttributeProto.name = 1, f = 2, i = 3, s = 4, t = 5, g = 6, floats = 7, ints = 8, type = 20;
TensorProto.dims = 1, data_type = 2, float_data = 4, int32_data = 5, int64_data = 7, name = 8, raw_data = 9;
ValueInfoProto.name = 1.
"""
import struct

import numpy as np


def _varint(b, i):
    v, s = 0, 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, i
        s += 7


def fields(b):
    """Yield (field number, wire type, value) of one message; length-delimited values as memoryview slices."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = bytes(b[i:i + 8]); i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]; i += ln
        elif wt == 5:
            v = bytes(b[i:i + 4]); i += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield f, wt, v


def _sint(v):  # int64 carried as a two's-complement varint
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v):
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_sint(x))
    return out


_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def parse_tensor(b):
    dims, dt, raw, name, fl, i32, i64 = [], 1, None, "", [], [], []
    for f, wt, v in fields(b):
        if f == 1:
            dims += _packed_varints(v) if wt == 2 else [_sint(v)]
        elif f == 2:
            dt = v
        elif f == 4:
            fl += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 5:
            i32 += _packed_varints(v) if wt == 2 else [_sint(v)]
        elif f == 7:
            i64 += _packed_varints(v) if wt == 2 else [_sint(v)]
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
    np_t = _DT[dt]
    if raw is not None:
        a = np.frombuffer(raw, np.dtype(np_t).newbyteorder("<")).astype(np_t)
    elif fl:
        a = np.array(fl, np_t)
    elif i64:
        a = np.array(i64, np_t)
    elif i32:
        a = np.array(i32, np_t)
    else:
        a = np.zeros(0, np_t)
    return name, a.reshape(dims) if dims else (a.reshape(()) if a.size == 1 else a)


def parse_attr(b):
    name, val, ints, floats = "", None, [], []
    for f, wt, v in fields(b):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _sint(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5:
            val = parse_tensor(v)[1]
        elif f == 6:
            val = parse_graph(v)
        elif f == 7:
            floats += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            ints += _packed_varints(v) if wt == 2 else [_sint(v)]
    if val is None:
        val = ints if ints else floats
    return name, val


def parse_node(b):
    n = {"input": [], "output": [], "name": "", "op": "", "attr": {}}
    for f, wt, v in fields(b):
        if f == 1:
            n["input"].append(bytes(v).decode())
        elif f == 2:
            n["output"].append(bytes(v).decode())
        elif f == 3:
            n["name"] = bytes(v).decode()
        elif f == 4:
            n["op"] = bytes(v).decode()
        elif f == 5:
            k, a = parse_attr(v)
            n["attr"][k] = a
    return n


def _value_name(b):
    for f, wt, v in fields(b):
        if f == 1:
            return bytes(v).decode()
    return ""


def parse_graph(b):
    g = {"name": "", "nodes": [], "init": {}, "inputs": [], "outputs": []}
    for f, wt, v in fields(b):
        if f == 1:
            g["nodes"].append(parse_node(v))
        elif f == 2:
            g["name"] = bytes(v).decode()
        elif f == 5:
            k, a = parse_tensor(v)
            g["init"][k] = a
        elif f == 11:
            g["inputs"].append(_value_name(v))
        elif f == 12:
            g["outputs"].append(_value_name(v))
    return g


def load(path):
    b = memoryview(open(path, "rb").read())
    for f, wt, v in fields(b):
        if f == 7:
            return parse_graph(v)
    raise ValueError("no graph in " + path)


# ---- writer (tests: synthetic models of the same structure) ----
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = v & 0x7F
        v >>= 7
        out.append(c | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _enc_varint(field << 3 | 2) + _enc_varint(len(payload)) + payload


def _vi(field, v):
    return _enc_varint(field << 3) + _enc_varint(v)


def enc_tensor(name, a):
    a = np.asarray(a)
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7, np.dtype(np.int32): 6}[a.dtype]
    out = b"".join(_vi(1, d) for d in a.shape) + _vi(2, dt) + _ld(8, name.encode()) + _ld(9, a.astype(a.dtype.newbyteorder("<")).tobytes())
    return out


def enc_attr(name, v):
    out = _ld(1, name.encode())
    if isinstance(v, float):
        out += _enc_varint(2 << 3 | 5) + struct.pack("<f", v) + _vi(20, 1)
    elif isinstance(v, int):
        out += _vi(3, v) + _vi(20, 2)
    elif isinstance(v, bytes):
        out += _ld(4, v) + _vi(20, 3)
    elif isinstance(v, np.ndarray):
        out += _ld(5, enc_tensor("", v)) + _vi(20, 4)
    elif isinstance(v, dict):
        out += _ld(6, enc_graph(v)) + _vi(20, 5)
    elif isinstance(v, (list, tuple)):
        out += b"".join(_vi(8, int(x)) for x in v) + _vi(20, 7)
    else:
        raise TypeError(type(v))
    return out


def enc_node(n):
    out = b"".join(_ld(1, s.encode()) for s in n["input"]) + b"".join(_ld(2, s.encode()) for s in n["output"])
    out += _ld(3, n.get("name", "").encode()) + _ld(4, n["op"].encode())
    out += b"".join(_ld(5, enc_attr(k, v)) for k, v in n.get("attr", {}).items())
    return out


def enc_graph(g):
    out = b"".join(_ld(1, enc_node(n)) for n in g["nodes"]) + _ld(2, g.get("name", "g").encode())
    out += b"".join(_ld(5, enc_tensor(k, a)) for k, a in g.get("init", {}).items())
    out += b"".join(_ld(11, _ld(1, s.encode())) for s in g.get("inputs", []))
    out += b"".join(_ld(12, _ld(1, s.encode())) for s in g.get("outputs", []))
    return out


def save(path, graph):
    with open(path, "wb") as f:
        f.write(_vi(1, 8) + _ld(7, enc_graph(graph)))

"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

Minimal protobuf wire-format reader for ONNX ``ModelProto`` files (no ``onnx``
package in this image).  Field numbers follow SURVEY.md Appendix C, which was
verified against the reference's tiny fixtures
(``/root/reference/tests/stable-diffusion/*.onnx``).

Only what the oracle needs: graph nodes (with attributes), initializers (raw_data,
typed repeated fields and single-file external data as written by
``scripts/sd2pyke.py:306-316``), graph inputs/outputs names.

The product has its own, independent C++ reader in
``diffusers_amd/csrc/onnx_reader.cpp``; the two are cross-checked in tests.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# --------------------------------------------------------------------------- wire primitives


def _varint(buf: memoryview, pos: int):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            break
        shift += 7
    return result, pos


def _fields(buf: memoryview):
    """Yield (field_number, wire_type, value) where value is int (wt 0/1/5) or memoryview (wt 2)."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fn, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v, wt) -> List[int]:
    if wt == 0:
        return [_signed64(v)]
    out = []
    pos = 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed64(x))
    return out


# --------------------------------------------------------------------------- data model

DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


@dataclass
class Tensor:
    name: str = ""
    dims: List[int] = field(default_factory=list)
    data_type: int = 0
    raw: Optional[bytes] = None
    float_data: List[float] = field(default_factory=list)
    int32_data: List[int] = field(default_factory=list)
    int64_data: List[int] = field(default_factory=list)
    external: Dict[str, str] = field(default_factory=dict)
    data_location: int = 0

    def numpy(self, base_dir: str = ".") -> np.ndarray:
        dt = DTYPES[self.data_type]
        shape = tuple(self.dims)
        if self.data_location == 1 or self.external:
            path = os.path.join(base_dir, self.external["location"])
            off = int(self.external.get("offset", 0))
            cnt = int(np.prod(shape)) if shape else 1
            arr = np.fromfile(path, dtype=dt, count=cnt, offset=off)
            return arr.reshape(shape)
        if self.raw is not None:
            return np.frombuffer(self.raw, dtype=dt).reshape(shape).copy()
        if self.data_type == 1:
            return np.asarray(self.float_data, dtype=np.float32).reshape(shape)
        if self.data_type == 7:
            return np.asarray(self.int64_data, dtype=np.int64).reshape(shape)
        if self.data_type == 10:  # fp16 bit patterns carried in int32_data (float16.py:93-108)
            return np.asarray(self.int32_data, dtype=np.uint16).view(np.float16).reshape(shape)
        if self.data_type in (6, 9, 2, 3):
            return np.asarray(self.int32_data, dtype=dt).reshape(shape)
        raise ValueError(f"tensor {self.name}: no data")


@dataclass
class Attribute:
    name: str = ""
    f: Optional[float] = None
    i: Optional[int] = None
    s: Optional[bytes] = None
    t: Optional[Tensor] = None
    floats: List[float] = field(default_factory=list)
    ints: List[int] = field(default_factory=list)
    type: int = 0


@dataclass
class Node:
    op_type: str = ""
    name: str = ""
    domain: str = ""
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    attrs: Dict[str, Attribute] = field(default_factory=dict)

    def attr_i(self, name, default=None):
        a = self.attrs.get(name)
        return default if a is None or a.i is None else a.i

    def attr_f(self, name, default=None):
        a = self.attrs.get(name)
        return default if a is None or a.f is None else a.f

    def attr_ints(self, name, default=None):
        a = self.attrs.get(name)
        return default if a is None else list(a.ints)

    def attr_s(self, name, default=None):
        a = self.attrs.get(name)
        return default if a is None or a.s is None else a.s.decode()


@dataclass
class Model:
    ir_version: int = 0
    producer: str = ""
    opsets: Dict[str, int] = field(default_factory=dict)
    nodes: List[Node] = field(default_factory=list)
    initializers: Dict[str, Tensor] = field(default_factory=dict)
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    base_dir: str = "."

    def weight(self, name: str) -> np.ndarray:
        return self.initializers[name].numpy(self.base_dir)


def _parse_tensor(buf: memoryview) -> Tensor:
    t = Tensor()
    for fn, wt, v in _fields(buf):
        if fn == 1:
            t.dims.extend(_packed_varints(v, wt))
        elif fn == 2:
            t.data_type = v
        elif fn == 4:
            if wt == 5:
                t.float_data.append(struct.unpack("<f", v)[0])
            else:
                t.float_data.extend(np.frombuffer(bytes(v), dtype="<f4").tolist())
        elif fn == 5:
            t.int32_data.extend(_packed_varints(v, wt))
        elif fn == 7:
            t.int64_data.extend(_packed_varints(v, wt))
        elif fn == 8:
            t.name = bytes(v).decode()
        elif fn == 9:
            t.raw = bytes(v)
        elif fn == 13:
            k = val = ""
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    k = bytes(v2).decode()
                elif f2 == 2:
                    val = bytes(v2).decode()
            t.external[k] = val
        elif fn == 14:
            t.data_location = v
    return t


def _parse_attr(buf: memoryview) -> Attribute:
    a = Attribute()
    for fn, wt, v in _fields(buf):
        if fn == 1:
            a.name = bytes(v).decode()
        elif fn == 2:
            a.f = struct.unpack("<f", v)[0]
        elif fn == 3:
            a.i = _signed64(v)
        elif fn == 4:
            a.s = bytes(v)
        elif fn == 5:
            a.t = _parse_tensor(v)
        elif fn == 7:
            if wt == 5:
                a.floats.append(struct.unpack("<f", v)[0])
            else:
                a.floats.extend(np.frombuffer(bytes(v), dtype="<f4").tolist())
        elif fn == 8:
            a.ints.extend(_packed_varints(v, wt))
        elif fn == 20:
            a.type = v
    return a


def _parse_node(buf: memoryview) -> Node:
    n = Node()
    for fn, wt, v in _fields(buf):
        if fn == 1:
            n.inputs.append(bytes(v).decode())
        elif fn == 2:
            n.outputs.append(bytes(v).decode())
        elif fn == 3:
            n.name = bytes(v).decode()
        elif fn == 4:
            n.op_type = bytes(v).decode()
        elif fn == 5:
            a = _parse_attr(v)
            n.attrs[a.name] = a
        elif fn == 7:
            n.domain = bytes(v).decode()
    return n


def _value_info_name(buf: memoryview) -> str:
    for fn, wt, v in _fields(buf):
        if fn == 1:
            return bytes(v).decode()
    return ""


def load(path: str) -> Model:
    with open(path, "rb") as f:
        data = memoryview(f.read())
    m = Model(base_dir=os.path.dirname(os.path.abspath(path)))
    for fn, wt, v in _fields(data):
        if fn == 1:
            m.ir_version = v
        elif fn == 2:
            m.producer = bytes(v).decode()
        elif fn == 8:
            dom, ver = "", 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = bytes(v2).decode()
                elif f2 == 2:
                    ver = v2
            m.opsets[dom] = ver
        elif fn == 7:
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    m.nodes.append(_parse_node(v2))
                elif f2 == 5:
                    t = _parse_tensor(v2)
                    m.initializers[t.name] = t
                elif f2 == 11:
                    m.inputs.append(_value_info_name(v2))
                elif f2 == 12:
                    m.outputs.append(_value_info_name(v2))
    return m

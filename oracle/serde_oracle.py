"""CPU restatement of the reference's V1 container — TEST INFRASTRUCTURE ONLY (never imported by the product).

Follows mesh_to_sdf/src/serde.rs:75-116 (what is serialized: SerializeVersion::V1(SerializeSdf::{Generic,
Grid}), structs SerializeGeneric {query_points, distances} / SerializeGrid {grid, distances}, Grid
{first_cell, cell_size, cell_count} grid.rs:30-37) and serde.rs:161-166 (`rmp_serde::to_vec`: compact
form — externally tagged enums as one-entry maps, structs as arrays, f32 as `ca`, integers shortest).
rmp-serde 1.x is a Cargo dependency that is not in the reference tree; its wire format is MessagePack,
restated here twice, independently:
  * `pack_with_msgpack`: python-msgpack (an independent MessagePack implementation) on the nested object,
  * `pack_with_numpy`:   explicit byte layout, vectorised (usable at 512^3, preserves NaN payloads).
PINNED byte for byte on the reference's golden files tests/sdf_grid_v1.bin and tests/sdf_generic_v1.bin
(serde.rs:314-374; copies under tests/golden/) by tests/test_serde_cpu.py.
"""
import struct

import numpy as np


def pack_with_msgpack(kind, *, grid=None, query_points=None, distances=()):
    import msgpack

    d = [float(x) for x in np.asarray(distances, np.float32)]
    if kind == "Grid":
        first, size, count = grid
        body = [[[float(np.float32(v)) for v in first], [float(np.float32(v)) for v in size], [int(c) for c in count]], d]
    else:
        body = [[[float(v) for v in p] for p in np.asarray(query_points, np.float32).reshape(-1, 3)], d]
    return msgpack.packb({"V1": {kind: body}}, use_single_float=True)


def _uint(v):
    v = int(v)
    if v < 128:
        return bytes([v])
    if v <= 0xFF:
        return b"\xcc" + struct.pack(">B", v)
    if v <= 0xFFFF:
        return b"\xcd" + struct.pack(">H", v)
    if v <= 0xFFFFFFFF:
        return b"\xce" + struct.pack(">I", v)
    return b"\xcf" + struct.pack(">Q", v)


def _array(n):
    if n < 16:
        return bytes([0x90 | n])
    if n <= 0xFFFF:
        return b"\xdc" + struct.pack(">H", n)
    if n <= 0xFFFFFFFF:
        return b"\xdd" + struct.pack(">I", n)
    raise ValueError("SerializationFailed")


def _f32(v):
    return b"\xca" + np.asarray(v, np.float32).astype(">f4").tobytes()


def _f32_records(a):
    a = np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))
    rec = np.empty((a.size, 5), np.uint8)
    rec[:, 0] = 0xCA
    rec[:, 1:] = a.view(np.uint32).astype(">u4").view(np.uint8).reshape(-1, 4)
    return rec.tobytes()


def _point_records(q):
    q = np.ascontiguousarray(np.asarray(q, np.float32).reshape(-1, 3))
    rec = np.empty((q.shape[0], 16), np.uint8)
    rec[:, 0] = 0x93
    be = q.view(np.uint32).astype(">u4").view(np.uint8).reshape(-1, 3, 4)
    for c in range(3):
        rec[:, 1 + 5 * c] = 0xCA
        rec[:, 2 + 5 * c : 6 + 5 * c] = be[:, c, :]
    return rec.tobytes()


def _str(s):
    b = s.encode()
    assert len(b) < 32
    return bytes([0xA0 | len(b)]) + b


def pack_with_numpy(kind, *, grid=None, query_points=None, distances=()):
    d = np.asarray(distances, np.float32).reshape(-1)
    out = [b"\x81", _str("V1"), b"\x81", _str(kind), _array(2)]
    if kind == "Grid":
        first, size, count = grid
        out += [_array(3), _array(3)] + [_f32(v) for v in first] + [_array(3)] + [_f32(v) for v in size]
        out += [_array(3)] + [_uint(c) for c in count]
    else:
        q = np.asarray(query_points, np.float32).reshape(-1, 3)
        out += [_array(q.shape[0]), _point_records(q)]
    out += [_array(d.size), _f32_records(d)]
    return b"".join(out)


def unpack(data):
    """-> ("Grid", (first, size, count), distances) | ("Generic", query_points, distances), via python-msgpack."""
    import msgpack

    top = msgpack.unpackb(data, raw=False, strict_map_key=False)
    (ver, inner), = top.items()
    assert ver == "V1"
    (kind, body), = inner.items()
    if kind == "Grid":
        (first, size, count), dist = body
        return kind, (np.array(first, np.float32), np.array(size, np.float32), [int(c) for c in count]), np.array(dist, np.float32)
    q, dist = body
    return kind, np.array(q, np.float32).reshape(-1, 3), np.array(dist, np.float32)

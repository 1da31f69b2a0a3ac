"""Builds small glTF / GLB files for the ingestion tests (hierarchies, TRS / matrix nodes, shared meshes,
u8/u16/u32 indices, byteStride, sparse accessors, missing indices, data: URIs, required extensions)."""
import base64
import json
import struct

import numpy as np


class Builder:
    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "buffers": [{}], "bufferViews": [], "accessors": [], "meshes": [], "nodes": [],
                    "scenes": [], "scene": 0}

    def view(self, data: bytes, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin += data
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, arr, kind, stride_pad=0, byte_offset=0):
        arr = np.ascontiguousarray(arr)
        ct = {np.dtype(np.uint8): 5121, np.dtype(np.uint16): 5123, np.dtype(np.uint32): 5125, np.dtype(np.float32): 5126}[arr.dtype]
        rows = arr.reshape(arr.shape[0], -1)
        elem = rows.shape[1] * arr.dtype.itemsize
        if stride_pad:
            stride = elem + stride_pad
            raw = b"".join(r.tobytes() + b"\xAB" * stride_pad for r in rows)
            bv = self.view(b"\xCD" * byte_offset + raw, stride)
        else:
            bv = self.view(b"\xCD" * byte_offset + rows.tobytes())
        a = {"bufferView": bv, "componentType": ct, "count": int(arr.shape[0]), "type": kind}
        if byte_offset:
            a["byteOffset"] = byte_offset
        self.doc["accessors"].append(a)
        return len(self.doc["accessors"]) - 1

    def sparse_positions(self, base, idx, vals, with_base=True):
        a = {"componentType": 5126, "count": int(base.shape[0]), "type": "VEC3",
             "sparse": {"count": int(len(idx)),
                        "indices": {"bufferView": self.view(np.asarray(idx, np.uint16).tobytes()), "componentType": 5123},
                        "values": {"bufferView": self.view(np.asarray(vals, np.float32).tobytes())}}}
        if with_base:
            a["bufferView"] = self.view(np.asarray(base, np.float32).tobytes())
        self.doc["accessors"].append(a)
        return len(self.doc["accessors"]) - 1

    def mesh(self, prims):
        """prims: list of (position accessor, index accessor or None)"""
        ps = []
        for pa, ia in prims:
            p = {"attributes": {"POSITION": pa}}
            if ia is not None:
                p["indices"] = ia
            ps.append(p)
        self.doc["meshes"].append({"primitives": ps})
        return len(self.doc["meshes"]) - 1

    def node(self, mesh=None, children=(), matrix=None, t=None, r=None, s=None):
        n = {}
        if mesh is not None:
            n["mesh"] = mesh
        if children:
            n["children"] = list(children)
        if matrix is not None:
            n["matrix"] = [float(x) for x in matrix]
        if t is not None:
            n["translation"] = [float(x) for x in t]
        if r is not None:
            n["rotation"] = [float(x) for x in r]
        if s is not None:
            n["scale"] = [float(x) for x in s]
        self.doc["nodes"].append(n)
        return len(self.doc["nodes"]) - 1

    def scene(self, roots):
        self.doc["scenes"].append({"nodes": list(roots)})

    def glb(self):
        doc = dict(self.doc)
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        bn = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
        total = 12 + 8 + len(js) + 8 + len(bn)
        return struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bn), 0x004E4942) + bn

    def gltf_embedded(self):
        doc = dict(self.doc)
        doc["buffers"] = [{"byteLength": len(self.bin), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()}]
        return json.dumps(doc, indent=1).encode()


def quat(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    return [*(axis * np.sin(angle / 2)), float(np.cos(angle / 2))]


def zoo(seed=0):
    """A scene graph exercising every code path; returns the Builder."""
    rng = np.random.default_rng(seed)
    b = Builder()
    tri = rng.standard_normal((30, 3)).astype(np.float32)
    quad = rng.standard_normal((8, 3)).astype(np.float32)
    m0 = b.mesh([(b.accessor(tri, "VEC3"), b.accessor(rng.integers(0, 30, 60).astype(np.uint16), "SCALAR"))])
    m1 = b.mesh([(b.accessor(quad, "VEC3", stride_pad=8, byte_offset=4), b.accessor(rng.integers(0, 8, 12).astype(np.uint8), "SCALAR"))])
    m2 = b.mesh([(b.accessor(tri[:9], "VEC3"), None)])                                   # no indices
    m3 = b.mesh([(b.sparse_positions(quad, [1, 5], rng.standard_normal((2, 3))), b.accessor(rng.integers(0, 8, 9).astype(np.uint32), "SCALAR"))])
    m4 = b.mesh([(b.accessor(tri[:6], "VEC3"), b.accessor(np.arange(6, dtype=np.uint16), "SCALAR")),   # two primitives: the last stays
                 (b.accessor(quad[:4], "VEC3"), b.accessor(np.array([0, 1, 2, 2, 1, 3], np.uint16), "SCALAR"))])
    m5 = b.mesh([(b.sparse_positions(np.zeros((5, 3), np.float32), [0, 4], rng.standard_normal((2, 3)), with_base=False), None)])
    leaf_a = b.node(mesh=m0, t=[1, 2, 3], r=quat([1, 2, 3], 0.7), s=[1.5, 0.5, 2])
    leaf_b = b.node(mesh=m1, matrix=np.array([[0, 0, -0.5, 0], [0, 0.5, 0, 0], [0.5, 0, -0.0, 0], [0.5, 0.5, 0.5, 1]], np.float32).reshape(-1))
    leaf_c = b.node(mesh=m0, r=quat([0, 1, 0], -2.1))                                    # m0 instanced twice
    chain3 = b.node(mesh=m2, s=[0.1, 0.2, 0.3])
    chain2 = b.node(children=[chain3], r=quat([1, 0, 0], 1.0))                            # transform-only chain (simplify_tree)
    chain1 = b.node(children=[chain2], t=[-4, 0, 9])
    group = b.node(children=[leaf_a, leaf_b], t=[10, 0, 0], s=[2, 2, 2])
    holder = b.node(mesh=m3, children=[leaf_c, group], r=quat([0, 0, 1], 0.3))            # a mesh node with children
    empty = b.node()                                                                     # no mesh, no children
    b.scene([holder, chain1, empty])
    b.scene([b.node(mesh=m4, t=[0, 0, 1]), b.node(mesh=m5)])                               # a second scene
    return b

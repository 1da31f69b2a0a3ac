"""CPU restatement of the reference client's glTF ingestion — TEST INFRASTRUCTURE ONLY.

Follows mesh_to_sdf_client/src/gltf/mod.rs:56-174 (load_scene / load / flatten_hierarchy),
gltf/scene/mod.rs:56-160 (ModelNode, read_node, simplify_tree), gltf/scene/model/mod.rs:247-262
(positions + indices), pbr/model.rs:29-32 (missing indices) with python's json module and numpy.
Third-party pieces restated from their published algorithms (crates absent from the tree):
gltf 1.4.1 `Transform::matrix()` (T*R*S, cgmath-style quaternion matrix) and glam 0.29 `Mat4 * Mat4`.
PINNING: the reference's loader tests only assert counts and load success/failure
(gltf/mod.rs:200-411: cube.glb 1 scene / 1 model; suzanne.glb 1 scene, 1 model; dragon.glb must fail) —
reproduced in tests/test_gltf_cpu.py.  Transform values are "parity unpinned" (no reference vectors).
"""
import base64
import json
import os
import struct

import numpy as np

F = np.float32
_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}
SUPPORTED_REQUIRED = {"KHR_lights_punctual"}


class GltfError(Exception):
    pass


def mat_mul(a, b):
    """glam Mat4 * Mat4, column-major (4,4) arrays m[col][row], f32, no FMA."""
    r = np.zeros((4, 4), F)
    for j in range(4):
        acc = a[0] * b[j][0]
        acc = (acc + a[1] * b[j][1]).astype(F)
        acc = (acc + a[2] * b[j][2]).astype(F)
        acc = (acc + a[3] * b[j][3]).astype(F)
        r[j] = acc
    return r


def trs_matrix(t, q, s):
    t, q, s = np.asarray(t, F), np.asarray(q, F), np.asarray(s, F)
    qx, qy, qz, qs = q
    x2, y2, z2 = F(qx + qx), F(qy + qy), F(qz + qz)
    xx2, xy2, xz2 = F(x2 * qx), F(x2 * qy), F(x2 * qz)
    yy2, yz2, zz2 = F(y2 * qy), F(y2 * qz), F(z2 * qz)
    sy2, sz2, sx2 = F(y2 * qs), F(z2 * qs), F(x2 * qs)
    one = F(1)
    R = np.eye(4, dtype=F)
    R[0][:3] = [F(F(one - yy2) - zz2), F(xy2 + sz2), F(xz2 - sy2)]
    R[1][:3] = [F(xy2 - sz2), F(F(one - xx2) - zz2), F(yz2 + sx2)]
    R[2][:3] = [F(xz2 + sy2), F(yz2 - sx2), F(F(one - xx2) - yy2)]
    T = np.eye(4, dtype=F)
    T[3][:3] = t
    S = np.eye(4, dtype=F)
    S[0][0], S[1][1], S[2][2] = s
    return mat_mul(mat_mul(T, R), S)


def _load_doc(path):
    data = open(path, "rb").read()
    blob = None
    if data[:4] == b"glTF":
        _, version, length = struct.unpack_from("<III", data, 0)
        if version != 2:
            raise GltfError("version")
        off, doc = 12, None
        while off + 8 <= length:
            clen, ctype = struct.unpack_from("<II", data, off)
            chunk = data[off + 8 : off + 8 + clen]
            if ctype == 0x4E4F534A and doc is None:
                doc = json.loads(chunk.decode("utf-8"))
            elif ctype == 0x004E4942 and blob is None:
                blob = chunk
            off += 8 + clen
        if doc is None:
            raise GltfError("no JSON chunk")
    else:
        try:
            doc = json.loads(data.decode("utf-8"))
        except Exception as e:  # noqa: BLE001
            raise GltfError(str(e))
    for x in doc.get("extensionsRequired", []):
        if x not in SUPPORTED_REQUIRED:
            raise GltfError("unsupported required extension " + x)
    bufs = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if i != 0 or blob is None:
                raise GltfError("buffer without data")
            bufs.append(blob)
        elif uri.startswith("data:"):
            bufs.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            bufs.append(open(os.path.join(os.path.dirname(path) or ".", uri), "rb").read())
    return doc, bufs


def _view(doc, bufs, bv, extra, dt, ncomp, count):
    v = doc["bufferViews"][bv]
    start = v.get("byteOffset", 0) + extra
    elem = np.dtype(dt).itemsize * ncomp
    stride = v.get("byteStride", 0) or elem
    buf = bufs[v["buffer"]]
    out = np.empty((count, ncomp), dt)
    for i in range(count):
        out[i] = np.frombuffer(buf, dt, ncomp, start + i * stride)
    return out


def read_accessor(doc, bufs, idx):
    acc = doc["accessors"][idx]
    dt, ncomp, count = _COMP[acc["componentType"]], _NCOMP[acc["type"]], acc["count"]
    if "bufferView" in acc:
        out = _view(doc, bufs, acc["bufferView"], acc.get("byteOffset", 0), dt, ncomp, count)
    else:
        out = np.zeros((count, ncomp), dt)
    sp = acc.get("sparse")
    if sp:
        n = sp["count"]
        ii = _view(doc, bufs, sp["indices"]["bufferView"], sp["indices"].get("byteOffset", 0), _COMP[sp["indices"]["componentType"]], 1, n)
        vv = _view(doc, bufs, sp["values"]["bufferView"], sp["values"].get("byteOffset", 0), dt, ncomp, n)
        out[ii[:, 0].astype(np.int64)] = vv
    return out


def load(path):
    """-> (n_scenes, models {mesh_index: (positions (N,3) f32, indices u32)}, instances [(mesh_index, mat4 (4,4) col-major)])"""
    doc, bufs = _load_doc(path)
    models = {}
    for mi, mesh in enumerate(doc.get("meshes", [])):
        for prim in mesh["primitives"]:            # the last primitive stays (sequential insertion order)
            if "POSITION" not in prim.get("attributes", {}):
                raise GltfError("The model primitive doesn't contain positions")
            pos = read_accessor(doc, bufs, prim["attributes"]["POSITION"]).astype(F)
            if "indices" in prim:
                ind = read_accessor(doc, bufs, prim["indices"])[:, 0].astype(np.uint32)
            else:
                ind = np.arange(pos.shape[0], dtype=np.uint32)
            models[mi] = (pos, ind)

    def node_matrix(n):
        if "matrix" in n:
            return np.array([np.float32(np.float64(x)) for x in n["matrix"]], F).reshape(4, 4)
        return trs_matrix(n.get("translation", [0, 0, 0]), n.get("rotation", [0, 0, 0, 1]), n.get("scale", [1, 1, 1]))

    def read_node(idx):
        n = doc["nodes"][idx]
        return {"model": n.get("mesh"), "m": node_matrix(n), "children": [read_node(c) for c in n.get("children", [])]}

    def simplify(node):
        for c in node["children"]:
            simplify(c)
        if node["model"] is None and len(node["children"]) == 1:
            child = node["children"][0]
            child["m"] = mat_mul(node["m"], child["m"])
            node.clear()
            node.update(child)

    def flatten(node, parent, out):
        t = mat_mul(parent, node["m"])
        for c in node["children"]:
            flatten(c, t, out)
        if node["model"] is not None:
            out.append((node["model"], t))

    instances = []
    for sc in doc.get("scenes", []):
        root = {"model": None, "m": np.eye(4, dtype=F), "children": []}
        for ni in sc.get("nodes", []):
            root["children"].append({"model": None, "m": np.eye(4, dtype=F), "children": [read_node(ni)]})
        simplify(root)
        flatten(root, np.eye(4, dtype=F), instances)
    return len(doc.get("scenes", [])), models, instances

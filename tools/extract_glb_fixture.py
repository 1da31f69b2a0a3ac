#!/usr/bin/env python3
"""Extract POSITION + indices of one glTF-binary mesh primitive into a small .npz fixture.

Used once, in the build container, to turn the reference's bundled test asset
`/root/reference/mesh_to_sdf/assets/suzanne.glb` (Suzanne, Blender Foundation — credit per
the reference's assets/README.md) into `tests/golden/suzanne.npz` (vertices f32 [V,3],
indices u32 [I]).  The fixture is DATA (a mesh the reference's own tests load,
generic/default.rs:85, generic/bvh.rs:156, ...), not reference source.  /root/reference does
not exist on the GPU box, so tests only read the committed .npz.

usage: tools/extract_glb_fixture.py <in.glb> <out.npz> [mesh_index] [primitive_index]
"""
import json
import struct
import sys

import numpy as np

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def load_glb(path):
    data = open(path, "rb").read()
    magic, version, length = struct.unpack_from("<III", data, 0)
    assert magic == 0x46546C67 and version == 2, "not a glTF 2.0 binary"
    off = 12
    gltf, blob = None, None
    while off < length:
        clen, ctype = struct.unpack_from("<II", data, off)
        chunk = data[off + 8 : off + 8 + clen]
        if ctype == 0x4E4F534A:
            gltf = json.loads(chunk.decode("utf-8"))
        elif ctype == 0x004E4942:
            blob = chunk
        off += 8 + clen
    return gltf, blob


def read_accessor(gltf, blob, idx):
    acc = gltf["accessors"][idx]
    view = gltf["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMPONENT[acc["componentType"]])
    ncomp = _NCOMP[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or dt.itemsize * ncomp
    count = acc["count"]
    if stride == dt.itemsize * ncomp:
        arr = np.frombuffer(blob, dtype=dt, count=count * ncomp, offset=start).reshape(count, ncomp)
    else:
        arr = np.stack(
            [np.frombuffer(blob, dtype=dt, count=ncomp, offset=start + i * stride) for i in range(count)]
        )
    return arr.copy()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    mi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    pi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    gltf, blob = load_glb(src)
    prim = gltf["meshes"][mi]["primitives"][pi]
    assert prim.get("mode", 4) == 4, "expected TRIANGLES"
    pos = read_accessor(gltf, blob, prim["attributes"]["POSITION"]).astype(np.float32)
    idx = read_accessor(gltf, blob, prim["indices"]).reshape(-1).astype(np.uint32)
    # sanity: node transforms (if any) are reported, not applied — suzanne has none.
    xf = [n for n in gltf.get("nodes", []) if n.get("mesh") == mi and any(k in n for k in ("matrix", "rotation", "translation", "scale"))]
    print(f"{src}: mesh {mi} '{gltf['meshes'][mi].get('name')}' prim {pi}: {pos.shape[0]} verts, {idx.size // 3} tris, "
          f"index dtype {gltf['accessors'][prim['indices']]['componentType']}, node transforms: {xf}")
    np.savez_compressed(dst, vertices=pos, indices=idx)


if __name__ == "__main__":
    main()

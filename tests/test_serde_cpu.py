"""V1 container (serde.rs:75-221) without a GPU: the oracle against the reference's golden files, and the
host half of the C ABI (sizes, envelope parsing) against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from mesh_to_sdf_amd import _lib
from mesh_to_sdf_amd._lib import M2SGrid, M2SSdfInfo
from oracle import serde_oracle as so

GOLD = os.path.join(os.path.dirname(__file__), "golden")
GRID = ([1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7, 8, 9])          # serde.rs:353
GRID_D = np.arange(7 * 8 * 9, dtype=np.float32)              # serde.rs:354-356
GEN_Q = np.array([[1, 2, 3], [6, 5, 4]], np.float32)         # serde.rs:316-319
GEN_D = np.array([1.0, 3.0], np.float32)                     # serde.rs:320


def gold(name):
    return open(os.path.join(GOLD, name), "rb").read()


@pytest.mark.parametrize("pack", [so.pack_with_msgpack, so.pack_with_numpy])
def test_oracle_reproduces_the_reference_golden_files(pack):
    assert pack("Grid", grid=GRID, distances=GRID_D) == gold("sdf_grid_v1.bin")
    assert pack("Generic", query_points=GEN_Q, distances=GEN_D) == gold("sdf_generic_v1.bin")


def test_oracle_restatements_agree_on_header_width_boundaries():
    rng = np.random.default_rng(5)
    for n in (0, 1, 15, 16, 17, 65535, 65536, 70001):
        d = rng.standard_normal(n).astype(np.float32)
        for count in ([1, 1, 1], [127, 128, 255], [256, 65535, 65536], [2**32 - 1, 2**32, 2**40]):
            g = ([0.5, -1.5, 2.5], [0.1, 0.2, -0.3], count)
            assert so.pack_with_msgpack("Grid", grid=g, distances=d) == so.pack_with_numpy("Grid", grid=g, distances=d)
    for nq in (0, 1, 15, 16, 300):
        q = rng.standard_normal((nq, 3)).astype(np.float32)
        d = rng.standard_normal(nq + 3).astype(np.float32)
        assert so.pack_with_msgpack("Generic", query_points=q, distances=d) == so.pack_with_numpy("Generic", query_points=q, distances=d)


def mk_grid(first, size, count):
    g = M2SGrid()
    for k in range(3):
        g.first_cell[k], g.cell_size[k], g.cell_count[k] = first[k], size[k], count[k]
    return g


def test_encoded_size_matches_the_golden_files_and_the_oracle():
    L = _lib.lib()
    assert L.m2s_sdf_grid_encoded_size(C.byref(mk_grid(*GRID)), GRID_D.size) == len(gold("sdf_grid_v1.bin")) == 2571
    assert L.m2s_sdf_generic_encoded_size(2, 2) == len(gold("sdf_generic_v1.bin")) == 58
    for n in (0, 15, 16, 65535, 65536):
        for count in ([1, 1, 1], [127, 128, 255], [256, 65535, 65536], [2**32 - 1, 2**32, 2**40]):
            g = ([0.5, -1.5, 2.5], [0.1, 0.2, -0.3], count)
            want = len(so.pack_with_numpy("Grid", grid=g, distances=np.zeros(n, np.float32)))
            assert L.m2s_sdf_grid_encoded_size(C.byref(mk_grid(*g)), n) == want
        assert L.m2s_sdf_generic_encoded_size(n, n + 1) == len(
            so.pack_with_numpy("Generic", query_points=np.zeros((n, 3), np.float32), distances=np.zeros(n + 1, np.float32)))
    # MessagePack arrays stop at 2^32-1 elements: rmp-serde fails there (SerializationFailed)
    assert L.m2s_sdf_grid_encoded_size(C.byref(mk_grid(*GRID)), 2**32) == 0
    assert L.m2s_sdf_generic_encoded_size(2**32, 1) == 0
    assert L.m2s_sdf_grid_encoded_size(C.byref(mk_grid(*GRID)), 2**32 - 1) == 53 + 5 * (2**32 - 1)   # array32 header


def probe(data):
    info = M2SSdfInfo()
    buf = np.frombuffer(data, np.uint8)
    rc = _lib.lib().m2s_sdf_probe(buf.ctypes.data, buf.size, C.byref(info), None)
    return rc, info


def test_probe_reads_the_golden_envelopes():
    rc, info = probe(gold("sdf_grid_v1.bin"))
    assert rc == 0 and info.kind == 1 and info.canonical == 1
    assert list(info.grid.first_cell) == [1, 2, 3] and list(info.grid.cell_size) == [4, 5, 6]
    assert list(info.grid.cell_count) == [7, 8, 9]
    assert info.n_distances == 504 and info.distances_offset == 2571 - 5 * 504 and info.n_queries == 0
    rc, info = probe(gold("sdf_generic_v1.bin"))
    assert rc == 0 and info.kind == 0 and info.canonical == 1
    assert info.n_queries == 2 and info.n_distances == 2
    assert info.queries_offset == 15 and info.distances_offset == 15 + 32 + 1
    info2 = M2SSdfInfo()
    assert _lib.lib().m2s_sdf_probe_file(os.path.join(GOLD, "sdf_grid_v1.bin").encode(), C.byref(info2)) == 0
    assert info2.n_distances == 504 and list(info2.grid.cell_count) == [7, 8, 9]


def test_probe_accepts_other_number_forms_and_flags_them():
    import msgpack

    # f64 payload (what python-msgpack writes by default): serde's f32 visitor accepts it
    data = msgpack.packb({"V1": {"Generic": [[[1.0, 2.0, 3.0]], [0.25, 7]]}})
    rc, info = probe(data)
    assert rc == 0 and info.kind == 0 and info.canonical == 0 and info.n_queries == 1 and info.n_distances == 2
    data = msgpack.packb({"V1": {"Grid": [[[1.0, 2.0, 3.0], [4, 5, 6], [7, 8, 9]], [1.5] * 20]}})
    rc, info = probe(data)
    assert rc == 0 and info.kind == 1 and info.canonical == 0 and info.n_distances == 20
    assert list(info.grid.cell_size) == [4, 5, 6]


@pytest.mark.parametrize("data", [b"\xc1", b"\x81", b"\x81\xa2V2\x81\xa4Grid\x92", b"\x81\xa2V1\x81\xa5Grids\x92",
                                  b"\x81\xa2V1\x81\xa4Grid\x93", gold("sdf_grid_v1.bin")[:30],
                                  gold("sdf_generic_v1.bin")[:14], b"not msgpack at all"])
def test_probe_rejects_what_rmp_serde_rejects(data):
    rc, _ = probe(data)
    assert rc == _lib.ERR_BAD_ARG
    assert "DeserializationFailed" in _lib.last_error()


def test_probe_file_io_error():
    info = M2SSdfInfo()
    assert _lib.lib().m2s_sdf_probe_file(b"/nonexistent/dir/sdf.bin", C.byref(info)) == _lib.ERR_IO
    assert "IoError" in _lib.last_error()


def test_truncated_payload_is_not_canonical():
    rc, info = probe(gold("sdf_grid_v1.bin")[:-1])
    assert rc == 0 and info.canonical == 0      # decode will walk it and fail like rmp-serde (unexpected EOF)

"""V1 container (serde.rs:75-221) on the GPU: byte-exact against the reference's golden files and the oracle
(integer/byte work: the bar is bit-exact), through the C ABI.  Mirrors the reference's own serde tests
(serde.rs:228-374).  Needs a real MI355X: run with `-m gpu`."""
import ctypes as C
import os

import numpy as np
import pytest

from mesh_to_sdf_amd import Grid, _lib
from mesh_to_sdf_amd.serde import (DeserializeGeneric, DeserializeGrid, SerdeError, SerializeGeneric, SerializeGrid,
                                   SerializeSdf, deserialize, read_from_file, save_to_file, serialize)
from oracle import serde_oracle as so

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    return open(os.path.join(GOLD, name), "rb").read()


def nasty_floats(n, seed):
    """Random bit patterns: NaN payloads, infinities, denormals, -0 all survive a byte-exact container."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    a[: min(n, 6)] = np.array([0x00000000, 0x80000000, 0x7F800000, 0xFF800001, 0x00000001, 0x7FC00123], np.uint32)[: min(n, 6)]
    return a.view(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_serde_generic_roundtrip_like_the_reference():   # serde.rs:228-252
    queries = np.array([[1, 2, 3], [6, 5, 4]], np.float32)
    distances = np.array([1.0, 3.0], np.float32)
    data = serialize(SerializeSdf.Generic(SerializeGeneric(queries, distances)))
    assert data == gold("sdf_generic_v1.bin")
    de = deserialize(data)
    assert isinstance(de, DeserializeGeneric)
    assert np.array_equal(de.query_points, queries) and np.array_equal(de.distances, distances)


def test_serde_grid_roundtrip_like_the_reference():      # serde.rs:255-279
    grid = Grid.new([1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7, 8, 9])
    distances = np.arange(grid.get_total_cell_count(), dtype=np.float32)
    data = serialize(SerializeSdf.Grid(SerializeGrid(grid, distances)))
    assert data == gold("sdf_grid_v1.bin")
    de = deserialize(data)
    assert isinstance(de, DeserializeGrid)
    assert de.grid == grid and np.array_equal(de.distances, distances)


def test_backward_compatibility_golden_files(tmp_path):  # serde.rs:314-374
    de = read_from_file(os.path.join(GOLD, "sdf_generic_v1.bin"))
    assert np.array_equal(de.query_points, [[1, 2, 3], [6, 5, 4]]) and np.array_equal(de.distances, [1.0, 3.0])
    de = read_from_file(os.path.join(GOLD, "sdf_grid_v1.bin"))
    assert de.grid == Grid.new([1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7, 8, 9])
    assert np.array_equal(de.distances, np.arange(504, dtype=np.float32))


def test_serde_file(tmp_path):                            # serde.rs:282-311
    import torch

    path = tmp_path / "sdf.bin"
    queries = nasty_floats(3 * 1000, 1).reshape(-1, 3)
    distances = nasty_floats(1003, 2)
    save_to_file(SerializeSdf.Generic(SerializeGeneric(queries, distances)), path)
    assert path.read_bytes() == so.pack_with_numpy("Generic", query_points=queries, distances=distances)
    de = read_from_file(path)
    assert np.array_equal(bits(de.query_points), bits(queries)) and np.array_equal(bits(de.distances), bits(distances))
    # device-resident result straight to a file, and back onto the device
    grid = Grid.new([0.5, -1, 2], [0.1, 0.2, -0.3], [31, 17, 5])
    d = nasty_floats(31 * 17 * 5, 3)
    save_to_file(SerializeGrid(grid, torch.from_numpy(d.copy()).cuda()), path)
    assert path.read_bytes() == so.pack_with_numpy("Grid", grid=([0.5, -1, 2], [0.1, 0.2, -0.3], [31, 17, 5]), distances=d)
    de = read_from_file(path, device="cuda")
    assert de.distances.is_cuda and de.grid == grid and np.array_equal(bits(de.distances.cpu().numpy()), bits(d))
    with pytest.raises(SerdeError) as e:
        save_to_file(SerializeGrid(grid, d), "/nonexistent/dir/x.bin")
    assert e.value.code == _lib.ERR_IO


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 255, 4099, 65535, 65536, 1_000_003])
def test_grid_payload_byte_exact_every_header_width(n):
    g = ([0.5, -1.5, 2.5], [0.1, 0.2, -0.3], [n, 1, 1])
    d = nasty_floats(n, n + 7)
    data = serialize(SerializeGrid(Grid.new(*g), d))
    assert data == so.pack_with_numpy("Grid", grid=g, distances=d)
    de = deserialize(data)
    assert np.array_equal(bits(de.distances), bits(d))
    if n <= 4099 and not np.isnan(d).any():   # python floats cannot carry NaN payloads through msgpack
        assert data == so.pack_with_msgpack("Grid", grid=g, distances=d)


@pytest.mark.parametrize("nq,nd", [(0, 0), (1, 1), (2, 5), (15, 15), (16, 16), (17, 300), (1000, 999), (65536, 65536), (300_001, 7)])
def test_generic_payload_byte_exact(nq, nd):
    q = nasty_floats(3 * nq, nq + 11).reshape(-1, 3)
    d = nasty_floats(nd, nd + 13)
    data = serialize(SerializeGeneric(q, d))
    assert data == so.pack_with_numpy("Generic", query_points=q, distances=d)
    de = deserialize(data)
    assert np.array_equal(bits(de.query_points), bits(q)) and np.array_equal(bits(de.distances), bits(d))


def test_every_output_alignment_device_side():
    """The payload may start at any byte offset of the caller's buffer: 16 alignments x several sizes, straight
    through the C ABI with device pointers; guard bytes on either side must stay untouched."""
    import torch

    from mesh_to_sdf_amd.serde import _opts

    L = _lib.lib()
    for n in (1, 2, 3, 7, 33, 1000):
        q = nasty_floats(3 * n, n).reshape(-1, 3)
        d = nasty_floats(n + 1, n + 1)
        want = np.frombuffer(so.pack_with_numpy("Generic", query_points=q, distances=d), np.uint8)
        tq, td = torch.from_numpy(q.copy()).cuda(), torch.from_numpy(d.copy()).cuda()
        for shift in range(16):
            buf = torch.full((want.size + 64,), 0xEE, dtype=torch.uint8, device="cuda")
            o = _opts(buf)
            w = C.c_size_t(0)
            rc = L.m2s_sdf_encode_generic(tq.data_ptr(), n, td.data_ptr(), n + 1, buf.data_ptr() + 16 + shift, want.size,
                                          C.byref(w), C.byref(o))
            assert rc == 0 and w.value == want.size
            got = buf.cpu().numpy()
            assert np.array_equal(got[16 + shift : 16 + shift + want.size], want), (n, shift)
            assert (got[: 16 + shift] == 0xEE).all() and (got[16 + shift + want.size :] == 0xEE).all(), (n, shift)
            # decode from the same misaligned device bytes
            oq = torch.empty((n, 3), dtype=torch.float32, device="cuda")
            od = torch.empty(n + 1, dtype=torch.float32, device="cuda")
            rc = L.m2s_sdf_decode(buf.data_ptr() + 16 + shift, want.size, oq.data_ptr(), od.data_ptr(), C.byref(o))
            assert rc == 0, _lib.last_error()
            assert np.array_equal(bits(oq.cpu().numpy()), bits(q)) and np.array_equal(bits(od.cpu().numpy()), bits(d))


def test_device_tensors_stay_on_the_device():
    import torch

    grid = Grid.new([0, 0, 0], [1, 1, 1], [64, 64, 64])
    d = torch.from_numpy(nasty_floats(64**3, 99).copy()).cuda()
    data = serialize(SerializeGrid(grid, d))
    assert data.is_cuda and data.dtype == torch.uint8
    assert data.cpu().numpy().tobytes() == so.pack_with_numpy("Grid", grid=([0, 0, 0], [1, 1, 1], [64, 64, 64]), distances=d.cpu().numpy())
    de = deserialize(data)
    assert de.distances.is_cuda and torch.equal(de.distances.view(torch.int32), d.view(torch.int32)) and de.grid == grid


def test_other_number_forms_decode_like_serde():
    """serde's f32 visitor accepts f64 and integers; python-msgpack writes those by default."""
    import msgpack

    data = msgpack.packb({"V1": {"Generic": [[[1.0, 2.5, -3], [4, 5, 6.125]], [0.1, 7, -2, 1e40]]}})
    de = deserialize(data)
    assert np.array_equal(de.query_points, np.array([[1, 2.5, -3], [4, 5, 6.125]], np.float32))
    with np.errstate(over="ignore"):
        assert np.array_equal(de.distances, np.array([0.1, 7, -2, 1e40], np.float64).astype(np.float32))
    data = msgpack.packb({"V1": {"Grid": [[[1.0, 2.0, 3.0], [4, 5, 6], [2, 3, 4]], list(range(24))]}})
    de = deserialize(data)
    assert de.grid == Grid.new([1, 2, 3], [4, 5, 6], [2, 3, 4]) and np.array_equal(de.distances, np.arange(24, dtype=np.float32))
    # same total length as the fixed-width layout but a foreign tag inside: the kernels flag it, the scalar reader decides
    good = bytearray(serialize(SerializeGrid(Grid.new([1, 2, 3], [4, 5, 6], [5, 1, 1]), np.arange(5, dtype=np.float32))))
    bad = bytes(good[:-5]) + b"\xc0" + bytes(good[-4:])
    with pytest.raises(SerdeError) as e:
        deserialize(bad)
    assert "DeserializationFailed" in str(e.value)
    with pytest.raises(SerdeError):
        deserialize(bytes(good[:-1]))      # unexpected end of input
    de = deserialize(bytes(good) + b"\x00\x01")   # trailing bytes are not read (from_slice stops after the value)
    assert np.array_equal(de.distances, np.arange(5, dtype=np.float32))


def test_crafted_header_with_huge_counts_is_rejected_not_allocated():
    """A 30-byte container whose array headers promise 2^32 - 1 elements (rmp-serde caps such preallocation): the decoder must
    answer DeserializationFailed — not allocate 48 GB, not let std::bad_alloc cross the C ABI."""
    L = _lib.lib()
    hdr = b"\x81\xa2V1\x81\xa7Generic\x92"
    crafted = hdr + b"\xdd\xff\xff\xff\xff" + b"\x93\x01\x02\x03"          # array32 of 2^32-1 points, one point present (ints: scalar reader)
    q = np.zeros(16, np.float32)
    d = np.zeros(16, np.float32)
    buf = np.frombuffer(crafted, np.uint8).copy()
    assert L.m2s_sdf_decode(buf.ctypes.data, buf.size, q.ctypes.data, d.ctypes.data, None) == _lib.ERR_BAD_ARG
    crafted = hdr + b"\x91\x93\x01\x02\x03" + b"\xdd\xff\xff\xff\xff\x01"   # one point, then 2^32-1 distances, one present
    buf = np.frombuffer(crafted, np.uint8).copy()
    assert L.m2s_sdf_decode(buf.ctypes.data, buf.size, q.ctypes.data, d.ctypes.data, None) == _lib.ERR_BAD_ARG
    with pytest.raises(SerdeError):
        deserialize(crafted)


def test_capacity_and_argument_errors():
    from mesh_to_sdf_amd.serde import _opts

    L = _lib.lib()
    g = Grid.new([1, 2, 3], [4, 5, 6], [7, 8, 9])
    d = np.zeros(504, np.float32)
    out = np.zeros(2571, np.uint8)
    o = _opts(None)
    assert L.m2s_sdf_encode_grid(C.byref(g._g), d.ctypes.data, 504, out.ctypes.data, 2570, None, C.byref(o)) == _lib.ERR_BAD_ARG
    assert "2571" in _lib.last_error()
    assert L.m2s_sdf_encode_grid(C.byref(g._g), None, 504, out.ctypes.data, 2571, None, C.byref(o)) == _lib.ERR_BAD_ARG
    assert L.m2s_sdf_encode_grid(C.byref(g._g), d.ctypes.data, 504, out.ctypes.data, 2571, None, C.byref(o)) == 0
    assert out.tobytes() == so.pack_with_numpy("Grid", grid=([1, 2, 3], [4, 5, 6], [7, 8, 9]), distances=d)


def test_512_cubed_container_roundtrip_on_device():
    """BASELINE size: a 512^3 result (512 MiB) -> 671 MB container -> back, all in HBM; bit-exact identity, and the
    container equals the oracle's bytes (compared through a checksum of 64-bit words to keep host time small)."""
    import torch

    n = 512**3
    grid = Grid.new([0.1, 0.2, 0.3], [0.01, 0.01, 0.01], [512, 512, 512])
    gen = torch.Generator(device="cuda").manual_seed(7)
    d = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda", generator=gen).view(torch.float32)
    data = serialize(SerializeGrid(grid, d))
    assert data.numel() == _lib.lib().m2s_sdf_grid_encoded_size(C.byref(grid._g), n) == 59 + 5 * n   # cd-form counts, array32
    de = deserialize(data)
    assert torch.equal(de.distances.view(torch.int32), d.view(torch.int32)) and de.grid == grid
    want = np.frombuffer(so.pack_with_numpy("Grid", grid=([0.1, 0.2, 0.3], [0.01, 0.01, 0.01], [512, 512, 512]),
                                            distances=d.cpu().numpy()), np.uint8)
    got = data.cpu().numpy()
    assert got.size == want.size and np.array_equal(got[:4096], want[:4096]) and np.array_equal(got[-4096:], want[-4096:])
    k = got.size // 8 * 8
    assert int(got[:k].view(np.uint64).sum(dtype=np.uint64)) == int(want[:k].view(np.uint64).sum(dtype=np.uint64))
    assert np.array_equal(got, want)

"""Loader for the gfx950 kernel library (mesh_to_sdf_amd/libm2s_hip.so, C ABI of include/m2s.h).

There is deliberately NO fallback: if the HIP library is missing or cannot be loaded, every
compute entry point raises.  The CPU oracle under oracle/ is test infrastructure and is never
imported from here.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("M2S_LIB") or os.path.join(_HERE, "libm2s_hip.so")   # M2S_LIB: A/B builds of the same library (tools/)
CSRC = os.path.join(_HERE, "csrc")

M2S_OK = 0
ERR_BAD_ARG, ERR_NAN, ERR_EMPTY_MESH, ERR_HIP, ERR_IO = -1, -2, -3, -4, -5
MEM_HOST, MEM_DEVICE = 0, 1


class M2SGrid(C.Structure):
    _fields_ = [("first_cell", C.c_float * 3), ("cell_size", C.c_float * 3), ("cell_count", C.c_uint64 * 3)]


class M2STimings(C.Structure):
    _fields_ = [
        ("accel_build_ms", C.c_float),
        ("sign_ms", C.c_float),
        ("distance_ms", C.c_float),
        ("total_ms", C.c_float),
        ("seed_ms", C.c_float),
        ("reserved_f", C.c_float),
        ("n_triangles", C.c_uint64),
        ("n_units", C.c_uint64),
        ("distance_launches", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class M2SOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("stream", C.c_void_p),
        ("mem_kind", C.c_int32),
        ("algorithm", C.c_int32),
        ("x_begin", C.c_uint64),
        ("x_end", C.c_uint64),
        ("timings", C.POINTER(M2STimings)),
        ("synchronous", C.c_int32),
        ("stream_mode", C.c_int32),
        # version 0.2 (struct_size >= sizeof): context lane, peer outputs
        ("lane", C.c_int32),
        ("n_peer_out", C.c_uint32),
        ("peer_out", C.POINTER(C.c_void_p)),
        ("peer_mode", C.c_int32),
        ("x_period", C.c_uint32),
    ]


OPTS_V1_SIZE = 56
MAX_PEERS = 15
PEER_PUSH, PEER_STORE, PEER_TRAIL = 0, 1, 2
XCHG_AUTO, XCHG_PEER, XCHG_RCCL, XCHG_NONE = 0, 1, 2, 3
IPC_HANDLE_BYTES = 64


class M2SMultiOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_devices", C.c_int32),
        ("devices", C.POINTER(C.c_int32)),
        ("mem_kind", C.c_int32),
        ("exchange", C.c_int32),
        ("peer_mode", C.c_int32),
        ("algorithm", C.c_int32),
        ("timings", C.POINTER(M2STimings)),
        ("wall_ms", C.POINTER(C.c_float)),
        ("exchange_used", C.POINTER(C.c_int32)),
        ("partition", C.c_int32),
        ("reserved", C.c_int32),
        # version 0.3
        ("partition_used", C.POINTER(C.c_int32)),
        ("slabs", C.POINTER(C.c_uint64)),
    ]


MULTI_OPTS_V1_SIZE, MULTI_OPTS_V2_SIZE = 56, 64
PART_AUTO, PART_CONTIGUOUS, PART_INTERLEAVED, PART_ADAPTIVE = 0, 1, 2, 3


class M2SSdfInfo(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("canonical", C.c_int32),
        ("grid", M2SGrid),
        ("n_queries", C.c_uint64),
        ("n_distances", C.c_uint64),
        ("queries_offset", C.c_uint64),
        ("distances_offset", C.c_uint64),
    ]


class M2SInstance(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p),
        ("n_vertices", C.c_size_t),
        ("vertex_stride", C.c_size_t),
        ("indices", C.c_void_p),
        ("n_indices", C.c_size_t),
        ("transform", C.c_float * 16),
    ]


class M2SGltfInfo(C.Structure):
    _fields_ = [("n_scenes", C.c_uint64), ("n_models", C.c_uint64), ("n_instances", C.c_uint64),
                ("n_vertices", C.c_uint64), ("n_indices", C.c_uint64)]


# every symbol include/m2s.h declares
EXPORTS = [
    "m2s_generate_sdf",
    "m2s_generate_grid_sdf",
    "m2s_grid_from_bounding_box",
    "m2s_grid_cell_center",
    "m2s_grid_cell_idx",
    "m2s_triangle_count",
    "m2s_warmup",
    "m2s_version",
    "m2s_device_count",
    "m2s_last_error",
    "m2s_release_workspace",
    "m2s_mesh_create",
    "m2s_mesh_destroy",
    "m2s_mesh_triangle_count",
    "m2s_mesh_generate_grid_sdf",
    "m2s_mesh_generate_sdf",
    "m2s_mesh_drain_timings",
    "m2s_sdf_grid_encoded_size",
    "m2s_sdf_generic_encoded_size",
    "m2s_sdf_encode_grid",
    "m2s_sdf_encode_generic",
    "m2s_sdf_probe",
    "m2s_sdf_decode",
    "m2s_sdf_save_grid",
    "m2s_sdf_save_generic",
    "m2s_sdf_probe_file",
    "m2s_sdf_read_file",
    "m2s_order_cells_by_distance",
    "m2s_merge_instances",
    "m2s_gltf_open",
    "m2s_gltf_instances",
    "m2s_gltf_close",
    "m2s_generate_grid_sdf_multi",
    "m2s_generate_sdf_multi",
    "m2s_slab_bounds",
    "m2s_balanced_slabs",
    "m2s_interleaved_slab",
    "m2s_peer_bandwidth",
    "m2s_shared_alloc",
    "m2s_shared_free",
    "m2s_ipc_export",
    "m2s_ipc_open",
    "m2s_ipc_close",
    "m2s_tuning_set",
    "m2s_tuning_describe",
]


def build(force=False):
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return SO_PATH


def _prototypes():
    """name -> (restype, argtypes) of every symbol include/m2s.h declares."""
    return {
        "m2s_generate_sdf": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(M2SOpts)]),
        "m2s_generate_grid_sdf": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(M2SGrid), C.c_int, C.c_void_p, C.POINTER(M2SOpts)]),
        "m2s_grid_from_bounding_box": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(M2SGrid)]),
        "m2s_grid_cell_center": (None, [C.POINTER(M2SGrid), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]),
        "m2s_grid_cell_idx": (C.c_uint64, [C.POINTER(M2SGrid), C.POINTER(C.c_uint64)]),
        "m2s_triangle_count": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_int, C.c_int]),
        "m2s_warmup": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t]),
        "m2s_version": (C.c_int, []),
        "m2s_device_count": (C.c_int, []),
        "m2s_last_error": (C.c_char_p, []),
        "m2s_release_workspace": (None, []),
        "m2s_mesh_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(M2SOpts), C.POINTER(C.c_void_p)]),
        "m2s_mesh_destroy": (None, [C.c_void_p]),
        "m2s_mesh_triangle_count": (C.c_size_t, [C.c_void_p]),
        "m2s_mesh_generate_grid_sdf": (C.c_int, [C.c_void_p, C.POINTER(M2SGrid), C.c_int, C.c_void_p, C.POINTER(M2SOpts)]),
        "m2s_mesh_generate_sdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(M2SOpts)]),
        "m2s_mesh_drain_timings": (C.c_int, [C.c_void_p, C.POINTER(M2STimings)]),
        "m2s_sdf_grid_encoded_size": (C.c_size_t, [C.POINTER(M2SGrid), C.c_size_t]),
        "m2s_sdf_generic_encoded_size": (C.c_size_t, [C.c_size_t, C.c_size_t]),
        "m2s_sdf_encode_grid": (C.c_int, [C.POINTER(M2SGrid), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(M2SOpts)]),
        "m2s_sdf_encode_generic": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(M2SOpts)]),
        "m2s_sdf_probe": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(M2SSdfInfo), C.POINTER(M2SOpts)]),
        "m2s_sdf_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(M2SOpts)]),
        "m2s_sdf_save_grid": (C.c_int, [C.c_char_p, C.POINTER(M2SGrid), C.c_void_p, C.c_size_t, C.POINTER(M2SOpts)]),
        "m2s_sdf_save_generic": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(M2SOpts)]),
        "m2s_sdf_probe_file": (C.c_int, [C.c_char_p, C.POINTER(M2SSdfInfo)]),
        "m2s_sdf_read_file": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(M2SOpts)]),
        "m2s_order_cells_by_distance": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_float), C.POINTER(M2SOpts)]),
        "m2s_merge_instances": (C.c_int, [C.POINTER(M2SInstance), C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(M2SOpts)]),
        "m2s_gltf_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(M2SGltfInfo)]),
        "m2s_gltf_instances": (C.c_int, [C.c_void_p, C.POINTER(M2SInstance), C.c_size_t]),
        "m2s_gltf_close": (None, [C.c_void_p]),
        "m2s_generate_grid_sdf_multi": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(M2SGrid), C.c_int, C.POINTER(C.c_void_p), C.POINTER(M2SMultiOpts)]),
        "m2s_generate_sdf_multi": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(M2SMultiOpts)]),
        "m2s_slab_bounds": (None, [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "m2s_peer_bandwidth": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "m2s_balanced_slabs": (C.c_int, [C.c_uint64, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
        "m2s_interleaved_slab": (C.c_int, [C.POINTER(M2SGrid), C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "m2s_shared_alloc": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
        "m2s_shared_free": (C.c_int, [C.c_void_p, C.c_int]),
        "m2s_ipc_export": (C.c_int, [C.c_void_p, C.c_char_p]),
        "m2s_ipc_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
        "m2s_ipc_close": (C.c_int, [C.c_void_p, C.c_int]),
        "m2s_tuning_set": (C.c_int, [C.c_char_p, C.c_char_p]),
        "m2s_tuning_describe": (C.c_int, [C.c_char_p, C.c_int]),
    }


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(mesh_to_sdf_amd has no CPU fallback)"
            )
        # torch ships its own libamdhip64.so (same SONAME, libamdhip64.so.7).  Two HIP/HSA runtimes in
        # one process cannot both open the GPU, so when torch is installed it must be loaded FIRST:
        # this library's NEEDED libamdhip64.so.7 then binds to the runtime torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        older = bool(os.environ.get("M2S_LIB"))   # an A/B build of an OLDER library (tools/exp_ab.py) may lack the newest entry points
        for name, (restype, argtypes) in _prototypes().items():
            if older and not hasattr(L, name):
                continue                      # only the missing symbol is skipped: everything the old library has keeps its types
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def set_knob(name, value=None):
    """m2s_tuning_set: one of the run-time knobs of csrc/tuning.h (value None = back to its default).  The library reads the
    environment only once, at its first use; the tests and tools that switch walk flavours between calls go through here."""
    L = lib()
    if not hasattr(L, "m2s_tuning_set"):              # an older A/B library reads its knobs from the environment per call
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(value)
        return
    rc = L.m2s_tuning_set(name.encode(), None if value is None else str(value).encode())
    if rc != M2S_OK:
        raise ValueError(last_error())


class knobs:
    """with knobs(M2S_LANE_WALK=1, M2S_BRUTE_MAX=0): ...   — sets the knobs, puts the previous values back afterwards (nests)."""

    def __init__(self, **kw):
        self.kw = kw
        self.before = {}

    def __enter__(self):
        if self.kw and hasattr(lib(), "m2s_tuning_describe"):
            now = describe_knobs()
            self.before = {k: now.get(k) for k in self.kw}
        for k, v in self.kw.items():
            set_knob(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_knob(k, self.before.get(k))
        return False


def describe_knobs():
    L = lib()
    buf = C.create_string_buffer(4096)
    L.m2s_tuning_describe(buf, 4096)
    return dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)


def last_error():
    return lib().m2s_last_error().decode("utf-8", "replace")

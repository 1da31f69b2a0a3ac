"""mesh_to_sdf_amd — MI355X-native hot path of Azkellas/mesh_to_sdf (generate_sdf / generate_grid_sdf).

The compute lives in csrc/ (hand-written HIP for gfx950 behind the C ABI of include/m2s.h);
this package is the thin host-side mirror of the reference's public interface
(mesh_to_sdf/src/lib.rs:146-311, generate/grid.rs:265-270, grid.rs:30-141).
"""
from .api import (AccelerationMethod, Exchange, Grid, M2SError, M2SPanic, Mesh, Partition, PeerMode, SharedGrid, SignMethod,  # noqa: F401
                  Topology, generate_grid_sdf, generate_grid_sdf_multi, generate_sdf, generate_sdf_multi, interleaved_slab, slab_bounds, balanced_slabs, peer_bandwidth, warmup)
from ._lib import M2STimings  # noqa: F401
from . import serde  # noqa: F401,E402  (mesh_to_sdf::serde, serde.rs)

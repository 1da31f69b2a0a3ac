"""Multi-GPU generate_grid_sdf: one process per GPU, x-slab shards, one RCCL all-gather.

Every voxel depends only on the (replicated) mesh, so the grid shards with no data-path
exchange.  The reference's output layout is x-slowest (`idx = z + y*nz + x*ny*nz`, grid.rs:122-124),
so a slab of cells along cell-axis 0 is one CONTIGUOUS range of the output and the final gather is
an in-place `all_gather_into_tensor` (backend "nccl" == RCCL over xGMI on ROCm).  Each rank marks
the sign planes for the whole grid (hits are slab independent: t is measured from cell 0 of the
line, generate/grid.rs:570-617) — O(T) work, no halo, no exchange.
"""
from typing import Callable, Optional, Tuple

from .api import Grid, SignMethod, Topology, generate_grid_sdf


def slab_bounds(nx: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous x-slab [x0, x1) of rank `rank`; sizes differ by at most one cell layer."""
    base, rem = divmod(nx, world)
    x0 = rank * base + min(rank, rem)
    return x0, x0 + base + (1 if rank < rem else 0)


def gather_slabs(out, nx: int, row_cells: int, group=None):
    """Make every rank's `out` (the whole grid, flat) complete: rank r has filled its slab."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return out
    rank = dist.get_rank(group)
    if nx % world == 0:
        x0, x1 = slab_bounds(nx, world, rank)
        dist.all_gather_into_tensor(out, out[x0 * row_cells : x1 * row_cells], group=group)  # in place
    else:  # uneven slabs: one broadcast per slab
        for r in range(world):
            x0, x1 = slab_bounds(nx, world, r)
            if x1 > x0:
                dist.broadcast(out[x0 * row_cells : x1 * row_cells], src=dist.get_global_rank(group, r) if group else r, group=group)
    return out


def generate_grid_sdf_sharded(vertices, indices: Topology, grid: Grid, sign_method: SignMethod = SignMethod.Raycast, *,
                              group=None, out=None, compute_slab: Optional[Callable] = None, timings=None,
                              gather: bool = True):
    """generate_grid_sdf over all ranks of `group` (default: the world).  `vertices`/`indices` are
    this rank's copies (CUDA tensors for the HIP path); returns the full grid on every rank.

    compute_slab(out, x0, x1) may replace the slab computation (the CPU tests inject the oracle
    there so the sharding/gather logic runs under gloo without a GPU)."""
    import torch
    import torch.distributed as dist

    nx, ny, nz = grid.get_cell_count()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    x0, x1 = slab_bounds(nx, world, rank)
    if out is None:
        dev = vertices.device if hasattr(vertices, "device") else "cpu"
        out = torch.empty(nx * ny * nz, dtype=torch.float32, device=dev)
    if compute_slab is not None:
        compute_slab(out, x0, x1)
    elif x1 > x0:
        generate_grid_sdf(vertices, indices, grid, sign_method, x_slab=(x0, x1), out=out, timings=timings)
    if gather and world > 1:
        gather_slabs(out, nx, ny * nz, group)
    return out

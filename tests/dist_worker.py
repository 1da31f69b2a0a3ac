"""Worker of the multi-process -m gpu tests (tests/test_gpu_multi.py), launched through torch.distributed.run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    backend = "nccl" if mode == "nccl_one_rank" else os.environ.get("M2S_DIST_BACKEND", "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)

    from mesh_to_sdf_amd import Grid, PeerMode, SignMethod, Topology, generate_grid_sdf, meshes
    from mesh_to_sdf_amd.distributed import PeerGrid, generate_grid_sdf_sharded

    v, idx = meshes.named("blob-6k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    g = Grid.from_bounding_box(lo, hi, [75, 40, 36] if mode != "ipc_peer" else [128, 44, 40])   # 128 = 2 ranks x 2 chunks x 32 layers
    dv = torch.as_tensor(v, device=f"cuda:{dev}")
    di = torch.as_tensor(idx.astype(np.int64), device=f"cuda:{dev}")
    topo = Topology.TriangleList(di)
    want = generate_grid_sdf(dv, topo, g, SignMethod.Raycast)

    if mode == "ipc_peer":
        pg = PeerGrid(g.get_total_cell_count(), dev)
        for pm in (PeerMode.Push, PeerMode.Store, PeerMode.Trail, PeerMode.Push):
            pg.tensor.fill_(float("nan"))
            torch.cuda.synchronize()
            dist.barrier()
            for interleave in (True, False):
                pg.tensor.fill_(float("nan"))
                torch.cuda.synchronize()
                dist.barrier()
                out = generate_grid_sdf_sharded(dv, topo, g, SignMethod.Raycast, peer_grid=pg, peer_mode=pm, interleave=interleave)
                assert out.data_ptr() == pg.tensor.data_ptr()
                assert torch.equal(out.view(torch.int32), want.view(torch.int32)), f"rank {rank} mode {pm} interleave {interleave}"
        pg.close()
        print(f"ipc_peer ok rank {rank}", flush=True)
    elif mode == "nccl_one_rank":
        for chunks in (1, 3):
            out = generate_grid_sdf_sharded(dv, topo, g, SignMethod.Raycast, chunks=chunks)
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int32), want.view(torch.int32)), chunks
        from mesh_to_sdf_amd.distributed import generate_sdf_sharded
        q = torch.as_tensor(meshes.uniform_queries(lo, hi, 3001), device=f"cuda:{dev}")
        from mesh_to_sdf_amd import generate_sdf
        assert torch.equal(generate_sdf_sharded(dv, topo, q), generate_sdf(dv, topo, q))
        print("nccl_one_rank ok", flush=True)
    else:
        raise SystemExit(f"unknown mode {mode}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: Mvoxels/s of generate_grid_sdf on a 512^3 grid,
100k-triangle watertight mesh (blob-100k), SignMethod::Raycast, at N GPUs.

One "step" = one complete generate_grid_sdf call through the C ABI with the mesh and the output
resident in HBM (device pointers): topology flatten + LBVH build + sign planes + nearest-triangle
kernel for this rank's contiguous x-slab, and its delivery so that the full grid is on every GPU
(N = 1: one m2s_generate_grid_sdf call; N > 1: mesh_to_sdf_amd/distributed.py — peer writes over xGMI
through m2s_opts.peer_out, or chunked RCCL all-gathers with M2S_EXCHANGE=rccl).
The work is FIXED as N grows (strong scaling): rank r computes cells x in [r*512/N, (r+1)*512/N).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel (k_packet, nearest triangle +
fused sign) timed with HIP events on the stream it is launched on; `cpu_baseline` is the CPU
oracle's multi-threaded restatement of the reference algorithm (oracle/, kind "port": the Rust
reference cannot be built in this image) on a bounded sample, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP multiplexes the streams of a process onto 4 hardware queues by default.  A rank of the sharded run uses torch's
# stream, two piece streams and RCCL's: one more stream anywhere and two of them share a queue and serialise (measured:
# +0.4 ms per step of an 8-GPU rank).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the host driver only supports dmabuf IPC: RCCL and the peer exchange (m2s_ipc_*) map each other's buffers through it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--mesh", default="blob-100k")
    ap.add_argument("--sign", default="Raycast", choices=["Raycast", "Normal"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child runs after the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=150.0, help="budget for the CPU baseline run (512^3 needs ~40-60 s)")
    ap.add_argument("--chunks", type=int, default=0, help="x-chunks per step whose all-gathers overlap compute (0 = auto)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="another BASELINE.json config on ONE GPU instead of the headline (0): 2 = 256^3 blob-100k Raycast, 3 = 10 M queries "
                         "RtreeBvh, 4 = 512^3 blob-1M Raycast, 5 = 1024^3 sheet-100k Normal; same JSON line with that config's roofline")
    args = ap.parse_args()
    if args.config in (2, 4, 5):
        args.mesh, args.grid, args.sign = {2: ("blob-100k", 256, "Raycast"), 4: ("blob-1M", 512, "Raycast"), 5: ("sheet-100k", 1024, "Normal")}[args.config]
        args.cpu_seconds = min(args.cpu_seconds, 45.0)   # a bounded sample (<= 384^3) of the config's own mesh / sign rule beside its line
    return args


def bench_queries(args):
    """BASELINE config 3: generate_sdf on 10 M uniform random queries x blob-100k, AccelerationMethod::RtreeBvh, one GPU,
    queries and distances resident in HBM; one step = one complete m2s_generate_sdf call (build + Morton sort + walk)."""
    import torch

    from mesh_to_sdf_amd import AccelerationMethod, M2STimings, Topology, generate_sdf, meshes

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    v, idx = meshes.named("blob-100k")
    lo, hi = meshes.extended_bbox(v, 0.1)
    nq = 10_000_000
    dq = torch.as_tensor(meshes.uniform_queries(lo, hi, nq), device=dev)
    dv = torch.as_tensor(v, device=dev)
    topo = Topology.TriangleList(torch.as_tensor(idx.astype(np.int64), device=dev).to(torch.int32))
    tims = []
    for _ in range(args.warmup):
        generate_sdf(dv, topo, dq, AccelerationMethod.RtreeBvh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t = M2STimings()
        generate_sdf(dv, topo, dq, AccelerationMethod.RtreeBvh, timings=t)
        tims.append(t)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dist_ms = float(np.mean([t.distance_ms for t in tims]))      # Morton sort + seed lattice + k_packet<LIST, UNSIGNED, RAYS3>
    b_alg = 16.0 * nq + 12.0 * v.shape[0] + 12.0 * int(tims[0].n_triangles)   # 16 B per query + the mesh once
    achieved = b_alg / (dist_ms * 1e-3) / 1e9
    traffic = traffic_src = None
    if not args.no_live_pmc:
        live, why_not = live_pmc(["--config", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc"])
        if live is not None:
            traffic = 2.0 * live["FETCH_SIZE"] * 1024.0 + live["WRITE_SIZE"] * 1024.0
            traffic_src = (f"live: rocprofv3 --pmc child runs of this command after the timed region, k_packet only (the query sort and seed lattice in front "
                           f"of it are not in this figure), {live['FETCH_SIZE_dispatches']} launches; FETCH_SIZE {live['FETCH_SIZE'] * 1024 / 1e6:.1f} MB as reported, "
                           f"doubled (gfx950 half-count); WRITE_SIZE {live['WRITE_SIZE'] * 1024 / 1e6:.1f} MB")
        elif why_not != "child run":
            traffic_src = f"live measurement unavailable ({why_not})"
    extra = {}
    if not args.no_cpu_baseline:
        # "CPU BVH nearest + 3-ray sign" beside the 10 M-query run (generic/rtree_bvh.rs:79-174): the oracle's
        # BVH-accelerated exact search + best-of-three rays, all host threads, on the first 1 M queries of the same set; the same
        # sample also checks the GPU result bit for bit
        import oracle as orc

        ns = 1_000_000
        qs = meshes.uniform_queries(lo, hi, nq)[:ns]
        cores = orc.hardware_threads()
        t1 = time.perf_counter()
        want = orc.generate_sdf(v, idx, qs, accel=3, fast=True, threads=cores)
        dt = time.perf_counter() - t1
        got = generate_sdf(dv, topo, dq[:ns].contiguous(), AccelerationMethod.RtreeBvh).cpu().numpy()
        extra["cpu_baseline"] = {"value": round(ns / dt / 1e6, 4), "unit": "Mqueries/s", "cores": cores, "kind": "port",
                                 "sample": f"the first {ns} of the {nq} queries, same mesh: C++ restatement of the reference's exact nearest search + "
                                           f"best-of-three rays with a BVH (oracle/, fast path), {cores} threads, {dt:.2f} s"}
        extra["oracle_check"] = {"queries": ns, "bit_identical": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))),
                                 "sign_mismatches": int(np.count_nonzero(np.signbit(got) != np.signbit(want)))}
    emit(json.dumps({
        "metric": "Mqueries/s for generate_sdf (10M random queries, 100k tris, RtreeBvh)", "value": round(nq * args.steps / elapsed / 1e6, 2),
        "unit": "Mqueries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 3: generate_sdf, 10 000 000 uniform queries in the extended bbox, blob-100k, AccelerationMethod::RtreeBvh "
                               "(nearest distance + best of three rays), inputs/outputs resident in HBM", "queries": nq, "mesh": "blob-100k"},
        "phases_ms": {"accel_build": round(float(np.mean([t.accel_build_ms for t in tims])), 4), "sort_seed_walk": round(dist_ms, 4)},
        "roofline": {"bound": "hbm", "kernel": "k_packet<LIST, MODE_UNSIGNED, SIGN_RAYS3> (+ the query sort and seed lattice in front of it)",
                     "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": b_alg, "avg_launch_ms": round(dist_ms, 4)},
        **extra,
    }))


def live_pmc(child_args, kernel="k_packet"):
    """HBM-side traffic and VALU issue of the dominant kernel, MEASURED for this very invocation: after the timed region the same
    command is run again as a child under `rocprofv3 --pmc`, one pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one
    pass: /opt/skills/guides/MI355X_MICROARCH.md), and the per-dispatch sums of the kernel's rows are averaged.  Returns
    (dict, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("M2S_BENCH_CHILD") == "1":
        return None, "child run"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    got = {}
    # a plain kernel trace first: the PROFILER's average duration of the kernel (what SURVEY.md section 8(d) prices the roofline with; the
    # HIP-event time of the timed region sits beside it) — counters would lengthen the launches, so it has a run of its own
    d = tempfile.mkdtemp(prefix="m2s_trace_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__)] + child_args
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", M2S_BENCH_CHILD="1"), capture_output=True, text=True, timeout=150)
        by_name = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"]:
                    by_name.setdefault(row["Kernel_Name"], []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        # ONE variant: the instantiation that carries the most time (a step launches one k_packet<...> form; follow-up rounds of a split
        # walk and other forms of a forced run must not dilute the average)
        name = max(by_name, key=lambda k: sum(x[1] for x in by_name[k])) if by_name else None
        durs = by_name.get(name, [])
        if r.returncode == 0 and len(durs) > 1:
            durs = [x[1] for x in sorted(durs)][1:]            # without the first launch (the child's warm-up step)
            got["rocprof_avg_ms"] = sum(durs) / len(durs) / 1e6
            got["rocprof_dispatches"] = len(durs)
            got["rocprof_kernel_name"] = name
    except Exception:   # noqa: BLE001
        pass
    finally:
        shutil.rmtree(d, ignore_errors=True)
    for counters in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES"):
        d = tempfile.mkdtemp(prefix="m2s_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                                                                sys.executable, os.path.abspath(__file__)] + child_args
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", M2S_BENCH_CHILD="1"), capture_output=True, text=True, timeout=150)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counters} failed (rc {r.returncode}): {r.stderr[-300:]}"
            acc = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if kernel in row["Kernel_Name"]:
                        per = acc.setdefault(row["Counter_Name"], {})
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            for c in counters.split():
                if c not in acc:
                    return None, f"no {c} rows for {kernel}"
                got[c] = sum(acc[c].values()) / len(acc[c])
                got[c + "_dispatches"] = len(acc[c])
        except Exception as e:   # noqa: BLE001
            return None, f"{type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return got, None


def cpu_baseline(v, idx, lo, hi, sign, budget_s, headline_n):
    """Times the oracle's faithful multi-threaded restatement of generate/grid.rs:265-642 (one heap
    per thread, like rayon::current_num_threads()) on the SAME mesh and bbox: directly on the headline
    grid when the time budget allows (512^3: ~40 s, ~2 GB), else on the largest cubic sample that fits.
    Returns (json dict, n, the propagation's output on the n^3 grid) — the output feeds `reference_parity`."""
    import oracle as orc
    from mesh_to_sdf_amd import meshes

    cores = orc.hardware_threads()

    def run(n, th):
        first, size, cnt = meshes.grid_from_bounding_box(lo, hi, [n, n, n])
        t0 = time.perf_counter()
        # heaps == threads: the reference makes one heap per rayon thread (generate/grid.rs:318-339)
        ref = orc.generate_grid_sdf(v, idx, first, size, cnt, sign=sign, semantics=orc.PROPAGATE, heaps=th, threads=th)
        return time.perf_counter() - t0, ref

    # The port does not scale to every core count (more heaps = more redundant propagation, and the
    # algorithm has serial O(N^3) passes, as the reference does): probe a few thread counts on 128^3 and
    # report the FASTEST configuration, so the baseline is the strongest CPU number, not the weakest;
    # the all-cores figure (what rayon's default pool would use) is printed beside it.
    probe = {}
    for th in sorted({min(cores, t) for t in (16, 32, 64, cores)}):
        probe[th] = run(128, th)[0]
    best_th = min(probe, key=probe.get)
    rate = 128 ** 3 / probe[best_th]
    n, dt, ref = 128, probe[best_th], None
    for cand in (headline_n, 384, 256, 192):
        if cand <= headline_n and cand ** 3 / rate * 1.25 <= budget_s:
            n = cand
            dt, ref = run(cand, best_th)
            break
    if ref is None:
        dt, ref = run(n, best_th)
    all_cores = None
    if cores != best_th:
        m = min(n, 256)
        all_cores = round(m ** 3 / run(m, cores)[0] / 1e6, 4)
    res = {
        "value": round(n ** 3 / dt / 1e6, 4),
        "unit": "Mvoxels/s",
        "cores": best_th,
        "kind": "port",
        "sample": f"{n}^3 grid ({n ** 3 / headline_n ** 3:.4f} of the {headline_n}^3 voxels), same mesh and bbox, C++ restatement of the reference's "
                  f"3-phase propagation algorithm (oracle/), {best_th} threads = fastest of {sorted(probe)} on this {cores}-thread host, {dt:.2f} s",
        "all_cores_value": all_cores,
        "all_cores": cores,
        "all_cores_note": f"rayon-equivalent pool (one heap per hardware thread), {min(n, 256)}^3 sample",
        "probe_128_s": {str(k): round(x, 3) for k, x in sorted(probe.items())},
    }
    return res, n, ref


_JSON_FD = None


def _stdout_is_for_the_json_line_only():
    """RCCL prints a version banner on STDOUT when a communicator is created (and libraries below us may print more): the driver
    reads one JSON line from stdout, so file descriptor 1 is pointed at stderr for the whole run and the JSON line is written
    to a duplicate of the original stdout at the end."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line: str):
    sys.stdout.flush()
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


def main():
    args = parse()
    _stdout_is_for_the_json_line_only()
    if args.config == 3:
        return bench_queries(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Three ways to run (all through the C ABI of include/m2s.h):
    #   N = 1                                   one m2s_generate_grid_sdf call per step (the reference's call)
    #   N > 1 under torch.distributed.run       one process per GPU; every rank computes its contiguous x-slab with one call and
    #                                           delivers it by peer writes over xGMI (m2s_opts.peer_out on IPC-mapped buffers,
    #                                           M2S_EXCHANGE=peer, default) or by chunked in-place RCCL all-gathers (=rccl)
    #   N > 1 started as plain `python bench.py` one process: m2s_generate_grid_sdf_multi, one host thread per device
    in_process = world == 1 and args.gpus > 1
    n_gpus = args.gpus if in_process else world
    local_dev = local_rank % max(torch.cuda.device_count(), 1)   # one rank per GPU; the modulo only matters in 1-GPU tests
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    force_pg = os.environ.get("M2S_FORCE_COLLECTIVES", "0") == "1"   # test hook: 1-rank RCCL group
    if world > 1 or (force_pg and "MASTER_ADDR" in os.environ):
        backend = os.environ.get("M2S_DIST_BACKEND", "nccl")   # "nccl" == RCCL over xGMI; gloo only for 1-GPU logic tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mesh_to_sdf_amd import (Exchange, Grid, M2STimings, PeerMode, SignMethod, Topology, generate_grid_sdf,
                                 generate_grid_sdf_multi, meshes)
    from mesh_to_sdf_amd.distributed import PeerGrid, generate_grid_sdf_sharded, slab_bounds

    v, idx = meshes.named(args.mesh)
    lo, hi = meshes.extended_bbox(v, 0.1)
    n = args.grid
    grid = Grid.from_bounding_box(lo, hi, [n, n, n])
    sign = SignMethod[args.sign]
    dv = torch.as_tensor(v, device=dev)
    di = torch.as_tensor(idx.astype(np.int64), device=dev).to(torch.int32)
    topo = Topology.TriangleList(di)
    x0, x1 = slab_bounds(n, world, rank)
    sharded = world > 1 or force_pg
    peer_mode = PeerMode[os.environ.get("M2S_PEER_MODE", "Push")]
    exchange = os.environ.get("M2S_EXCHANGE", "peer") if sharded else ("in-process" if in_process else "none")

    def all_agree(ok):
        if not (dist.is_available() and dist.is_initialized()) or world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    pg, why = None, None
    if sharded and exchange == "peer":
        try:
            pg = PeerGrid(n ** 3, local_dev)
        except Exception as e:   # noqa: BLE001  (IPC mapping not permitted on this host, ...)
            why = f"{type(e).__name__}: {e}"
        if not all_agree(pg is not None):
            if pg is not None:
                pg.close(sync=False)      # not every rank has one: no collective here
            pg, exchange = None, "rccl"
    out = pg.tensor if pg is not None else torch.empty(n ** 3, dtype=torch.float32, device=dev)
    outs = devices = None
    if in_process:
        devices = list(range(args.gpus))
        if os.environ.get("M2S_BENCH_DEVICES"):                   # e.g. "0,0,0,0": the in-process path on a 1-GPU box
            devices = [int(d) for d in os.environ["M2S_BENCH_DEVICES"].split(",")]
        outs = [out] + [torch.empty(n ** 3, dtype=torch.float32, device=f"cuda:{d}") for d in devices[1:]]

    def measure(exchange):
        chunks = args.chunks if args.chunks > 0 else (4 if exchange == "rccl" else 1)

        def step():
            # one complete call: LBVH build, sign planes, seed passes, cut lists, nearest-triangle launches (+ delivery of the slab)
            if in_process:
                info = {}
                generate_grid_sdf_multi(dv, topo, grid, sign, devices=devices, outs=outs, exchange=Exchange.Auto, peer_mode=peer_mode, info=info)
                return info["timings"][0]
            t = M2STimings()
            if not sharded:
                generate_grid_sdf(dv, topo, grid, sign, out=out, timings=t)
            elif exchange == "peer":
                generate_grid_sdf_sharded(dv, topo, grid, sign, peer_grid=pg, peer_mode=peer_mode, timings=t)
            else:
                # this rank's x-pieces through the persistent mesh + the all-gathers (asynchronous per chunk, overlapping the
                # next chunk's compute)
                _, mesh = generate_grid_sdf_sharded(dv, topo, grid, sign, out=out, chunks=chunks, return_mesh=True)
                t = mesh.drain_timings()   # waits for the launches of this step, reads their HIP-event durations
                mesh.close()
            return t

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tims = []
        t0 = time.perf_counter()
        for k in range(args.steps):
            tims.append(step())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = float(te.item())
        return elapsed, tims, chunks

    elapsed, tims, chunks = measure(exchange)

    # N > 1: the delivered grid must be the grid (checked after the timed region, every rank against its own single-GPU run)
    verified = None
    if sharded or in_process:
        ref = torch.empty(n ** 3, dtype=torch.float32, device=dev)
        generate_grid_sdf(dv, topo, grid, sign, out=ref)
        ok = all(bool(torch.equal(o.to(dev).view(torch.int32), ref.view(torch.int32))) for o in (outs if in_process else [out]))
        verified = all_agree(ok)
        if not verified and exchange == "peer":      # never report a number for a wrong grid: measure the RCCL path instead
            why = "peer exchange delivered a grid that differs from the single-GPU result"
            exchange = "rccl"
            out = torch.empty(n ** 3, dtype=torch.float32, device=dev)
            elapsed, tims, chunks = measure(exchange)
            verified = all_agree(bool(torch.equal(out.view(torch.int32), ref.view(torch.int32))))
        del ref

    # N > 1: what every rank spent where (means over the timed steps), and what its links give — so that ONE run on a multi-GPU
    # node says which delivery (push / store / trail, peer vs RCCL) to choose instead of a bare number
    per_rank = link_probe = None
    if sharded or in_process:
        def rank_line(ts, wall_ms):
            dev_total = float(np.mean([t.total_ms for t in ts]))
            return {"accel_build_ms": round(float(np.mean([t.accel_build_ms for t in ts])), 4), "seeds_and_cut_ms": round(float(np.mean([t.seed_ms for t in ts])), 4),
                    "walk_ms": round(float(np.mean([t.distance_ms for t in ts])), 4), "walk_launches": int(ts[0].distance_launches),
                    "device_total_ms": round(dev_total, 4), "step_wall_ms": round(wall_ms, 4),
                    # after the last walk's kernels: the exposed part of the delivery (last piece's push / gather), the barrier wait for
                    # the slowest rank, host overhead
                    "after_walk_ms": round(wall_ms - dev_total, 4)}
        if in_process:
            info = {}
            generate_grid_sdf_multi(dv, topo, grid, sign, devices=devices, outs=outs, exchange=Exchange.Auto, peer_mode=peer_mode, info=info)
            per_rank = [dict(rank_line([t], info["wall_ms"]), shard=k, device=devices[k], slab=list(info["slabs"][k])) for k, t in enumerate(info["timings"])]
            exchange_ran = info["exchange"]
        else:
            mine = dict(rank_line(tims, elapsed * 1e3 / args.steps), rank=rank)
            per_rank = [None] * world
            if world > 1:
                dist.all_gather_object(per_rank, mine)
            else:
                per_rank = [mine]
            exchange_ran = exchange
        # A probe must never cost the run its number — and in the one-process-per-GPU form every rank must issue the same collectives
        # whatever happened to ITS probe: the error is caught into the rank's own entry, the gather stands outside the try.
        probe_cells = min(n ** 3, 16 << 20)                     # 64 MB to each peer
        if in_process and len(set(devices)) > 1:
            try:
                from mesh_to_sdf_amd import peer_bandwidth
                each, together = peer_bandwidth(outs[0], [o for o, d in zip(outs[1:], devices[1:]) if d != devices[0]], probe_cells)
                link_probe = {"from_device": devices[0], "payload_mb": probe_cells * 4 >> 20, "gbps_per_peer": [round(x, 1) for x in each], "gbps_all_peers_at_once": round(together, 1)}
            except Exception as e:   # noqa: BLE001
                link_probe = {"error": f"{type(e).__name__}: {e}"}
        elif pg is not None and world > 1:
            try:
                from mesh_to_sdf_amd import peer_bandwidth
                each, together = peer_bandwidth(pg.tensor, pg.peers, probe_cells)
                mine = {"rank": rank, "payload_mb": probe_cells * 4 >> 20, "gbps_per_peer": [round(x, 1) for x in each], "gbps_all_peers_at_once": round(together, 1)}
            except Exception as e:   # noqa: BLE001
                mine = {"rank": rank, "error": f"{type(e).__name__}: {e}"}
            link_probe = [None] * world
            dist.all_gather_object(link_probe, mine)

    # phase breakdown of one extra, untimed, synchronous one-shot call on this rank's whole slab
    ph = M2STimings()
    scratch_out = out if pg is None else torch.empty(n ** 3, dtype=torch.float32, device=dev)
    xs = slab_bounds(n, n_gpus, 0) if in_process else (x0, x1)
    generate_grid_sdf(dv, topo, grid, sign, x_slab=xs, out=scratch_out, timings=ph)
    del scratch_out
    world_label = n_gpus

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        voxels = n ** 3
        value = voxels * args.steps / elapsed / 1e6
        launches = int(sum(t.distance_launches for t in tims))
        dist_ms = float(sum(t.distance_ms for t in tims)) / max(launches, 1)   # average launch of the dominant kernel
        build_ms = float(np.mean([t.accel_build_ms for t in tims]))
        sign_ms, seed_ms, total_ms = float(ph.sign_ms), float(ph.seed_ms), float(ph.total_ms)
        n_tris = int(tims[0].n_triangles)
        slab_voxels = float(sum(t.n_units for t in tims)) / max(launches, 1)      # voxels one launch processes
        # algorithmic bytes of one launch of the dominant kernel:
        # 4 B per voxel written + the mesh read once (12 B per vertex + 12 B per triangle)
        b_alg = 4.0 * slab_voxels + 12.0 * v.shape[0] + 12.0 * n_tris
        achieved = b_alg / (dist_ms * 1e-3) / 1e9
        prof_ms = None
        prof_name = None
        traffic = None
        valu_frac = None
        pmc_src = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if world_label == 1 and launches == args.steps and not args.no_live_pmc:
            child = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc", "--grid", str(n), "--mesh", args.mesh, "--sign", args.sign]
            live, why_not = live_pmc(child)
            if live is not None and live.get("rocprof_avg_ms"):
                # the roofline is priced with the profiler's average launch duration; the event time stays beside it
                prof_ms = float(live["rocprof_avg_ms"])
                prof_name = live.get("rocprof_kernel_name")
                achieved = b_alg / (prof_ms * 1e-3) / 1e9
            if live is not None:
                # KiB -> bytes; FETCH_SIZE doubled: gfx950's rocprofv3 tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, HBM section)
                traffic = 2.0 * live["FETCH_SIZE"] * 1024.0 + live["WRITE_SIZE"] * 1024.0
                valu_frac = round(live["SQ_INSTS_VALU"] / (1024.0 * (live["GRBM_GUI_ACTIVE"] / 8.0) / 2.0), 3)
                pmc_src = (f"live: rocprofv3 --pmc child runs of this command after the timed region, one pass per counter group, averaged over "
                           f"{live['FETCH_SIZE_dispatches']} launches; FETCH_SIZE {live['FETCH_SIZE'] * 1024 / 1e6:.1f} MB as reported, doubled (gfx950 half-count); "
                           f"WRITE_SIZE {live['WRITE_SIZE'] * 1024 / 1e6:.1f} MB")
            elif why_not != "child run":
                pmc_src = f"live measurement unavailable ({why_not})"
        # fallback: the PMC figure of the last committed passes, per launch over the WHOLE 512^3 grid (1-GPU, 1-launch step only)
        if traffic is None and os.path.exists(pmc) and world_label == 1 and launches == args.steps and n == 512 and args.mesh == "blob-100k":
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("k_packet_hbm_bytes_per_launch")
                pmc_src = (pmc_src + "; " if pmc_src else "") + f"static: profiles/pmc_latest.json ({pj.get('source', 'rocprofv3 --pmc')}), not collected in this run"
                valu_frac = pj.get("valu_issue_frac")
            except Exception:
                traffic = None
        delivery = {"none": "single GPU, nothing to deliver",
                    "peer": f"each rank writes its slab into every peer's buffer over xGMI, m2s_opts.peer_out / {peer_mode.name}, IPC-mapped, then a barrier; no collective",
                    "rccl": f"{chunks} chunked in-place RCCL all-gathers overlapping the next chunk's compute",
                    "in-process": f"peer writes over xGMI ({peer_mode.name}) or RCCL where peers cannot map each other"}[exchange]
        res = {
            "metric": ("Mvoxels/s for generate_grid_sdf (512^3, 100k tris, Raycast)" if args.config == 0 else
                       f"Mvoxels/s for generate_grid_sdf (BASELINE config {args.config}: {n}^3, {args.mesh}, {args.sign}, one GPU)"),
            "value": round(value, 2),
            "unit": "Mvoxels/s",
            "n_gpus": world_label,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"generate_grid_sdf {n}^3 grid, {args.mesh} ({n_tris} tris, {v.shape[0]} verts), SignMethod::{args.sign}, "
                            f"contiguous x-slabs over {world_label} GPU(s), whole grid resident on every GPU at the end of a step "
                            f"(delivery: {delivery}), inputs/outputs resident in HBM",
                "grid": [n, n, n],
                "mesh": args.mesh,
                "sign_method": args.sign,
                "parallelism": f"xslab{world_label}",
                "processes": ("1 (m2s_generate_grid_sdf_multi: one host thread per device)" if in_process else f"{world} (one per GPU)"),
                "exchange": exchange,
                "exchange_fallback_reason": why,
                "gather_verified": verified,
                "gather_chunks": chunks,
                # rccl exchange: a rank's x-pieces run on two alternating streams, so consecutive launches of the dominant kernel
                # overlap and their individual durations (roofline.avg_launch_ms) are longer than when run alone
                "piece_streams": (int(os.environ.get("M2S_PIECE_STREAMS", "2")) if exchange == "rccl" and chunks > 1 else 1),
            },
            "phases_ms": {"accel_build": round(build_ms, 4), "sign_planes": round(sign_ms, 4), "seed_passes": round(seed_ms, 4),
                          "distance_per_launch": round(dist_ms, 4), "launches_per_step": launches // max(args.steps, 1),
                          "one_shot_device_total": round(total_ms, 4)},
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_packet<GRID, MODE_UNSIGNED, SIGN_GRID_PLANE>" if args.sign == "Raycast" else "k_packet<GRID, MODE_NORMAL_FOLD, SIGN_NONE>"),
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic,
                # "live: ..." = measured by rocprofv3 --pmc child runs of this invocation; "static: ..." = the last committed passes
                "traffic_source": pmc_src,
                "algorithmic_bytes_per_launch": b_alg,
                # avg_launch_ms: ALWAYS the HIP-event time of the dominant kernel's launches inside the timed region (same meaning in every
                # environment and in every round's JSON).  rocprof_avg_launch_ms: the rocprofv3 --kernel-trace average of that exact kernel
                # variant in a 3-launch child run of this command (null without rocprofv3).  `achieved` / `frac` are priced with the profiler's
                # figure when there is one (the slower, conservative one), else with the event time: achieved_basis says which.
                "avg_launch_ms": round(dist_ms, 4),
                "rocprof_avg_launch_ms": round(prof_ms, 4) if prof_ms is not None else None,
                "rocprof_kernel_name": prof_name,
                "achieved_basis": "rocprof_avg_launch_ms" if prof_ms is not None else "avg_launch_ms",
                "achieved_by_events": round(b_alg / (dist_ms * 1e-3) / 1e9, 3),
                # context, from the same PMC passes: the kernel's real ceiling is VALU issue
                "valu_issue_frac_pmc": valu_frac,
            },
        }
        if per_rank is not None:
            res["per_rank"] = per_rank
            res["config"]["exchange_ran"] = exchange_ran
            res["link_probe"] = link_probe
        if world_label == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"], rn, ref = cpu_baseline(v, idx, lo, hi, int(sign), args.cpu_seconds, n)
            # What a user switching from the reference sees (SURVEY.md header fact 2, §8c): this library's exact minimum
            # against the reference's propagation semantics (the CPU run above, kept) on the same grid.
            from mesh_to_sdf_amd.report import reference_parity
            rgrid = Grid.from_bounding_box(lo, hi, [rn, rn, rn])
            ours = generate_grid_sdf(dv, topo, rgrid, sign).cpu().numpy()
            res["reference_parity"] = {"grid": [rn, rn, rn], "mesh": args.mesh, "sign_method": args.sign,
                                       "reference": f"oracle PROPAGATE (generate/grid.rs:383-558 restated), {res['cpu_baseline']['cores']} heaps",
                                       **reference_parity(ours, ref, normal_sign=(args.sign == "Normal"))}
            del ours, ref
        if world_label == 1 and not args.no_cpu_baseline and args.config == 0:
            # the PCIe-inclusive drop-in call (host pointers in/out), reported for DESIGN.md; never `value`
            host_out = np.empty(n ** 3, np.float32)
            times = []
            for _ in range(3):   # the first call also allocates the pinned ring and faults the output pages in
                t1 = time.perf_counter()
                generate_grid_sdf(v, Topology.TriangleList(idx), grid, sign, out=host_out)
                times.append((time.perf_counter() - t1) * 1e3)
            res["host_pointer_call_ms"] = round(min(times[1:]), 2)
            res["host_pointer_first_call_ms"] = round(times[0], 2)
        emit(json.dumps(res))
    if pg is not None:
        pg.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

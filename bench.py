#!/usr/bin/env python3
"""bench.py -- segmented points/sec of the end-to-end VGS hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1 launched through torch.distributed.run,
one rank per GPU).  A step = one full pass of the hot path (voxelize -> features -> adjacency -> local
cuts -> merge -> per-point labels) over one synthetic scene that is already resident in HBM.
N = 1 workload: BASELINE.json configs[2] "URB10M" (10 M points, voxel 0.1 m, Task_File_VGS defaults) --
the configuration the metric is quoted on.  N > 1: weak scaling, every rank owns one 10 M-point tile of
the URB80M layout (configs[4]); tiles are segmented on one shared grid and boundary segments are merged
with one all-gather of boundary records (vgs-svgs-segmentation_amd/dist.py).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def cpu_baseline(v, xyz_full, params, target_points=150_000):
    """The oracle in the reference's own arithmetic and data flow (RefMath, faithful: by-value vectors,
    n x n matrix, std::sort of n^2 weights), one thread, on a bounded spatial crop of the same scene."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import refcpu_py as R

    # crop: a square window centred on the strongest facade column of the scene (most points above 1 m in a
    # 0.5 m x 0.5 m cell), grown until it holds about target_points points: ground + facade, the mix the scene is made of
    hi = xyz_full[:, 2] > 1.0
    nb = max(8, int((xyz_full[:, 0].max() - xyz_full[:, 0].min()) / 0.5))
    H, xe, ye = np.histogram2d(xyz_full[hi, 0], xyz_full[hi, 1], bins=nb)
    i, j = np.unravel_index(np.argmax(H), H.shape)
    x0, y0 = 0.5 * (xe[i] + xe[i + 1]), 0.5 * (ye[j] + ye[j + 1])
    a = 0.5
    while True:
        m = (np.abs(xyz_full[:, 0] - x0) < a) & (np.abs(xyz_full[:, 1] - y0) < a)
        if m.sum() >= target_points or a > 30:
            break
        a *= 1.1
    sample = np.ascontiguousarray(xyz_full[m])
    rp = R.vgs_params(voxel_size=params.voxel_size, graph_size=params.graph_size, sig_p=params.sig_p, sig_n=params.sig_n,
                      sig_o=params.sig_o, sig_e=params.sig_e, sig_c=params.sig_c, sig_w=params.sig_w, cut_thred=params.cut_thred,
                      points_min=params.points_min, adjacency_min=params.adjacency_min, voxels_min=params.voxels_min,
                      math=0, flavour=0)
    t = time.perf_counter()
    res = R.run_vgs(sample, rp)
    dt = time.perf_counter() - t
    return {"value": sample.shape[0] / dt, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": f"{sample.shape[0]} points: {2 * a:.1f} m x {2 * a:.1f} m crop of the same scene, oracle RefMath + faithful "
                      f"flavour (n x n matrix, by-value vectors, std::sort), {dt:.1f} s, {res.pair_evals} pair evaluations"}


def profiled_traffic(n_points):
    """HBM bytes (and VALU wave instructions) per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r02_traffic.json,
    written by tools/collect_profiles.sh from FETCH_SIZE + WRITE_SIZE of the same bench command); None if the profile
    is missing or was taken on another workload."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for tag in ("r02", "r01"):   # the newest round's profile that exists
        try:
            with open(os.path.join(prof, f"{tag}_traffic.json")) as f:
                t = json.load(f)
            if int(t.get("points", -1)) != int(n_points):
                continue
            return float(t["hbm_bytes_per_launch"]), t.get("valu_wave_instructions_per_launch")
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=10_000_000, help="points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    import vgs_svgs_segmentation_amd as v

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test-only overrides (a 1-GPU box cannot run RCCL with 2 ranks): VGS_BENCH_BACKEND=gloo VGS_BENCH_SINGLE_DEVICE=1
    backend = os.environ.get("VGS_BENCH_BACKEND", "nccl")
    if os.environ.get("VGS_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    p = v.default_params(2, voxel_size=0.1, device=local_rank)
    n_per = args.points
    if world == 1:
        xyz = v.scenes.urban_scene(n_per) if n_per == 10_000_000 else v.scenes.urban_scene(n_per, nominal=n_per)
        workload = f"URB10M: {n_per} pts synthetic urban scene, VGS, voxel 0.1 m, graph 0.5 m, Task_File_VGS defaults"
    else:
        tiles = {2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(world, (world, 1))
        xyz = v.scenes.tiled_urban_scene(n_per * world, tiles=tiles, tile_index=rank)
        workload = f"URB{n_per * world // 1_000_000}M: {tiles[0]}x{tiles[1]} tiles of {n_per} pts, VGS, voxel 0.1 m, one tile per GPU"

    d_xyz = torch.from_numpy(xyz).to(dev)  # inputs resident in HBM before the timed region
    torch.cuda.synchronize(dev)

    if world == 1:
        eng = v.Engine(p)
        eng.set_points_device(d_xyz.data_ptr(), d_xyz.shape[0], 12, keep=d_xyz)

        def step():
            eng.run()
        runner = eng
    else:
        from vgs_svgs_segmentation_amd.dist import TiledSegmenter
        seg = TiledSegmenter(p, dist, tiles=tiles, rank=rank, world=world, pitch=50.0 * (n_per / 10_000_000) ** 0.5)
        seg.set_points_device(d_xyz, xyz)

        def step():
            seg.run()
        runner = seg.engine

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kern_ms = []
    stage_acc = {}
    for _ in range(args.steps):
        step()
        st = runner.stage_times()
        kern_ms.append(st["localcut_bulk"])
        for k, val in st.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + val
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        c = runner.counts()
        N, V, E = c["points"], c["voxels"], c["adj"]
        total_points = n_per * world * args.steps
        # algorithmic bytes of one pass: 28 B/point + 48 B/voxel + 4 B/adjacency entry
        # The dominant kernel is the bulk class of the local cut, one launch per step over class_a of the used voxels:
        # its share of the run's algorithmic bytes over its own duration (HIP events on its stream, VGS_T_LOCALCUT_BULK).
        alg_run = 28 * N + 48 * V + 4 * E
        alg_bytes = int(alg_run * (c["class_a"] / max(c["used"], 1)))
        k_avg_ms = sum(kern_ms) / max(len(kern_ms), 1)
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        traffic, valu = profiled_traffic(N) if world == 1 else (None, None)
        # the kernel is VALU-issue bound: wave64 instructions take 4 cycles on the 1024 16-lane SIMDs (2.4 GHz peak clock)
        valu_frac = (valu * 4.0 / 1024.0 / 2.4e9) / (k_avg_ms * 1e-3) if (valu and k_avg_ms > 0) else None
        out = {
            "metric": "segmented points/sec (end-to-end VGS)",
            "value": total_points / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "points_per_gpu": n_per, "voxels": V, "used_voxels": c["used"], "adjacency_entries": E,
                       "segments": c["kept"], "pair_evaluations": c["pairs"],
                       "parallelism": "single GPU" if world == 1 else f"{world} spatial tiles, shared grid, one all-gather of boundary labels"},
            "roofline": {"bound": "hbm", "kernel": "k_localcut_wave<96,448,1> (local affinity graph + threshold-merge cut, bulk class)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_avg_ms,
                         "valu_issue_frac": valu_frac,
                         "algorithmic_bytes_per_step": alg_run,
                         "end_to_end_frac": alg_run / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "pair_evals_per_s": c["pairs"] / (k_avg_ms * 1e-3) if k_avg_ms > 0 else 0.0},
            "stage_ms": {k: val / args.steps for k, val in stage_acc.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(v, xyz, p)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

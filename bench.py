#!/usr/bin/env python3
"""bench.py -- segmented points/sec of the end-to-end VGS hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  A step = one full pass of the hot path (voxelize -> features
-> adjacency -> local cuts -> merge -> per-point labels) over one synthetic scene.

N = 1 workload: BASELINE.json configs[2] "URB10M" (10 M points, voxel 0.1 m, Task_File_VGS defaults) -- the
configuration the metric is quoted on.  Two timed regions, both K steps after W warm-up steps:
  * `value` / `ms_per_step`: inputs resident in HBM when the region starts, labels left in HBM (the task contract's
    definition of `value`);
  * `host_to_host`: SURVEY.md 8d / BASELINE.md 2 -- float32 xyz in HOST memory in, int32 labels in HOST memory out, H2D and
    D2H inside the region.  The clouds come as a sequence: cloud k+1's upload and cloud k-1's label download run on copy
    streams beside cloud k's stages (vgs_stage_points / vgs_commit_points / vgs_get_point_labels_async, include/vgs.h);
    `latency_ms_median` is one cloud on its own (set_points -> run -> labels, nothing overlapped, median of 5).
N > 1: weak scaling, every rank owns one 10 M-point tile of the URB80M layout (configs[4]); tiles are segmented on one
shared grid and boundary segments are merged with one all-gather of boundary records.  The driver is the NATIVE one
(include/vgs_tiles.h = csrc/tiles.cpp, libvgs_tiles.so, through vgs-svgs-segmentation_amd/tiles_native.py): every rank creates
its own ncclComm_t (ncclGetUniqueId on rank 0, the id broadcast through the process group that also carries the barrier and
the max-over-ranks reduction), hands it to vgs_tiles_create(VGS_TILES_COMM_RCCL, ...) and times vgs_tiles_run; the line carries
the communicator's own rank count (ncclCommCount), per-rank stage times and the exchange time.  `--python-twin` runs the
test harness instead (vgs-svgs-segmentation_amd/dist.py, torch.distributed collectives) -- same protocol, same labels.
Launched by the driver through torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment); when it is NOT
(plain `python bench.py --gpus N`), this script starts that launcher itself as a child process BEFORE anything touches
the GPU, passes its output through and exits with its status.
Prints ONE JSON line on rank 0.

`--dry-run` (CPU, no GPU, gloo): every rank joins the process group, builds its tile of the scene and takes part in the
same collectives with stand-in payloads; the printed line carries n_gpus and "dry_run": true.  It checks the launch path.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
# VALU issue rates measured on this pool with tools/valu_roof.hip (hand-placed registers; profiles/r04_valu_roof.txt): a SIMD issues
# one wave64 VALU instruction per 2.2 cycles for the full-rate classes, per 4.1 cycles for the half-rate ones (every compare, select,
# DPP form, v_readlane, shift, bit-field op, integer multiply, conversion, any VOP3 with an SGPR source or two sources in one VGPR bank)
# and per 8.1 cycles for transcendentals.  The counters do not separate the classes, so ONE figure is made of the two things that
# exist (tools/valu_mix.py -> profiles/rNN_valu_mix.json): the static full/half/quarter mix of each phase of the bulk kernel's ISA x
# the dynamic SQ_INSTS_VALU of that phase (tools/pmc_phases.sh) = mean issue cycles per instruction; valu_issue_frac = instructions
# per launch x that mean / 2.39 GHz / 1024 SIMDs / the kernel's duration of THIS run.
VALU_CLOCK_GHZ = 2.39
N_SIMD = 1024


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start torch.distributed.run as a CHILD (never exec, and
    before any HIP call of this process), one rank per GPU, and hand its exit status back."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def crop_tile(xyz, frac, work_per_point=None):
    """A contiguous tile that holds `frac` of the points: a strip across the whole scene in y (it crosses every kind of lot:
    buildings, poles, trees, open ground), 1/frac candidate positions in x.  With a per-point work measure the strip whose share
    of the work is closest to its share of the points is taken (a representative tile), otherwise the middle one."""
    import numpy as np
    k = max(int(round(1.0 / frac)), 1)
    edges = np.quantile(xyz[:, 0], np.linspace(0.0, 1.0, k + 1))
    which = np.clip(np.searchsorted(edges, xyz[:, 0], side="right") - 1, 0, k - 1)
    pick = k // 2
    if work_per_point is not None:
        share = np.bincount(which, weights=work_per_point, minlength=k) / max(float(work_per_point.sum()), 1e-30)
        cnt = np.bincount(which, minlength=k) / float(xyz.shape[0])
        pick = int(np.argmin(np.abs(share / np.maximum(cnt, 1e-30) - 1.0)))
    m = which == pick
    return np.ascontiguousarray(xyz[m]), float(edges[pick]), float(edges[pick + 1])


def cpu_baseline(xyz_full, params, n_all_full, point_voxel, frac=0.05):
    """The oracle in the reference's own arithmetic and data flow (RefMath, faithful: by-value vectors, n x n matrix,
    std::sort of n^2 weights), one thread, on a contiguous tile with >= 5 % of the scene, extrapolated to the scene by the
    work that dominates it, sum of n_i^2 over the used voxels; the lean flavour (same results, unique pairs,
    no per-pair allocations) runs on the same tile as the context row of BASELINE.md 2."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import refcpu_py as R

    # work measure per point: n^2 of its voxel, shared by the voxel's points (n = neighbours inside graph_size, from the GPU run)
    n2 = n_all_full.astype(np.float64) ** 2
    pts_in_vox = np.maximum(np.bincount(point_voxel[point_voxel >= 0], minlength=n2.size), 1)
    wpp = np.where(point_voxel >= 0, (n2 / pts_in_vox)[np.maximum(point_voxel, 0)], 0.0)
    sample, x_lo, x_hi = crop_tile(xyz_full, frac, wpp)
    kw = dict(voxel_size=params.voxel_size, graph_size=params.graph_size, sig_p=params.sig_p, sig_n=params.sig_n,
              sig_o=params.sig_o, sig_e=params.sig_e, sig_c=params.sig_c, sig_w=params.sig_w, cut_thred=params.cut_thred,
              points_min=params.points_min, adjacency_min=params.adjacency_min, voxels_min=params.voxels_min)
    out = {}
    sum_n2_full = float(n2.sum())
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for name, math, flavour, threads in (("faithful", 0, 0, 1), ("lean", 0, 1, 1), ("lean_all", 0, 1, cores)):
        t = time.perf_counter()
        res = R.run_vgs(sample, R.vgs_params(math=math, flavour=flavour, threads=threads, **kw))
        dt = time.perf_counter() - t
        off, _ = res.lists("adjacency")
        n_i = np.diff(off).astype(np.float64) * (res.nodes()["used"] != 0)   # the local graphs are built for the used voxels (VS:384)
        sum_n2 = float((n_i ** 2).sum())
        scale = sum_n2_full / max(sum_n2, 1.0)
        out[name] = dict(dt=dt, points=int(sample.shape[0]), voxels=int(res.V), used=int((n_i > 0).sum()), sum_n2=sum_n2, scale=scale,
                         scene_seconds=dt * scale, pair_evals=int(res.pair_evals))
    f, l, la = out["faithful"], out["lean"], out["lean_all"]
    n_full = xyz_full.shape[0]
    return {"value": n_full / max(f["scene_seconds"], 1e-9), "unit": "points/s", "cores": 1, "kind": "port",
            "sample": (f"{f['points']} points ({100.0 * f['points'] / n_full:.1f} % of the scene, {f['used']} used voxels): contiguous strip "
                       f"x in [{x_lo:.2f}, {x_hi:.2f}) m across the whole scene (of {int(round(1 / frac))} such strips the one whose share of sum n_i^2 "
                       f"is closest to its share of the points), oracle RefMath + faithful flavour (n x n matrix, by-value vectors, std::sort) in "
                       f"{f['dt']:.1f} s = {f['points'] / f['dt']:.0f} points/s on the tile; whole scene extrapolated by sum n_i^2 over the used "
                       f"voxels (scene {sum_n2_full:.4g} from the GPU run / tile {f['sum_n2']:.4g} = x{f['scale']:.2f}; by points it would be "
                       f"x{n_full / f['points']:.2f}) -> {f['scene_seconds']:.0f} s"),
            "sample_value": f["points"] / f["dt"],
            "cpu_lean": {"value": n_full / max(l["scene_seconds"], 1e-9), "unit": "points/s", "cores": 1,
                         "sample": f"same tile, lean flavour (unique pairs, no per-pair allocations), {l['dt']:.1f} s, same extrapolation"},
            "cpu_lean_all_cores": {"value": n_full / max(la["scene_seconds"], 1e-9), "unit": "points/s", "cores": cores,
                                   "sample": f"same tile, lean flavour with the nodes' local cuts on {cores} threads (OpenMP; voxelisation and radius search "
                                             f"stay on one), {la['dt']:.1f} s, same extrapolation"}}


def profiled_traffic(n_points):
    """HBM bytes (and VALU wave instructions) per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_traffic.json, written by tools/collect_round.sh from FETCH_SIZE + WRITE_SIZE of the same bench command), the
    commit they were taken at, and the kernel's mean issue cycles per VALU instruction (profiles/rNN_valu_mix.json);
    None if a profile is missing or was taken on another workload."""
    prof = os.path.join(ROOT, "profiles")
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):   # the newest round's profile that exists
        try:
            with open(os.path.join(prof, f"{tag}_traffic.json")) as f:
                t = json.load(f)
            if int(t.get("points", -1)) != int(n_points):
                continue
            mix = None
            try:
                with open(os.path.join(prof, f"{tag}_valu_mix.json")) as f:
                    mix = json.load(f)
            except (OSError, ValueError):
                pass
            return {"bytes": float(t["hbm_bytes_per_launch"]), "valu": t.get("valu_wave_instructions_per_launch"), "source": f"profiles/{tag}_traffic.json",
                    "commit": t.get("commit"), "cycles_per_instr": mix.get("mean_cycles_per_instruction") if mix else None,
                    "mix_source": f"profiles/{tag}_valu_mix.json" if mix else None, "mix_commit": mix.get("commit") if mix else None}
        except (OSError, ValueError, KeyError):
            continue
    return None


def dry_run(args, world, rank):
    """No GPU: the launch path, the rank environment and the collectives of the tiled driver, with stand-in payloads."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from importlib import import_module
    scenes = import_module("vgs_svgs_segmentation_amd").scenes
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if os.environ.get("VGS_BENCH_FAIL_RANK") == str(rank):   # tests: a failing rank must fail the whole launch
        raise RuntimeError(f"rank {rank}: failure requested by VGS_BENCH_FAIL_RANK")
    tiles = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(world, (world, 1))
    n_per = args.points
    xyz = scenes.tiled_urban_scene(n_per * world, tiles=tiles, tile_index=rank) if world > 1 else scenes.urban_scene(n_per)
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        if world > 1:
            # the collectives of one tiled step: bounding boxes, boundary records, (barrier)
            bb = torch.tensor([float(v) for v in (*xyz.min(0), *xyz.max(0))], dtype=torch.float64)
            allbb = [torch.zeros_like(bb) for _ in range(world)]
            dist.all_gather(allbb, bb)
            rec = torch.full((16,), rank, dtype=torch.int64)
            allrec = [torch.zeros_like(rec) for _ in range(world)]
            dist.all_gather(allrec, rec)
            assert [int(r[0]) for r in allrec] == list(range(world))
    native_kept = None
    if (args.native or world > 1) and not args.python_twin:
        # the native driver's side of the exchange that needs no GPU: libvgs_tiles.so loads, and its boundary union-find
        # (vgs_tiles_merge_boundary) runs on stand-in records gathered over the process group -- every rank must get the same tables
        import ctypes as C
        from importlib import import_module
        tn = import_module("vgs_svgs_segmentation_amd.tiles_native")
        L = tn.lib()
        m = 16
        mine = np.zeros(2 + 3 * m, dtype=np.int64)
        mine[0], mine[1] = m, 5 + rank                                   # records, purely local segments
        mine[2:2 + m] = 1000 * (np.arange(m) // 4) + np.arange(m) % 4     # codes shared by every rank: the segments cross every border
        mine[2 + m:2 + 2 * m] = 7 * (np.arange(m) // 4) + rank            # four local roots
        mine[2 + 2 * m:] = 2 + (np.arange(m) // 4)                        # owned voxels of each root
        allr = [torch.zeros(mine.size, dtype=torch.int64) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, torch.from_numpy(mine))
        else:
            allr = [torch.from_numpy(mine)]
        recs = [a.numpy() for a in allr]
        off = np.arange(world + 1, dtype=np.int64) * m
        code = np.concatenate([r[2:2 + m] for r in recs]).astype(np.uint64)
        root = np.concatenate([r[2 + m:2 + 2 * m] for r in recs]).astype(np.int32)
        cnt = np.concatenate([r[2 + 2 * m:] for r in recs]).astype(np.int32)
        kl = np.array([r[1] for r in recs], dtype=np.int64)
        base, uoff = np.zeros(world, np.int64), np.zeros(world + 1, np.int64)
        uroot, ulabel = np.zeros(world * m, np.int32), np.zeros(world * m, np.int32)
        kept = C.c_int64(0)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        st = L.vgs_tiles_merge_boundary(world, P(off), P(code), P(root), P(cnt), P(kl), 1, P(base), P(uoff), P(uroot), P(ulabel), C.byref(kept))
        assert st == 0, st
        native_kept = int(kept.value)
        assert native_kept == int(kl.sum()) + 4, (native_kept, kl)      # four segments span all ranks
        if world > 1:
            ks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(ks, torch.tensor([native_kept], dtype=torch.int64))
            assert len({int(k.item()) for k in ks}) == 1, ks
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "segmented points/sec (end-to-end VGS)", "value": None, "unit": "points/s", "n_gpus": world, "ranks": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "backend": "gloo",
                          "driver": "native (libvgs_tiles.so: boundary merge on stand-in records)" if native_kept is not None else "python twin",
                          "native_merge_kept": native_kept,
                          "config": {"workload": f"dry run: {tiles[0]}x{tiles[1]} tiles of {xyz.shape[0]} pts", "points_per_gpu": n_per}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=10_000_000, help="points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--no-clusters", action="store_true", help="skip the getClusterIdx timing (extras, outside the metric)")
    ap.add_argument("--clusters-reference", action="store_true", help="also time getClusterIdx in the reference's own element order")
    ap.add_argument("--dry-run", action="store_true", help="CPU only: exercise the launch path and the collectives (gloo), no engine")
    ap.add_argument("--native", action="store_true", help="tiled runs through libvgs_tiles.so (the default for --gpus > 1; with --gpus 1: a one-rank communicator)")
    ap.add_argument("--python-twin", action="store_true", help="tiled runs through the Python twin of the native driver (dist.py)")
    args = ap.parse_args()

    have_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not have_launcher:
        sys.exit(spawn_ranks(args))     # nothing in this process has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        dry_run(args, world, rank)
        return

    import numpy as np
    import torch

    import vgs_svgs_segmentation_amd as v

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
        xyz = v.scenes.urban_scene(n_per)    # other sizes: the same density on a smaller ground (extents scale with sqrt(n))
        workload = f"URB10M: {n_per} pts synthetic urban scene, VGS, voxel 0.1 m, graph 0.5 m, Task_File_VGS defaults"
    else:
        tiles = {2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(world, (world, 1))
        xyz = v.scenes.tiled_urban_scene(n_per * world, tiles=tiles, tile_index=rank)
        workload = f"URB{n_per * world // 1_000_000}M: {tiles[0]}x{tiles[1]} tiles of {n_per} pts, VGS, voxel 0.1 m, one tile per GPU"

    native = (args.native or world > 1) and not args.python_twin
    d_xyz = None
    if not native:
        d_xyz = torch.from_numpy(xyz).to(dev)  # inputs resident in HBM before the timed region
    torch.cuda.synchronize(dev)

    tiles_drv, rccl_comm, rccl_ranks = None, None, None
    if native:
        from vgs_svgs_segmentation_amd import tiles_native as tn
        tiles = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}.get(world, (world, 1))
        pitch = 50.0 * (n_per / 10_000_000) ** 0.5
        if backend == "nccl":
            # every rank creates its own communicator for the driver: the id comes from rank 0 through the process group
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(tn.rccl_unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(uid, src=0)
            rccl_comm = tn.RcclComm(bytes(uid.cpu().numpy().tobytes()), rank, world, local_rank)
            rccl_ranks = rccl_comm.count()
            tiles_drv = tn.NativeTiles(p, tn.COMM_RCCL, rccl_comm.handle, rank, world, tiles, pitch, keep=rccl_comm)
        else:
            # test harness (two processes on ONE GPU, which RCCL refuses): the driver's collectives over the caller's gloo group
            def _ag(send, recv):
                t = torch.from_numpy(np.array(send, copy=True))
                outs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(outs, t)
                torch.from_numpy(recv).copy_(torch.cat(outs))

            def _bc(buf, root):
                dist.broadcast(torch.from_numpy(buf), src=root)

            tiles_drv = tn.NativeTiles.with_callbacks(p, rank, world, tiles, pitch, _ag, _bc)
        tiles_drv.set_points(xyz)       # tile + halo uploaded once: inputs resident in HBM before the timed region

        def step():
            tiles_drv.run()
        runner = tiles_drv
    elif world == 1:
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
    kern_ms, step_ms = [], []
    stage_acc, tiles_acc = {}, {}
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append((time.perf_counter() - ts) * 1e3)   # a step ends with a wait for its last kernel (the stage's read-back)
        st = runner.stage_times()
        kern_ms.append(st["localcut_bulk"])
        for k, val in st.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + val
        if tiles_drv is not None:
            for k, val in tiles_drv.times().items():
                tiles_acc[k] = tiles_acc.get(k, 0.0) + val
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    per_rank = None
    if tiles_drv is not None:
        _free, _total = torch.cuda.mem_get_info(dev)   # device memory in use on this rank's GPU behind the timed steps: tile + halo, every grow-only table
        mine = {"rank": rank, "device": local_rank, "points": int(xyz.shape[0]), "hbm_in_use_gb": (_total - _free) / 1e9, "hbm_total_gb": _total / 1e9, **tiles_drv.info(),
                "stage_ms": {k: val / args.steps for k, val in stage_acc.items()}, "tiles_ms": {k: val / args.steps for k, val in tiles_acc.items()}}
        per_rank = [mine]
        if dist is not None:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)

    # ---- host -> host (single GPU): pinned staging buffers, uploads and downloads beside the stages -----------------
    h2h = None
    if world == 1 and not native and not args.no_host_to_host:
        hx = v.pinned_empty(xyz.shape, np.float32)        # the caller's cloud in host memory (a PCD reader's buffer)
        hx[...] = xyz
        hl = [v.pinned_empty((xyz.shape[0],), np.int32) for _ in range(2)]
        eng2 = v.Engine(p)

        def h2h_sequence(k_steps):
            eng2.stage_points(hx)
            for k in range(k_steps):
                eng2.commit_points()
                if k + 1 < k_steps:
                    eng2.stage_points(hx)             # the next cloud's upload runs beside this cloud's stages
                eng2.run()
                eng2.point_labels_async(hl[k & 1])    # this cloud's labels go out beside the next cloud's stages
            eng2.wait_labels()

        h2h_sequence(max(args.warmup, 1))
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        h2h_sequence(args.steps)
        h2h_elapsed = time.perf_counter() - t1
        labels_dev = eng.point_labels()
        same = bool(np.array_equal(hl[(args.steps - 1) & 1], labels_dev))
        # one cloud on its own: nothing overlapped
        lat = []
        for _ in range(5 + 1):
            ts = time.perf_counter()
            eng2.set_points(hx)
            eng2.run()
            eng2.point_labels_async(hl[0])
            eng2.wait_labels()
            lat.append((time.perf_counter() - ts) * 1e3)
        lat = sorted(lat[1:])
        eng2.close()
        # the same sequence dealt to TWO contexts (one host thread each, their own streams and staging buffers): the latency-bound
        # tail of one cloud's step (hand-over kernels, crossValidation, union-find) runs beside the bulk of the other cloud's
        import threading
        eng3 = v.Engine(p)
        hl3 = [v.pinned_empty((xyz.shape[0],), np.int32) for _ in range(2)]

        def seq_on(e, bufs, k_steps):
            if k_steps <= 0:
                return
            e.stage_points(hx)
            for k in range(k_steps):
                e.commit_points()
                if k + 1 < k_steps:
                    e.stage_points(hx)
                e.run()
                e.point_labels_async(bufs[k & 1])
            e.wait_labels()

        thread_errors = []

        def guarded(bar, e, b, k):
            bar.wait()
            try:
                seq_on(e, b, k)
            except Exception as ex:  # noqa: BLE001  (reported below: a failed leg must not print a number)
                thread_errors.append(ex)

        def two_contexts(k_steps):
            ka, kb = (k_steps + 1) // 2, k_steps // 2
            bar = threading.Barrier(3)
            th = [threading.Thread(target=guarded, args=(bar, e, b, k)) for e, b, k in ((eng2, hl, ka), (eng3, hl3, kb))]
            for t in th:
                t.start()
            bar.wait()
            t_start = time.perf_counter()
            for t in th:
                t.join()
            return time.perf_counter() - t_start

        eng2 = v.Engine(p)                      # (the first one was closed above)
        two_contexts(2)
        torch.cuda.synchronize(dev)
        k2 = max(args.steps, 2)
        two_elapsed = two_contexts(k2)
        if thread_errors:
            raise thread_errors[0]
        two_same = bool(np.array_equal(hl3[((k2 // 2) - 1) & 1], labels_dev)) if k2 // 2 > 0 else True
        eng2.close(); eng3.close()
        h2h = {"value": n_per * args.steps / h2h_elapsed, "unit": "points/s", "ms_per_step": h2h_elapsed / args.steps * 1e3,
               "two_contexts": {"value": n_per * k2 / two_elapsed, "unit": "points/s", "ms_per_step": two_elapsed / k2 * 1e3, "steps": k2,
                                "labels_equal_device_resident_run": two_same,
                                "mode": "the same host-to-host sequence dealt to two contexts (two host threads, own streams): two clouds in flight"},
               "steps": args.steps, "mode": "sequence of clouds: pinned host xyz in -> pinned host labels out, cloud k+1's H2D and cloud k-1's D2H "
                                            "beside cloud k's stages (the first upload and the last download are inside the timed region)",
               "latency_ms_median": lat[len(lat) // 2], "latency_note": "one cloud, nothing overlapped: set_points + run + labels to host, median of 5 after 1",
               "labels_equal_device_resident_run": same}

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
        prof = profiled_traffic(N) if (world == 1 and not native) else None
        traffic, traffic_src = (prof["bytes"], prof["source"]) if prof else (None, None)
        # the kernel works on-chip (VALU issue + LDS latency): its VALU wave instructions at the issue cost of ITS OWN instruction mix
        valu_frac = None
        if prof and prof["valu"] and prof["cycles_per_instr"] and k_avg_ms > 0:
            valu_frac = (prof["valu"] * prof["cycles_per_instr"] / (VALU_CLOCK_GHZ * 1e9) / N_SIMD) / (k_avg_ms * 1e-3)
        sm = sorted(step_ms)
        out = {
            "metric": "segmented points/sec (end-to-end VGS)",
            "value": total_points / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "ranks": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median": sm[len(sm) // 2],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "value_definition": "inputs resident in HBM, labels left in HBM (task contract); host_to_host is the SURVEY 8d metric",
            "config": {"workload": workload, "points_per_gpu": n_per, "voxels": V, "used_voxels": c["used"], "adjacency_entries": E,
                       "segments": c["kept"], "pair_evaluations": c["pairs"],
                       "parallelism": "single GPU" if world == 1 else f"{world} spatial tiles, shared grid, one all-gather of boundary labels ({'RCCL, native driver' if rccl_comm is not None else backend})"},
            "roofline": {"bound": "hbm", "kernel": "k_localcut_wave<96,448,1,false,true> (local affinity graph + threshold-merge cut, bulk class, one-word sort keys)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_avg_ms,
                         "traffic_commit": prof["commit"] if prof else None,
                         "valu_issue_frac": valu_frac, "valu_cycles_per_wave_instr": prof["cycles_per_instr"] if prof else None,
                         "valu_mix_source": prof["mix_source"] if prof else None, "valu_mix_commit": prof["mix_commit"] if prof else None,
                         "algorithmic_bytes_per_step": alg_run,
                         "end_to_end_frac": alg_run / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "pair_evals_per_s": c["pairs"] / (k_avg_ms * 1e-3) if k_avg_ms > 0 else 0.0},
            "stage_ms": {k: val / args.steps for k, val in stage_acc.items()},
        }
        if tiles_drv is not None:
            out["driver"] = {"kind": "native: libvgs_tiles.so (csrc/tiles.cpp) through ctypes",
                             "communicator": "RCCL: ncclComm_t created by this process (ncclGetUniqueId / ncclCommInitRank), ncclAllGather / ncclBroadcast inside the driver"
                                             if rccl_comm is not None else f"caller's callbacks over torch.distributed/{backend} (test harness: ranks share a GPU)",
                             "rccl_ranks": rccl_ranks, "exchange_ms": max(r["tiles_ms"]["exchange"] for r in per_rank),
                             "grid_ms": max(r["tiles_ms"]["grid"] for r in per_rank), "per_rank": per_rank}
            out["ranks"] = rccl_ranks if rccl_ranks is not None else world
        elif world > 1:
            out["driver"] = {"kind": "python twin: vgs-svgs-segmentation_amd/dist.py (test harness)"}
        if world == 1 and not native and not args.no_clusters:
            # getClusterIdx, the reference's end product (voxel_segmentation.h:117; test:74-76) -- outside the metric, beside it:
            # the lists made on the device and left in HBM (csrc/clusters.hip), the same copied to host memory (vgs_get_clusters),
            # and, on request, the reference's own element order (host walk over lists the device compacts: csrc/cutorder.hip)
            cl = {}
            runner.clusters_device()           # (warm-up: the list buffers are grow-only allocations of the context, like every table)
            runner.run()                       # fresh labels: the lists are cached per run
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter(); runner.clusters_device(); cl["device_ms"] = (time.perf_counter() - t1) * 1e3
            t1 = time.perf_counter(); off, idx = runner.clusters("voxel_id"); cl["to_host_ms"] = (time.perf_counter() - t1) * 1e3
            cl["clusters"], cl["points_listed"] = int(off.shape[0] - 1), int(off[-1])
            cl["order"] = "clusters by ascending smallest voxel id, inside a cluster ascending voxel id then point index"
            if args.clusters_reference:
                t1 = time.perf_counter(); off_r, idx_r = runner.clusters("reference"); cl["reference_order_to_host_ms"] = (time.perf_counter() - t1) * 1e3
                cl["reference_order_same_sets"] = bool(np.array_equal(off, off_r))
            out["clusters_ms"] = cl
        if h2h is not None:
            out["host_to_host"] = h2h
        if world == 1 and not native and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(xyz, p, runner.adjacency_counts(), runner.point_voxel())
        print(json.dumps(out))
    if tiles_drv is not None:
        tiles_drv.close()
    if rccl_comm is not None:
        rccl_comm.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

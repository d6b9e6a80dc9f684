#!/usr/bin/env python3
"""Randomised differential test, GPU path against the oracle (DevMath + lean): random scenes, sizes and parameters for a time
budget; every run compares the connect lists after the local cut and the point labels.

usage: fuzz_parity.py [seconds] [seed] [method 2|3] [wide] [--case K] [--threads T] [--dump DIR]

* the case generator (`cases`) is a pure function of (seed, method, wide): `--case K` replays exactly the K-th case of a campaign
  (1-based, non-finite and repeated points included) and nothing else, `tools/fuzz_find.py` finds K for a logged `start` line;
* every case prints its wall time (engine / oracle), and the campaign ALWAYS ends with a summary line -- also when its time budget
  or a SIGTERM from `timeout` cuts it: the case in flight is then reported as `unfinished`, not silently lost (round 3 left four
  campaigns without a verdict that way);
* the oracle's local cuts run on `--threads` cores (default: all; identical results, tests/test_oracle_kat.py)."""
import os, signal, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np


def grid_supervoxels(xyz, seed_size, r):
    cell = np.floor(xyz.astype(np.float64) / seed_size).astype(np.int64)
    cell -= cell.min(0)
    code = (cell[:, 0] * 4096 + cell[:, 1]) * 4096 + cell[:, 2]
    _, inv = np.unique(code, return_inverse=True)
    labels = (inv + 1).astype(np.int32)
    labels[r.random(labels.size) < 0.01] = 0
    return labels, int(labels.max())


def fuzzy(n, seed, sigma):
    r = np.random.default_rng(seed)
    side = np.sqrt(n / 6000.0)
    x, y = r.random(n) * side, r.random(n) * side
    z = 0.3 * np.sin(2.0 * x) * np.cos(1.5 * y) + r.normal(0, sigma, n) + 2.0
    return np.stack([x + 0.011, y + 0.017, z], axis=1).astype(np.float32)


def slab(n, seed, thick):
    r = np.random.default_rng(seed)
    side = np.sqrt(n / (2500.0 * max(thick / 0.1, 0.3)))
    x = r.random(n) * side + 0.013; y = r.random(n) * side + 0.027
    z = r.random(n) * thick + 0.02 * np.sin(3.0 * x) + 1.0
    return np.stack([x, y, z], axis=1).astype(np.float32)


def cases(seed, method=2, wide=False, scenes=None, build=True):
    """Yields dict(kind, n, seed, kw, xyz, labels, max_label) forever.  The order of the rng draws is the campaign format: do not
    reorder (logged campaigns are replayed through it).  build=False skips the scene generation where no later draw depends on it."""
    if scenes is None:
        import vgs_svgs_segmentation_amd as v
        scenes = v.scenes
    rng = np.random.default_rng(seed)
    while True:
        kind = rng.choice(["urban", "town", "pc", "fuzzy", "slab"])
        n = int(rng.integers(20_000, 90_000))
        sseed = int(rng.integers(0, 1 << 30))
        gen = None
        if kind == "urban": gen = lambda n=n, s=sseed: scenes.urban_scene(n, seed=s)
        elif kind == "town": gen = lambda n=n, s=sseed: scenes.town_scene(n, seed=s)
        elif kind == "pc": gen = lambda n=n, s=sseed: scenes.pc_scene(n, seed=s)
        elif kind == "fuzzy":
            sg = float(rng.choice([0.01, 0.03, 0.06, 0.1])); gen = lambda n=n, s=sseed, sg=sg: fuzzy(n, s, sg)
        else:
            th = float(rng.choice([0.05, 0.2, 0.35])); gen = lambda n=n, s=sseed, th=th: slab(n, s, th)
        kw = dict(voxel_size=float(rng.choice([0.06, 0.08, 0.1, 0.15])), graph_size=float(rng.choice([0.3, 0.4, 0.5, 0.6])),
                  cut_thred=float(rng.choice([0.1, 0.3, 0.5, 0.7, 0.9])), sig_w=float(rng.choice([1.0, 2.0])),
                  sig_n=float(rng.choice([0.2, 0.5])), sig_p=float(rng.choice([0.1, 0.2, 0.4])))
        cut_to = None
        if wide and method == 2:
            kw["voxel_size"] = float(rng.choice([0.04, 0.05, 0.06]))
            kw["graph_size"] = float(rng.choice([0.3, 0.4, 0.5]))
            cut_to = min(n, 50_000)
        if kw["graph_size"] / kw["voxel_size"] > (10.5 if wide else 8.0):
            kw["graph_size"] = (10.0 if wide else 8.0) * kw["voxel_size"]
        holes = False
        if method == 2 and rng.random() < 0.5:   # the size filters, and inputs with holes: non-finite points, repeated points
            kw.update(points_min=int(rng.choice([3, 5, 10, 20])), voxels_min=int(rng.choice([1, 3, 8])), adjacency_min=int(rng.choice([1, 3, 6])))
            holes = rng.random() < 0.5
        need_xyz = build or holes or method == 3      # later draws depend on the cloud's size
        xyz = None
        if need_xyz:
            xyz = gen()
            if cut_to is not None:
                xyz = xyz[:cut_to]
        n_out = cut_to if cut_to is not None else n
        if holes:
            xyz = xyz.copy()
            xyz[rng.random(xyz.shape[0]) < 0.002] = np.nan
            xyz = np.concatenate([xyz, xyz[rng.integers(0, xyz.shape[0], xyz.shape[0] // 50)]])
        labels = max_label = seed_size = None
        if method == 3:
            kw = dict(graph_size=float(rng.choice([0.4, 0.5, 0.8, 1.2])), cut_thred=kw["cut_thred"], sig_w=kw["sig_w"], sig_n=kw["sig_n"], sig_p=kw["sig_p"])
            seed_size = float(rng.choice([0.15, 0.25, 0.4]))
            labels, max_label = grid_supervoxels(xyz, seed_size, rng)
        yield dict(kind=str(kind), n=n_out, seed=sseed, kw=kw, xyz=xyz, labels=labels, max_label=max_label, holes=holes)


def check_case(v, oracle, c, method, threads, full_order):
    """Runs one case on the GPU and through the oracle.  Returns (ok, summary string, messages)."""
    from helpers import oracle_params, ragged_sets
    msgs = []
    p = v.default_params(method, **c["kw"])
    xyz = c["xyz"]
    t0 = time.time()
    e = v.Engine(p); e.set_points(xyz)
    if method == 3:
        e.set_supervoxel_labels(c["labels"], c["max_label"]); e.svgs_segment()
    else:
        e.run()
    t_gpu = time.time() - t0
    t0 = time.time()
    rp = oracle_params(oracle, p, threads=threads)
    ref = oracle.run_svgs_from_labels(xyz, c["labels"], c["max_label"], rp) if method == 3 else oracle.run_vgs(xyz, rp)
    t_cpu = time.time() - t0
    ok = True
    tag = f"{c['kind']} {c['n']} {c['seed']} {c['kw']}"
    for which in ("connect_cut", "connect_final"):
        off, idx = e.lists(which); roff, ridx = ref.lists(which)
        if not (np.array_equal(off, roff) and ragged_sets(off, idx) == ragged_sets(roff, ridx)):
            ok = False
            msgs.append(f"MISMATCH {which} {tag}")
    if not np.array_equal(e.point_labels(), ref.labels()[0]):
        ok = False
        msgs.append(f"MISMATCH labels {tag}")
    # element order (round 3): connect lists in merge-history order, getClusterIdx in the reference's DFS order; full adjacency lists
    if ok and full_order:
        for which in ("connect_cut", "connect_final"):
            off, idx = e.lists(which, "reference"); roff, ridx = ref.lists(which)
            if not (np.array_equal(off, roff) and np.array_equal(idx, ridx)):
                ok = False
                msgs.append(f"MISMATCH order {which} {tag}")
        co, ci = e.clusters("reference"); rco, rci = ref.lists("clusters_points")
        if not (np.array_equal(co, rco) and np.array_equal(ci, rci)):
            ok = False
            msgs.append(f"MISMATCH cluster order {tag}")
        ao, ai = e.lists("adjacency"); rao, rai = ref.lists("adjacency")
        if not (np.array_equal(ao, rao) and np.array_equal(ai, rai)):
            ok = False
            msgs.append(f"MISMATCH adjacency (all voxels) {tag}")
    sc = e.schedule_counters()
    nmax = int(e.adjacency_counts().max(initial=0))
    s = (f"{c['kind']} n={c['n']} used={e.counts()['used']} nmax={nmax} handed={sc['handed_over']}/{sc['handed_over_large']} banded={sc['banded']} "
         f"sent_on={sc['dense_sent_on']} gpu={t_gpu:.2f}s oracle={t_cpu:.1f}s {'ok' if ok else 'BAD'}")
    return ok, s, msgs


def main():
    import vgs_svgs_segmentation_amd as v
    import refcpu_py as oracle
    args = [a for a in sys.argv[1:]]
    opt = {}
    for name in ("--case", "--threads", "--dump"):
        if name in args:
            k = args.index(name); opt[name] = args[k + 1]; del args[k:k + 2]
    budget = float(args[0]) if len(args) > 0 else 240.0
    seed = int(args[1]) if len(args) > 1 else 1
    method = int(args[2]) if len(args) > 2 else 2     # 3: SVGS from a grid labelling (everything behind pcl::SupervoxelClustering)
    wide = len(args) > 3 and args[3] == "wide"        # search balls of 6-10 voxels: the multi-wavefront classes and their hand-overs
    only = int(opt["--case"]) if "--case" in opt else None
    threads = int(opt.get("--threads", os.cpu_count() or 1))
    state = dict(runs=0, bad=0, skipped=0, current=None, t0=time.time())

    def summary(why):
        cur = f", unfinished: case {state['current']}" if state["current"] else ""
        print(f"{state['runs']} runs, {state['bad']} mismatches, {state['skipped']} skipped, {time.time() - state['t0']:.0f} s ({why}){cur}", flush=True)

    def on_term(sig, frm):
        summary(f"cut by signal {sig}")
        os._exit(3 if state["bad"] else 2)
    signal.signal(signal.SIGTERM, on_term)
    signal.signal(signal.SIGINT, on_term)

    t_end = time.time() + budget
    for k, c in enumerate(cases(seed, method, wide, build=(only is None)), 1):
        if only is not None:
            if k < only:
                continue
            if k > only:
                break
            if c["xyz"] is None:        # skipped builds are for the cases before the one asked for; rebuild this one
                c = next(x for j, x in enumerate(cases(seed, method, wide), 1) if j == only)
        elif time.time() >= t_end:
            break
        state["current"] = f"{k} ({c['kind']} {c['n']} {c['seed']})"
        print("start", k, c["kind"], c["n"], c["seed"], c["kw"], flush=True)
        if "--dump" in opt:
            os.makedirs(opt["--dump"], exist_ok=True)
            np.savez_compressed(os.path.join(opt["--dump"], f"case_{seed}_{method}_{int(wide)}_{k}.npz"), xyz=c["xyz"])
        try:
            ok, s, msgs = check_case(v, oracle, c, method, threads, full_order=(k % 2 == 1))
        except v.VgsError as ex:
            state["skipped"] += 1
            print("skip", k, c["kind"], c["n"], c["kw"], str(ex)[:80], flush=True)
            continue
        for m in msgs:
            print(m, flush=True)
        state["runs"] += 1; state["bad"] += 0 if ok else 1
        print(f"run {k} {s}", flush=True)
        state["current"] = None
    summary("done")
    sys.exit(1 if state["bad"] else 0)


if __name__ == "__main__":
    main()

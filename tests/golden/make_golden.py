#!/usr/bin/env python3
"""Generates the golden vectors of tests/golden/ with the CPU oracle (RefMath = the reference's arithmetic,
lean flavour; plus the DevMath labels the HIP path must reproduce bit for bit).

The reference ships no fixtures and cannot be built here, so these vectors are produced by the
oracle in this container; they pin the oracle against silent changes and give the GPU tests a fixed target.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import refcpu_py as R  # noqa: E402
import vgs_svgs_segmentation_amd as v  # noqa: E402

CASES = {
    # name: (scene function, n, oracle params)
    "town_20k": (v.scenes.town_scene, 20_000, dict()),
    "urban_30k_v010": (v.scenes.urban_scene, 30_000, dict(voxel_size=0.1)),
    "pc_20k_v005_g025": (v.scenes.pc_scene, 20_000, dict(voxel_size=0.05, graph_size=0.25)),
}


def grid_supervoxels(xyz, seed):
    """A plain seed-size grid as the supervoxel labelling (what matters downstream of pcl::SupervoxelClustering), with some
    unassigned points (label 0, SS:303); getMaxLabel() is the largest label, whose supervoxel the reference drops (SS:313)."""
    cell = np.floor(xyz.astype(np.float64) / seed).astype(np.int64)
    cell -= cell.min(0)
    code = (cell[:, 0] * 4096 + cell[:, 1]) * 4096 + cell[:, 2]
    _, inv = np.unique(code, return_inverse=True)
    labels = (inv + 1).astype(np.int32)
    rng = np.random.default_rng(5)
    labels[rng.random(labels.size) < 0.01] = 0
    return labels, int(labels.max())


SVGS_CASES = {
    # name: (scene function, n, labelling, oracle params): method 3, Task_File_SVGS.txt values unless overridden
    "svgs_urban_40k_grid": (v.scenes.urban_scene, 40_000, "grid", dict()),
    "svgs_pc_80k_vccs": (v.scenes.pc_scene, 80_000, "vccs", dict()),      # supervoxels in PCL's order (the engine's default since round 6)
    "svgs_pc_80k_vccs0": (v.scenes.pc_scene, 80_000, "vccs0", dict()),    # the synchronous variant (vccs_mode 0)
    "svgs_town_30k_grid_cut03": (v.scenes.town_scene, 30_000, "grid", dict(cut_thred=0.3, sig_w=2.0, graph_size=0.6)),
}


def main_svgs(out_dir):
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for name, (fn, n, how, kw) in SVGS_CASES.items():
        if only and name not in only:
            continue
        xyz = fn(n)
        p0 = R.svgs_params(**kw)
        if how == "grid":
            labels, max_label = grid_supervoxels(xyz, p0.seed_size)
        elif how == "vccs0":
            labels, max_label = R.vccs(xyz, p0)     # the oracle's restatement of this repo's synchronous VCCS-style stage (unpinned against PCL)
        else:
            labels, max_label = R.vccs_pcl(xyz, p0)  # ... of pcl::SupervoxelClustering's own order (unpinned against PCL; the default)
        rec = {"xyz": xyz, "sv_label": labels, "max_label": np.int32(max_label)}
        for math in (0, 1):
            r = R.run_svgs_from_labels(xyz, labels, max_label, R.svgs_params(math=math, flavour=1, **kw))
            nd = r.nodes()
            pl, nc = r.labels()
            tag = "ref" if math == 0 else "dev"
            if math == 0:
                off, idx = r.lists("sv_points")
                rec.update(sv_start=off.astype(np.int64), sv_point_idx=idx.astype(np.int32))
                aoff, _ = r.lists("adjacency")
                rec.update(adj_len=np.diff(aoff).astype(np.int32))
            rec[f"centroid_{tag}"] = nd["centroid"]
            rec[f"normal_{tag}"] = nd["normal"]
            rec[f"eigen_{tag}"] = nd["eigen"]
            rec[f"point_label_{tag}"] = pl
            rec[f"node_cluster_{tag}"] = nc
            rec[f"clusters_{tag}"] = np.array([r.clusters_num, r.kept_clusters], dtype=np.int32)
        # the faithful data flow (n x n matrix, std::sort) in the reference's arithmetic: the partition the reference would give
        rf = R.run_svgs_from_labels(xyz, labels, max_label, R.svgs_params(math=0, flavour=0, **kw))
        rec["point_label_ref_faithful"] = rf.labels()[0]
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec,
                            params_keys=np.array(list(kw.keys())), params_vals=np.array(list(kw.values()), dtype=np.float64))
        print(name, xyz.shape, "supervoxels", rec["centroid_ref"].shape[0], "clusters", rec["clusters_ref"], rec["clusters_dev"])


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    if "--svgs-only" not in sys.argv:
        main_vgs(out_dir)
    main_svgs(out_dir)


def main_vgs(out_dir):
    for name, (fn, n, kw) in CASES.items():
        xyz = fn(n)
        rec = {"xyz": xyz, "params": np.array(sorted(kw.items()), dtype=object) if kw else np.array([], dtype=object)}
        for math in (0, 1):
            r = R.run_vgs(xyz, R.vgs_params(math=math, flavour=1, **kw))
            t = r.voxel_table()
            nd = r.nodes()
            pl, nc = r.labels()
            tag = "ref" if math == 0 else "dev"
            if math == 0:
                rec.update(key=t["key"], start=t["start"], point_voxel=t["point_voxel"], bbox=r.bbox(), depth=np.int32(r.depth),
                           used=nd["used"])
                off, idx = r.lists("adjacency")
                rec.update(adj_len=np.diff(off).astype(np.int32))
            rec[f"centroid_{tag}"] = nd["centroid"]
            rec[f"normal_{tag}"] = nd["normal"]
            rec[f"eigen_{tag}"] = nd["eigen"]
            rec[f"point_label_{tag}"] = pl
            rec[f"node_cluster_{tag}"] = nc
            rec[f"clusters_{tag}"] = np.array([r.clusters_num, r.kept_clusters], dtype=np.int32)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **{k: val for k, val in rec.items() if k != "params"},
                            params_keys=np.array(list(kw.keys())), params_vals=np.array(list(kw.values()), dtype=np.float64))
        print(name, xyz.shape, "V", rec["key"].shape[0], "clusters", rec["clusters_ref"], rec["clusters_dev"])


if __name__ == "__main__":
    main()

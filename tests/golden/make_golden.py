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


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
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

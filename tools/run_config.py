#!/usr/bin/env python3
"""Run one BASELINE.json configuration on the GPU and print ONE bench-style JSON line (counts, ms per step, stage times).
usage: tools/run_config.py c1|c2|c3|c3n|c4|c4s|xl [points] [steps]      (not the bench contract: bench.py times configs[2])"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vgs_svgs_segmentation_amd as v

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if cfg == "c2":
    xyz, p, name = v.scenes.pc_scene(n or 1_000_000), v.default_params(2, voxel_size=0.05), "PC1M: planar+cylinder scene, VGS, voxel 0.05 m, graph 0.5 m"
elif cfg == "c3":
    xyz, p, name = v.scenes.urban_scene(n or 10_000_000), v.default_params(2, voxel_size=0.1), "URB10M: urban scene, VGS, voxel 0.1 m, graph 0.5 m"
elif cfg in ("c4", "c4p"):   # BASELINE config 4; supervoxels in PCL's own order are the default since round 6 (c4p: the name rounds 3-5 used for it)
    xyz, p, name = v.scenes.urban_scene(n or 10_000_000), v.default_params(3), "URB10M: urban scene, SVGS (Task_File_SVGS.txt: voxel 0.05 m, seed 0.25 m, graph 0.5 m), supervoxels in PCL's own order (vccs_mode 1, the default)"
elif cfg == "c4s":
    xyz, p, name = v.scenes.urban_scene(n or 10_000_000), v.default_params(3, vccs_mode=0), "URB10M: urban scene, SVGS with the synchronous supervoxel variant (vccs_mode 0: an approximation, not within P2 of the default)"
elif cfg == "c3n":
    xyz, p, name = v.scenes.noisy_surface_scene(n or 5_000_000), v.default_params(2, voxel_size=0.1), "C3N: undulating surface, 3 cm range noise, VGS, voxel 0.1 m, graph 0.5 m"
elif cfg == "xl":
    xyz, p, name = v.scenes.solid_block_scene(n or 500_000), v.default_params(2, voxel_size=0.05, graph_size=0.5), "BLOCK: solid cube, VGS, voxel 0.05 m, graph 0.5 m (neighbourhoods up to 4159 voxels)"
elif cfg == "c1":
    xyz, p, name = v.scenes.town_scene(n or 500_000), v.default_params(2), "TOWN stand-in: VGS, Task_File_VGS.txt defaults"
else:
    raise SystemExit("unknown configuration " + cfg)
eng = v.Engine(p)
eng.set_points(xyz)
ms, acc = [], {}
for it in range(steps + 1):
    t = time.perf_counter()
    if cfg in ("c4", "c4p", "c4s"):
        eng.supervoxels()   # run() keeps supervoxel labels once they exist: createSupervoxels is part of every step
    eng.run()
    dt = (time.perf_counter() - t) * 1e3
    if it == 0:
        continue            # warm-up (allocations, table uploads)
    ms.append(dt)
    for k, x in eng.stage_times().items():
        acc[k] = acc.get(k, 0.0) + x
c = eng.counts()
ms.sort()
print(json.dumps({"metric": "segmented points/sec (end-to-end, inputs resident in HBM)", "config": {"workload": name, "points": int(xyz.shape[0]), "id": cfg},
                  "value": xyz.shape[0] / (sum(ms) / len(ms) * 1e-3), "unit": "points/s", "ms_per_step": sum(ms) / len(ms), "ms_per_step_median": ms[len(ms) // 2],
                  "steps": steps, "n_gpus": 1, "dtype": "f32", "data": "synthetic", "counts": c,
                  "stage_ms": {k: x / steps for k, x in acc.items()}, "schedule": eng.schedule_counters()}))

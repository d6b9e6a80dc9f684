import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import vgs_svgs_segmentation_amd as v
for name, xyz, kw in (("urban1M", v.scenes.urban_scene(1_000_000), dict(voxel_size=0.1)), ("urban3M", v.scenes.urban_scene(3_000_000), dict(voxel_size=0.1)),
                      ("pc1M v.05", v.scenes.pc_scene(1_000_000), dict(voxel_size=0.05)), ("town500k", v.scenes.town_scene(500_000), dict())):
    e = v.Engine(v.default_params(2, **kw)); e.set_points(xyz)
    for it in range(3):
        t = time.perf_counter(); e.run(); dt = time.perf_counter() - t
    c = e.counts(); s = e.schedule_counters()
    print(f"{name}: {dt*1e3:.1f} ms used {c['used']} adj/used {c['adj']/c['used']:.0f} handed {s['handed_over']} large {s['handed_over_large']} sent_on {s['dense_sent_on']} classes a {c['class_a']} bc {c['class_bc']} d {c['class_d']}", {k: round(x, 2) for k, x in e.stage_times().items() if x}, flush=True)

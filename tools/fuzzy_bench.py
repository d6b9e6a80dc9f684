import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import vgs_svgs_segmentation_amd as v
def fuzzy(n, seed, sigma):
    rng = np.random.default_rng(seed)
    side = np.sqrt(n / 6000.0)
    x, y = rng.random(n) * side, rng.random(n) * side
    z = 0.3 * np.sin(2.0 * x) * np.cos(1.5 * y) + rng.normal(0, sigma, n) + 2.0
    return np.stack([x + 0.011, y + 0.017, z], axis=1).astype(np.float32)
for sigma in (0.01, 0.03, 0.08):
    xyz = fuzzy(5_000_000, 1, sigma)
    e = v.Engine(v.default_params(2, voxel_size=0.1)); e.set_points(xyz)
    for it in range(3):
        t = time.perf_counter(); e.run(); dt = time.perf_counter() - t
    c = e.counts(); s = e.schedule_counters()
    print(f"sigma {sigma}: {dt*1e3:.1f} ms used {c['used']} adj/used {c['adj']/c['used']:.0f} pairs {c['pairs']/1e6:.0f}M handed {s['handed_over']} large {s['handed_over_large']} sent_on {s['dense_sent_on']}", {k: round(x, 2) for k, x in e.stage_times().items()}, flush=True)

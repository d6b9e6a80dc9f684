"""Contexts are independent: several of them driven from different host threads at the same time (a stream of clouds with
more than one in flight, tools/frames_in_flight.py) must give what each gives alone.  The library keeps no state outside
a context; this test would catch a shared scratch buffer, table or counter."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_contexts_in_parallel_threads(gpu):
    jobs = [(gpu.scenes.urban_scene(300_000), gpu.default_params(2, voxel_size=0.1)),
            (gpu.scenes.town_scene(200_000), gpu.default_params(2)),
            (gpu.scenes.urban_scene(250_000), gpu.default_params(3)),                    # SVGS: supervoxels, then the same cut
            (gpu.scenes.urban_scene(300_000), gpu.default_params(2, voxel_size=0.1, cut_thred=0.5))]
    alone = []
    for xyz, p in jobs:
        e = gpu.Engine(p); e.set_points(xyz); e.run()
        alone.append((e.point_labels().copy(), e.counts()))
    results = [None] * len(jobs)
    errors = []
    start = threading.Barrier(len(jobs))

    def work(k):
        try:
            xyz, p = jobs[k]
            e = gpu.Engine(p)
            start.wait()
            for _ in range(4):   # several runs each, so that the stages of different contexts interleave in many ways
                e.set_points(xyz)
                e.run()
            results[k] = (e.point_labels().copy(), e.counts())
        except Exception as ex:   # noqa: BLE001 -- reported below, in the main thread
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, (lab, cnt) in enumerate(results):
        np.testing.assert_array_equal(lab, alone[k][0])
        for key in ("voxels", "used", "adj", "clusters", "kept"):
            assert cnt[key] == alone[k][1][key], (k, key)

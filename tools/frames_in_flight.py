#!/usr/bin/env python3
"""Throughput with several clouds in flight: F contexts, one host thread each, every thread runs K steps of the bench
workload on its own context (its own streams).  Prints points/s for F = 1, 2, 3."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vgs_svgs_segmentation_amd as v

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
xyz = v.scenes.urban_scene(N)
dev = torch.from_numpy(xyz).cuda()
p = v.default_params(2, voxel_size=0.1)


def worker(eng, k, bar):
    bar.wait()
    for _ in range(k):
        eng.set_points_device(dev.data_ptr(), N, 12, keep=dev)
        eng.run()


for F in (1, 2, 3):
    engs = [v.Engine(p) for _ in range(F)]
    for e in engs:   # warm up: allocations, tables
        e.set_points_device(dev.data_ptr(), N, 12, keep=dev); e.run()
    torch.cuda.synchronize()
    bar = threading.Barrier(F + 1)
    th = [threading.Thread(target=worker, args=(e, K, bar)) for e in engs]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    labels = [e.point_labels() for e in engs]
    same = all(np.array_equal(labels[0], l) for l in labels[1:])
    print(f"F={F}: {F * K} steps in {dt * 1e3:.1f} ms -> {dt * 1e3 / (F * K):.2f} ms/step, {F * K * N / dt / 1e9:.3f} Gpts/s, labels equal across contexts: {same}", flush=True)
    del engs

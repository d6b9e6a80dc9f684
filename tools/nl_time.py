#!/usr/bin/env python3
"""Kernel time of the near-list builder on URB10M (A/B of library variants: VGS_LIB=...): runs the step under no profiler and reads the stage clock of the local cut up to the bulk launch is not available, so this prints k_near_lists from rocprofv3 stats when run under it; plain: prints the localcut stage time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vgs_svgs_segmentation_amd as v
xyz = v.scenes.urban_scene(10_000_000)
eng = v.Engine(v.default_params(2, voxel_size=0.1))
eng.set_points(xyz)
for it in range(4):
    eng.run()
print(os.environ.get("VGS_LIB", "base"), eng.stage_times())

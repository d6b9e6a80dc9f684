#!/bin/bash
# A/B of the early hand-over (VGS_EARLY_HO=N persistent workgroups of the dense hand-over kernel dispatched before the bulk class): step and
# bulk-kernel time on the bench's configuration, two runs each
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 0 64 128 256 512 0 256; do
  VGS_EARLY_HO=$n timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-to-host | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d.get('stage_ms', {})
print('VGS_EARLY_HO=$n', 'step %.3f ms' % d['ms_per_step'], 'bulk kernel %.3f' % d['roofline']['kernel_ms'], 'localcut %.3f' % s.get('localcut', 0), 'merge %.3f' % s.get('merge', 0), 'localcut_kernel %.3f' % s.get('localcut_kernel', 0))"
done

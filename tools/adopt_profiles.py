#!/usr/bin/env python3
"""Copy what tools/collect_round.sh left under gpurun_out/profiles/ into profiles/ (tracked) and merge its manifest (file -> commit).
usage (container): tools/adopt_profiles.py r05"""
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(root, "gpurun_out", "profiles"), os.path.join(root, "profiles")
man_path = os.path.join(dst, f"{tag}_MANIFEST.json")
man = json.load(open(man_path)) if os.path.exists(man_path) else {}
new = json.load(open(os.path.join(src, f"{tag}_MANIFEST.json")))
for f, commit in new.items():
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
        man[f] = commit
json.dump(dict(sorted(man.items())), open(man_path, "w"), indent=1)
print({c: sum(1 for x in man.values() if x == c) for c in set(man.values())})

#!/usr/bin/env python3
"""Finds the campaign (seed, method, wide) and the case index K of logged `start <kind> <n> <scene seed>` lines of tools/fuzz_parity.py,
by replaying the case generator on the CPU (no GPU, no oracle).  usage: fuzz_find.py kind:n:seed [kind:n:seed ...] [--seeds A-B] [--depth D]"""
import sys
from fuzz_parity import cases

args = sys.argv[1:]
lo, hi, depth = 1, 100, 400
if "--seeds" in args:
    k = args.index("--seeds"); lo, hi = map(int, args[k + 1].split("-")); del args[k:k + 2]
if "--depth" in args:
    k = args.index("--depth"); depth = int(args[k + 1]); del args[k:k + 2]
want = {(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in args}
for seed in range(lo, hi + 1):
    for wide in (False, True):
        for k, c in enumerate(cases(seed, 2, wide, build=False), 1):
            if k > depth:
                break
            key = (c["kind"], c["n"], c["seed"])
            if key in want:
                print(f"{key}: seed {seed} method 2 wide {int(wide)} case {k} holes {c['holes']} kw {c['kw']}", flush=True)

"""Cases of the randomised differential campaigns (tools/fuzz_parity.py) that deserve to be permanent: the four of round 3 whose
campaigns were cut (by their time budget, inside a slow single-thread oracle run) before a verdict was printed -- fuzzy 66184 pts
(seed 118680320), slab 64509 (418974415), slab 61975 (522626661, with non-finite and repeated points), pc 50000 (420246468; search
ball of ten voxels).  Each is replayed exactly from its campaign (seed, method, wide, case index: tools/fuzz_find.py found them by
replaying the generator) and held to the campaign's own bar: connect lists after the cut and after closestCheck, point labels,
the reference's element order, getClusterIdx and the full adjacency lists identical to the oracle (DevMath + lean)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu

CASES = [
    # campaign seed, method, wide, case index, what the log's `start` line said
    (101, 2, False, 76, ("fuzzy", 66184, 118680320)),
    (111, 2, False, 78, ("slab", 64509, 418974415)),
    (41, 2, False, 210, ("slab", 61975, 522626661)),
    (62, 2, True, 66, ("pc", 50000, 420246468)),
]


@pytest.mark.parametrize("seed,method,wide,k,logged", CASES, ids=[f"{c[4][0]}_{c[4][1]}" for c in CASES])
def test_unfinished_round3_case(gpu, oracle, seed, method, wide, k, logged):
    import fuzz_parity
    c = next(x for j, x in enumerate(fuzz_parity.cases(seed, method, wide, build=False), 1) if j == k)
    if c["xyz"] is None:
        c = next(x for j, x in enumerate(fuzz_parity.cases(seed, method, wide), 1) if j == k)
    assert (c["kind"], c["n"], c["seed"]) == logged
    ok, summary, msgs = fuzz_parity.check_case(gpu, oracle, c, method, threads=os.cpu_count() or 1, full_order=True)
    print(summary)
    assert ok, msgs

"""bench.py's config-4 block with the IF-rank prefetch on / off and paired anneals on / off (round 6, profiles/r06_rank_prefetch_start_ab.txt).
    python tools/config4_prefetch_ab.py <prefetch_ranks 0|1> <pair 0|1>"""
import os
import sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, batch
PF = float(sys.argv[1])
class S(Solver):
    def __init__(self, device=0):
        super().__init__(device)
        self.set_option("prefetch_ranks", PF)
s = S(0)
for rep in range(3):
    r = batch.bench_block(s, 0, 1, None, "cpu", pair=bool(int(sys.argv[2])))
    print("prefetch", PF, "pair", sys.argv[2], r["wall_s"], r["per_rank"][0]["solve_s"], r["per_rank"][0]["anneal_device_ms"])

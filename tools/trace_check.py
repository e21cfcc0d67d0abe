#!/usr/bin/env python3
"""Does a rocprofv3 kernel trace agree with the bench line taken under it?

    python tools/trace_check.py <trace_dir>/t_kernel_stats.csv <bench.json> [...pairs]

Prints, per pair, the step kernel's average duration in the trace next to `roofline.avg_launch_us`, and the trace's
total time of that kernel per SA step next to `us_per_step_kernel` (the trace holds every launch of the process, the
bench line the timed regions only; `process_totals` is what connects the two)."""
import csv
import json
import sys


def bench_line(path):
    for line in open(path):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise SystemExit(f"{path}: no bench line")


def main(argv):
    for stats, bench in zip(argv[0::2], argv[1::2]):
        b = bench_line(bench)
        # the full instantiation (`k_cluster<3, 4, 2>`): the process also runs other instantiations (config 2's side figure)
        name = b["roofline"]["kernel"].split("::")[-1]
        allrows = list(csv.DictReader(open(stats)))
        rows = [r for r in allrows if name in r["Name"]]
        calls = sum(int(r["Calls"]) for r in rows)
        total_us = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3
        # the same instantiation's twin for ranges that hold two-point minimiser steps (k_cluster_tp, round 5): part of the process's
        # steps run in it (the untimed whole anneal's final stage), so the per-step figure takes both; the average stays k_cluster's
        twin = name.replace("k_cluster<", "k_cluster_tp<")
        trows = [r for r in allrows if twin != name and twin in r["Name"]]
        tcalls = sum(int(r["Calls"]) for r in trows)
        ttotal_us = sum(float(r["TotalDurationNs"]) for r in trows) / 1e3
        t = b["process_totals"]
        print(f"{bench}: --steps {b['steps']} --warmup {b['warmup']}  kernel {b['roofline']['kernel']}")
        print(f"  trace : {calls} launches, average {total_us / calls:.2f} us, total {total_us / 1e3:.3f} ms"
              + (f"; its twin {twin.split('<')[0]}: {tcalls} launches, average {ttotal_us / tcalls:.2f} us, total {ttotal_us / 1e3:.3f} ms" if tcalls else "")
              + f" -> {(total_us + ttotal_us) / t['sa_steps']:.4f} us per SA step over the {t['sa_steps']} steps of the process")
        print(f"  bench : roofline.avg_launch_us {b['roofline']['avg_launch_us']:.2f} ({b['roofline']['launches_per_region']} launches per region),"
              f" us_per_step_kernel {b['us_per_step_kernel']:.4f}, us_per_step_device (event-bracketed region) {b['us_per_step_device']:.4f}")
        print(f"  ratio : per-step trace / bench = {(total_us + ttotal_us) / t['sa_steps'] / b['us_per_step_kernel']:.4f}")


if __name__ == "__main__":
    main(sys.argv[1:])

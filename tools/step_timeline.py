#!/usr/bin/env python3
"""Where a replayed bench step's time goes: reads a rocprofv3 --kernel-trace csv of `bench.py --steps K` and prints, for the steps of
the timed stretch (the longest run of consecutive kNN -> EdgeConv -> conv5 periods), every kernel's median duration and the median
idle gap in front of it on the main chain.  Usage: step_timeline.py <..._kernel_trace.csv>"""
import csv
import statistics
import sys


def short(n):
    for key in ("knn_mfma", "edgeconv_f16b", "conv_f16_kernel", "chamfer_fwd", "chamfer_loss", "conv5_persist", "mfma_sustained"):
        if key in n:
            return key
    return n.split("(")[0][-40:]


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # a step = from one knn_mfma start to the next
    starts = [i for i, r in enumerate(rows) if r[2] == "knn_mfma"]
    steps = []
    for a, b in zip(starts, starts[1:]):
        seg = rows[a:b]
        names = [s[2] for s in seg]
        if "edgeconv_f16b" in names and any(n.startswith("conv") for n in names) and len(seg) <= 8:
            steps.append((rows[b][0] - rows[a][0], seg))
    if not steps:
        print("no steps found")
        return
    periods = sorted(p for p, _ in steps)
    med = periods[len(periods) // 2]
    good = [(p, s) for p, s in steps if p < 1.15 * med]          # replayed steps, not the eager ones
    print(f"{len(steps)} steps, {len(good)} within 15 % of the median period {med / 1e3:.1f} us "
          f"(p10 {periods[len(periods) // 10] / 1e3:.1f}, p90 {periods[9 * len(periods) // 10] / 1e3:.1f})")
    dur, gap = {}, {}
    for p, seg in good:
        chain_end = None
        for s, e, n in seg:
            dur.setdefault(n, []).append(e - s)
            if n.startswith("chamfer_fwd"):
                continue
            if chain_end is not None:
                gap.setdefault(n, []).append(s - chain_end)
            chain_end = e if chain_end is None else max(chain_end, e)
        gap.setdefault("(next knn)", []).append(seg[0][0] + p - chain_end)
    tot = 0.0
    for n in dur:
        d = statistics.median(dur[n]) / 1e3
        g = statistics.median(gap[n]) / 1e3 if n in gap else float("nan")
        print(f"  {n:22s} dur {d:8.2f} us   gap in front {g:7.2f} us")
        if not n.startswith("chamfer_fwd"):
            tot += d + (g if g == g else 0)
    g = statistics.median(gap["(next knn)"]) / 1e3
    print(f"  {'(to next step)':22s} {'':16s} gap {g:7.2f} us   chain sum {tot + g:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])

"""Gaps between consecutive pass launches from a `rocprofv3 --kernel-trace` CSV: how much of a frame the GPU sits idle between
kernels (sequential mode) and how the two streams interleave (two frames in flight).  usage: python scripts/gpu_launch_gaps.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void srd::", "")))
rows.sort()
rows = [r for r in rows if "kernel<0>" in r[2] or "tile_order" in r[2]]
rows = rows[len(rows) // 3:]          # steady state
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - max(r[1] for r in rows[:i + 1]) for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print("launches %d, span %.3f ms, sum of kernel durations %.3f ms, idle gaps: %d, total %.3f ms (%.1f %% of the span), mean %.1f us, max %.1f us" % (
    len(rows), span / 1e6, busy / 1e6, len(pos), sum(pos) / 1e6, 100.0 * sum(pos) / span, (sum(pos) / max(len(pos), 1)) / 1e3, max(pos + [0]) / 1e3))
for s, e, n in rows[:8]:
    print("  %-22s start +%.3f ms  dur %.3f ms" % (n, (s - rows[0][0]) / 1e6, (e - s) / 1e6))

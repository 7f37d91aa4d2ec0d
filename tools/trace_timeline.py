"""Developer tool: timeline of one band step from a rocprofv3 --kernel-trace CSV (tools/profile_stats.sh): kernels of the step in start
order with stream, duration and the idle gap on their stream in front of them.

    python tools/trace_timeline.py gpurun_out/prof_<tag>/stats/*/*_kernel_trace.csv [step-index-from-the-end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_band_take" in r["Kernel_Name"]]
i0, i1 = marks[-back - 1], marks[-back]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
tot_gap = 0.0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  stream  kernel")
for r in rows[i0:i1]:
    s, e, st = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"]
    gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
    last_end[st] = e
    name = r["Kernel_Name"].replace("mopt::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {st:>6}  {name[:70]}")
print("step length (take -> next take):", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us")

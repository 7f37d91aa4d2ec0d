"""Reads a rocprofv3 kernel-trace csv: kernels of the last burst (bursts are separated by > 5 ms), start offsets and durations in us."""
import csv, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > 5_000_000:
        bursts.append(cur); cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
b = bursts[-1]
t0 = b[0][0]
busy = 0
for s, e, k, q in b:
    print("%8.1f %7.1f  q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k))
print("kernels", len(b), "span %.1f us" % ((b[-1][1] - t0) / 1e3), "sum of durations %.1f us" % (sum(e - s for s, e, _, _ in b) / 1e3))

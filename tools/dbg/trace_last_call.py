"""Reads a rocprofv3 kernel-trace (rocpd .db): kernels of the last burst (bursts are separated by > 5 ms), start offsets and durations in us."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in c.execute("select * from kernels")]
rows = sorted(((r["start"], r["end"], r["name"].split("(")[0][:72], r.get("stream_id", r.get("queue_id"))) for r in rows))
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > 5_000_000:
        bursts.append(cur); cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
b = bursts[-1]
t0 = b[0][0]
for s, e, k, q in b:
    print("%8.1f %7.1f  s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k))
print("kernels", len(b), "span %.1f us" % ((b[-1][1] - t0) / 1e3), "sum of durations %.1f us" % (sum(e - s for s, e, _, _ in b) / 1e3))

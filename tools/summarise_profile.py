"""Condense the rocprofv3 passes written by tools/profile_bench.sh into per-kernel rows:
launches, average duration (kernel-trace pass), FETCH_SIZE and WRITE_SIZE per launch (PMC passes, raw counter units = KB)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return r[0] if r else None


rows = defaultdict(lambda: {"n": 0, "ns": 0.0, "fetch": 0.0, "nf": 0, "write": 0.0, "nw": 0})
f = find("stats", "*kernel_trace.csv")
if f:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        rows[k]["n"] += 1
        rows[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for sub, key, cnt in (("fetch", "fetch", "nf"), ("write", "write", "nw")):
    f = find(sub, "*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            rows[k][key] += float(r["Counter_Value"])
            rows[k][cnt] += 1
tot = sum(v["ns"] for v in rows.values()) or 1.0
print(f"{'kernel':40s} {'launches':>9s} {'avg_us':>10s} {'time_%':>7s} {'FETCH_KB/launch':>16s} {'WRITE_KB/launch':>16s}")
for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ns"]):
    if v["n"] == 0:
        continue
    fe = v["fetch"] / v["nf"] if v["nf"] else float("nan")
    wr = v["write"] / v["nw"] if v["nw"] else float("nan")
    print(f"{k[:40]:40s} {v['n']:9d} {v['ns'] / v['n'] / 1e3:10.2f} {100 * v['ns'] / tot:7.2f} {fe:16.1f} {wr:16.1f}")

# whole-run totals of the counter passes (raw FETCH_SIZE is doubled: gfx950 counts wide reads at half their bytes)
tf = sum(v["fetch"] for v in rows.values()); tw = sum(v["write"] for v in rows.values())
# band steps of the run = launches of the band's post kernel (the run also steps the band while it builds its input trajectory)
steps = int(os.environ.get("MOPT_PROFILE_STEPS", "0")) or rows.get("mopt::k_neb_post", {"n": 0})["n"]
if steps and (tf or tw):
    print(f"\nall kernels of the run: FETCH {tf / 1024:.1f} MB raw, WRITE {tw / 1024:.1f} MB; per band step over the {steps} band steps of the run (trajectory "
          f"generation and warm-up included): 2 x FETCH + WRITE = {(2 * tf + tw) / 1024 / steps:.1f} MB")

#!/bin/bash
# Copy the files of one end-of-round run (tools/end_of_round.sh <tag>, merged back into gpurun_out/) into profiles/ under the round's
# prefix, write the per-launch list of the tridiagonalisation and refresh profiles/traffic.json from the PMC summary.
#   bash tools/collect_profiles.sh r04k
set -e
TAG=${1:?tag}
S=gpurun_out/$TAG; P=gpurun_out/prof_$TAG; D=profiles
cp $S/bench_32x100.json $D/${TAG}_bench_32x100.json
cp $S/bench_64x200.json $D/${TAG}_bench_64x200.json
for K in 4 8 16; do cp $S/bench_${K}x100.json $D/${TAG}_bench_${K}x100_shard_shape.json; done
cp $S/bench_16x14.json $D/${TAG}_bench_16x14.json
cp $S/bench_32x100_60steps.json $D/${TAG}_bench_32x100_60steps.json
for A in 7 11 22; do cp $S/bench_single_${A}atoms.json $D/${TAG}_bench_single_geometry_${A}atoms.json; done
for A in 7 22; do cp $S/bench_single_${A}atoms_general_pipeline.json $D/${TAG}_bench_single_geometry_${A}atoms_general_pipeline.json; done
cp $S/single.txt $D/${TAG}_single_geometry_plugin_classes.txt
grep -v amdgpu.ids $S/timeline.txt > $D/${TAG}_timeline_32x100.txt
cp $S/pytest.log $D/${TAG}_gpu_suite.log
cp "$(ls -t $P/stats/*/*_kernel_stats.csv | head -1)" $D/${TAG}_bench_32x100_kernel_stats.csv      # (newest pass when the profile was repeated)
cp $P/bench_under_stats.json $D/${TAG}_bench_32x100_under_rocprof.json
cp $P/summary.txt $D/${TAG}_bench_32x100_summary_with_pmc.txt
cp "$(ls -t $S/stats_64x200/*/*_kernel_stats.csv | head -1)" $D/${TAG}_bench_64x200_kernel_stats.csv
cp $S/bench_64x200_under_stats.json $D/${TAG}_bench_64x200_under_rocprof.json
python3 - "$TAG" <<'PY'
import csv, glob, json, sys, os
tag = sys.argv[1]
sys.path.insert(0, os.getcwd())
import bench
rows = list(csv.DictReader(open(max(glob.glob(f"gpurun_out/prof_{tag}/stats/*/*_kernel_trace.csv"), key=os.path.getmtime))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(f"profiles/{tag}_k_tri_persist3_per_launch_us.txt", "w") as fh:
    fh.write("# duration of every k_tri_persist3 launch of the default bench command under rocprofv3 --kernel-trace, in launch order (us); <..., true>: early stop on (identity-start workload), <..., false>: exact-start workload\n")
    for r in rows:
        if "k_tri_persist3" in r["Kernel_Name"]:
            fh.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:9.1f}  {r['Kernel_Name'].split('(')[0]}\n")
# PMC figures of the dominant kernel
fig = {}
for line in open(f"gpurun_out/prof_{tag}/summary.txt"):
    if "k_tri_persist3" in line:
        parts = line.split()
        key = "true" if "true" in line else "fals"
        fig[key] = (int(parts[-5]), float(parts[-4]), float(parts[-2]), float(parts[-1]))
t = json.load(open("profiles/traffic.json"))
old = t.get("k_tri_persist3")
if old and old.get("round") != tag:
    t["k_tri_persist3_" + old["round"]] = old
n, avg, f, w = fig["true"]
n2, avg2, f2, w2 = fig["fals"]
t["k_tri_persist3"] = {"fetch_kb_per_launch": f, "write_kb_per_launch": w, "round": tag, "workload": "32x100:identity", "source_sha1": bench.library_stamp(),
                       "note": f"k_tri_persist3<5,3,16,true> (early stop on), {n} launches of the default bench command, {avg:.2f} us average under the profiler "
                               f"(profiles/{tag}_k_tri_persist3_per_launch_us.txt); the variant without the test (exact-start workload): fetch {f2:.1f} KB, write {w2:.1f} KB, "
                               f"{avg2:.2f} us; profiles/{tag}_bench_32x100_summary_with_pmc.txt (counter passes run with the cross-stream waits switched off, see k_tridiag.hip serialized_dispatch)"}
json.dump(t, open("profiles/traffic.json", "w"), indent=2)
print("traffic.json:", t["k_tri_persist3"])
PY
ls $D | grep "^${TAG}_"

#!/bin/bash
# Run on the MI355X box from the repo root:  bash tools/end_of_round.sh <tag>
# The numbers a round is closed with: GPU test suite, smoke, bench lines (headline, 64 x 200, per-rank shard shapes, 16 x 14,
# 60-step average), single-geometry timings and the three rocprofv3 passes of tools/profile_bench.sh; everything under gpurun_out/<tag>/.
TAG=${1:-eor}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python3 bench.py > $OUT/bench_32x100.json 2> $OUT/bench.err && echo "headline done"
python3 bench.py --images 64 --atoms 200 --steps 10 --cpu-budget 40 > $OUT/bench_64x200.json 2>> $OUT/bench.err && echo "64x200 done"
for K in 4 8 16; do python3 bench.py --images $K --atoms 100 --cpu-budget 0 --no-cpu-baseline --no-second-workload > $OUT/bench_${K}x100.json 2>> $OUT/bench.err; done
python3 bench.py --images 16 --atoms 14 --no-cpu-baseline > $OUT/bench_16x14.json 2>> $OUT/bench.err
python3 bench.py --steps 60 --no-cpu-baseline --no-second-workload > $OUT/bench_32x100_60steps.json 2>> $OUT/bench.err && echo "bench lines done"
python3 tools/time_single_geometry.py > $OUT/single.txt 2>&1
# BASELINE configs 2 / 3 (+ 11 atoms): one line each with the oracle classes on one host core beside it
for A in 7 11 22; do python3 bench.py --single-geometry $A --steps 60 --warmup 5 > $OUT/bench_single_${A}atoms.json 2>> $OUT/bench.err; done
# the general pipeline at the same sizes (one-workgroup kernels switched off), for the comparison in DESIGN
for A in 7 22; do MOPT_SMALL=0 python3 bench.py --single-geometry $A --steps 40 --warmup 5 > $OUT/bench_single_${A}atoms_general_pipeline.json 2>> $OUT/bench.err; done
MOPT_TIMELINE=1 python3 bench.py --no-cpu-baseline --no-second-workload > $OUT/timeline_bench.json 2> $OUT/timeline.txt
bash tools/profile_bench.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
# kernel statistics of the 64 x 200 line (its roofline object must be recomputable from a committed file)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stats_64x200 -- python3 $OLDPWD/bench.py --images 64 --atoms 200 --steps 6 --warmup 2 --no-cpu-baseline --no-second-workload > $OLDPWD/$OUT/bench_64x200_under_stats.json 2> $OLDPWD/$OUT/stats_64x200.err )
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        sw = d.get("second_workload") or {}
        print(f.split("/")[-1], round(d["ms_per_step"], 4), "exact:", round(sw.get("ms_per_step", 0), 4), "cpu:", (d.get("cpu_baseline") or {}).get("value"),
              {k: round(v["gpu_ms"], 4) for k, v in (d.get("classes") or {}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/single.txt

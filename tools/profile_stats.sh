#!/bin/bash
# Run on the MI355X box from the repo root:  bash tools/profile_stats.sh <tag> [bench.py arguments]
# One rocprofv3 --kernel-trace --stats pass over bench.py; per-kernel table (launches, average us, share) to gpurun_out/prof_<tag>/summary.txt
set -e
TAG=${1:-run}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py "$@" > $OUT/bench_under_stats.json 2> $OUT/stats.err
cd $ROOT
python3 tools/summarise_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt

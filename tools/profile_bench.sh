#!/bin/bash
# Run on the MI355X box from the repo root:  bash tools/profile_bench.sh <tag>
# Three separate rocprofv3 passes over the same bench command (timing, then one PMC counter per pass -- FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950), outputs under gpurun_out/prof_<tag>/.
set -e
TAG=${1:-run}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=${MOPT_PROFILE_ARGS:-"--no-cpu-baseline"}      # the default bench command (20 steps, 3 warm-up): per-kernel averages comparable with its own line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_fetch.json 2> $OUT/fetch.err
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_under_write.json 2> $OUT/write.err
echo "write pass done"
cd $ROOT
python3 tools/summarise_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt

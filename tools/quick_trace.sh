#!/bin/bash
# kernel trace of one resident-only run (MTSV_LANES=1), summary to gpurun_out/<name>_kernel_stats.csv
# usage (on the GPU box): tools/quick_trace.sh NAME [extra bench args]
set -e
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/qt_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MTSV_LANES=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --resident-only "$@" > $OUT/bench.json 2> $OUT/trace.err
cd $ROOT
python3 tools/profile_summary.py trace $OUT/trace $ROOT/gpurun_out/${NAME}_kernel_stats.csv
rm -rf $OUT/trace
head -16 $ROOT/gpurun_out/${NAME}_kernel_stats.csv
tail -c 600 $OUT/bench.json

#!/bin/bash
# One GPU-box call that produces everything profiles/ quotes for a workload (default config2):
# the plain bench line, the rocprofv3 kernel trace of `bench.py --no-extras`, and FETCH_SIZE / WRITE_SIZE
# in separate counter passes (profiled runs use MTSV_LANES=1; the plain bench line uses the default lanes).  Output under gpurun_out/prof_<workload>/ .
set -e
WL=${1:-config2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$WL
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 bench.py --workload $WL > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
# one lane for the profiled runs: with the default three lanes kernels of different parts overlap and a
# kernel's duration in the trace includes the time it shared the device
export MTSV_LANES=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace done"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/fetch.err
echo "fetch done"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/write.err
echo "write done"
cd $ROOT
python3 tools/profile_summary.py trace $OUT/trace $OUT/kernel_stats.csv
python3 tools/profile_summary.py pmc $OUT/fetch $OUT/write $WL $OUT/hbm_pmc.json $OUT/hbm_traffic.json
rm -rf $OUT/fetch $OUT/write   # raw counter CSVs are large

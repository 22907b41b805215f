#!/bin/bash
# One GPU-box call that produces everything profiles/ quotes for a workload (default config2): the plain bench
# line (host bases in -> host hits out, default lanes), then -- with MTSV_LANES=1 and `bench.py --resident-only`,
# so that every launch is part of a whole-batch pass and kernels of different lanes do not overlap in the trace --
# the rocprofv3 kernel trace, FETCH_SIZE / WRITE_SIZE in separate counter passes, and the SQ counters of the
# pipeline kernels.  Output under gpurun_out/prof_<workload>/ .
set -e
WL=${1:-config2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$WL
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 bench.py --workload $WL > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
export MTSV_LANES=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --resident-only > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "trace done"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --resident-only > /dev/null 2> $OUT/fetch.err
echo "fetch done"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --resident-only > /dev/null 2> $OUT/write.err
echo "write done"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o p -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --resident-only > /dev/null 2> $OUT/sq.err
echo "sq done"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d $OUT/tlb -o p -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --resident-only > /dev/null 2> $OUT/tlb.err
echo "tlb done"
cd $ROOT
python3 tools/profile_summary.py trace $OUT/trace $OUT/kernel_stats.csv
python3 tools/profile_summary.py pmc $OUT/fetch $OUT/write $WL $OUT/hbm_pmc.json $OUT/hbm_traffic.json
python3 tools/profile_summary.py sq $OUT/sq $OUT/trace $OUT/sq_counters.txt $OUT/hbm_traffic.json
python3 tools/profile_summary.py tlb $OUT/tlb $OUT/tlb_counters.txt
rm -rf $OUT/fetch $OUT/write $OUT/sq $OUT/tlb   # raw counter CSVs are large

#!/bin/bash
# Same-box comparison of the host path's transfer forms (boxes of the pool differ by more than the forms do): ten steps
# each after one warm-up, a process per setting, twice over.
#     gpurun -- 'bash tools/ab_host_path.sh'
run() { env "$@" timeout -k 10 200 python3 tools/host_path_trace.py --warm 10 --sweep "X=0" 2>/dev/null | tail -1 | python3 -c "
import json,sys,statistics
d=json.loads(sys.stdin.read()); m=d['step_ms'][1:]; print(' median %.2f min %.2f max %.2f ms per step' % (statistics.median(m), min(m), max(m)))"; }
for rep in 1 2; do
echo -n "plain bytes, k_normalise      "; run MTSV_H2D_PLAIN=1
echo -n "4-bit codes,  8 pack threads  "; run MTSV_PACK_THREADS=8
echo -n "4-bit codes, 10 pack threads  "; run MTSV_PACK_THREADS=10
echo -n "4-bit codes, 12 pack threads  "; run MTSV_PACK_THREADS=12
done
grep "nr_throttled\|nr_periods" /sys/fs/cgroup/cpu.stat

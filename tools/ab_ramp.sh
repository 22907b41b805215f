#!/bin/bash
# Same-box comparison of the host path's range policy (MTSV_RAMP_FLOOR: reads of the first ranges; SLICE: workspace reads,
# i.e. three times the largest range): ten steps each after one warm-up, a process per setting, twice over.
run() { env "$@" timeout -k 10 200 python3 tools/host_path_trace.py --warm 10 --sweep "X=0" 2>/dev/null | tail -1 | python3 -c "
import json,sys,statistics
d=json.loads(sys.stdin.read()); m=d['step_ms'][1:]; print(' median %.2f min %.2f max %.2f' % (statistics.median(m), min(m), max(m)))"; }
for rep in 1 2; do
echo -n "floor 256K            "; run MTSV_RAMP_FLOOR=262144
echo -n "floor 768K            "; run MTSV_RAMP_FLOOR=786432
echo -n "floor 1M              "; run MTSV_RAMP_FLOOR=1048576
echo -n "floor 1.5M            "; run MTSV_RAMP_FLOOR=1572864
echo -n "floor 1M, slice 4M    "; run MTSV_RAMP_FLOOR=1048576 SLICE=4194304
echo -n "floor 1.5M, slice 5M  "; run MTSV_RAMP_FLOOR=1572864 SLICE=5242880
done

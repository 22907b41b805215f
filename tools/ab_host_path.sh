run() { env "$@" timeout -k 10 200 python3 tools/host_path_trace.py --warm 10 --sweep "X=0" 2>/dev/null | tail -1 | python3 -c "
import json,sys,statistics
d=json.loads(sys.stdin.read()); m=d['step_ms'][1:]; print(' median %.2f min %.2f max %.2f' % (statistics.median(m), min(m), max(m)))"; }
for rep in 1 2; do
echo -n "plain      "; run MTSV_H2D_PLAIN=1
echo -n "avx2-12    "; run MTSV_PACK_AVX2=1 MTSV_PACK_THREADS=12
echo -n "avx512-12  "; run MTSV_PACK_THREADS=12
echo -n "avx2-10    "; run MTSV_PACK_AVX2=1 MTSV_PACK_THREADS=10
echo -n "avx512-10  "; run MTSV_PACK_THREADS=10
echo -n "avx512-14  "; run MTSV_PACK_THREADS=14
done
grep -c . /sys/fs/cgroup/cpu.stat; grep "nr_throttled\|nr_periods" /sys/fs/cgroup/cpu.stat

set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, mtsv_tools_amd as M
n = 8_000_000
ix = M.MGIndex.synth(0x6D747376, 64, 4, 270000, threads=32)
ix.write("/tmp/cli.idx")
bases, off = M.synth_reads(ix, 5, n, 150)
b = bases.reshape(n, 150)
q = b"I" * 150
with open("/tmp/cli.fastq", "wb") as f:
    for lo in range(0, n, 100000):
        f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b[i].tobytes(), q) for i in range(lo, min(n, lo + 100000))))
PY
MTSV_TRACE=1 MTSV_CLI_TIMING=1 mtsv_tools_amd/bin/mtsv-binner --fastq /tmp/cli.fastq -i /tmp/cli.idx -m /tmp/cli.out --force-overwrite --batch-reads 1048576 > gpurun_out/r3_clitrace.out 2> gpurun_out/r3_clitrace.err
grep -c "run_host. entered" gpurun_out/r3_clitrace.err

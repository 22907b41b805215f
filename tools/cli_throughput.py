"""Measure the end-to-end rate of bin/mtsv-binner on a synthetic FASTQ (plain and gzip)."""
import gzip, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mtsv_tools_amd as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ix = M.MGIndex.synth(0x6D747376, 64, 4, 270000, threads=32)
ix.write("/tmp/cli.idx")
bases, off = M.synth_reads(ix, 5, n, 150)
b = bases.reshape(n, 150)
t0 = time.time()
with open("/tmp/cli.fastq", "wb") as f:
    q = b"I" * 150
    for lo in range(0, n, 100000):
        f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b[i].tobytes(), q) for i in range(lo, min(n, lo + 100000))))
print("fastq written", time.time() - t0, os.path.getsize("/tmp/cli.fastq") / 1e6, "MB")
subprocess.check_call("gzip -1 -k -f /tmp/cli.fastq", shell=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtsv_tools_amd", "bin", "mtsv-binner")
for path in ("/tmp/cli.fastq", "/tmp/cli.fastq.gz"):
    for extra in ([], ["--batch-reads", "262144"]):
        t0 = time.time()
        subprocess.check_call([exe, "--fastq", path, "-i", "/tmp/cli.idx", "-m", "/tmp/cli.out", "--force-overwrite", *extra],
                              stdout=subprocess.DEVNULL)
        dt = time.time() - t0
        print(f"{os.path.basename(path)} {extra}: {dt:.2f} s  {n / dt / 1e6:.2f} M reads/s  lines={sum(1 for _ in open('/tmp/cli.out'))}")

"""Ingest rate of bin/mtsv-binner --parse-only on a gzip-compressed synthetic FASTQ: zlib's one stream
(MTSV_SERIAL_GZIP=1) against the parallel inflater (pgzip.hpp), over helper-thread counts.  No GPU needed."""
import os, subprocess, sys, time
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtsv_tools_amd", "bin", "mtsv-binner")
import numpy as np
rng = np.random.default_rng(3)
path = "/tmp/gzrate.fastq"
t0 = time.time()
with open(path, "wb") as f:
    # 100 000 distinct records (bases and qualities drawn with numpy: the generation used to take longer than
    # everything it measures), repeated with running read names
    nb = 100000
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(nb, 150))]
    quals = np.frombuffer(b"FFFFFFF:,#", dtype=np.uint8)[rng.integers(0, 10, size=(nb, 150))]
    block = [(seqs[i].tobytes().decode(), quals[i].tobytes().decode()) for i in range(nb)]
    for lo in range(0, n, nb):
        f.write("".join(f"@read{lo + k} 1:N:0:ACGT\n{s}\n+\n{q}\n" for k, (s, q) in enumerate(block[: min(nb, n - lo)])).encode())
subprocess.check_call(f"gzip -6 -k -f {path}", shell=True)
print(f"{n} reads, {os.path.getsize(path) / 1e6:.0f} MB FASTQ, {os.path.getsize(path + '.gz') / 1e6:.0f} MB gzip -6, made in {time.time() - t0:.0f} s", flush=True)
def run(env):
    e = dict(os.environ, MTSV_PARSE_NOHASH="1", **env)
    t0 = time.time()
    out = subprocess.run([exe, "--parse-only", "--fastq", path + ".gz"], env=e, capture_output=True, text=True, check=True).stdout
    dt = time.time() - t0
    print(f"{env}: {dt:.2f} s = {n / dt / 1e6:.2f} M reads/s  ({out.strip().splitlines()[-1][:30]})", flush=True)
run({"MTSV_SERIAL_GZIP": "1", "MTSV_HOST_THREADS": "8"})
for th in (4, 8, 16, 32, 64):
    run({"MTSV_HOST_THREADS": str(th)})
run({"MTSV_HOST_THREADS": "32", "MTSV_PGZIP_CHUNK": str(2 << 20)})
run({"MTSV_HOST_THREADS": "32", "MTSV_PGZIP_CHUNK": str(8 << 20)})

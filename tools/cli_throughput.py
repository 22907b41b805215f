"""Measure the end-to-end rate of bin/mtsv-binner on a synthetic FASTQ (plain and gzip).

    python3 tools/cli_throughput.py [n_reads]      (CLI_QUICK=1: skip the parse-only / thread-count legs)

Every run prints the command line's own per-stage times (MTSV_CLI_TIMING=1) and the CPU time it used."""
import os
import re
import resource
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
do_gz = n <= 10_000_000
if do_gz:
    subprocess.check_call("gzip -1 -k -f /tmp/cli.fastq", shell=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtsv_tools_amd", "bin", "mtsv-binner")


def run(path, extra=(), env=None):
    e = dict(os.environ, **(env or {}))
    if os.path.exists("/tmp/cli.out") and not os.environ.get("CLI_KEEP_OUT"):
        os.remove("/tmp/cli.out")   # a fresh results file, as in a first run (closing a truncated 0.9 GB file costs ext4 70 ms)
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    out = subprocess.run([exe, "--fastq", path, "-i", "/tmp/cli.idx", "-m", "/tmp/cli.out", "--force-overwrite", *extra],
                         stdout=subprocess.PIPE, text=True, env=dict(e, MTSV_CLI_TIMING="1"), check=True).stdout
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    secs = float(re.search(r"Took ([0-9.]+) seconds", out).group(1))   # queries only (after index load + upload)
    lines = sum(1 for _ in open("/tmp/cli.out"))
    print(f"{os.path.basename(path)} {list(extra)} {env or ''}: wall {dt:.2f} s, queries {secs:.2f} s = {n / secs / 1e6:.2f} M reads/s, "
          f"lines={lines}, cpu {cpu:.1f} s = {cpu / dt:.1f} cores", flush=True)
    return lines


if not os.environ.get("CLI_QUICK"):
    for env in ({"MTSV_SERIAL_INGEST": "1"}, {"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"}):
        t0 = time.time()
        subprocess.check_call([exe, "--parse-only", "--fastq", "/tmp/cli.fastq"], stdout=subprocess.DEVNULL,
                              env=dict(os.environ, MTSV_PARSE_NOHASH="1", **env))
        print(f"parse-only {env}: {time.time() - t0:.2f} s", flush=True)
    ref = run("/tmp/cli.fastq", env={"MTSV_SERIAL_INGEST": "1", "MTSV_HOST_THREADS": "1"})
    for env in ({"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"}):
        assert run("/tmp/cli.fastq", env=env) == ref
for br in ("65536", "131072", "262144"):
    run("/tmp/cli.fastq", ["--batch-reads", br])
for wk in ("1", "2", "3"):
    run("/tmp/cli.fastq", env={"MTSV_CLI_WORKERS": wk})
for _ in range(2):
    run("/tmp/cli.fastq", env={"MTSV_INGEST_PREAD": "1"})   # blocks copied out of the page cache with pread before they are parsed
    run("/tmp/cli.fastq")
run("/tmp/cli.fastq", env={"MTSV_CLI_PACKED": "1"})     # the bases packed to 4-bit codes on the host before they cross PCIe
run("/tmp/cli.fastq", env={"MTSV_CLI_COLD": "1"})       # workspaces not sized / warmed before the clock starts
for gr in ("262144", "1048576"):
    run("/tmp/cli.fastq", env={"MTSV_CLI_GROUP_READS": gr})
    run("/tmp/cli.fastq", env={"MTSV_CLI_GROUP_READS": gr, "MTSV_CLI_WORKERS": "3"})
run("/tmp/cli.fastq")
run("/tmp/cli.fastq", env={"MTSV_CLI_MARKS": "1"})   # the same with the time line of the run on stderr
run("/tmp/cli.fastq", ["--batch-reads", "262144"], env={"MTSV_CLI_PAGEABLE": "1"})
run("/tmp/cli.fastq", ["--devices", "0,0"])
if do_gz:
    run("/tmp/cli.fastq.gz")

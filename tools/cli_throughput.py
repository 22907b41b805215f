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
do_gz = n <= 10_000_000
if do_gz:
    subprocess.check_call("gzip -1 -k -f /tmp/cli.fastq", shell=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtsv_tools_amd", "bin", "mtsv-binner")
import re


def run(path, extra=(), env=None):
    import resource
    e = dict(os.environ, **(env or {}))
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    out = subprocess.run([exe, "--fastq", path, "-i", "/tmp/cli.idx", "-m", "/tmp/cli.out", "--force-overwrite", *extra],
                         stdout=subprocess.PIPE, text=True, env=dict(e, MTSV_CLI_TIMING="1"), check=True).stdout
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    print(f"    cpu {cpu:.2f} s over {dt:.2f} s wall = {cpu / dt:.1f} cores (user {ru1.ru_utime - ru0.ru_utime:.2f}, sys {ru1.ru_stime - ru0.ru_stime:.2f})")
    q = float(re.search(r"Took ([0-9.]+) seconds", out).group(1))   # queries only (after index load + upload)
    lines = sum(1 for _ in open("/tmp/cli.out"))
    print(f"{os.path.basename(path)} {list(extra)} {env or ''}: wall {dt:.2f} s, queries {q:.2f} s = {n / q / 1e6:.2f} M reads/s, lines={lines}", flush=True)
    return lines


for env in (() if os.environ.get("CLI_QUICK") else ({"MTSV_SERIAL_INGEST": "1"}, {"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"})):
    t0 = time.time()
    subprocess.check_call([exe, "--parse-only", "--fastq", "/tmp/cli.fastq"], stdout=subprocess.DEVNULL, env=dict(os.environ, MTSV_PARSE_NOHASH="1", **env))
    print(f"parse-only {env}: {time.time() - t0:.2f} s", flush=True)
for br in ("131072", "262144", "524288", "1048576"):
    run("/tmp/cli.fastq", ["--batch-reads", br])
run("/tmp/cli.fastq", ["--batch-reads", "262144"], env={"MTSV_CLI_PAGEABLE": "1"})
run("/tmp/cli.fastq", ["--devices", "0,0"])
run("/tmp/cli.fastq", ["--devices", "0,0,0"])
if do_gz:
    subprocess.check_call("gzip -1 -k -f /tmp/cli.fastq", shell=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtsv_tools_amd", "bin", "mtsv-binner")
import re


def run(path, extra=(), env=None):
    import resource
    e = dict(os.environ, **(env or {}))
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    out = subprocess.run([exe, "--fastq", path, "-i", "/tmp/cli.idx", "-m", "/tmp/cli.out", "--force-overwrite", *extra],
                         stdout=subprocess.PIPE, text=True, env=dict(e, MTSV_CLI_TIMING="1"), check=True).stdout
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    print(f"    cpu {cpu:.2f} s over {dt:.2f} s wall = {cpu / dt:.1f} cores (user {ru1.ru_utime - ru0.ru_utime:.2f}, sys {ru1.ru_stime - ru0.ru_stime:.2f})")
    q = float(re.search(r"Took ([0-9.]+) seconds", out).group(1))   # queries only (after index load + upload)
    lines = sum(1 for _ in open("/tmp/cli.out"))
    print(f"{os.path.basename(path)} {list(extra)} {env or ''}: wall {dt:.2f} s, queries {q:.2f} s = {n / q / 1e6:.2f} M reads/s, lines={lines}", flush=True)
    return lines


for env in (() if os.environ.get("CLI_QUICK") else ({"MTSV_SERIAL_INGEST": "1"}, {"MTSV_HOST_THREADS": "2"}, {"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"})):
    t0 = time.time()
    subprocess.check_call([exe, "--parse-only", "--fastq", "/tmp/cli.fastq"], stdout=subprocess.DEVNULL, env=dict(os.environ, MTSV_PARSE_NOHASH="1", **env))
    print(f"parse-only {env}: {time.time() - t0:.2f} s", flush=True)
if not os.environ.get("CLI_QUICK"):
    ref = run("/tmp/cli.fastq", env={"MTSV_SERIAL_INGEST": "1", "MTSV_HOST_THREADS": "1"})
    for env in ({"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"}):
        assert run("/tmp/cli.fastq", env=env) == ref
for br in ("131072", "262144", "524288", "1048576", "2097152"):
    run("/tmp/cli.fastq", ["--batch-reads", br])
run("/tmp/cli.fastq", ["--batch-reads", "524288", "--devices", "0,0"])
run("/tmp/cli.fastq", ["--batch-reads", "1048576", "--devices", "0,0"])
if do_gz:
    subprocess.check_call("gzip -1 -k -f /tmp/cli.fastq", shell=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtsv_tools_amd", "bin", "mtsv-binner")
import re


def run(path, extra=(), env=None):
    import resource
    e = dict(os.environ, **(env or {}))
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    out = subprocess.run([exe, "--fastq", path, "-i", "/tmp/cli.idx", "-m", "/tmp/cli.out", "--force-overwrite", *extra],
                         stdout=subprocess.PIPE, text=True, env=dict(e, MTSV_CLI_TIMING="1"), check=True).stdout
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    print(f"    cpu {cpu:.2f} s over {dt:.2f} s wall = {cpu / dt:.1f} cores (user {ru1.ru_utime - ru0.ru_utime:.2f}, sys {ru1.ru_stime - ru0.ru_stime:.2f})")
    q = float(re.search(r"Took ([0-9.]+) seconds", out).group(1))   # queries only (after index load + upload)
    lines = sum(1 for _ in open("/tmp/cli.out"))
    print(f"{os.path.basename(path)} {list(extra)} {env or ''}: wall {dt:.2f} s, queries {q:.2f} s = {n / q / 1e6:.2f} M reads/s, lines={lines}", flush=True)
    return lines


for env in (() if os.environ.get("CLI_QUICK") else ({"MTSV_SERIAL_INGEST": "1"}, {"MTSV_HOST_THREADS": "2"}, {"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"})):
    t0 = time.time()
    subprocess.check_call([exe, "--parse-only", "--fastq", "/tmp/cli.fastq"], stdout=subprocess.DEVNULL, env=dict(os.environ, MTSV_PARSE_NOHASH="1", **env))
    print(f"parse-only {env}: {time.time() - t0:.2f} s", flush=True)
if not os.environ.get("CLI_QUICK"):
    ref = run("/tmp/cli.fastq", env={"MTSV_SERIAL_INGEST": "1", "MTSV_HOST_THREADS": "1"})
    for env in ({"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"}):
        assert run("/tmp/cli.fastq", env=env) == ref
run("/tmp/cli.fastq", ["--batch-reads", "262144"])
for lanes in ("1", "2"):
    for dev in ("0,0", "0,0,0", "0,0,0,0"):
        run("/tmp/cli.fastq", ["--devices", dev], env={"MTSV_LANES": lanes})
run("/tmp/cli.fastq", ["--devices", "0,0,0", "--batch-reads", "524288"], env={"MTSV_LANES": "1"})
run("/tmp/cli.fastq", ["--devices", "0,0,0", "--batch-reads", "131072"], env={"MTSV_LANES": "1"})
if do_gz:
    subprocess.check_call("gzip -1 -k -f /tmp/cli.fastq", shell=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtsv_tools_amd", "bin", "mtsv-binner")
import re


def run(path, extra=(), env=None):
    import resource
    e = dict(os.environ, **(env or {}))
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    out = subprocess.run([exe, "--fastq", path, "-i", "/tmp/cli.idx", "-m", "/tmp/cli.out", "--force-overwrite", *extra],
                         stdout=subprocess.PIPE, text=True, env=dict(e, MTSV_CLI_TIMING="1"), check=True).stdout
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    print(f"    cpu {cpu:.2f} s over {dt:.2f} s wall = {cpu / dt:.1f} cores (user {ru1.ru_utime - ru0.ru_utime:.2f}, sys {ru1.ru_stime - ru0.ru_stime:.2f})")
    q = float(re.search(r"Took ([0-9.]+) seconds", out).group(1))   # queries only (after index load + upload)
    lines = sum(1 for _ in open("/tmp/cli.out"))
    print(f"{os.path.basename(path)} {list(extra)} {env or ''}: wall {dt:.2f} s, queries {q:.2f} s = {n / q / 1e6:.2f} M reads/s, lines={lines}", flush=True)
    return lines


for env in (() if os.environ.get("CLI_QUICK") else ({"MTSV_SERIAL_INGEST": "1"}, {"MTSV_HOST_THREADS": "2"}, {"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"})):
    t0 = time.time()
    subprocess.check_call([exe, "--parse-only", "--fastq", "/tmp/cli.fastq"], stdout=subprocess.DEVNULL, env=dict(os.environ, MTSV_PARSE_NOHASH="1", **env))
    print(f"parse-only {env}: {time.time() - t0:.2f} s", flush=True)
if not os.environ.get("CLI_QUICK"):
    ref = run("/tmp/cli.fastq", env={"MTSV_SERIAL_INGEST": "1", "MTSV_HOST_THREADS": "1"})
    for env in ({"MTSV_HOST_THREADS": "4"}, {"MTSV_HOST_THREADS": "8"}, {"MTSV_HOST_THREADS": "16"}):
        assert run("/tmp/cli.fastq", env=env) == ref
run("/tmp/cli.fastq", ["--batch-reads", "262144"])
run("/tmp/cli.fastq", ["--batch-reads", "262144"], env={"MTSV_HOST_THREADS": "8"})
run("/tmp/cli.fastq", ["--batch-reads", "262144"], env={"MTSV_HOST_THREADS": "4"})
run("/tmp/cli.fastq", ["--batch-reads", "1048576"], env={"MTSV_HOST_THREADS": "6"})
run("/tmp/cli.fastq", ["--batch-reads", "262144"], env={"MTSV_CLI_PAGEABLE": "1"})
run("/tmp/cli.fastq", ["--batch-reads", "524288"])
run("/tmp/cli.fastq", ["--batch-reads", "1048576"])
if os.environ.get("CLI_TWO_WORKERS"):
    run("/tmp/cli.fastq", ["--devices", "0,0"])
    run("/tmp/cli.fastq", ["--devices", "0,0", "--batch-reads", "524288"])
    run("/tmp/cli.fastq", ["--devices", "0,0,0"])
if do_gz:
    run("/tmp/cli.fastq.gz")

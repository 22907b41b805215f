import os, re, subprocess, sys, time
sys.path.insert(0, os.getcwd())
exe = "mtsv_tools_amd/bin/mtsv-binner"
n = 32_000_000
def run(env, extra=()):
    if os.path.exists("/tmp/cli.out"): os.remove("/tmp/cli.out")
    e = dict(os.environ, **env)
    out = subprocess.run([exe, "--fastq", "/tmp/cli.fastq", "-i", "/tmp/cli.idx", "-m", "/tmp/cli.out", *extra], stdout=subprocess.PIPE, text=True, env=e, check=True).stdout
    return float(re.search(r"Took ([0-9.]+) seconds", out).group(1))
cfgs = [({}, ()), ({"MTSV_CLI_WORKERS": "3"}, ()), ({"MTSV_CLI_WORKERS": "3", "MTSV_CLI_GROUP_READS": "1048576"}, ()), ({"MTSV_CLI_GROUP_READS": "1048576"}, ()),
        ({"MTSV_CLI_WORKERS": "4"}, ()), ({}, ("-t", "12")), ({"MTSV_CLI_WORKERS": "3"}, ("-t", "12")), ({"MTSV_CLI_WORKERS": "3", "MTSV_CLI_GROUP_READS": "786432"}, ())]
for rep in range(2):
    for env, extra in cfgs:
        s = run(env, extra)
        print(rep, env, extra, f"{s:.3f} s = {n / s / 1e6:.1f} M reads/s", flush=True)

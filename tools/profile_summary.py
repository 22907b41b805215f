"""Summarise rocprofv3 CSV output of a `bench.py --no-extras` run into the JSON files kept under profiles/.

  python tools/profile_summary.py trace  <dir with *_kernel_trace.csv>  out.csv
      per-kernel calls / total / per-pipeline-pass duration (a pass = one k_search launch)
  python tools/profile_summary.py pmc <dir FETCH_SIZE run> <dir WRITE_SIZE run> workload out.json traffic.json
      per-kernel FETCH_SIZE / WRITE_SIZE sums and the per-step HBM bytes bench.py quotes as roofline.traffic
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROUND = os.environ.get("MTSV_PROFILE_ROUND", "r03")
COMMIT = os.environ.get("MTSV_PROFILE_COMMIT", "unknown")


def norm(name):
    name = name.replace("void ", "").replace("mtsv::(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    if name.startswith("k_search_listed"):
        return name  # (the second kernel of the stage: the slots with an N in their table part)
    return "k_search" if name.startswith("k_search") else name  # k_search_fast<KK> / k_search: one launch per pass


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[0]


def trace(d, out):
    dur, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(find(d, "kernel_trace.csv"))):
        k = norm(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        calls[k] += 1
    passes = max(1, calls.get("k_search", 1))
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,ms_per_pipeline_pass,avg_ms_per_call\n")
        for k in sorted(dur, key=lambda k: -dur[k]):
            in_pipeline = k.startswith("k_") and k not in ("k_expand_sa", "k_kmer_level", "k_kmer_level1")
            f.write(f"{k},{calls[k]},{dur[k]:.3f},{dur[k] / passes if in_pipeline else float('nan'):.3f},{dur[k] / calls[k]:.4f}\n")
    print(f"{passes} pipeline passes; wrote {out}")


def pmc(dfetch, dwrite, workload, out, traffic):
    per = {}
    passes = 1
    for name, d in (("FETCH_SIZE", dfetch), ("WRITE_SIZE", dwrite)):
        acc, calls = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if r["Counter_Name"] != name:
                continue
            k = norm(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"])
            calls[k] += 1
        passes = max(1, calls.get("k_search", 1))
        per[name + "_KB"] = {k: {"sum": acc[k], "dispatches": calls[k]} for k in acc}
    def pipeline(kind):
        return sum(v["sum"] for k, v in per[kind].items()
                   if k.startswith("k_") and k not in ("k_expand_sa", "k_kmer_level", "k_kmer_level1")) * 1024 / passes
    fetch, write = pipeline("FETCH_SIZE_KB"), pipeline("WRITE_SIZE_KB")
    def kernel_bytes(prefix):
        f = sum(v["sum"] for k, v in per["FETCH_SIZE_KB"].items() if k.startswith(prefix))
        w = sum(v["sum"] for k, v in per["WRITE_SIZE_KB"].items() if k.startswith(prefix))
        return (f + w) * 1024 / passes
    summary = {
        "workload": workload, "dev_flags": 0, "commit": COMMIT, "k_sw_pairs_bytes_per_step": kernel_bytes("k_sw_pairs"),
        "k_edit_myers_bytes_per_step": kernel_bytes("k_edit_myers"),
        "k_search_bytes_per_step": kernel_bytes("k_search"),
        "source": f"{os.path.join('profiles', ROUND + '_' + workload + '_hbm_pmc.json')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, `MTSV_LANES=1 bench.py --resident-only`; per-step = per-kernel sums / pipeline passes in the profiled process)",
        "fetch_bytes_per_step_raw": fetch, "write_bytes_per_step": write, "hbm_bytes_per_step": fetch + write,
        "note": "FETCH_SIZE is in KiB and, per MI355X_MICROARCH.md, under-reports wide coalesced streams by 2x on gfx950; this path's 16-B-per-lane random gathers are uncalibrated, so the raw counter is quoted (with the 2x correction: fetch*2+write)",
        "hbm_bytes_per_step_with_2x_fetch_correction": 2 * fetch + write,
    }
    json.dump({"per_kernel": per, "summary": summary, "pipeline_passes_in_process": passes}, open(out, "w"), indent=1)
    json.dump(summary, open(traffic, "w"), indent=1)
    print(f"{passes} passes: fetch {fetch / 1e9:.1f} GB, write {write / 1e9:.1f} GB per step")


def sq(dsq, dtrace, out, traffic=None):
    """SQ counters per pipeline kernel (sums over its launches in the profiled process) next to its duration in the
    same process' kernel trace; VALU rate = SQ_INSTS_VALU / 1024 SIMDs / duration."""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(find(dsq, "counter_collection.csv"))):
        acc[norm(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(find(dsq, "kernel_trace.csv"))):
        dur[norm(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    with open(out, "w") as f:
        f.write("MTSV_LANES=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY "
                "SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES\n  -- python3 bench.py --steps 1 --warmup 0 --resident-only     (one whole-batch pipeline pass; "
                "sums over a kernel's launches; durations from the same profiled process)\n"
                "VALU rate = SQ_INSTS_VALU / 1024 SIMDs / duration; architectural ceiling 600 wave-instr/us/SIMD (one wave64 VALU op per 4 cycles at 2.4 GHz)\n")
        for k in sorted(dur, key=lambda k: -dur[k]):
            if not k.startswith("k_") or k in ("k_expand_sa", "k_kmer_level", "k_kmer_level1"):
                continue
            c = acc[k]
            valu = c.get("SQ_INSTS_VALU", 0.0)
            rate = valu / 1024 / (dur[k] * 1e3) if dur[k] else 0.0
            wait = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
            f.write(f"{k:28s} {dur[k]:7.2f} ms  VALU {valu:.3g} ({rate:.0f}/us/SIMD)  SALU {c.get('SQ_INSTS_SALU', 0):.3g}  "
                    f"LDS {c.get('SQ_INSTS_LDS', 0):.3g}  VMEM_RD {c.get('SQ_INSTS_VMEM_RD', 0):.3g}  SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES {wait:.2f}\n")
    print("wrote", out)
    if traffic and os.path.exists(traffic):  # what bench.py quotes as roofline.valu.issued_frac
        t = json.load(open(traffic))
        passes = 1  # the SQ run is one step
        my = [k for k in dur if k.startswith("k_edit_myers")]
        t["k_edit_myers_valu_per_step"] = sum(acc[k].get("SQ_INSTS_VALU", 0.0) for k in my) / passes
        t["k_edit_myers_ms_per_step"] = sum(dur[k] for k in my) / passes
        json.dump(t, open(traffic, "w"), indent=1)


def tlb(dtlb, out):
    """TCP_UTCL1 (the per-CU first-level address translation cache) requests / hits / misses per pipeline kernel, one pass"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(find(dtlb, "counter_collection.csv"))):
        acc[norm(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(find(dtlb, "kernel_trace.csv"))):
        dur[norm(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    with open(out, "w") as f:
        f.write("MTSV_LANES=1 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum\n"
                "  -- python3 bench.py --steps 1 --warmup 0 --resident-only     (one whole-batch pipeline pass; commit " + COMMIT + ")\n")
        for k in sorted(dur, key=lambda k: -dur[k]):
            if not k.startswith("k_") or k in ("k_expand_sa", "k_kmer_level", "k_kmer_level1"):
                continue
            c = acc[k]
            req, hit, miss = c.get("TCP_UTCL1_REQUEST_sum", 0.0), c.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0.0), c.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0.0)
            f.write(f"{k:28s} {dur[k]:7.2f} ms  UTCL1 requests {req:.3g}  hits {hit:.3g}  misses {miss:.3g}  miss ratio {miss / req if req else float('nan'):.3f}  "
                    f"misses per us {miss / (dur[k] * 1e3) if dur[k] else 0:.0f}\n")
    print("wrote", out)


if __name__ == "__main__":
    if sys.argv[1] == "tlb":
        tlb(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "sq":
        sq(*sys.argv[2:6])
    else:
        pmc(*sys.argv[2:7])

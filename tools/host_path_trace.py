#!/usr/bin/env python3
"""Timeline of the host path (mtsv_batch_run_host) on one workload: warm steps, then a pause and ONE traced step
(MTSV_TRACE=1 lines on stderr); run it under `rocprofv3 --kernel-trace --memory-copy-trace` and feed the CSVs to
`tools/timeline.py` for the per-stream picture of that last step.

    python3 tools/host_path_trace.py [--workload config2] [--warm 3]
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
import mtsv_tools_amd as M  # noqa: E402
from mtsv_tools_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--verify-mode", type=int, default=0)
    ap.add_argument("--resident-sweep", default="", help="';'-separated settings 'K=V,K=V': the RESIDENT step (reads in HBM) under each")
    ap.add_argument("--sweep", default="", help="';'-separated settings, each 'K=V,K=V' (environment of a fresh workspace; "
                    "SLICE = workspace reads): prints the step times of each instead of tracing one")
    args = ap.parse_args()
    n_taxa, gis, seq_len, n_reads, read_len, desc = B.WORKLOADS[args.workload]
    if args.reads:
        n_reads = args.reads
    idx_path = f"/tmp/mtsv_bench_{args.workload}.idx"
    if not os.path.exists(idx_path):
        M.set_build_device(0)
        ixb = M.MGIndex.synth(B.SEED_DB, n_taxa, gis, seq_len, threads=min(32, os.cpu_count() or 8))
        M.set_build_device(-1)
        ixb.write(idx_path)
        ixb.close()
    ix = M.MGIndex.load(idx_path)
    ix.to_device(0, 0)
    bases, off = M.synth_reads(ix, seed=1000, n_reads=n_reads, read_len=read_len)
    params = M.default_params()
    pinned = M.HostBuffer(len(bases))
    pinned.array[:] = bases
    hb = None

    def workspace(slice_reads=0):
        slice_reads = slice_reads or int(os.environ.get("MTSV_BENCH_SLICE", 0)) or M.bin_batch_slice_reads(n_reads)
        b_ = M.Batch(ix, 0, min(n_reads, slice_reads), min(len(bases), slice_reads * (read_len + 8)))
        b_.set_verify_mode(args.verify_mode)
        return b_

    def step():
        t0 = time.perf_counter()
        L._check(M.lib().mtsv_batch_run_host(hb.h, pinned.array.ctypes.data, off.ctypes.data, n_reads, ctypes.byref(params)))
        out_p, out_n = ctypes.c_void_p(), ctypes.c_uint64()
        L._check(M.lib().mtsv_batch_download(hb.h, ctypes.byref(out_p), ctypes.byref(out_n)))
        M.lib().mtsv_hits_free(out_p)
        return (time.perf_counter() - t0) * 1e3

    if args.resident_sweep:
        for setting in args.resident_sweep.split(";"):
            kv = dict(x.split("=") for x in setting.split(",") if x)
            for k, v in kv.items():
                os.environ[k] = v
            rb = M.Batch(ix, 0, n_reads, len(bases))
            rb.upload(bases, off)
            rb.run(params)
            ms = []
            for _ in range(4):
                t0 = time.perf_counter()
                rb.run(params)
                ms.append(round((time.perf_counter() - t0) * 1e3, 2))
            st = rb.stats()
            print(json.dumps({"resident_setting": setting, "step_ms": ms, "best_ms": min(ms), "n_lanes": st["n_lanes"], "n_passes": st["n_passes"]}), flush=True)
            rb.close()
            for k in kv:
                del os.environ[k]
    if args.sweep:
        # the resident rate of the same kernels on this box, to compare boxes: reads in HBM, default lanes
        rb = M.Batch(ix, 0, n_reads, len(bases))
        rb.upload(bases, off)
        rb.run(params)
        t0 = time.perf_counter()
        for _ in range(3):
            rb.run(params)
        print(json.dumps({"resident_ms": round((time.perf_counter() - t0) / 3 * 1e3, 2)}), flush=True)
        rb.close()
        for setting in args.sweep.split(";"):
            kv = dict(x.split("=") for x in setting.split(",") if x)
            sl = int(kv.pop("SLICE", 0))
            for k, v in kv.items():
                os.environ[k] = v
            hb = workspace(sl)
            ms = [round(step(), 2) for _ in range(args.warm + 4)]
            print(json.dumps({"setting": setting, "step_ms": ms, "best_ms": min(ms[1:]), "n_lanes": hb.stats()["n_lanes"]}), flush=True)
            hb.close()
            for k in kv:
                del os.environ[k]
        pinned.close()
        return
    hb = workspace(int(os.environ.get("SLICE", 0)))
    warm = [round(step(), 2) for _ in range(args.warm)]
    time.sleep(0.1)
    os.environ["MTSV_TRACE"] = "1"
    traced = step()
    del os.environ["MTSV_TRACE"]
    print(json.dumps({"warm_ms": warm, "traced_ms": round(traced, 2), "stats": hb.stats()["stage_ms"]}))
    hb.close()
    pinned.close()


if __name__ == "__main__":
    main()

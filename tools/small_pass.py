#!/usr/bin/env python3
"""One small resident pass, repeated: per-kernel floors (run under rocprofv3 --kernel-trace --stats).
    MTSV_LANES=1 python3 tools/small_pass.py --reads 84000 [--max-candidates 1]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import mtsv_tools_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--reads", type=int, default=84000)
ap.add_argument("--max-candidates", type=int, default=-1)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
n_taxa, gis, seq_len, _, read_len, _ = B.WORKLOADS[args.workload]
idx_path = f"/tmp/mtsv_bench_{args.workload}.idx"
if not os.path.exists(idx_path):
    M.set_build_device(0)
    ixb = M.MGIndex.synth(B.SEED_DB, n_taxa, gis, seq_len, threads=min(32, os.cpu_count() or 8))
    M.set_build_device(-1)
    ixb.write(idx_path)
    ixb.close()
ix = M.MGIndex.load(idx_path)
ix.to_device(0, 0)
bases, off = M.synth_reads(ix, seed=1000, n_reads=args.reads, read_len=read_len)
params = M.default_params(max_candidates=args.max_candidates) if args.max_candidates >= 0 else M.default_params()
b = M.Batch(ix, 0, args.reads, len(bases))
b.upload(bases, off)
for _ in range(args.reps):
    b.run(params)
st = b.stats()
print(json.dumps({"reads": args.reads, "max_candidates": args.max_candidates, "stage_ms": st["stage_ms"],
                  **{k: st[k] for k in ("sw_diag_ms", "sw_bound_ms", "sw_sweep_ms", "edit_ms", "n_candidates", "n_verified", "n_sw_bound_refuted", "n_sw_passed", "n_hits")}}))
b.close()

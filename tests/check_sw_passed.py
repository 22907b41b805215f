#!/usr/bin/env python3
"""The SW predicate of index.rs:406 on a benchmark workload, in aggregate: the number of candidates the prefilter
kernels pass on (stats n_sw_passed) against the number of edit distances the oracle computes (n_edit), with the
first-round bounds as a kernel of their own (k_sw_diag) and inside k_sw_pairs, on reads that are not settled by
their N count.  Usage: tests/check_sw_passed.py [config1|config2] [n_reads]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mtsv_tools_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

SEED_DB = 0x6D747376
WORKLOADS = {"config0": (8, 2, 17500, 100), "config1": (256, 4, 270_000, 150), "config2": (1024, 4, 674_000, 150)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config1"
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    n_taxa, gis, seq_len, L = WORKLOADS[name]
    path = f"/tmp/mtsv_bench_{name}.idx"
    if not (os.path.exists(path) and int.from_bytes(open(path, "rb").read(8), "little") == n_taxa * gis * seq_len + 1):
        M.set_build_device(0)
        ix = M.MGIndex.synth(SEED_DB, n_taxa, gis, seq_len, threads=min(32, os.cpu_count() or 8))
        M.set_build_device(-1)
        ix.write(path)
        ix.close()
    ix = M.MGIndex.load(path)
    ix.to_device(0)
    bases, off = M.synth_reads(ix, seed=1000, n_reads=n_reads, read_len=L)
    mp, op = M.default_params(), O.default_params()
    ED = math.ceil(L * op.edit_rate)
    rows = bases.reshape(n_reads, L)
    keep = (~np.isin(rows, np.frombuffer(b"ACGTacgt", dtype=np.uint8))).sum(axis=1) <= ED
    fb = np.ascontiguousarray(rows[keep]).reshape(-1)
    fo = np.arange(int(keep.sum()) + 1, dtype=np.uint64) * L
    print(f"{name}: {int(keep.sum())} of {n_reads} reads kept (N count <= {ED})", flush=True)
    got = {}
    for prepass in ("1", "0"):
        os.environ["MTSV_SW_PREPASS"] = prepass
        b = M.Batch(ix, 0, len(fo) - 1, len(fb))
        b.upload(fb, fo)
        b.run(mp)
        st = b.stats()
        got[prepass] = (st["n_verified"], st["n_sw_passed"], st["n_hits"])
        b.close()
        print("prepass", prepass, got[prepass], flush=True)
    want, ctr = O.Index.read(path).bin_batch(fb, fo, op, threads=min(16, os.cpu_count() or 8))
    print("oracle ", (ctr["n_sw"], ctr["n_edit"], len(want)), flush=True)
    ok = all(g == (ctr["n_sw"], ctr["n_edit"], len(want)) for g in got.values())
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""The library as the command line drives it: a stream of host->host calls on a megaread each, from one, two and three
workers (a workspace each) on the same device.

    python3 tools/call_stream.py [--workload quarter] [--reads 8388608] [--call-reads 1048576] [--block-reads 131072]

A call is mtsv_batch_run_host_parts on `call-reads` reads in pieces of `block-reads` (page-locked, like the command
line's blocks) followed by mtsv_batch_download without the copy into numpy (the array is freed as it comes)."""
import argparse, ctypes as C, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench as B
import mtsv_tools_amd as M
from mtsv_tools_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="quarter")
ap.add_argument("--reads", type=int, default=8 << 20)
ap.add_argument("--call-reads", type=int, default=1 << 20)
ap.add_argument("--block-reads", type=int, default=1 << 17)
ap.add_argument("--workers", default="1,2,3")
ap.add_argument("--verify-mode", type=int, default=1, help="0 reference order, 1 edit distance first (the command line's default)")
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
hb = M.HostBuffer(len(bases))
hb.array[:] = bases
params = M.default_params()
lib = L.lib()

# blocks: (address of the bases, offsets array relative to the block)
blocks = []
for lo in range(0, args.reads, args.block_reads):
    hi = min(args.reads, lo + args.block_reads)
    o = np.ascontiguousarray(off[lo:hi + 1] - off[lo])
    blocks.append((hb.ptr + int(off[lo]), o, hi - lo))
per_call = max(1, args.call_reads // args.block_reads)
calls = [blocks[i:i + per_call] for i in range(0, len(blocks), per_call)]


def one_call(batch, grp):
    k = len(grp)
    bp = (C.c_void_p * k)(*[g[0] for g in grp])
    op = (C.c_void_p * k)(*[g[1].ctypes.data for g in grp])
    nr = (C.c_uint64 * k)(*[g[2] for g in grp])
    L._check(lib.mtsv_batch_run_host_parts(batch.h, k, bp, op, nr, C.byref(params)))
    out, n = C.c_void_p(), C.c_uint64()
    L._check(lib.mtsv_batch_download(batch.h, C.byref(out), C.byref(n)))
    lib.mtsv_hits_free(out)
    return n.value


res = {"workload": args.workload, "reads": args.reads, "call_reads": args.call_reads, "block_reads": args.block_reads, "runs": []}
for nw in [int(x) for x in args.workers.split(",")]:
    batches = [M.Batch(ix, 0, args.call_reads, args.call_reads * read_len) for _ in range(nw)]
    for b in batches:
        b.set_verify_mode(args.verify_mode)
    for b in batches:  # warm: pools, workspaces
        one_call(b, calls[0])
        one_call(b, calls[0])
    nxt = [0]
    lock = threading.Lock()
    per = [[] for _ in range(nw)]
    hits = [0]

    def work(w):
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(calls):
                return
            t0 = time.perf_counter()
            h = one_call(batches[w], calls[i])
            per[w].append((time.perf_counter() - t0) * 1e3)
            with lock:
                hits[0] += h

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(w,)) for w in range(nw)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    allc = sorted(x for p in per for x in p)
    res["runs"].append({"workers": nw, "seconds": round(dt, 4), "reads_per_s": round(args.reads / dt), "calls": len(allc),
                        "call_ms_median": round(allc[len(allc) // 2], 2), "call_ms_min": round(allc[0], 2), "call_ms_max": round(allc[-1], 2),
                        "hits": hits[0]})
    print(json.dumps(res["runs"][-1]), flush=True)
    for b in batches:
        b.close()
# the same reads in ONE call
b = M.Batch(ix, 0, min(args.reads, 1 << 22), min(args.reads, 1 << 22) * read_len)
b.set_verify_mode(args.verify_mode)
one_call(b, blocks)
t0 = time.perf_counter()
h = one_call(b, blocks)
dt = time.perf_counter() - t0
res["one_call"] = {"seconds": round(dt, 4), "reads_per_s": round(args.reads / dt), "hits": h}
print(json.dumps(res["one_call"]), flush=True)
b.close()
hb.close()

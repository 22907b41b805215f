#!/usr/bin/env python3
"""bench.py -- reads/s of the mtsv-binner hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole hot path (H2D of the reads -> seed search -> locate -> coalesce ->
SW + edit verify -> hit gather -> D2H of the hits) over one batch of synthetic reads handed over in
host memory, as SURVEY.md 8(d) defines the metric; the index is resident in HBM, replicated per GPU,
and reads shard across ranks with no collective on the data path (weak scaling: every rank processes
its own batch of the same size).  The device-resident rate of the same kernels is reported beside it
(`device_resident`).  Rank 0 prints ONE JSON line.

Default workload = the configuration BASELINE.json's metric is quoted on: config2.
Workloads (BASELINE.json configs; generators of SURVEY.md 8(d), seeds fixed):
    config1  1M x 150 bp reads vs "1 GB MG-index"  (n = 2.76e8 symbols, 256 taxa x 4 GIs x 270 kb)
    config2  10M x 150 bp reads vs "10 GB MG-index" (n = 2.76e9)  -- several minutes of host-side index build
    config0  10k x 100 bp reads vs "1 MB MG-index" (n = 2.8e5)  -- plumbing / quick check
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_taxa, gis_per_taxon, seq_len, n_reads, read_len, description)
    "config0": (8, 2, 17500, 10_000, 100, "10k x 100bp reads vs 1MB MG-index (n=2.8e5)"),
    "config1": (256, 4, 270_000, 1_000_000, 150, "1M x 150bp reads vs 1GB MG-index (n=2.76e8)"),
    "quarter": (64, 4, 270_000, 1_000_000, 150, "1M x 150bp reads vs 250MB MG-index (n=6.9e7)"),
    "config2": (1024, 4, 674_000, 10_000_000, 150, "10M x 150bp reads vs 10GB MG-index (n=2.76e9)"),
}
SEED_DB = 0x6D747376
PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(ctr, n_reads):
    """bytes = 64*(2X+S) + 8H + W + L + 24R, per read, split per stage."""
    per = {k: ctr[k] / n_reads for k in ("X", "S", "H", "W", "R", "Lsum")}
    stage = {
        "search": 64.0 * 2 * per["X"] + per["Lsum"],
        "locate": 64.0 * per["S"] + 8.0 * per["H"],
        "verify": per["W"],
        "gather": 24.0 * per["R"],
    }
    stage["total"] = sum(stage.values())
    return stage, per


def host_description():
    """nproc, CPU model, and what this process may actually use of them (affinity, cgroup quota)."""
    d = {"nproc": os.cpu_count()}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                d["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    d["affinity_cpus"] = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    d["cgroup_cpu_quota"] = quota
    return d


CHUNK_SPEC = (128, 4, 674_000, 10_000_000, 150)  # per chunk: taxa, GIs per taxon, sequence length; reads of the whole job, read length


def merge_by_read(per_chunk):
    """per-chunk hit arrays -> one list ordered by (read, chunk): what mtsv_bin_batch_chunks returns"""
    allh = np.concatenate(per_chunk)
    chunk = np.concatenate([np.full(len(h), c) for c, h in enumerate(per_chunk)])
    return allh[np.lexsort((chunk, allh["read"]))]


def main_chunks(args):
    """BASELINE config 5 (SURVEY 8(e) Mode B): 8 MG-index chunks of 1/8 of the 10 GB database each, one per GPU, every
    chunk sees every read; the per-chunk hit lists are merged per read (what mtsv-collapse makes of the per-chunk result
    files, README.md:189, collapse.rs:597-625).  A step = host bases in -> merged host hits out for ALL reads.
      one process (--gpus 1): all chunks resident on this GPU, one call of mtsv_bin_batch_chunks per step (the merge is
          inside the timed call);
      torchrun, N ranks: rank r holds chunk r of N on its own GPU and bins every read against it (timed: max over
          ranks); afterwards every rank writes its result lines and rank 0 merges the N files with bin/mtsv-collapse,
          timed separately (`collapse_s`) -- the reference's own workflow for this configuration."""
    import ctypes
    import subprocess

    import torch
    import torch.distributed as dist

    import mtsv_tools_amd as M
    from mtsv_tools_amd import _lib as L

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available() or M.device_count() < 1:
        sys.exit("bench.py needs a HIP device: libmtsv_amd has no CPU path")
    shared_gpu = os.environ.get("MTSV_BENCH_SHARE_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if shared_gpu else "nccl", **({} if shared_gpu else {"device_id": torch.device("cuda", local_rank)}))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_taxa, gis, seq_len, n_reads, read_len = CHUNK_SPEC
    if args.reads:
        n_reads = args.reads
    n_chunks = world if world > 1 else args.chunks
    mine = [rank] if world > 1 else list(range(n_chunks))
    ncpu = os.cpu_count() or 8
    paths = [f"/tmp/mtsv_bench_config5_c{c}of{n_chunks}.idx" for c in range(n_chunks)]
    t0 = time.time()
    for c in mine:  # untimed setup: every chunk is its own synthetic database (seed per chunk), built on this rank's GPU
        expect_n = n_taxa * gis * seq_len + 1
        ok = False
        try:
            ok = os.path.exists(paths[c]) and int.from_bytes(open(paths[c], "rb").read(8), "little") == expect_n
        except OSError:
            pass
        if not ok:
            M.set_build_device(local_rank)
            ixb = M.MGIndex.synth(SEED_DB + 101 * (c + 1), n_taxa, gis, seq_len, threads=min(32, ncpu))
            M.set_build_device(-1)
            ixb.write(paths[c] + ".tmp")
            ixb.close()
            os.replace(paths[c] + ".tmp", paths[c])
    t_build = time.time() - t0
    barrier()
    chunks = {}
    for c in mine:
        chunks[c] = M.MGIndex.load(paths[c])
        chunks[c].to_device(local_rank, args.dev_flags)
    # reads: an equal share sampled from every chunk (seeded), concatenated in chunk order; every rank ends up with all of them
    share = [n_reads // n_chunks + (1 if c < n_reads % n_chunks else 0) for c in range(n_chunks)]
    parts = {c: M.synth_reads(chunks[c], seed=2000 + c, n_reads=share[c], read_len=read_len)[0] for c in mine}
    if world > 1:
        buf = [torch.empty(share[c] * read_len, dtype=torch.uint8, device="cpu" if shared_gpu else "cuda") for c in range(n_chunks)]
        mine_t = torch.from_numpy(parts[rank]).to(buf[0].device)
        for c in range(n_chunks):  # (shares differ by at most one read: broadcast per chunk instead of a padded all_gather)
            if c == rank:
                buf[c].copy_(mine_t)
            dist.broadcast(buf[c], src=c)
        bases = torch.cat(buf).cpu().numpy()
        del buf
    else:
        bases = np.concatenate([parts[c] for c in range(n_chunks)])
    off = (np.arange(n_reads + 1, dtype=np.uint64) * read_len)
    params = M.default_params()
    pinned = M.HostBuffer(len(bases))
    pinned.array[:] = bases
    bases_p, off_p = pinned.array.ctypes.data, off.ctypes.data
    HIT_FIELDS = ("read", "tax_id", "gi", "edit", "strand", "offset")

    if world == 1:
        handles = (ctypes.c_void_p * n_chunks)(*[chunks[c].h for c in range(n_chunks)])
        devs = (ctypes.c_int * n_chunks)(*([local_rank] * n_chunks))

        def step(keep=False):
            out_p, out_n = ctypes.c_void_p(), ctypes.c_uint64()
            L._check(M.lib().mtsv_bin_batch_chunks(handles, devs, n_chunks, bases_p, off_p, n_reads, ctypes.byref(params),
                                                   ctypes.byref(out_p), ctypes.byref(out_n)))
            if keep:
                return L._hits_from(out_p, out_n.value)
            M.lib().mtsv_hits_free(out_p)
    else:
        slice_reads = M.bin_batch_slice_reads(n_reads)
        hb = M.Batch(chunks[rank], local_rank, min(n_reads, slice_reads), min(len(bases), slice_reads * (read_len + 8)))

        def step(keep=False):
            L._check(M.lib().mtsv_batch_run_host(hb.h, bases_p, off_p, n_reads, ctypes.byref(params)))
            out_p, out_n = ctypes.c_void_p(), ctypes.c_uint64()
            L._check(M.lib().mtsv_batch_download(hb.h, ctypes.byref(out_p), ctypes.byref(out_n)))
            if keep:
                return L._hits_from(out_p, out_n.value)
            M.lib().mtsv_hits_free(out_p)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    each_ms = []
    for _ in range(args.steps):
        t_s = time.perf_counter()
        step()
        each_ms.append(round((time.perf_counter() - t_s) * 1e3, 2))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    hits = step(keep=True)  # untimed: what the checks and the collapse leg look at

    # ---- parity on a sample of the reads (every (n/ns)-th read, so that every chunk's reads are in it) ----
    ns = min(args.cpu_sample // 4, n_reads)
    idx = (np.arange(ns, dtype=np.int64) * (n_reads // ns))
    sample = bases.reshape(n_reads, read_len)[idx].reshape(-1)
    soff = (np.arange(ns + 1, dtype=np.uint64) * read_len)
    remap = np.full(n_reads, -1, dtype=np.int64)
    remap[idx] = np.arange(ns)
    mine_sample = hits[remap[hits["read"].astype(np.int64)] >= 0].copy()
    mine_sample["read"] = remap[mine_sample["read"].astype(np.int64)]
    parity, cpu, collapse = "not checked", None, None
    gathered = [mine_sample]
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine_sample, gathered, dst=0)
        # the reference's workflow: a results file per chunk, merged by mtsv-collapse
        ids = [f"r{i}" for i in range(n_reads)] if n_reads <= 2_000_000 else None
        if ids is not None:
            res = f"/tmp/mtsv_bench_config5_results_{rank}.txt"
            open(res, "w").write(M.format_results(hits, ids, False))
            dist.barrier()
            if rank == 0:
                t0 = time.perf_counter()
                subprocess.check_call([os.path.join(ROOT, "mtsv_tools_amd", "bin", "mtsv-collapse"), "-o", "/tmp/mtsv_bench_config5_collapsed.txt"]
                                      + [f"/tmp/mtsv_bench_config5_results_{r}.txt" for r in range(world)])
                collapse = {"collapse_s": time.perf_counter() - t0, "files": world,
                            "lines": sum(1 for _ in open("/tmp/mtsv_bench_config5_collapsed.txt"))}
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as O
        got = merge_by_read(gathered) if world > 1 else mine_sample
        cores = max(1, min(ncpu, args.cpu_threads or 16))
        t0 = time.perf_counter()
        per = [O.Index.read(paths[c]).bin_batch(sample, soff, O.default_params(), threads=cores)[0] for c in range(n_chunks)]
        dt = time.perf_counter() - t0
        want = merge_by_read(per)
        same = len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in HIT_FIELDS)
        parity = f"{'identical' if same else 'MISMATCH'} on {ns} sampled reads x {n_chunks} chunks ({len(want)} merged hits)"
        cpu = {"value": ns / dt, "unit": "reads/s", "cores": cores, "kind": "port", "host": host_description(),
               "sample": f"{ns} reads (every {n_reads // ns}-th) against all {n_chunks} chunks one after the other, oracle/libmtsv_oracle.so, {cores} OpenMP threads, {dt:.1f} s"}
    if world > 1:
        dist.barrier()
    if rank == 0:
        if "MISMATCH" in parity:
            print(json.dumps({"error": "parity", "parity": parity}))
            raise SystemExit("bench: merged GPU hits differ from the oracle's: " + parity)
        info = chunks[mine[0]].info()
        out = {"metric": "reads/sec (whole node), 150bp reads vs MG-index", "value": n_reads * args.steps / elapsed, "unit": "reads/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32/i16", "data": "synthetic",
               "value_region": ("one call of mtsv_bin_batch_chunks per step: host bases in -> per-read merged host hits out, all chunks on one GPU"
                                if world == 1 else "every rank: mtsv_batch_run_host + mtsv_batch_download of ALL reads against its chunk; max over ranks; "
                                                   "the merge of the per-chunk results (mtsv-collapse) is timed separately"),
               "step_ms_each": each_ms,
               "config": {"workload": f"config5: {n_reads} x {read_len}bp reads vs {n_chunks} MG-index chunks of n={info['n']:.3g} each",
                          "mode": "chunks (SURVEY 8(e) Mode B)", "chunks": n_chunks, "reads_per_step": n_reads, "read_len": read_len,
                          "chunk_symbols": info["n"], "chunk_hbm_bytes": info["device_bytes"], "chunk_kmer_k": info.get("kmer_k"),
                          "parallelism": f"{n_chunks} database chunks x {world} GPU(s), reads broadcast, hits merged per read; no collective on the data path"},
               "n_hits_merged": int(len(hits)) if world == 1 else None, "collapse": collapse,
               "cpu_baseline": cpu, "parity": parity, "setup_s": {"chunk_build": t_build}}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("MTSV_BENCH_WORKLOAD", "config2"))
    ap.add_argument("--reads", type=int, default=0, help="override reads per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="reads timed on the CPU oracle")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (default: every CPU this process may use -- "
                    "the GPU's NUMA node, capped by the cgroup's CPU quota)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the device-resident leg and the other verify order (profiling runs)")
    ap.add_argument("--resident-only", action="store_true",
                    help="profiling mode: upload the reads once and time K resident runs (mtsv_batch_run) only -- every launch in the "
                         "process then belongs to a whole-batch pipeline pass (tools/profile_round.sh); the line is marked as such")
    ap.add_argument("--pageable-input", action="store_true",
                    help="time the steps on reads in ordinary (pageable) host memory instead of memory from mtsv_host_alloc")
    ap.add_argument("--verify-mode", type=int, default=0, help="0: reference order (SW + edit per candidate), 1: edit first")
    ap.add_argument("--dev-flags", type=int, default=0, help="MTSV_DEV_* flags (1: sampled SA only, 2: no k-mer table)")
    ap.add_argument("--mode", default="reads", choices=["reads", "chunks"],
                    help="reads (default): BASELINE configs 1-4, index replicated per GPU, reads sharded.  chunks: BASELINE config 5, "
                         "the database in chunks (one per GPU), every chunk sees every read, hits merged per read")
    ap.add_argument("--chunks", type=int, default=8, help="--mode chunks on one GPU: database chunks, all resident on that GPU")
    args = ap.parse_args()
    if args.mode == "chunks":
        return main_chunks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
        args.gpus = world

    import torch
    import torch.distributed as dist

    import mtsv_tools_amd as M

    if not torch.cuda.is_available() or M.device_count() < 1:
        sys.exit("bench.py needs a HIP device: libmtsv_amd has no CPU path")
    # rehearsal knob: several ranks sharing one GPU (gloo; NCCL refuses duplicate devices) to exercise
    # the rank logic on a one-GPU box.  Never set by the driver.
    shared_gpu = os.environ.get("MTSV_BENCH_SHARE_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Host threads of a rank (uploaders, lanes, staging copies) stay on the NUMA node of their GPU: on a
    # two-socket node eight ranks staging 25 GB/s each should not cross the socket interconnect.  Best effort.
    affinity = None
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node >= 0 and os.environ.get("MTSV_BENCH_NO_AFFINITY") != "1":
            cpus = set()
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            cpus &= set(os.sched_getaffinity(0))
            if cpus:
                os.sched_setaffinity(0, cpus)
                affinity = f"NUMA node {node} of GPU {bdf} ({len(cpus)} CPUs)"
    except Exception:
        pass

    # The library packs the reads to 4-bit codes on the host before they cross PCIe when the process has the CPUs for it
    # (host_pack.hpp: ten threads at most, nine at least, else the plain bytes go).  It sees the affinity mask and the
    # cgroup's quota, not the other ranks that share them: every rank gets its share.
    if world > 1 and "MTSV_PACK_THREADS" not in os.environ:
        hd = host_description()
        usable = hd["affinity_cpus"] * (world if affinity else 1)   # (the mask was narrowed to this GPU's node above)
        if hd.get("cgroup_cpu_quota"):
            usable = min(usable, int(round(hd["cgroup_cpu_quota"])))
        os.environ["MTSV_PACK_THREADS"] = str(max(1, min(10, usable // world - 4)))

    n_taxa, gis, seq_len, n_reads, read_len, desc = WORKLOADS[args.workload]
    if args.reads:
        n_reads = args.reads
    idx_path = f"/tmp/mtsv_bench_{args.workload}.idx"
    ncpu = os.cpu_count() or 8
    build_threads = max(4, min(32, ncpu // max(1, world) if world > 1 else ncpu))

    # ---- index: rank 0 builds + writes the MG-index file, every rank loads it (drop-in format) ----
    t0 = time.time()
    index_cached = False
    if rank == 0:
        # the synthetic database is a pure function of (workload, SEED_DB): reuse a file left by an
        # earlier run on this box (the loader re-validates every invariant of the file anyway)
        expect_n = n_taxa * gis * seq_len + 1
        try:
            if os.path.exists(idx_path) and int.from_bytes(open(idx_path, "rb").read(8), "little") == expect_n:
                index_cached = True
        except OSError:
            pass
        if not index_cached:
            # untimed setup: suffix array by prefix doubling on this rank's GPU (MTSV_BENCH_HOST_BUILD=1: host threads)
            if os.environ.get("MTSV_BENCH_HOST_BUILD") != "1":
                M.set_build_device(local_rank)
            ixb = M.MGIndex.synth(SEED_DB, n_taxa, gis, seq_len, threads=min(32, ncpu))
            M.set_build_device(-1)
            ixb.write(idx_path + ".tmp")
            ixb.close()
            os.replace(idx_path + ".tmp", idx_path)
    t_build = time.time() - t0
    barrier()
    t0 = time.time()
    ix = M.MGIndex.load(idx_path)
    t_load = time.time() - t0
    t0 = time.time()
    ix.to_device(local_rank, args.dev_flags)
    t_upload = time.time() - t0
    info = ix.info()

    # ---- reads: each rank its own shard (different seed), in host memory ----
    import ctypes
    from mtsv_tools_amd import _lib as L
    bases, off = M.synth_reads(ix, seed=1000 + rank, n_reads=n_reads, read_len=read_len)
    params = M.default_params()
    HIT_FIELDS = ("read", "tax_id", "gi", "edit", "strand", "offset")

    def same_hits(a, b):
        return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in HIT_FIELDS)

    if args.resident_only:
        b = M.Batch(ix, local_rank, n_reads, len(bases))
        b.set_verify_mode(args.verify_mode)
        b.upload(bases, off)
        for _ in range(args.warmup):
            b.run(params)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            b.run(params)
        barrier()
        dt = time.perf_counter() - t0
        st = b.stats()
        print(json.dumps({"mode": "resident-only (profiling; not the benchmark line)", "metric": "reads/sec, reads resident in HBM",
                          "value": n_reads * args.steps / dt, "unit": "reads/s", "ms_per_step": dt / args.steps * 1e3,
                          "n_lanes": st.get("n_lanes", 1), "stage_ms": st["stage_ms"], "sw_prefilter_ms": st.get("sw_prefilter_ms"),
                          "config": {"workload": f"{args.workload}: {desc}", "reads": n_reads, "verify_mode": args.verify_mode},
                          "device_counters": {k: st[k] for k in ("n_seed_slots", "n_seed_hits", "n_candidates", "n_verified",
                                                                  "window_bytes", "n_hits", "sw_cell_pairs", "n_sw_passed")}}))
        b.close()
        return

    # ---- timed region = SURVEY 8(d): host bases in -> host hits out, through the C ABI ----
    # (mtsv_batch_run_host + mtsv_batch_download = what mtsv_bin_batch does on its cached workspace: the reads are
    # staged through pinned memory in slices, the slices run on the workspace's lanes, the hits arrive in pooled
    # pinned memory; the workspace is the one mtsv_bin_batch would create for this batch)
    slice_reads = int(os.environ.get("MTSV_BENCH_SLICE", 0)) or M.bin_batch_slice_reads(n_reads)
    hb = M.Batch(ix, local_rank, min(n_reads, slice_reads), min(len(bases), slice_reads * (read_len + 8)))
    hb.set_verify_mode(args.verify_mode)
    # The reads of the timed steps lie in host memory from the library's allocator (mtsv_host_alloc: page-locked, what a
    # host integrating the library parses its reads into -- INTEGRATION.md): the copy engine reads them in place.  The
    # same steps on an ordinary numpy array (staged through the library's own page-locked buffers) are timed after
    # the headline steps and reported as `pageable_input`.
    pinned = None
    if not args.pageable_input:
        try:
            pinned = M.HostBuffer(len(bases))
            pinned.array[:] = bases
        except M.MtsvError as e:  # no page-locked memory to be had: time the staged route and say so
            print(f"bench: mtsv_host_alloc failed ({e}); timing reads in ordinary memory", file=sys.stderr)
            pinned = None
            args.pageable_input = True
    bases_p, off_p = (pinned.array if pinned else bases).ctypes.data, off.ctypes.data

    split = [0.0, 0.0, 0.0]  # seconds inside mtsv_batch_run_host / mtsv_batch_download / mtsv_hits_free over the timed steps

    def host_step(keep=False):
        t_a = time.perf_counter()
        L._check(M.lib().mtsv_batch_run_host(hb.h, bases_p, off_p, n_reads, ctypes.byref(params)))
        t_b = time.perf_counter()
        out_p, out_n = ctypes.c_void_p(), ctypes.c_uint64()
        L._check(M.lib().mtsv_batch_download(hb.h, ctypes.byref(out_p), ctypes.byref(out_n)))
        t_c = time.perf_counter()
        if keep:
            return L._hits_from(out_p, out_n.value)  # copies, then frees
        M.lib().mtsv_hits_free(out_p)
        t_d = time.perf_counter()
        split[0] += t_b - t_a
        split[1] += t_c - t_b
        split[2] += t_d - t_c
        return None

    for _ in range(args.warmup):
        host_step()
    barrier()
    split[:] = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    host_stage = None
    each_ms = []
    for _ in range(args.steps):
        t_s = time.perf_counter()
        host_step()
        each_ms.append(round((time.perf_counter() - t_s) * 1e3, 2))
        sth = hb.stats()
        if host_stage is None:
            host_stage = dict(sth["stage_ms"])
        else:
            for k, v in sth["stage_ms"].items():
                host_stage[k] += v
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st_host = hb.stats()
    split_timed = list(split)  # (the pageable leg below calls host_step again)
    hits = host_step(keep=True)  # untimed: the hits the parity checks below look at
    pageable = None
    if pinned is not None and not args.no_extras and rank == 0:
        bases_p = bases.ctypes.data
        host_step()
        t1 = time.perf_counter()
        for _ in range(3):
            host_step()
        pg_dt = (time.perf_counter() - t1) / 3
        pageable = {"reads_per_s": n_reads / pg_dt, "ms_per_step": pg_dt * 1e3,
                    "note": "the same step on reads in an ordinary numpy array (3 runs after 1 warm-up): one more host memcpy per slice"}
        hits_pg = host_step(keep=True)
        if not same_hits(hits_pg, hits):
            raise SystemExit("bench: page-locked and pageable input returned different hits")
        del hits_pg
    hb.close()
    if pinned is not None:
        pinned.close()

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- device-resident rate (reads already in HBM, hits left there): the kernels alone, for the roofline ----
    resident = None
    alt = None
    stage_ms = None
    stage_note = None
    overlapped = None
    st = st_host
    if not args.no_extras:
        batch = M.Batch(ix, local_rank, n_reads, len(bases))
        batch.set_verify_mode(args.verify_mode)
        batch.upload(bases, off)
        batch.run(params)
        t1 = time.perf_counter()
        for _ in range(3):
            batch.run(params)
        res_dt = (time.perf_counter() - t1) / 3
        st = batch.stats()
        res_hits = batch.download()
        if not same_hits(res_hits, hits):
            raise SystemExit("bench: resident path and host path returned different hits")
        resident = {"reads_per_s": n_reads / res_dt, "ms_per_step": res_dt * 1e3, "n_lanes": st.get("n_lanes", 1),
                    "note": "mtsv_batch_run on reads already in HBM, hits left in HBM (3 runs after 1 warm-up); not `value`"}
        del res_hits
        # the other evaluation order of the two acceptance predicates, for information (never `value`)
        other = 1 - args.verify_mode
        batch.set_verify_mode(other)
        batch.run(params)
        t1 = time.perf_counter()
        for _ in range(3):
            batch.run(params)
        alt_dt = time.perf_counter() - t1
        alt_hits = batch.download()
        alt = {"verify_mode": ["reference", "edit_first"][other], "device_resident_reads_per_s": 3 * n_reads / alt_dt,
               "stage_ms": batch.stats()["stage_ms"], "hits_identical_to_timed_mode": bool(same_hits(alt_hits, hits))}
        if not alt["hits_identical_to_timed_mode"]:
            raise SystemExit("bench: the two verification orders returned different hits")
        del alt_hits
        batch.close()
    # per-stage kernel times: one untimed pass of the same reads through a single-lane workspace, so that a
    # stage's HIP-event span holds its own kernels only (the timed steps overlap lanes)
    os.environ["MTSV_LANES"] = "1"
    b1 = M.Batch(ix, local_rank, n_reads, len(bases))
    del os.environ["MTSV_LANES"]
    b1.set_verify_mode(args.verify_mode)
    b1.upload(bases, off)
    b1.run(params)
    b1.run(params)
    st1 = b1.stats()
    stage_ms = dict(st1["stage_ms"])
    b1.close()
    stage_note = ("HIP events on the lane's own stream; one untimed pass of the same reads through a single-lane resident "
                  f"workspace (MTSV_LANES=1); the timed steps ran {st_host.get('n_lanes', 1)} overlapping lanes over host slices")
    overlapped = {"n_lanes": st_host.get("n_lanes", 1), "stage_ms_summed_over_lanes_per_step": {k: v / args.steps for k, v in host_stage.items()}}

    value = n_reads * args.steps * world / elapsed
    step_ms = elapsed / args.steps * 1e3

    # ---- CPU baseline + algorithmic-byte counters from the oracle on a bounded sample ----
    cpu = None
    ctr_per = None
    parity = "not checked"
    stage_bytes = None
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        ns = min(args.cpu_sample, n_reads)
        oix = O.Index.read(idx_path)
        host = host_description()
        usable = host["affinity_cpus"]
        if host.get("cgroup_cpu_quota"):
            usable = max(1, min(usable, int(round(host["cgroup_cpu_quota"]))))
        cores = max(1, min(ncpu, args.cpu_threads or usable))
        t0 = time.perf_counter()
        ohits, ctr = oix.bin_batch(bases[: ns * read_len], off[: ns + 1], O.default_params(), threads=cores)
        dt = time.perf_counter() - t0
        cpu = {"value": ns / dt, "unit": "reads/s", "cores": cores, "kind": "port", "host": host,
               "cores_note": "every CPU this process may use: the CPUs of the GPU's NUMA node (affinity_cpus), capped by the cgroup's CPU quota",
               "sample": f"first {ns} reads of rank 0's batch, oracle/libmtsv_oracle.so (reference layout: byte BWT, "
                         f"Occ k=64, SA s=32, emulated striped SW, full-matrix edit DP), {cores} OpenMP threads, {dt:.1f} s"}
        stage_bytes, ctr_per = algorithmic_bytes(ctr, ns)
        # the reference's default thread count (-t 4, src/bin/mtsv-binner.rs:62) on a quarter of the sample
        ns4 = max(1, ns // 4)
        t0 = time.perf_counter()
        oix.bin_batch(bases[: ns4 * read_len], off[: ns4 + 1], O.default_params(), threads=4)
        cpu["value_4_threads"] = ns4 / (time.perf_counter() - t0)
        g = hits[hits["read"] < ns]
        same = len(g) == len(ohits) and all(np.array_equal(g[f], ohits[f]) for f in
                                            ("read", "tax_id", "gi", "edit", "strand", "offset"))
        parity = f"{'identical' if same else 'MISMATCH'} on {ns} sampled reads ({len(ohits)} hits)"

    if "MISMATCH" in parity:
        print(json.dumps({"error": "parity", "parity": parity}))
        raise SystemExit("bench: GPU hits differ from the oracle's: " + parity)

    # ---- roofline ----
    # Per-kernel times of one pass: HIP events on the lane's own stream, single-lane resident workspace (they agree
    # with the rocprofv3 kernel trace under profiles/).  The largest kernel is priced against the roof that binds it:
    #   k_search_fast  -- random 64-byte gathers in HBM (k-mer table entry + rank blocks): bound "hbm".  achieved =
    #                     the bytes its own layout must touch per seed / its duration; `traffic` = FETCH_SIZE +
    #                     WRITE_SIZE of the kernel from the PMC passes of the profiled run (profiles/hbm_traffic.json,
    #                     which names the commit it was taken at); SURVEY 8(d)'s figure for the reference layout
    #                     (64*2*X + L per read) sits beside it -- the table answers KK of a seed's K steps with one
    #                     gather, so that figure is a work rate, not a fraction of the roof.
    #   k_edit_myers   -- the bit-vector recurrences (edit-distance bound of the SW prefilter + edit distance): integer
    #                     VALU issue.  Peak = the architectural 1024 SIMDs x one wave64 VALU instruction per 4 cycles
    #                     at 2.4 GHz = 600 /us/SIMD; useful work = columns advanced (device counter) x W words x the
    #                     12 instructions one word step of the recurrence needs on this ISA / 64 lanes.
    VALU_PEAK = 1024 * 600e6  # wave-instructions per second
    K_seed = params.seed_size
    kk = info.get("kmer_k", 0)
    W_words = (read_len + 31) // 32
    kernels_ms = {"k_search_fast + k_search_listed": stage_ms["search"], "k_thin + scan": stage_ms["thin_scan"], "k_expand / k_locate": stage_ms["expand"] + stage_ms["locate"],
                  "k_coalesce (+ _mid, _heavy)": stage_ms["coalesce"], "k_sw_diag": st1.get("sw_diag_ms", 0.0),
                  "k_edit_myers (prefilter bound)": st1.get("sw_bound_ms", 0.0), "k_sw_pairs": st1.get("sw_sweep_ms", 0.0),
                  "k_edit_myers (edit distance)": st1.get("edit_ms", 0.0), "scan + k_gather": stage_ms["gather"]}
    prof = None
    tr = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tr):
        try:
            t = json.load(open(tr))
            if t.get("workload") == args.workload and t.get("dev_flags") == args.dev_flags:
                prof = t
        except Exception:
            prof = None
    search_ms = stage_ms["search"]
    seeds = st1["n_seed_slots"]
    alg_search = seeds * (8 + 2 * 64 * max(0, K_seed - kk) + K_seed) if kk else None
    roof = {"bound": "hbm", "kernel": f"k_search_fast<{kk}> + k_search_listed, the search stage (FMIndex::backward_search, index.rs:305)" if kk else "k_search",
            "kernel_ms": search_ms, "share_of_resident_step": search_ms / stage_ms["total"] if stage_ms.get("total") else None,
            "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "kernels_ms": kernels_ms}
    if alg_search and search_ms > 0:
        roof["algorithmic_bytes_per_launch"] = alg_search
        roof["algorithmic_model"] = (f"{seeds} seeds x (8 B k-mer table entry + {K_seed - kk} FM step(s) x 2 rank blocks of 64 B + {K_seed} B of read codes); "
                                     "seeds whose table interval is empty skip the rank blocks, so this is an upper bound of what must be touched")
        roof["achieved"] = alg_search / (search_ms * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / PEAK_HBM_GBS
    if prof and prof.get("k_search_bytes_per_step"):
        roof["traffic"] = prof["k_search_bytes_per_step"]
        roof["traffic_GBs_over_this_runs_time"] = prof["k_search_bytes_per_step"] / (search_ms * 1e-3) / 1e9
        roof["traffic_source"] = {"file": "profiles/hbm_traffic.json", "from_profile_of_commit": prof.get("commit"),
                                  "how": prof.get("source")}
    my_ms = st1.get("sw_bound_ms", 0.0) + st1.get("edit_ms", 0.0)
    valu = {"bound": "valu", "kernel": "k_edit_myers (bound mode + edit distance; index.rs:401-410, align.rs:28-85)", "kernel_ms": my_ms,
            "peak": VALU_PEAK / 1e9, "unit": "G wave-instr/s",
            "peak_source": "architectural: 1024 SIMDs x one wave64 VALU instruction per 4 cycles at 2.4 GHz (600 /us/SIMD); "
                           "tools/valu_rate.hip measures 540-600 on this chip (profiles/r01_valu_issue_rate.txt)"}
    if my_ms > 0 and st1.get("myers_columns"):
        useful = st1["myers_columns"] * W_words * 12 / 64.0
        valu["achieved"] = useful / (my_ms * 1e-3) / 1e9
        valu["frac"] = valu["achieved"] / valu["peak"]
        valu["useful_work"] = {"myers_columns": st1["myers_columns"], "words_per_column": W_words, "valu_per_word_step": 12, "lanes_per_wave": 64}
    if prof and prof.get("k_edit_myers_valu_per_step") and my_ms > 0:
        valu["issued_frac"] = prof["k_edit_myers_valu_per_step"] / (prof.get("k_edit_myers_ms_per_step") or my_ms) / 1e-3 / VALU_PEAK
        valu["issued_source"] = {"file": "profiles/hbm_traffic.json", "from_profile_of_commit": prof.get("commit"),
                                 "how": "SQ_INSTS_VALU of the k_edit_myers launches / their duration in the same profiled process"}
    roof["valu"] = valu
    if stage_bytes is not None:
        res_ms = resident["ms_per_step"] if resident else stage_ms["total"]
        hbm = {"survey_8d_bytes_per_read": stage_bytes["total"],
               "survey_8d_GBs_timed_region": stage_bytes["total"] * n_reads / (step_ms * 1e-3) / 1e9,
               "survey_8d_GBs_resident_step": stage_bytes["total"] * n_reads / (res_ms * 1e-3) / 1e9,
               "survey_8d_search_bytes_per_read": stage_bytes["search"],
               "survey_8d_search_GBs": stage_bytes["search"] * n_reads / (search_ms * 1e-3) / 1e9 if search_ms > 0 else None,
               "note": "SURVEY 8(d) bytes = what the reference's own layout must touch (64*(2X+S) + 8H + W + L + 24R); the resident k-mer table "
                       "and full suffix array avoid most of them, so these are work rates, not fractions of the 8 TB/s roof"}
        if prof:
            hbm["measured_traffic_bytes_per_step"] = prof["hbm_bytes_per_step"]
            hbm["measured_frac_of_peak_resident_step"] = prof["hbm_bytes_per_step"] / (res_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
            hbm["measured_traffic_source"] = {"file": "profiles/hbm_traffic.json", "from_profile_of_commit": prof.get("commit"), "how": prof.get("source")}
        roof["survey_8d"] = hbm

    out = {
        "metric": "reads/sec (whole node), 150bp reads vs MG-index",
        "value": value,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32/i16",
        "data": "synthetic",
        "value_region": "SURVEY 8(d): wall clock over host bases in -> host hits out through the C ABI (mtsv_batch_run_host + "
                        "mtsv_batch_download = mtsv_bin_batch on a warm workspace), H2D of reads and D2H of hits included; "
                        "index load/upload excluded",
        "timed_calls_ms_per_step": {"mtsv_batch_run_host": split_timed[0] / args.steps * 1e3, "mtsv_batch_download": split_timed[1] / args.steps * 1e3,
                                    "mtsv_hits_free": split_timed[2] / args.steps * 1e3},
        "step_ms_each": each_ms,
        "device_resident": resident,
        "host_input": "ordinary (pageable) memory" if args.pageable_input else "page-locked memory from mtsv_host_alloc",
        "host_transfer": ({"form": "4-bit codes, packed on the host chunk by chunk (csrc/host_pack.cpp), expanded by k_unpack",
                           "pack_threads": M.lib().mtsv_host_pack_threads()} if M.lib().mtsv_host_pack_threads()
                          else {"form": "plain bytes (MTSV_H2D_PLAIN, or fewer than nine CPUs to pack with), k_normalise on the device",
                                "pack_threads": 0}),
        "pageable_input": pageable,
        "config": {"workload": f"{args.workload}: {desc}", "reads_per_gpu_per_step": n_reads, "read_len": read_len,
                   "index_symbols": info["n"], "index_file_bytes": os.path.getsize(idx_path),
                   "index_hbm_bytes": info["device_bytes"], "dev_flags": args.dev_flags, "workspace_reads": min(n_reads, slice_reads),
                   "host_cpu_affinity": affinity,
                   "verify_mode": ["reference (SW prefilter + edit distance per verified candidate)", "edit_first"][args.verify_mode],
                   "parallelism": f"reads sharded x{world}, index replicated, no collective"},
        "roofline": roof,
        "cpu_baseline": cpu,
        "stage_ms": stage_ms,
        "stage_ms_note": stage_note,
        "overlapped_lanes": overlapped,
        "counters_per_read": ctr_per,
        "device_counters": {k: st1[k] for k in ("n_seed_slots", "n_seed_hits", "lf_steps", "n_candidates",
                                                 "n_verified", "window_bytes", "n_hits", "n_passes", "sw_cell_pairs", "n_sw_passed",
                                                 "n_sw_bound_refuted", "myers_columns")},
        "parity": parity,
        "other_verify_order": alt,
        "setup_s": {"index_build": t_build, "index_file_reused": index_cached, "index_load": t_load, "index_pack_upload_accel": t_upload},
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""The configurations BASELINE.json names, at their full sizes, through the C ABI on the GPU.

    configs[0]  10k x 100 bp reads vs the "1 MB" MG-index      every read against the oracle
    configs[1]  1M  x 150 bp reads vs the "1 GB" MG-index      every read against the oracle; the GPU-built index file
                                                               is byte-identical to the host-built one (n = 2.76e8)
    configs[2]  10M x 150 bp reads vs the "10 GB" MG-index     >= 1 % of the reads against the oracle + the
                                                               size-independent properties (idempotence, shard
                                                               invariance, both verification orders)
    configs[3]  reads sharded over GPUs, index replicated      mtsv_bin_batch_multi with GPU 0 listed twice (no
                                                               8-GPU node here: the scaling itself is unmeasured)
    configs[4]  one index chunk per GPU, reads broadcast       mtsv_bin_batch_chunks, checked against the per-chunk oracle

Workload generators and seeds are bench.py's; the index files land where bench.py looks for
them, so a bench run on the same box reuses them.
"""
import hashlib
import os

import numpy as np
import pytest

import mtsv_tools_amd as M
from helpers import assert_same_hits
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SEED_DB = 0x6D747376
WORKLOADS = {  # = bench.py
    "config0": (8, 2, 17500, 10_000, 100),
    "config1": (256, 4, 270_000, 1_000_000, 150),
    "config2": (1024, 4, 674_000, 10_000_000, 150),
}


def _index(name, gpu_build=True):
    n_taxa, gis, seq_len, _, _ = WORKLOADS[name]
    path = f"/tmp/mtsv_bench_{name}.idx"
    expect_n = n_taxa * gis * seq_len + 1
    if not (os.path.exists(path) and int.from_bytes(open(path, "rb").read(8), "little") == expect_n):
        if gpu_build:
            M.set_build_device(0)
        ix = M.MGIndex.synth(SEED_DB, n_taxa, gis, seq_len, threads=min(32, os.cpu_count() or 8))
        M.set_build_device(-1)
        ix.write(path + ".tmp")
        ix.close()
        os.replace(path + ".tmp", path)
    return path


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def test_config0_every_read_against_the_oracle():
    path = _index("config0")
    ix = M.MGIndex.load(path)
    ix.to_device(0)
    _, _, _, n_reads, L = WORKLOADS["config0"]
    bases, off = M.synth_reads(ix, seed=1000, n_reads=n_reads, read_len=L)
    got = ix.bin_batch(bases, off, device=0)
    want, _ = O.Index.read(path).bin_batch(bases, off, threads=8)
    assert len(want) > 0.8 * n_reads
    assert_same_hits(got, want)
    ix.close()


def test_config1_every_read_against_the_oracle_and_builder_bytes(tmp_path):
    path = _index("config1")  # GPU builder (prefix doubling in HBM)
    n_taxa, gis, seq_len, n_reads, L = WORKLOADS["config1"]
    # the host suffix sort writes the same bytes at n = 2.76e8 (every benchmark index comes from the GPU builder)
    hx = M.MGIndex.synth(SEED_DB, n_taxa, gis, seq_len, threads=min(32, os.cpu_count() or 8))
    host_path = str(tmp_path / "host.idx")
    hx.write(host_path)
    hx.close()
    assert os.path.getsize(host_path) == os.path.getsize(path)
    assert _sha(host_path) == _sha(path)
    os.remove(host_path)

    ix = M.MGIndex.load(path)
    ix.to_device(0)
    bases, off = M.synth_reads(ix, seed=1000, n_reads=n_reads, read_len=L)
    got = ix.bin_batch(bases, off, device=0)
    want, ctr = O.Index.read(path).bin_batch(bases, off, threads=min(16, os.cpu_count() or 8))   # ~1 minute
    assert len(want) > 0.8 * n_reads
    assert_same_hits(got, want)
    # device work counters of the resident path equal the oracle's (the reference order prefilters the same candidates)
    b = M.Batch(ix, 0, n_reads, len(bases))
    b.upload(bases, off)
    b.run()
    st = b.stats()
    assert (st["n_verified"], st["window_bytes"], st["n_hits"]) == (ctr["n_sw"], ctr["W"], len(want))
    assert_same_hits(b.download(), want)
    b.close()
    # configs[3] semantics on this workload: reads in blocks over two workspaces of GPU 0, index replicated
    assert_same_hits(M.bin_batch_multi(ix, [0, 0], bases, off), want)
    ix.close()


def test_config2_sampled_oracle_and_properties():
    path = _index("config2")
    ix = M.MGIndex.load(path)
    ix.to_device(0)
    _, _, _, n_reads, L = WORKLOADS["config2"]
    bases, off = M.synth_reads(ix, seed=1000, n_reads=n_reads, read_len=L)
    whole = ix.bin_batch(bases, off, device=0)                     # host path, reference order
    assert len(whole) > 0.8 * n_reads
    # >= 1 % of the reads against the oracle: a contiguous block and every 97th read of another
    orc = O.Index.read(path)
    ns = 100_000
    want, _ = orc.bin_batch(bases[: ns * L], off[: ns + 1], threads=min(16, os.cpu_count() or 8))
    assert_same_hits(whole[whole["read"] < ns], want)
    pick = np.arange(5_000_000, 5_000_000 + 97 * 20_000, 97)
    pb = bases.reshape(n_reads, L)[pick].reshape(-1)
    poff = np.arange(len(pick) + 1, dtype=np.uint64) * L
    pwant, _ = orc.bin_batch(pb, poff, threads=min(16, os.cpu_count() or 8))
    sel = whole[np.isin(whole["read"], pick)]
    remap = {int(r): i for i, r in enumerate(pick)}
    sel = sel.copy()
    sel["read"] = np.array([remap[int(r)] for r in sel["read"]], dtype=np.uint64)
    assert_same_hits(sel, pwant)
    del orc
    # idempotence (same call again), resident path == host path, edit-first order == reference order
    b = M.Batch(ix, 0, n_reads, len(bases))
    b.upload(bases, off)
    for mode in (0, 1):  # MTSV_VERIFY_REFERENCE, MTSV_VERIFY_EDIT_FIRST
        b.set_verify_mode(mode)
        b.run()
        assert_same_hits(b.download(), whole)
    b.close()
    # shard invariance = configs[3]: blocks of reads on separate workspaces, concatenated (index replicated)
    assert_same_hits(M.bin_batch_multi(ix, [0, 0], bases, off), whole)
    h = n_reads // 2
    second = ix.bin_batch(bases[h * L:], off[h:] - off[h], device=0)
    second["read"] += h
    assert_same_hits(second, whole[whole["read"] >= h])
    ix.close()


def test_config4_chunks_one_per_device_entry(tmp_path):
    """eight index chunks (1/8-size each, independent seeds), every chunk sees every read, merged per read"""
    n_taxa, gis, seq_len = 32, 4, 84_000        # 8 chunks x 1.07e7 symbols
    chunks, per = [], []
    M.set_build_device(0)
    for c in range(8):
        ix = M.MGIndex.synth(SEED_DB + 1 + c, n_taxa, gis, seq_len, threads=8)
        chunks.append(ix)
    M.set_build_device(-1)
    n_reads, L = 200_000, 150
    # reads drawn from all chunks in turn
    parts = [M.synth_reads(ix, seed=77 + c, n_reads=n_reads // 8, read_len=L)[0] for c, ix in enumerate(chunks)]
    bases = np.concatenate(parts)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    got = M.bin_batch_chunks(chunks, [0] * 8, bases, off)
    ns = 20_000  # oracle on every 10th read, all chunks
    pick = np.arange(0, n_reads, 10)
    pb = bases.reshape(n_reads, L)[pick].reshape(-1)
    poff = np.arange(len(pick) + 1, dtype=np.uint64) * L
    for c, ix in enumerate(chunks):
        p = str(tmp_path / f"c{c}.idx")
        ix.write(p)
        per.append(O.Index.read(p).bin_batch(pb, poff, threads=8)[0])
        os.remove(p)
    allh = np.concatenate(per)
    order = np.lexsort((np.concatenate([np.full(len(h), c) for c, h in enumerate(per)]), allh["read"]))
    want = allh[order]
    sel = got[got["read"] % 10 == 0].copy()
    sel["read"] //= 10
    assert len(want) > 0.7 * len(pick) and ns == len(pick)
    assert_same_hits(sel, want)
    for ix in chunks:
        ix.close()


def test_host_batch_beyond_4_gib_of_bases():
    """configs[3] hands one GPU 12.5 M reads of a 100 M-read node batch; the library takes host batches of any size: the
    reads stream through input segments of at most 3 GiB (offsets inside a segment are 32-bit), two arenas taking turns.
    30 M x 150 bp reads = 4.5 GB of bases in ordinary (pageable) memory against the "1 GB" index: the first reads and
    reads beyond the 4 GiB mark against the oracle, and the part beyond the first segment against a call of its own."""
    path = _index("config1")
    ix = M.MGIndex.load(path)
    ix.to_device(0)
    L, part = 150, 10_000_000
    bases = np.concatenate([M.synth_reads(ix, seed=4000 + k, n_reads=part, read_len=L)[0] for k in range(3)])
    n_reads = 3 * part
    assert len(bases) > (1 << 32)
    off = np.arange(n_reads + 1, dtype=np.uint64) * L
    whole = ix.bin_batch(bases, off, device=0)
    assert len(whole) > 0.8 * n_reads and int(whole["read"].max()) > 29_900_000
    orc = O.Index.read(path)
    ns = 50_000
    want, _ = orc.bin_batch(bases[: ns * L], off[: ns + 1], threads=min(16, os.cpu_count() or 8))
    assert_same_hits(whole[whole["read"] < ns], want)
    first = 29_000_000                                    # byte 4.35e9: beyond 2^32
    tail_b, tail_o = bases[first * L: (first + ns) * L], off[: ns + 1]
    want, _ = orc.bin_batch(tail_b, tail_o, threads=min(16, os.cpu_count() or 8))
    sel = whole[(whole["read"] >= first) & (whole["read"] < first + ns)].copy()
    sel["read"] -= first
    assert_same_hits(sel, want)
    del orc
    h = 21_000_000                                        # inside the second segment
    second = ix.bin_batch(bases[h * L:], off[h:] - off[h], device=0)
    second["read"] += h
    assert_same_hits(second, whole[whole["read"] >= h])
    ix.close()

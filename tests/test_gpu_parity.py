"""-m gpu parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bit-exact: every field of every hit, in the reference's order."""
import os
import random

import numpy as np
import pytest

import helpers
import mtsv_tools_amd as M
from helpers import assert_same_hits
from mtsv_tools_amd import _lib
from oracle import oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALL_FLAGS = [M.DEV_SAMPLED_SA_ONLY | M.DEV_NO_KMER_TABLE, M.DEV_SAMPLED_SA_ONLY, M.DEV_NO_KMER_TABLE, M.DEV_DEFAULT]

PARAM_SETS = {
    "default": {},
    # SURVEY 8(c) stress set: every cut-off active
    "stress": dict(max_hits=5, tune_max_hits=2, max_candidates=3, max_assignments=1, min_seed=0.5),
    # short dense seeds: hundreds to thousands of hits per strand (sort / walk in L2 scratch), thinning
    "dense": dict(seed_size=10, seed_interval=3, max_hits=100000, tune_max_hits=30),
    "dense_nothin": dict(seed_size=11, seed_interval=4, max_hits=100000, tune_max_hits=100000, min_seed=0.1),
    "exact_only": dict(edit_rate=0.0),
    "loose": dict(edit_rate=0.3, max_candidates=40),
    "wrapped_threshold": dict(edit_rate=0.6),      # 2*ED > L: usize wrap of index.rs:406, nothing passes
    "all_seeds_needed": dict(min_seed=1.0),
    "one_assignment": dict(max_assignments=1),
    "two_candidates": dict(max_candidates=2),
    "big_seed": dict(seed_size=31, seed_interval=7),
    "seed_over_32": dict(seed_size=40, seed_interval=9),          # k_search's byte-wise path
    "seed_below_table": dict(seed_size=5, seed_interval=20, max_hits=400, tune_max_hits=100),  # shorter than the k-mer table
}


def both_params(**over):
    mp = M.default_params(**{("seed_interval" if k == "seed_gap" else k): v for k, v in over.items()})
    op = O.default_params(**{("seed_gap" if k == "seed_interval" else k): v for k, v in over.items()})
    return mp, op


@pytest.fixture(scope="module")
def small_db(tmp_path_factory):
    ix = M.MGIndex.synth(seed=21, n_taxa=16, gis_per_taxon=4, seq_len=5000)
    p = str(tmp_path_factory.mktemp("idx") / "small.idx")
    ix.write(p)
    return ix, O.Index.read(p)


@pytest.fixture(scope="module")
def tricky(tmp_path_factory):
    entries, gene, unit = helpers.tricky_db(seed=7)
    ix = M.MGIndex.build(entries, threads=4)
    p = str(tmp_path_factory.mktemp("idx") / "tricky.idx")
    ix.write(p)
    reads = helpers.tricky_reads(entries, gene, unit, seed=11, n_each=50, lengths=(100, 150, 64, 253))
    return ix, O.Index.read(p), reads


@pytest.mark.parametrize("flags", ALL_FLAGS)
@pytest.mark.parametrize("read_len", [100, 150])
def test_default_params_match_oracle(small_db, flags, read_len):
    ix, orc = small_db
    bases, off = M.synth_reads(ix, seed=5 + read_len, n_reads=3000, read_len=read_len)
    ix.to_device(0, flags)
    got = ix.bin_batch(bases, off, M.default_params(), device=0)
    want, _ = orc.bin_batch(bases, off, O.default_params(), threads=8)
    assert len(want) > 1000
    assert_same_hits(got, want)


@pytest.mark.parametrize("flags", [ALL_FLAGS[0], ALL_FLAGS[3]])
@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_adversarial_database_all_parameter_sets(tricky, pname, flags):
    """conserved genes (duplicate TaxIds, ties in the stable rank sort), tandem repeats (merged
    windows longer than the LDS ring, thousands of seed hits), N runs, bin junctions, reads of mixed
    length incl. shorter than a seed and empty, lower case and junk bytes"""
    ix, orc, reads = tricky
    mp, op = both_params(**PARAM_SETS[pname])
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0, flags)
    got = ix.bin_batch(bases, off, mp, device=0)
    want, ctr = orc.bin_batch(bases, off, op, threads=8)
    if pname == "wrapped_threshold":
        assert len(want) == 0
    elif pname != "exact_only":
        assert len(want) > 50
    assert_same_hits(got, want)


@pytest.mark.parametrize("pname", list(PARAM_SETS))
def test_edit_first_order_gives_identical_hits(tricky, pname):
    """MTSV_VERIFY_EDIT_FIRST: Myers bit-vector edit distance decides (reads <= 253 bases); mixed
    batches with longer reads fall back to the reference order.  Same hits either way."""
    ix, orc, reads = tricky
    mp, op = both_params(**PARAM_SETS[pname])
    ix.to_device(0)
    for subset in ([r for r in reads if len(r) <= 253], reads):
        bases, off = helpers.reads_to_batch(subset)
        want, _ = orc.bin_batch(bases, off, op, threads=8)
        b = M.Batch(ix, 0, len(subset), len(bases))
        b.set_verify_mode(1)
        b.upload(bases, off)
        b.run(mp)
        assert_same_hits(b.download(), want)
        b.close()


def test_mega_tandem_repeat_tens_of_thousands_of_hits_per_strand(tmp_path):
    """a 400-copy tandem repeat with dense short seeds: > 8192 seed hits per strand (bitonic sort in
    L2-resident scratch), one merged candidate window of ~20 kb (LDS ring refill in the sweep),
    thousands of hits thinned by tune_max_hits"""
    import random
    rng = random.Random(5)
    unit = helpers.rnd_seq(rng, 53)
    flank = lambda: helpers.rnd_seq(rng, 700)
    entries = [(7, 1, flank() + unit * 400 + flank()), (9, 2, flank() + unit * 30 + flank()), (9, 3, helpers.rnd_seq(rng, 3000))]
    ix = M.MGIndex.build(entries, threads=2)
    p = str(tmp_path / "t.idx")
    ix.write(p)
    orc = O.Index.read(p)
    rep = unit * 6
    reads = [helpers.mutate(rng, rep[s:s + 150], rng.randrange(0, 8)) for s in range(0, 60, 7)]
    reads += [helpers.revcomp(r) for r in reads[:4]] + [entries[0][2][600:750], entries[2][2][100:250]]
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    for over in (dict(seed_size=11, seed_interval=2, max_hits=100000, tune_max_hits=100000),
                 dict(seed_size=11, seed_interval=2, max_hits=100000, tune_max_hits=300),
                 dict(seed_size=14, seed_interval=5, max_hits=100000, tune_max_hits=100000, min_seed=0.2)):
        mp, op = both_params(**over)
        want, ctr = orc.bin_batch(bases, off, op, threads=8)
        b = M.Batch(ix, 0, len(reads), len(bases), max_hits_ws=4_000_000)
        b.upload(bases, off)
        b.run(mp)
        st = b.stats()
        assert st["n_seed_hits"] == ctr["H"] and st["n_candidates"] == ctr["n_cand"]
        if over["tune_max_hits"] == 100000 and over["seed_size"] == 11:
            assert st["n_seed_hits"] / (2 * len(reads)) > 8192  # the global-scratch path is really taken
            assert st["window_bytes"] / max(st["n_verified"], 1) > 2048  # and windows outgrow the LDS ring
        assert_same_hits(b.download(), want)
        assert len(want) >= len(reads) - 2
        b.close()


def test_tiny_index_of_the_reference_unit_test():
    """the literal 10-symbol database of index.rs:860-873: fewer rows than one rank block"""
    ix = M.MGIndex.build([(1, 10, b"ACGT"), (1, 11, b"TTAA"), (2, 20, b"GG")], occ_k=8, sa_s=8, threads=1)
    orc = O.Index.build([(1, 10, b"ACGT"), (1, 11, b"TTAA"), (2, 20, b"GG")], 8, 8)
    reads = [b"ACGT", b"TTAA", b"GG", b"ACGTTTAAGG", b"CGTTTA", b"TTAAACGT", b"NNNN", b"A"]
    bases, off = helpers.reads_to_batch(reads)
    for flags in ALL_FLAGS:
        ix.to_device(0, flags)
        for over in (dict(seed_size=2, seed_interval=1, edit_rate=0.3), dict(seed_size=4, seed_interval=1, edit_rate=0.0),
                     dict(seed_size=3, seed_interval=2, edit_rate=0.5, min_seed=1.0)):
            mp, op = both_params(**over)
            want, _ = orc.bin_batch(bases, off, op, threads=1)
            assert_same_hits(ix.bin_batch(bases, off, mp, device=0), want)


def test_committed_golden_result_lines(tricky):
    """the committed end-to-end vectors (tests/golden/e2e_*.results) through the HIP path + the
    product's formatter"""
    entries = []
    for line in open(os.path.join(GOLD, "e2e_db.fasta")):
        if line.startswith(">"):
            gi, tax = line[1:].split()[0].split("-")
        else:
            entries.append((int(tax), int(gi), line.strip().encode()))
    reads = [l.rstrip("\n").encode("latin-1") for l in open(os.path.join(GOLD, "e2e_reads.txt"), encoding="latin-1")]
    ix = M.MGIndex.build(entries, threads=4)
    ix.to_device(0)
    bases, off = helpers.reads_to_batch(reads)
    ids = [f"r{i}" for i in range(len(reads))]
    stress = dict(max_hits=5, tune_max_hits=2, max_candidates=3, max_assignments=1, min_seed=0.5)
    for name, over in (("default", {}), ("stress", stress)):
        hits = ix.bin_batch(bases, off, M.default_params(**over), device=0)
        for long_fmt in (False, True):
            want = open(os.path.join(GOLD, f"e2e_{name}{'_long' if long_fmt else ''}.results")).read()
            assert M.format_results(hits, ids, long_fmt) == want


def test_batch_split_when_hit_workspace_is_small(tricky):
    ix, orc, reads = tricky
    mp, op = both_params(**PARAM_SETS["dense_nothin"])
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want, _ = orc.bin_batch(bases, off, op, threads=8)
    b = M.Batch(ix, 0, len(reads), len(bases), max_hits_ws=40000)
    b.upload(bases, off)
    b.run(mp)
    st = b.stats()
    assert st["n_passes"] > 1
    assert_same_hits(b.download(), want)
    # a workspace that cannot hold a single read's seed hits is grown for that read (passes of one read)
    b2 = M.Batch(ix, 0, len(reads), len(bases), max_hits_ws=64)
    b2.upload(bases, off)
    b2.run(mp)
    assert b2.stats()["n_passes"] > 100
    assert_same_hits(b2.download(), want)
    b2.close()


def test_run_host_slices_match_oracle(tricky):
    """mtsv_batch_run_host: a workspace much smaller than the batch -> many double-buffered slices,
    cut by read count or by base count; hits (incl. the batch-wide read index) equal the oracle's."""
    ix, orc, reads = tricky
    mp, op = both_params()
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want, _ = orc.bin_batch(bases, off, op, threads=8)
    for max_reads, max_bases in ((97, 1 << 16), (len(reads), 4096), (len(reads) + 5, len(bases) + 5)):
        b = M.Batch(ix, 0, max_reads, max_bases)
        b.run_host(bases, off, mp)
        st = b.stats()
        assert st["n_reads"] == len(reads)
        if max_reads < len(reads) or max_bases < len(bases):
            assert st["n_passes"] > 2
        assert_same_hits(b.download(), want)
        # the same workspace again, then an empty batch
        b.run_host(bases, off, mp)
        assert_same_hits(b.download(), want)
        b.run_host(np.zeros(0, np.uint8), np.zeros(1, np.uint64), mp)
        assert len(b.download()) == 0
        b.close()


def test_run_host_errors_surface_from_the_uploader(small_db):
    ix, _ = small_db
    ix.to_device(0)
    b = M.Batch(ix, 0, 64, 1 << 16)
    reads = [b"ACGT" * 30] * 200 + [b"ACGT" * 10000] + [b"ACGT" * 30] * 50   # a 40000-base read in slice 3: beyond the tiled kernel's cells
    bases, off = helpers.reads_to_batch(reads)
    with pytest.raises(M.MtsvError) as e:
        b.run_host(bases, off)
    assert e.value.code == _lib.E_LIMIT
    bad = off.copy()
    bad[150] = bad[149] - 1                                                # offsets not ascending
    with pytest.raises(M.MtsvError) as e:
        b.run_host(bases, bad)
    assert e.value.code == _lib.E_ARG
    # and the workspace is still usable afterwards
    good, goff = helpers.reads_to_batch([b"ACGT" * 30] * 300)
    b.run_host(good, goff)
    b.close()


def test_limits_and_argument_errors(small_db):
    ix, _ = small_db
    ix.to_device(0)
    bases, off = helpers.reads_to_batch([b"ACGT" * 10000])  # 40000 bases: beyond the 16-bit cells of the tiled kernel
    with pytest.raises(M.MtsvError) as e:
        ix.bin_batch(bases, off, device=0)
    assert e.value.code == _lib.E_LIMIT
    ok, off2 = helpers.reads_to_batch([b"ACGT" * 30])
    for bad in (dict(edit_rate=1.5), dict(edit_rate=-0.1), dict(seed_size=0), dict(seed_interval=0)):
        with pytest.raises(M.MtsvError) as e:
            ix.bin_batch(ok, off2, M.default_params(**bad), device=0)
        assert e.value.code == _lib.E_ARG
    assert len(ix.bin_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64), device=0)) == 0  # empty batch


@pytest.mark.parametrize("read_len", [254, 300, 320, 321, 400, 512, 513, 700, 1400])
def test_long_reads_take_the_ssw_word_kernel_path(tricky, read_len):
    """reads of 254+ bases whose score reaches 254 make ssw_align rerun sw_sse2_word (ssw.c:789-792),
    whose lazy-F loop truncates vertical gaps at stripe boundaries: insertions planted there"""
    import random
    ix, orc, _ = tricky
    entries, gene, unit = helpers.tricky_db(seed=7)
    rng = random.Random(read_len)
    texts = [e[2].upper() for e in entries if len(e[2]) > read_len + 200]
    reads = []
    seg = (read_len + 7) // 8
    for i in range(150):
        t = rng.choice(texts)
        st = rng.randrange(0, len(t) - read_len - 40)
        r = bytearray(t[st:st + read_len + 30])
        for _ in range(rng.randrange(0, 4)):
            b = rng.randrange(1, 8) * seg + rng.randrange(-2, 3)
            k = rng.randrange(1, 7)
            if rng.random() < 0.6:
                r[b:b] = helpers.rnd_seq(rng, k)
            else:
                del r[b:b + k]
        for _ in range(rng.randrange(0, 8)):
            r[rng.randrange(read_len)] = rng.choice(b"ACGTN")
        r = bytes(r[:read_len])
        reads.append(r if i % 2 else helpers.revcomp(r))
    bases, off = helpers.reads_to_batch(reads)
    for over in ({}, dict(edit_rate=0.05), dict(edit_rate=0.02)):
        mp, op = both_params(**over)
        ix.to_device(0)
        got = ix.bin_batch(bases, off, mp, device=0)
        want, _ = orc.bin_batch(bases, off, op, threads=8)
        assert len(want) > 20
        assert_same_hits(got, want)


def test_stray_long_reads_do_not_fail_the_batch(tricky):
    """The reference has no read-length cap (index.rs:258-432; ssw.c handles any length): a few long reads
    (merged pairs, contigs) among short ones run in passes of their own through the tiled kernel and every
    other read keeps its hits; host-sliced and resident paths agree with the oracle."""
    ix, orc, short_reads = tricky
    entries, gene, unit = helpers.tricky_db(seed=7)
    rng = random.Random(99)
    texts = [e[2].upper() for e in entries if len(e[2]) > 1200]
    reads = list(short_reads[:400])
    strays = []
    for L in (321, 513, 800, 1100, 1100, 2 * 97 * 6):
        t = rng.choice(texts)
        st = rng.randrange(0, len(t) - min(L, len(t) - 1))
        r = helpers.mutate(rng, t[st:st + L], rng.randrange(0, 12))
        strays.append(r if rng.random() < 0.5 else helpers.revcomp(r))
    strays.append(helpers.mutate(rng, unit * 12, 5))   # long read inside the tandem repeat: thousands of seed hits
    strays.append(helpers.rnd_seq(rng, 900))           # long read without an origin
    strays.append(b"N" * 700)
    for k, r in enumerate(strays):  # first, last, adjacent and scattered positions
        reads.insert([0, len(reads), 7, 8, 150, 151, 152, 300, 301][k] if k < 9 else rng.randrange(len(reads)), r)
    bases, off = helpers.reads_to_batch(reads)
    assert max(len(r) for r in reads) > 1000
    mp, op = both_params()
    ix.to_device(0)
    want, _ = orc.bin_batch(bases, off, op, threads=8)
    long_idx = [i for i, r in enumerate(reads) if len(r) > 320]
    assert len(set(want["read"]) & set(long_idx)) >= 4  # the long reads do produce hits
    got = ix.bin_batch(bases, off, mp, device=0)
    assert_same_hits(got, want)
    b = M.Batch(ix, 0, max_reads=64, max_bases=20000)  # sliced host path: long reads land in different slices
    b.run_host(bases, off, mp)
    assert_same_hits(b.download(), want)
    b.close()
    b = M.Batch(ix, 0, max_reads=len(reads), max_bases=len(bases), max_hits_ws=1 << 12)  # tiny hit workspace: grown for the repeat read
    b.upload(bases, off)
    b.run(mp)
    assert_same_hits(b.download(), want)
    b.close()


def test_bin_batch_multi_shards_reads_over_device_entries(tricky, medium):
    """mtsv_bin_batch_multi (Mode A): contiguous read blocks over the listed devices, one workspace per entry;
    listing GPU 0 twice / three times gives the single-device hits, read indices included."""
    ix, orc, reads = tricky
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want = ix.bin_batch(bases, off, device=0)
    for devs in ([0], [0, 0], [0, 0, 0]):
        assert_same_hits(M.bin_batch_multi(ix, devs, bases, off), want)
    mix, mbases, moff = medium
    mix.to_device(0)
    mwant = mix.bin_batch(mbases, moff, device=0)
    assert_same_hits(M.bin_batch_multi(mix, [0, 0], mbases, moff), mwant)
    assert len(M.bin_batch_multi(ix, [0, 0], np.zeros(0, np.uint8), np.zeros(1, np.uint64))) == 0
    with pytest.raises(M.MtsvError):
        M.bin_batch_multi(ix, [0, 99], bases, off)  # no such device: the whole call fails, nothing leaks


def test_bin_batch_chunks_merges_hits_per_read(tmp_path):
    """mtsv_bin_batch_chunks (Mode B): three chunks of the database, every chunk sees every read; the merged
    list equals the oracle's per-chunk hits merged per read in chunk order, and its result lines equal the
    smallest edit per TaxId over the chunks (collapse.rs:597-602)."""
    entries, gene, unit = helpers.tricky_db(seed=7)
    reads = helpers.tricky_reads(entries, gene, unit, seed=5, n_each=30)
    bases, off = helpers.reads_to_batch(reads)
    chunks, per = [], []
    for c in range(3):
        ix = M.MGIndex.build(entries[c::3], threads=2)
        p = str(tmp_path / f"c{c}.idx")
        ix.write(p)
        chunks.append(ix)
        per.append(O.Index.read(p).bin_batch(bases, off, threads=8)[0])
    allh = np.concatenate(per)
    order = np.lexsort((np.concatenate([np.full(len(h), c) for c, h in enumerate(per)]), allh["read"]))  # stable: read, chunk
    want = allh[order]
    got = M.bin_batch_chunks(chunks, [0, 0, 0], bases, off)
    assert_same_hits(got, want)
    ids = [f"r{i}" for i in range(len(reads))]
    lines = M.format_results(got, ids, False).splitlines()
    best = {}
    for h in want:
        k = (int(h["read"]), int(h["tax_id"]))
        best[k] = min(best.get(k, 1 << 30), int(h["edit"]))
    assert len(lines) == len({k[0] for k in best})
    for line in lines:
        rid, rest = line.split(":")
        for tok in rest.split(","):
            t, e = tok.split("=")
            assert best[(int(rid[1:]), int(t))] == int(e)
    for ix in chunks:
        ix.close()


def test_repeated_calls_with_changing_batch_shapes(small_db):
    """mtsv_bin_batch reuses / regrows its cached workspace: sizes going up and down, different read
    lengths and parameter sets back to back, two devices-structure flavours, same answers"""
    import random
    ix, orc = small_db
    rng = random.Random(77)
    ix.to_device(0)
    for it in range(12):
        n = rng.choice([1, 7, 300, 2500, 40, 9000, 3])
        L = rng.choice([36, 100, 150, 250, 301])
        over = rng.choice([{}, dict(max_candidates=2), dict(seed_size=14, seed_interval=6), dict(edit_rate=0.05)])
        if it == 6:
            ix.to_device(0, M.DEV_SAMPLED_SA_ONLY)
        bases, off = M.synth_reads(ix, seed=1000 + it, n_reads=n, read_len=L)
        mp, op = both_params(**over)
        want, _ = orc.bin_batch(bases, off, op, threads=8)
        assert_same_hits(ix.bin_batch(bases, off, mp, device=0), want)


def test_counters_equal_oracle_counters(small_db):
    """the device counts the same work the reference does (SURVEY 8(d) accounting)"""
    ix, orc = small_db
    bases, off = M.synth_reads(ix, seed=77, n_reads=5000, read_len=150)
    ix.to_device(0, M.DEV_SAMPLED_SA_ONLY | M.DEV_NO_KMER_TABLE)
    b = M.Batch(ix, 0, 5000, len(bases))
    b.upload(bases, off)
    b.run(M.default_params())
    st = b.stats()
    _, ctr = orc.bin_batch(bases, off, O.default_params(), threads=8)
    assert st["n_seed_hits"] == ctr["H"]
    assert st["lf_steps"] == ctr["S"]
    assert st["n_candidates"] == ctr["n_cand"]
    assert st["n_verified"] == ctr["n_sw"]
    assert st["window_bytes"] == ctr["W"]
    assert st["n_hits"] == ctr["R"]
    assert 0 < st["n_sw_passed"] <= ctr["n_edit"]   # equal but for reads with more N than the tolerance (below)


@pytest.fixture(scope="module")
def medium():
    ix = M.MGIndex.synth(seed=0x6D747376, n_taxa=64, gis_per_taxon=4, seq_len=40000)  # n ~ 1e7
    ix.to_device(0)
    bases, off = M.synth_reads(ix, seed=4242, n_reads=400_000, read_len=150)
    return ix, bases, off


@pytest.mark.parametrize("k,extra", [("17", {}), ("17", dict(seed_size=17, seed_interval=9)), ("17", dict(seed_size=25)),
                                     ("17", dict(seed_size=16, seed_interval=7)),
                                     ("16", {}), ("15", {}), ("14", {}), ("13", {}), ("12", {}), ("11", {}),
                                     ("16", dict(seed_size=24, seed_interval=11)), ("12", dict(seed_size=20)),
                                     ("16", dict(seed_size=16, seed_interval=7)), ("13", dict(seed_size=22))])
def test_kmer_table_width_does_not_change_hits(tricky, monkeypatch, k, extra):
    """The k-mer interval table replaces the first k backward-search steps of a seed; 16 is the widest
    (32 GiB, chosen by itself for the 10 GB index), and a table as wide as the seed minus two leaves two
    FM steps.  Hits and counters must not depend on it.  Widths 12..16 with seeds of 16..24 symbols take
    k_search_fast (every instantiation here; seeds with an N in the table part go through k_search_slow),
    everything else the general kernel."""
    entries, _, _ = helpers.tricky_db()
    ix2 = M.MGIndex.build(entries, threads=4)
    monkeypatch.setenv("MTSV_KMER_K", k)
    ix2.to_device(0)
    monkeypatch.delenv("MTSV_KMER_K")
    _, orc, reads = tricky
    mp, op = both_params(**extra)
    bases, off = helpers.reads_to_batch(reads)
    want, ctr = orc.bin_batch(bases, off, op, threads=8)
    b = M.Batch(ix2, 0, len(reads), len(bases))
    b.upload(bases, off)
    b.run(mp)
    assert_same_hits(b.download(), want)
    assert b.stats()["n_seed_hits"] == ctr["H"]
    # the general kernel (every slot through the step-by-step search) agrees with the table-driven fast kernel
    monkeypatch.setenv("MTSV_SEARCH_GENERIC", "1")
    b.run(mp)
    assert_same_hits(b.download(), want)
    assert b.stats()["n_seed_hits"] == ctr["H"]
    b.close()
    ix2.close()


@pytest.mark.parametrize("max_len", [60, 96, 128, 150, 200, 253])
def test_prefilter_shortcuts_for_every_row_count(tricky, max_len, monkeypatch):
    """Every k_sw_pairs<R> instantiation (R is chosen from the longest read of the batch) with all three
    ways a candidate leaves the prefilter: decided on the seed diagonal without a sweep (substitutions
    only), passed or failed by the sweep (indels, junk), rejected by the N count.  Mixed lengths in one
    batch, both evaluation orders, hits and work counters against the oracle."""
    ix, orc, _ = tricky
    entries, gene, unit = helpers.tricky_db()
    texts = [e[2].upper() for e in entries if len(e[2]) > max_len + 50]
    rng = random.Random(max_len)
    reads = []
    for i in range(700):
        L = max_len if i % 3 == 0 else rng.randrange(max(20, max_len // 3), max_len + 1)
        src = gene if i % 2 == 0 and len(gene) > L else rng.choice(texts)
        st = rng.randrange(0, len(src) - L)
        r = bytearray(src[st:st + L])
        kind = i % 5
        if kind in (0, 1):        # substitutions only: the diagonal decides
            for _ in range(rng.randrange(0, int(L * 0.13) + 3)):
                r[rng.randrange(L)] = rng.choice(b"ACGT")
        elif kind == 2:           # indels: needs the sweep
            r = bytearray(helpers.mutate(rng, bytes(r), rng.randrange(1, 8)))[:max_len]
        elif kind == 3:           # N-rich
            for _ in range(rng.randrange(0, L // 4)):
                r[rng.randrange(L)] = ord("N")
        r = bytes(r)
        reads.append(r if rng.random() < 0.5 else helpers.revcomp(r))
    assert max(map(len, reads)) == max_len
    mp, op = both_params()
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want, ctr = orc.bin_batch(bases, off, op, threads=8)
    assert len(want) > 300
    b = M.Batch(ix, 0, len(reads), len(bases))
    b.upload(bases, off)
    swept = {}
    for mode in (0, 1, 0):
        b.set_verify_mode(mode)
        b.run(mp)
        assert_same_hits(b.download(), want)
        st = b.stats()
        if mode == 0:
            assert (st["n_verified"], st["window_bytes"]) == (ctr["n_sw"], ctr["W"])
            swept["bounds"] = st["sw_cell_pairs"]
    b.close()
    # The SW predicate itself (index.rs:406), candidate by candidate in aggregate: the number of candidates the
    # prefilter passes on equals the number of edit distances the reference computes.  Reads with more N than
    # the edit tolerance are left out: no candidate of theirs can be accepted, so the coalescing kernels account
    # their prefilter work without deciding it.
    import math
    few_n = [r for r in reads if sum(c not in b"ACGT" for c in r) <= math.ceil(len(r) * op.edit_rate)]
    assert len(few_n) > 400
    fb, fo = helpers.reads_to_batch(few_n)
    fwant, fctr = orc.bin_batch(fb, fo, op, threads=8)
    b = M.Batch(ix, 0, len(few_n), len(fb))
    b.upload(fb, fo)
    b.run(mp)
    assert_same_hits(b.download(), fwant)
    st = b.stats()
    assert (st["n_verified"], st["n_sw_passed"]) == (fctr["n_sw"], fctr["n_edit"]) and fctr["n_edit"] > len(fwant)
    b.close()
    # the instantiation without the lower bounds on the seed diagonal (every candidate that is not hopeless is
    # swept) decides every candidate the same way: same hits, same counters, more cells
    monkeypatch.setenv("MTSV_SW_DIAG", "0")
    monkeypatch.setenv("MTSV_SW_BOUND", "0")
    b = M.Batch(ix, 0, len(reads), len(bases))
    monkeypatch.delenv("MTSV_SW_DIAG")
    monkeypatch.delenv("MTSV_SW_BOUND")
    b.upload(bases, off)
    b.run(mp)
    assert_same_hits(b.download(), want)
    st = b.stats()
    assert (st["n_verified"], st["window_bytes"]) == (ctr["n_sw"], ctr["W"])
    assert st["sw_cell_pairs"] > swept["bounds"]
    b.close()


def test_lanes_do_not_change_hits_or_counters(medium, monkeypatch):
    """A resident batch above 98 304 reads runs as three concurrent lanes (own streams, host threads);
    one lane, two and three must give the same hits in the same order and the same work counters, for
    both evaluation orders and for the sliced host path."""
    ix, bases, off = medium
    ix.to_device(0)
    n = len(off) - 1
    ref = None
    for lanes in ("1", "2", "3"):
        monkeypatch.setenv("MTSV_LANES", lanes)
        b = M.Batch(ix, 0, n, len(bases))
        b.upload(bases, off)
        for mode in (0, 1):
            b.set_verify_mode(mode)
            b.run()
            st = b.stats()
            assert st["n_lanes"] == int(lanes)
            got = (b.download(), {k: st[k] for k in ("n_seed_hits", "n_candidates", "n_verified", "window_bytes", "n_hits")})
            if ref is None:
                ref = got
            assert_same_hits(got[0], ref[0])
            assert got[1] == ref[1]
        b.close()
        b = M.Batch(ix, 0, 120000, 120000 * 150)      # host path: slices of 120 k reads, each split over the lanes
        b.run_host(bases, off)
        assert_same_hits(b.download(), ref[0])
        b.close()


def test_prefilter_stages_decide_alike(medium, tricky, monkeypatch):
    """The SW prefilter of index.rs:406 runs as three kernels in the first round of a pass -- the lower bounds on the
    seed diagonal (k_sw_diag), the two-sided bound by the unit-cost edit distance under the SW matrix's matches
    (k_edit_myers in bound mode), the full-height sweep of what neither decides -- or, with the environment switches
    below, with the top-half sweep (k_sw_pairs TOP) in the place of the second, as two kernels or as one.  Every
    arrangement decides every candidate alike: same hits, same candidates examined, same number sent on to the edit
    distance."""
    ix, bases, off = medium
    tix, _, treads = tricky
    tb, to = helpers.reads_to_batch([r for r in treads if len(r) <= 253])
    ref = {}
    for prepass, top, bound in (("0", "0", "0"), ("1", "0", "0"), ("1", "1", "0"), ("1", "1", "1"), ("0", "0", "1")):
        monkeypatch.setenv("MTSV_SW_PREPASS", prepass)
        monkeypatch.setenv("MTSV_SW_TOP", top)
        monkeypatch.setenv("MTSV_SW_BOUND", bound)
        for name, (x, b_, o_) in (("medium", (ix, bases, off)), ("tricky", (tix, tb, to))):
            x.to_device(0)
            b = M.Batch(x, 0, len(o_) - 1, len(b_))
            b.upload(b_, o_)
            b.run()
            st = b.stats()
            got = (b.download(), {k: st[k] for k in ("n_candidates", "n_verified", "window_bytes", "n_sw_passed", "n_hits")})
            b.close()
            if name not in ref:
                ref[name] = got
                assert got[1]["n_sw_passed"] >= got[1]["n_hits"] > 0
            assert_same_hits(got[0], ref[name][0])
            assert got[1] == ref[name][1], (name, prepass, top, bound)
            if (prepass, top, bound) == ("1", "1", "0"):
                # the top-half sweep computes about half the cells of the full-height one (both count theirs)
                assert 0 < st["sw_cell_pairs"]


@pytest.mark.parametrize("edit_rate", [0.0, 0.04, 0.2, 0.3, 0.45])
def test_prefilter_kernels_at_other_tolerances(tricky, edit_rate):
    """The threshold L - 2*ED moves the geometry of every bound: at 0.3 an alignment below the top half of the rows
    alone can reach it (those candidates skip the top-half sweep), at 0.45 and above nearly every read is refuted
    late or wraps (2*ED > L), at 0 the window is the read's length and only an exact copy passes."""
    ix, orc, reads = tricky
    reads = [r for r in reads if len(r) <= 253]
    bases, off = helpers.reads_to_batch(reads)
    mp, op = both_params(edit_rate=edit_rate)
    ix.to_device(0)
    want, ctr = orc.bin_batch(bases, off, op, threads=8)
    b = M.Batch(ix, 0, len(reads), len(bases))
    b.upload(bases, off)
    for mode in (0, 1):
        b.set_verify_mode(mode)
        b.run(mp)
        assert_same_hits(b.download(), want)
        st = b.stats()
        if mode == 0:
            assert (st["n_verified"], st["window_bytes"]) == (ctr["n_sw"], ctr["W"])
    b.close()


def test_page_locked_input_is_read_in_place(medium, capfd, monkeypatch):
    """bases in memory from mtsv_host_alloc, or registered with mtsv_host_register, skip the staging copy of
    run_host (MTSV_TRACE names the route); the hits do not depend on where the bases lie"""
    ix, bases, off = medium
    ix.to_device(0)
    want = ix.bin_batch(bases, off, device=0)
    monkeypatch.setenv("MTSV_TRACE", "1")
    hb = M.HostBuffer(len(bases))
    hb.array[:] = bases
    capfd.readouterr()
    assert_same_hits(ix.bin_batch(hb.array, off, device=0), want)
    assert "page-locked" in capfd.readouterr().err
    hb.close()
    own = bases.copy()
    assert_same_hits(ix.bin_batch(own, off, device=0), want)
    assert "pageable" in capfd.readouterr().err
    M.host_register(own)
    assert_same_hits(ix.bin_batch(own, off, device=0), want)
    assert "page-locked" in capfd.readouterr().err
    M.host_unregister(own)
    with pytest.raises(M.MtsvError):
        M.host_unregister(own)


def test_run_host_recycles_the_lanes_result_arrays(medium, monkeypatch):
    """a host batch far larger than the workspace: every lane's device result array is reused once its hits have
    left for the host (tiny arrays here: dozens of wrap-arounds per lane), growth still works for a slice that
    needs more, and the hits equal the single-call result"""
    ix, bases, off = medium
    ix.to_device(0)
    want = ix.bin_batch(bases, off, device=0)
    for lanes, cap in (("1", "20000"), ("3", "40000"), ("3", "2000")):
        monkeypatch.setenv("MTSV_LANES", lanes)
        monkeypatch.setenv("MTSV_HITS_CAP", cap)     # initial entries of a lane's result array (grown when a slice needs more)
        b = M.Batch(ix, 0, 100_000 if lanes == "3" else 6000, 100_000 * 150)
        monkeypatch.delenv("MTSV_HITS_CAP")
        b.run_host(bases, off)
        assert_same_hits(b.download(), want)
        b.run_host(bases, off)                  # and again on the warm workspace
        assert_same_hits(b.download(), want)
        b.close()


def test_full_size_properties(medium, tmp_path):
    """size-independent properties at a batch the oracle cannot finish in seconds:
    idempotence, shard invariance (two halves == whole: the multi-GPU read sharding), strand
    symmetry (reverse-complemented reads give the same hits with the strands swapped), and a
    sampled bit-exact comparison with the oracle."""
    ix, bases, off = medium
    n = len(off) - 1
    whole = ix.bin_batch(bases, off, device=0)
    again = ix.bin_batch(bases, off, device=0)
    assert_same_hits(whole, again)
    assert len(whole) > 0.8 * n
    # shards
    h = n // 2
    a = ix.bin_batch(bases[: h * 150], off[: h + 1], device=0)
    b = ix.bin_batch(bases[h * 150:], off[h:] - off[h], device=0)
    b["read"] += h
    assert_same_hits(np.concatenate([a, b]), whole)
    # strand symmetry
    comp = np.full(256, ord("N"), np.uint8)
    for x, y in zip(b"ACGTacgt", b"TGCATGCA"):
        comp[x] = y
    rc = comp[bases.reshape(n, 150)[:, ::-1]].reshape(-1)
    r = ix.bin_batch(rc, off, device=0)
    key = lambda hh, flip: set(zip(hh["read"].tolist(), hh["tax_id"].tolist(), hh["gi"].tolist(), hh["edit"].tolist(),
                                   hh["offset"].tolist(), (hh["strand"] ^ flip).tolist()))
    assert key(whole, 0) == key(r, 1)
    # sampled oracle comparison
    p = str(tmp_path / "m.idx")
    ix.write(p)
    ns = 4000
    want, _ = O.Index.read(p).bin_batch(bases[: ns * 150], off[: ns + 1], threads=8)
    assert_same_hits(whole[whole["read"] < ns], want)


def test_gpu_index_builder_writes_the_same_bytes(tmp_path):
    """mtsv_set_build_device: suffix array by prefix doubling on the GPU -> byte-identical MG-index"""
    entries, _, _ = helpers.tricky_db(seed=13)
    big = M.MGIndex.synth(seed=3, n_taxa=32, gis_per_taxon=4, seq_len=30000)  # host build, n ~ 3.8e6
    a, b, c, d = (str(tmp_path / f"{x}.idx") for x in "abcd")
    M.MGIndex.build(entries, 64, 32, threads=4).write(a)
    big.write(c)
    try:
        M.set_build_device(0)
        M.MGIndex.build(entries, 64, 32, threads=4).write(b)
        M.MGIndex.synth(seed=3, n_taxa=32, gis_per_taxon=4, seq_len=30000).write(d)
        for k, s in ((3, 5), (128, 7)):
            e, f = str(tmp_path / "e.idx"), str(tmp_path / "f.idx")
            M.MGIndex.build(entries, k, s, threads=2).write(e)
            O.Index.build(entries, k, s).write(f)
            assert open(e, "rb").read() == open(f, "rb").read()
    finally:
        M.set_build_device(-1)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(c, "rb").read() == open(d, "rb").read()


def test_randomised_soak_short(tmp_path):
    """25 iterations of tests/fuzz_parity.py (random databases, read lengths 1..400, damage, parameters,
    device-structure flags; both evaluation orders against the oracle).  800 iterations over three seeds
    were run clean when this was committed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for it in range(25):
        fz.one(it, random.Random(7 * 100003 + it), str(tmp_path))


def test_reupload_with_other_flags_between_bin_batch_calls(small_db):
    """mtsv_bin_batch keeps a workspace on the index handle; mtsv_index_to_device with other flags retires the
    device structures that workspace points at.  bin_batch -> to_device(other flags) -> bin_batch on the same
    handle, in a cycle and twice over, must give the oracle's hits every time (a use-after-free on the cached
    workspace's mutex aborted the process here once, round 2)."""
    ix, orc = small_db
    bases, off = M.synth_reads(ix, seed=77, n_reads=2500, read_len=150)
    want, _ = orc.bin_batch(bases, off, O.default_params(), threads=8)
    assert len(want) > 1000
    for _ in range(2):
        for flags in (ALL_FLAGS[3], ALL_FLAGS[0], ALL_FLAGS[1], ALL_FLAGS[3], ALL_FLAGS[2]):
            ix.to_device(0, flags)
            assert_same_hits(ix.bin_batch(bases, off, M.default_params(), device=0), want)
            assert_same_hits(ix.bin_batch(bases, off, M.default_params(), device=0), want)  # the cached workspace


def _edited_read(rng, src, n_sub, n_del, n_ins):
    """src with n_sub substitutions, n_del bases of src left out and n_ins bases added, at distinct spread-out places
    (outside the first and last 20 bases, so that seeds at both ends keep the window where it belongs)."""
    s = bytearray(src)
    places = rng.sample(range(20, len(s) - 20, 3), n_sub + n_del + n_ins)
    ops = [0] * n_sub + [1] * n_del + [2] * n_ins
    for at, op in sorted(zip(places, ops), reverse=True):
        if op == 0:
            s[at] = rng.choice([c for c in b"ACGT" if c != s[at]])
        elif op == 1:
            del s[at]
        else:
            s.insert(at, rng.choice(b"ACGT"))
    return bytes(s)


def test_edit_distance_bound_of_the_prefilter_in_all_three_outcomes(tmp_path, monkeypatch):
    """k_edit_myers in bound mode decides the SW predicate of index.rs:406 from the unit-cost edit distance D under
    the SW matrix's matches: D <= ED passes, D > 2*ED refutes, in between the sweep decides.  Reads built to land in
    each class, candidates that k_sw_diag's seed-diagonal bounds cannot decide (several indels):
      * few edits with indels                      -> D <= ED: passes, a hit;
      * 14 bases of the reference left out + 8 substitutions: score = L - 2*8 - 14 >= L - 2*ED but D = 22 > ED
        -> undecided by the bound, passed by the sweep, refused by the edit distance (index.rs:410);
      * 26 substitutions: D = 26 in (ED, 2*ED], score = L - 52 < threshold -> undecided, refuted by the sweep;
      * unrelated reads sharing one 18-mer with the database -> D > 2*ED: refuted by the bound alone;
      * the same with N in the read (N matches N in the SW matrix and nothing in the edit distance).
    Hits, candidates examined and the number passed on to the edit distance equal the oracle's; the device counters
    show that every route was taken."""
    rng = random.Random(99)
    entries = [(10 + t, 500 + t, helpers.rnd_seq(rng, 6000)) for t in range(6)]
    # an N run inside the last sequence: reads across it hold N that face N
    body = bytearray(entries[5][2])
    body[3000:3012] = b"N" * 12
    entries[5] = (15, 505, bytes(body))
    ix = M.MGIndex.build(entries, threads=4)
    p = str(tmp_path / "bound.idx")
    ix.write(p)
    orc = O.Index.read(p)
    L = 150
    reads = []
    for i in range(400):
        t = entries[rng.randrange(6)][2]
        st = rng.randrange(0, len(t) - L - 40)
        kind = i % 5
        if kind == 0:
            r = _edited_read(rng, t[st:st + L + 4], 3, 4, 2)[:L]
        elif kind == 1:
            r = _edited_read(rng, t[st:st + L + 14], 8, 14, 0)[:L]
        elif kind == 2:
            r = _edited_read(rng, t[st:st + L], 26, 0, 0)
        elif kind == 3:
            r = bytearray(helpers.rnd_seq(rng, L))
            at = rng.randrange(0, 8) * 15
            r[at:at + 18] = t[st:st + 18]
            r = bytes(r)
        else:
            t5 = entries[5][2]
            st5 = rng.randrange(2900, 2990)
            r = _edited_read(rng, t5[st5:st5 + L + 3], 2, 3, 1)[:L]
        reads.append(r if rng.random() < 0.5 else helpers.revcomp(r))
    mp, op = both_params()
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want, ctr = orc.bin_batch(bases, off, op, threads=8)
    assert len(want) > 100
    ref = None
    for bound in ("1", "0"):
        monkeypatch.setenv("MTSV_SW_BOUND", bound)
        b = M.Batch(ix, 0, len(reads), len(bases))
        b.upload(bases, off)
        b.run(mp)
        assert_same_hits(b.download(), want)
        st = b.stats()
        assert (st["n_verified"], st["window_bytes"], st["n_sw_passed"]) == (ctr["n_sw"], ctr["W"], ctr["n_edit"])
        assert st["n_sw_passed"] > st["n_hits"]              # passed the prefilter, refused by the edit distance
        if bound == "1":
            assert st["n_sw_bound_refuted"] > 50             # refuted without a sweep
            assert st["sw_cell_pairs"] > 0                   # and some left to the sweep
            ref = st["sw_cell_pairs"]
        else:
            assert st["n_sw_bound_refuted"] == 0 and st["sw_cell_pairs"] > ref
        b.close()


def test_run_host_parts_equals_one_batch(tricky, medium):
    """mtsv_batch_run_host_parts: the same reads handed over in several pieces (uneven, an empty one among them, one in
    page-locked memory) give the hits of the one-piece call, read numbers running through the parts"""
    ix, orc, reads = tricky
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want, _ = orc.bin_batch(bases, off, both_params()[1], threads=8)
    cuts = [0, 7, 7, 60, len(reads) // 2, len(reads)]
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        pb, po = helpers.reads_to_batch(reads[a:b])
        parts.append((pb, po))
    b = M.Batch(ix, 0, 64, 1 << 16)            # a workspace far smaller than the batch
    b.run_host_parts(parts)
    assert_same_hits(b.download(), want)
    b.run_host_parts([])                       # no parts at all
    assert len(b.download()) == 0
    b.close()
    mix, mbases, moff = medium                 # 150-base reads: parts cut anywhere, one of them page-locked
    mix.to_device(0)
    n = len(moff) - 1
    whole = mix.bin_batch(mbases, moff, device=0)
    c1, c2 = n // 3 + 11, 2 * n // 3 + 5
    hb = M.HostBuffer(int(moff[c2] - moff[c1]))
    hb.array[:] = mbases[int(moff[c1]):int(moff[c2])]
    parts = [(mbases[: int(moff[c1])], moff[: c1 + 1]), (hb.array, moff[c1: c2 + 1] - moff[c1]),
             (mbases[int(moff[c2]):], moff[c2:] - moff[c2])]
    b = M.Batch(mix, 0, 300_000, 300_000 * 150)
    b.run_host_parts(parts)
    assert_same_hits(b.download(), whole)
    b.close()
    hb.close()


def test_one_lane_workspace_reserved_and_warmed_gives_the_same_hits(tricky, medium):
    """mtsv_batch_create_lanes + mtsv_batch_reserve_host (what mtsv-binner's workers use): a workspace of one lane, sized
    for its host batches and warmed on reads sampled from the index, returns the hits of the default workspace and of the
    oracle; the warm-up leaves nothing behind in the next call's results or counters"""
    ix, orc, reads = tricky
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want, octr = orc.bin_batch(bases, off, both_params()[1], threads=8)
    b = M.Batch(ix, 0, 4096, 1 << 20, lanes=1)
    b.reserve_host(len(reads), len(bases), warm_read_len=100)
    b.run_host(bases, off)
    assert_same_hits(b.download(), want)
    st = b.stats()
    assert st["n_lanes"] == 1 and st["n_reads"] == len(reads)
    assert st["n_verified"] == octr["n_sw"] and st["n_candidates"] == octr["n_cand"] and st["n_seed_hits"] == octr["H"]
    b.close()
    mix, mbases, moff = medium
    mix.to_device(0)
    n = len(moff) - 1
    whole = mix.bin_batch(mbases, moff, device=0)
    for lanes in (1, 2):
        b = M.Batch(mix, 0, n, n * 150, lanes=lanes)
        b.reserve_host(n, len(mbases), warm_read_len=150)
        b.reserve_host(n, len(mbases))           # a second time: nothing to do
        b.run_host(mbases, moff)
        assert_same_hits(b.download(), whole)
        assert b.stats()["n_lanes"] <= lanes
        b.close()
    with pytest.raises(M.MtsvError):
        M.Batch(mix, 0, n, n * 150, lanes=-1)


def test_batches_rich_in_n_overflow_the_listed_search_grid_and_run_again(small_db):
    """seeds with an N in their table part go to a list that a second kernel walks; its grid is sized from the share of
    such seeds in the passes before (an eighth at first), and a pass whose list is longer runs again.  Reads with an N
    every dozen bases (three seeds in four on the list), then clean reads, then N-rich ones again -- on one workspace, so
    that the share is learnt, unlearnt and exceeded again -- against the oracle"""
    ix, orc = small_db
    ix.to_device(0)
    rng = np.random.default_rng(3)
    clean, off = M.synth_reads(ix, seed=77, n_reads=6000, read_len=150)
    dirty = clean.copy()
    dirty[rng.random(len(dirty)) < 0.08] = ord("N")
    mp, op = both_params(edit_rate=0.2)     # (tolerant enough for some N-rich reads to be assigned)
    want_dirty, ctr = orc.bin_batch(dirty, off, op, threads=8)
    want_clean, _ = orc.bin_batch(clean, off, op, threads=8)
    assert len(want_dirty) > 100 and len(want_clean) > 4000
    b = M.Batch(ix, 0, 6000, len(clean))
    for bases, want in ((dirty, want_dirty), (clean, want_clean), (clean, want_clean), (dirty, want_dirty)):
        b.upload(bases, off)
        b.run(mp)
        assert_same_hits(b.download(), want)
    st = b.stats()
    assert st["n_seed_hits"] == ctr["H"] and st["n_candidates"] == ctr["n_cand"]   # the repeated pass is counted once
    b.run_host(dirty, off, mp)
    assert_same_hits(b.download(), want_dirty)
    b.close()


def test_host_batches_arrive_as_four_bit_codes_or_as_plain_bytes_with_the_same_hits(tricky, medium, monkeypatch):
    """run_host packs the bases to 4-bit codes on the host (host_pack.cpp) and k_unpack expands them on the device when
    the process has the CPUs for it (the GPU box has); MTSV_H2D_PLAIN=1 sends the bytes as they are and k_normalise maps
    them there.  Reads of mixed length (odd offsets: chunks share a byte), lower case and junk bytes, in ordinary and in
    page-locked memory, small input segments -- both forms against the oracle"""
    ix, orc, reads = tricky
    bases, off = helpers.reads_to_batch(reads)
    ix.to_device(0)
    want, _ = orc.bin_batch(bases, off, both_params()[1], threads=8)
    hb = M.HostBuffer(len(bases))
    hb.array[:] = bases
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("MTSV_H2D_PLAIN", "1")
        else:
            monkeypatch.delenv("MTSV_H2D_PLAIN", raising=False)
        for src in (bases, hb.array):
            b = M.Batch(ix, 0, 97, 1 << 14)
            b.run_host(src, off)
            assert_same_hits(b.download(), want)
            b.close()
        monkeypatch.setenv("MTSV_ARENA_BASES", "70000")        # several segments: the packed image starts anew in each
        b = M.Batch(ix, 0, 97, 1 << 14)
        b.run_host(hb.array, off)
        assert_same_hits(b.download(), want)
        b.close()
        monkeypatch.delenv("MTSV_ARENA_BASES")
    hb.close()
    mix, mbases, moff = medium
    mix.to_device(0)
    whole = mix.bin_batch(mbases, moff, device=0)            # (packed when the box has the CPUs)
    monkeypatch.setenv("MTSV_H2D_PLAIN", "1")
    assert_same_hits(mix.bin_batch(mbases, moff, device=0), whole)

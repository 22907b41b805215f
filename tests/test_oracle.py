"""CPU tests of the oracle: the reference's own known-answer tests and golden vectors, the
reference's ssw.c (oracle/_ref) and brute force pin the restatement before it is used as checker."""
import json
import os
import random

import numpy as np
import pytest

import helpers
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
HAY = b"ACGACTAGTTATAAAAATTCNACTCCANTTAGCTCCCTACTTTCCGAGAG"


# src/align.rs:100-170 -- the nine known-answer tests of Aligner::min_edit_distance
@pytest.mark.parametrize("needle,haystack,expected", [
    (b"TACGTCAGC", b"AACCCTATGTCATGCCTTGGA", 2),
    (HAY, HAY, 0),
    (b"AAAAAT", HAY, 0),
    (b"", HAY, 0),
    (b"*********", HAY, 9),
    (b"ACGT", b"ACGA", 1),
    (b"ANNGTTCNGNT", HAY, 5),
    (b"***GTTATAA", HAY, 3),
    (b"GTTATAA***", HAY, 3),
])
def test_min_edit_distance_reference_kats(needle, haystack, expected):
    assert O.min_edit_distance(needle, haystack) == expected


def test_candidate_indices_reference_cases():
    # src/index.rs:795-857: bin [100,200), read_len 50, edits 3
    s, e = O.candidate_indices(110, 1, 100, 200, 50, 3)
    assert s < e and s >= 100 and e <= 200 and e - s >= 50 + 2 * 3
    s, e = O.candidate_indices(180, 25, 100, 200, 50, 3)
    assert s < e and s >= 100 and e <= 200 and e - s >= 50 - 3
    assert O.candidate_indices(90, 1, 100, 200, 50, 3) is None  # seed_hits_fail
    # src/index.rs:721-769: second hit 115/3 keeps the start and extends the end
    s1, e1 = O.candidate_indices(110, 1, 0, 100000, 50, 3)
    s2, e2 = O.candidate_indices(115, 3, 0, 100000, 50, 3)
    assert s1 <= s2 < e1 and e2 > e1


def test_write_assignments_reference_vectors():
    # src/binner.rs:440-472
    hits = np.zeros(3, dtype=O.HIT_DTYPE)
    hits["tax_id"] = [2, 2, 5]
    hits["gi"] = [10, 11, 12]
    hits["offset"] = [3, 8, 1]
    hits["edit"] = [7, 4, 9]
    assert O.format_line("R1_1_0_0", hits) == "R1_1_0_0:2=4,5=9\n"
    hits = np.zeros(4, dtype=O.HIT_DTYPE)
    hits["tax_id"] = [2, 2, 2, 5]
    hits["gi"] = [10, 10, 11, 12]
    hits["offset"] = [3, 3, 8, 1]
    hits["edit"] = [7, 4, 6, 9]
    assert O.format_line("R1_1_0_0", hits, True) == "R1_1_0_0:2-10-3=4,2-11-8=6,5-12-1=9\n"
    assert O.format_line("R1", np.zeros(0, dtype=O.HIT_DTYPE)) == ""


def test_ssw_emulation_matches_golden_vectors_from_reference_ssw_c():
    vec = json.load(open(os.path.join(GOLD, "ssw_golden.json")))
    assert len(vec) > 500
    n_word = 0
    for v in vec:
        r, w = v["read"].encode(), v["ref"].encode()
        assert O.ssw_score(r, w) == v["score"], v
        exact = O.sw_exact(r, w)
        if len(r) <= 253:
            # byte kernel == textbook local alignment score (gap 1/1) for every read up to 253 bases
            assert exact == v["score"]
        n_word += exact >= 254
    assert n_word >= 10  # the word-kernel path is exercised


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs the upstream checkout)")
def test_ssw_emulation_matches_compiled_reference_live():
    rng = random.Random(99)
    for L in (40, 100, 150, 253, 254, 320):
        for it in range(60):
            w = helpers.rnd_seq(rng, L + rng.randrange(0, 70), b"ACGTN" if it % 3 == 0 else b"ACGT")
            st = rng.randrange(0, max(1, len(w) - L + 1))
            read = helpers.mutate(rng, w[st:st + L], rng.randrange(0, L // 4)) if it % 2 else helpers.rnd_seq(rng, L)
            if len(read) < 30:
                continue
            assert O.ssw_score(read, w) == O.ref_ssw_scores(read, [w])[0]


def test_edit_within_tolerance_implies_sw_over_threshold():
    # edits <= ED  =>  exact SW >= L - 2*ED (the SW step is a speed prefilter)
    rng = random.Random(3)
    for _ in range(300):
        L = rng.choice([50, 100, 150])
        w = helpers.rnd_seq(rng, L + 40, b"ACGTN")
        read = helpers.mutate(rng, w[20:20 + L], rng.randrange(0, 30))
        if not read:
            continue
        ed_max = int(np.ceil(len(read) * 0.13))
        p = read.replace(b"N", b".")
        if O.min_edit_distance(p, w) <= ed_max:
            assert O.sw_exact(read, w) >= len(read) - 2 * ed_max


def test_fm_search_and_locate_equal_brute_force():
    entries, gene, unit = helpers.tricky_db(seed=3)
    for k, s in ((64, 32), (5, 3), (128, 64)):
        ix = O.Index.build(entries, k, s)
        rng = random.Random(k)
        text = b"".join(e[2] for e in sorted(entries, key=lambda e: e[0]))
        pats = [gene[i:i + 18] for i in range(0, 200, 7)] + [unit[:18], unit[40:58], b"N" * 18, b"ACGTACGTACGTACGTAC"]
        pats += [helpers.rnd_seq(rng, rng.randrange(1, 12)) for _ in range(20)]
        for pat in pats:
            ok, lo, hi = ix.backward_search(pat)
            brute = sorted(ix.brute_find(pat).tolist())
            if not brute:
                assert not ok and lo == hi == 0
                continue
            assert ok and hi - lo == len(brute)
            assert sorted(ix.sa_get(r) for r in range(lo, hi)) == brute


def test_oracle_reproduces_committed_end_to_end_results():
    entries = []
    name = None
    for line in open(os.path.join(GOLD, "e2e_db.fasta")):
        if line.startswith(">"):
            gi, tax = line[1:].split()[0].split("-")
            name = (int(tax), int(gi))
        else:
            entries.append((name[0], name[1], line.strip().encode()))
    reads = [l.rstrip("\n").encode("latin-1") for l in open(os.path.join(GOLD, "e2e_reads.txt"), encoding="latin-1")]
    ix = O.Index.build(entries)
    bases, off = helpers.reads_to_batch(reads)
    stress = dict(max_hits=5, tune_max_hits=2, max_candidates=3, max_assignments=1, min_seed=0.5)
    for name, params in (("default", {}), ("stress", stress)):
        hits, ctr = ix.bin_batch(bases, off, O.default_params(**params), threads=4)
        for long_fmt in (False, True):
            got = "".join(O.format_line(f"r{r}", hits[hits["read"] == r], long_fmt) for r in range(len(reads)))
            want = open(os.path.join(GOLD, f"e2e_{name}{'_long' if long_fmt else ''}.results")).read()
            assert got == want
        assert ctr["R"] == len(hits)


def test_oracle_batch_is_thread_count_invariant_and_counts_work():
    entries, gene, unit = helpers.tricky_db(seed=5)
    ix = O.Index.build(entries)
    reads = helpers.tricky_reads(entries, gene, unit, seed=2, n_each=10)
    bases, off = helpers.reads_to_batch(reads)
    h1, c1 = ix.bin_batch(bases, off, threads=1)
    h4, c4 = ix.bin_batch(bases, off, threads=4)
    helpers.assert_same_hits(h1, h4)
    assert c1 == c4 and c1["X"] > 0 and c1["S"] > 0 and c1["W"] > 0

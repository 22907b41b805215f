"""Randomised parity soak: random small databases, reads of random lengths / damage, random parameters;
the HIP path (both evaluation orders, through the C ABI) against the CPU oracle, every field of every hit
and the work counters.  FUZZ_BIG=1 makes the batches large enough to run as concurrent lanes.  `python tests/fuzz_parity.py [iterations] [seed]`  (needs a GPU; ~1 s per iteration)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import helpers
import mtsv_tools_amd as M
from oracle import oracle as O


def rnd(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(alpha) for _ in range(n))


def one(it, rng, tmp):
    # database: a few taxa, shared segments, tandem repeats, N runs
    shared = rnd(rng, rng.randrange(200, 900))
    unit = rnd(rng, rng.randrange(5, 120))
    entries = []
    gi = 1
    for tax in rng.sample(range(1, 5000), rng.randrange(2, 9)):
        for _ in range(rng.randrange(1, 4)):
            body = bytearray(rnd(rng, rng.randrange(300, 4000)))
            if rng.random() < 0.6:
                s = bytearray(shared)
                for _ in range(rng.randrange(0, len(s) // 20 + 1)):
                    s[rng.randrange(len(s))] = rng.choice(b"ACGT")
                p = rng.randrange(0, len(body))
                body[p:p] = s
            if rng.random() < 0.3:
                p = rng.randrange(0, len(body))
                body[p:p] = unit * rng.randrange(2, 30)
            if rng.random() < 0.3:
                p = rng.randrange(0, len(body))
                body[p:p + rng.randrange(1, 60)] = b"N" * rng.randrange(1, 60)
            entries.append((tax, gi, bytes(body)))
            gi += 1
    rng.shuffle(entries)
    entries.sort(key=lambda e: e[0])          # bins must be grouped by ascending TaxId (index.rs:523-541)
    max_len = rng.choice((40, 75, 100, 150, 151, 200, 250, 253, 256, 257, 300, 400, 600, 1200))  # > 256: passes of the tiled kernel
    texts = [e[2] for e in entries]
    reads = []
    n_reads = rng.randrange(200, 1500) * (90 if os.environ.get("FUZZ_BIG") else 1)  # FUZZ_BIG: batches that run as lanes
    for _ in range(n_reads):
        L = max_len if rng.random() < 0.3 else rng.randrange(1, max_len + 1)
        t = rng.choice(texts)
        if len(t) <= L or rng.random() < 0.1:
            r = rnd(rng, L, b"ACGTN")
        else:
            st = rng.randrange(0, len(t) - L)
            r = bytearray(t[st:st + L])
            kind = rng.randrange(4)
            if kind == 0:
                for _ in range(rng.randrange(0, L // 5 + 1)):
                    r[rng.randrange(L)] = rng.choice(b"ACGTNacgtnX-")
            elif kind == 1:
                r = bytearray(helpers.mutate(rng, bytes(r), rng.randrange(0, L // 6 + 2)))[:max_len]
            r = bytes(r)
        reads.append(r if rng.random() < 0.5 else helpers.revcomp(r))
    over = {}
    if rng.random() < 0.7:
        over["edit_rate"] = rng.choice((0.0, 0.05, 0.1, 0.13, 0.2, 0.3, 0.5))
    if rng.random() < 0.5:
        over["seed_size"] = rng.choice((8, 12, 16, 18, 20, 21, 22, 24, 32, 33))
    if rng.random() < 0.5:
        over["seed_interval"] = rng.choice((1, 2, 5, 15, 30))
    if rng.random() < 0.4:
        over["min_seed"] = rng.choice((0.015, 0.1, 0.3, 0.6, 1.0))
    if rng.random() < 0.4:
        over["max_hits"] = rng.choice((1, 5, 50, 2000, 100000))
        over["tune_max_hits"] = rng.choice((1, 3, 20, 200))
    if rng.random() < 0.3:
        over["max_candidates"] = rng.choice((1, 2, 5, 50))
    if rng.random() < 0.3:
        over["max_assignments"] = rng.choice((1, 2, 5))
    ix = M.MGIndex.build(entries, occ_k=rng.choice((16, 64, 128)), sa_s=rng.choice((1, 8, 32)), threads=4)
    path = os.path.join(tmp, "f.idx")
    ix.write(path)
    orc = O.Index.read(path)
    # a k-mer table of 12 or 13 symbols (far wider than these small databases would get) sends the seeds of 16..24
    # symbols through k_search_fast and its in-workgroup queue for seeds with an N
    kk = rng.choice((None, None, "12", "13"))
    if kk:
        os.environ["MTSV_KMER_K"] = kk
    ix.to_device(0, rng.choice((0, 0, 0, 1, 2, 3)))
    os.environ.pop("MTSV_KMER_K", None)
    bases, off = helpers.reads_to_batch(reads)
    want, ctr = orc.bin_batch(bases, off, O.default_params(**over), threads=8)
    mp = M.default_params(**over)
    b = M.Batch(ix, 0, len(reads), len(bases))
    b.upload(bases, off)
    for mode in (0, 1):
        b.set_verify_mode(mode)
        b.run(mp)
        got = b.download()
        helpers.assert_same_hits(got, want)
        st = b.stats()
        # (with max_assignments the reference stops early; the device verifies every TaxId chain and applies the
        #  cut-off afterwards, so only the hits are comparable then)
        if mode == 0 and "max_assignments" not in over:
            assert (st["n_seed_hits"], st["n_candidates"], st["n_verified"], st["window_bytes"]) == \
                   (ctr["H"], ctr["n_cand"], ctr["n_sw"], ctr["W"]), (st, ctr)
    b.close()
    ix.close()
    return len(reads), len(want), max_len, over


if __name__ == "__main__":
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for it in range(n_it):
            rng = random.Random(seed0 * 100003 + it)
            try:
                info = one(it, rng, tmp)
            except Exception:
                print(f"FAILED at iteration {it} (seed {seed0})", flush=True)
                raise
            if it % 10 == 0:
                print(f"iteration {it}: reads {info[0]} hits {info[1]} max_len {info[2]} {info[3]}", flush=True)
    print(f"fuzz ok: {n_it} iterations")

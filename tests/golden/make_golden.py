#!/usr/bin/env python3
"""Regenerates the committed golden vectors under tests/golden/.

  ssw_golden.json     score1 of the REFERENCE's own striped Smith-Waterman (ssw/src/ssw.c compiled
                      into oracle/_ref/libssw_ref.so by oracle/Makefile), called exactly as
                      ssw/src/lib.rs:36-84 does, for seeded (read, window) pairs incl. N's and reads
                      of 254+ bases (word kernel).  Needs the upstream checkout; data only.
  e2e_db.fasta / e2e_reads.txt / e2e_*.results
                      a small seeded database + reads and the result lines of the CPU oracle for
                      default and stress parameters (the reference holds no end-to-end vector:
                      index.rs:239 "TODO test this function").
  tiny.idx / tiny.idx.json
                      an MG-index file written by the product's writer for the literal database of
                      the reference's own test (index.rs:860-873) and its decoded fields.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import helpers  # noqa: E402
import mtsv_tools_amd as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

STRESS = dict(max_hits=5, tune_max_hits=2, max_candidates=3, max_assignments=1, min_seed=0.5)


def ssw_vectors():
    rng = random.Random(20240229)
    out = []
    for L in (30, 50, 100, 150, 253, 254, 300):
        for it in range(110):
            w = helpers.rnd_seq(rng, L + rng.randrange(0, 90), b"ACGTN" if it % 4 == 0 else b"ACGT")
            st = rng.randrange(0, max(1, len(w) - L + 1))
            if it % 3 == 2:
                read = helpers.rnd_seq(rng, L)
            else:
                read = helpers.mutate(rng, w[st:st + L], rng.randrange(0, 2 * L // 5))
            if len(read) < 30:
                continue
            out.append({"read": read.decode(), "ref": w.decode(), "score": O.ref_ssw_scores(read, [w])[0]})
    return out


def main():
    if not O.ref_available():
        sys.exit("oracle/_ref/libssw_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    with open(os.path.join(HERE, "ssw_golden.json"), "w") as f:
        json.dump(ssw_vectors(), f, separators=(",", ":"))

    entries, gene, unit = helpers.tricky_db(seed=7)
    with open(os.path.join(HERE, "e2e_db.fasta"), "w") as f:
        for tax, gi, seq in entries:
            f.write(f">{gi}-{tax} synthetic\n{seq.decode()}\n")
    reads = helpers.tricky_reads(entries, gene, unit, seed=11, n_each=40, lengths=(100, 150))
    reads = [r for r in reads if len(r) > 0]
    with open(os.path.join(HERE, "e2e_reads.txt"), "w") as f:
        for r in reads:
            f.write(r.decode("latin-1") + "\n")
    oix = O.Index.build(entries)
    bases, off = helpers.reads_to_batch(reads)
    for name, params in (("default", {}), ("stress", STRESS)):
        hits, _ = oix.bin_batch(bases, off, O.default_params(**params), threads=8)
        for long_fmt in (False, True):
            lines = []
            for r in range(len(reads)):
                h = hits[hits["read"] == r]
                lines.append(O.format_line(f"r{r}", h, long_fmt))
            with open(os.path.join(HERE, f"e2e_{name}{'_long' if long_fmt else ''}.results"), "w") as f:
                f.write("".join(lines))

    # the literal database of index.rs:860-873
    tiny = [(1, 10, b"ACGT"), (1, 11, b"TTAA"), (2, 20, b"GG")]
    ix = M.MGIndex.build(tiny, occ_k=8, sa_s=8, threads=1)
    p = os.path.join(HERE, "tiny.idx")
    ix.write(p)
    raw = open(p, "rb").read()
    text = b"ACGTTTAAGG$"
    sa = sorted(range(len(text)), key=lambda i: text[i:])
    bwt = bytes(text[i - 1] if i else text[-1] for i in sa)
    with open(p + ".json", "w") as f:
        json.dump({"sequences": text.decode(), "bins": [[10, 1, 0, 4], [11, 1, 4, 8], [20, 2, 8, 10]],
                   "suffix_array": sa, "bwt": bwt.decode(), "k": 8, "s": 8, "bytes": len(raw)}, f)


if __name__ == "__main__":
    main()

"""Shared generators for the test-suite: adversarial databases and read sets that reach the
order-dependent corners of the reference's hot loop (duplicate TaxIds, tie-breaking in the stable
rank sort, merged windows, seed thinning, bin-boundary clipping, N handling)."""
import random

import numpy as np

FIELDS = ("read", "tax_id", "gi", "edit", "strand", "offset")
COMP = {65: 84, 67: 71, 71: 67, 84: 65}


def rnd_seq(rng, n, alpha=b"ACGT"):
    return bytes(rng.choice(alpha) for _ in range(n))


def mutate(rng, s, n_edits, alpha=b"ACGTN"):
    s = bytearray(s)
    for _ in range(n_edits):
        if not s:
            break
        op = rng.randrange(3)
        i = rng.randrange(len(s))
        if op == 0:
            s[i] = rng.choice(alpha)
        elif op == 1:
            del s[i]
        else:
            s.insert(i, rng.choice(b"ACGT"))
    return bytes(s)


def revcomp(s):
    return bytes(COMP.get(c, 78) for c in reversed(s.upper()))


def tricky_db(seed=7):
    """Returns (entries, gene, unit): entries = (tax, gi, seq) in database (file) order, gene = the
    conserved segment planted in many taxa, unit = the tandem-repeat unit."""
    rng = random.Random(seed)
    gene = rnd_seq(rng, 700)          # conserved gene present in many taxa / GIs
    unit = rnd_seq(rng, 97)           # tandem repeat unit
    entries = []
    gi = 1000
    taxa = [9, 2, 77, 40, 5, 123456, 31, 8, 4000000000, 17, 64, 3]
    for ti, tax in enumerate(taxa):
        for g in range(3):
            body = bytearray(rnd_seq(rng, rng.randrange(1500, 3000)))
            if ti < 8:  # conserved gene with 0..4 % divergence, several GIs per taxon
                at = rng.randrange(100, len(body) - 800)
                body[at:at + 700] = mutate(rng, gene, rng.randrange(0, 28), b"ACGT")[:700].ljust(700, b"A")
            if ti == 1 and g == 0:  # long tandem repeat: every seed hits ~40 sites, windows merge
                at = 50
                body[at:at + 97 * 40] = unit * 40
            if g == 1:  # runs of N and soft-masked / IUPAC bytes (index.rs:543-553)
                p = rng.randrange(0, len(body) - 200)
                body[p:p + rng.randrange(20, 120)] = b"N" * rng.randrange(20, 120)
                q = rng.randrange(0, len(body) - 60)
                body[q:q + 40] = bytes(body[q:q + 40]).lower()
                body[rng.randrange(len(body))] = ord("R")
            entries.append((tax, gi, bytes(body)))
            gi += rng.randrange(1, 50)
    # very short and empty sequences, same TaxId twice in different places of the file
    entries.append((2, 5, rnd_seq(rng, 40)))
    entries.append((2, 6, b""))
    entries.append((9, 7, rnd_seq(rng, 160)))
    entries.append((1, 8, rnd_seq(rng, 19)))
    rng.shuffle(entries)
    return entries, gene, unit


def reads_to_batch(reads):
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    return bases, off


def tricky_reads(entries, gene, unit, seed=11, n_each=60, lengths=(150,)):
    """Reads aimed at the corners: conserved gene (many TaxIds / duplicate TaxIds), tandem repeat
    (hundreds of seed hits, merged windows), plain sequence with 0..ED+5 edits, reverse strand,
    N-rich, lower case / junk bytes, bin-boundary spanning, too short for a seed, empty."""
    rng = random.Random(seed)
    text_by_entry = [e[2].upper() for e in entries]
    long_entries = [t for t in text_by_entry if len(t) > 400]
    reads = []
    for L in lengths:
        for _ in range(n_each):  # conserved gene
            st = rng.randrange(0, len(gene) - L) if len(gene) > L else 0
            r = mutate(rng, gene[st:st + L], rng.randrange(0, 26))
            reads.append(r if rng.random() < 0.5 else revcomp(r))
        for _ in range(n_each // 2):  # tandem repeat
            rep = unit * 5
            st = rng.randrange(0, len(rep) - L) if len(rep) > L else 0
            reads.append(mutate(rng, rep[st:st + L], rng.randrange(0, 12)))
        for _ in range(n_each):  # ordinary reads with edit counts around the tolerance
            t = rng.choice(long_entries)
            st = rng.randrange(0, len(t) - L)
            r = mutate(rng, t[st:st + L], rng.choice([0, 1, 3, 8, 15, 19, 20, 21, 25, 40]))
            if rng.random() < 0.5:
                r = revcomp(r)
            if rng.random() < 0.2:
                r = r.lower()
            if rng.random() < 0.1:
                r = bytes(c if rng.random() > 0.05 else rng.choice(b"nRYK-*.") for c in r)
            reads.append(r)
        for _ in range(n_each // 3):  # spanning the junction of two database sequences
            a, b = rng.sample(long_entries, 2)
            k = rng.randrange(20, L - 20)
            reads.append(a[len(a) - k:] + b[:L - k])
    reads += [b"", b"A", b"ACGTACGTACGTACGTA", b"ACGTACGTACGTACGTAC", b"ACGTACGTACGTACGTACG",
              b"N" * 60, b"NNNNNNNNNNNNNNNNNN" + gene[:100], gene[:120] + b"N" * 30,
              rnd_seq(rng, 253), gene[:253], gene[100:130]]
    rng.shuffle(reads)
    return reads


def assert_same_hits(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for f in FIELDS:
        bad = np.nonzero(got[f] != want[f])[0]
        assert len(bad) == 0, (f, bad[:5], got[bad[:5]], want[bad[:5]])

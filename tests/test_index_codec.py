"""MG-index file codec and builder: byte-identical to the oracle's restatement of MGIndex::new +
bincode, loud failures on malformed files, the reference's own literal database."""
import json
import os

import pytest

import helpers
import mtsv_tools_amd as M
from mtsv_tools_amd import _lib
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("k,s", [(64, 32), (16, 32), (3, 5), (128, 7), (1, 1)])
def test_builder_and_writer_are_byte_identical_to_oracle(tmp_path, k, s):
    entries, _, _ = helpers.tricky_db(seed=k + s)
    a, b = str(tmp_path / "o.idx"), str(tmp_path / "m.idx")
    O.Index.build(entries, k, s).write(a)
    M.MGIndex.build(entries, k, s, threads=3).write(b)
    assert open(a, "rb").read() == open(b, "rb").read()
    # each side reads the other's file
    info = M.MGIndex.load(a).info()
    assert info["occ_k"] == k and info["sa_s"] == s and info["n_bins"] == len(entries)
    O.Index.read(b)


def test_committed_tiny_index_matches_its_decoded_fields(tmp_path):
    meta = json.load(open(os.path.join(GOLD, "tiny.idx.json")))
    raw = open(os.path.join(GOLD, "tiny.idx"), "rb").read()
    assert len(raw) == meta["bytes"]
    # u64 n, then the sequences of index.rs:860-873 in TaxId order with the sentinel (index.rs:555)
    assert int.from_bytes(raw[:8], "little") == len(meta["sequences"])
    assert raw[8:8 + len(meta["sequences"])].decode() == meta["sequences"]
    p = 8 + len(meta["sequences"])
    assert int.from_bytes(raw[p:p + 8], "little") == 3
    for i, (gi, tax, start, end) in enumerate(meta["bins"]):  # gi, tax_id, start, end (index.rs:45-54)
        q = p + 8 + 24 * i
        assert int.from_bytes(raw[q:q + 4], "little") == gi
        assert int.from_bytes(raw[q + 4:q + 8], "little") == tax
        assert int.from_bytes(raw[q + 8:q + 16], "little") == start
        assert int.from_bytes(raw[q + 16:q + 24], "little") == end
    q = p + 8 + 72
    assert int.from_bytes(raw[q:q + 8], "little") == len(meta["bwt"])
    assert raw[q + 8:q + 8 + len(meta["bwt"])].decode() == meta["bwt"]
    # rebuilding gives the same bytes; the loader accepts them
    ix = M.MGIndex.build([(1, 10, b"ACGT"), (1, 11, b"TTAA"), (2, 20, b"GG")], occ_k=8, sa_s=8, threads=1)
    out = str(tmp_path / "t.idx")
    ix.write(out)
    assert open(out, "rb").read() == raw
    assert M.MGIndex.load(os.path.join(GOLD, "tiny.idx")).info()["n"] == 11


def test_lowercase_database_builds_the_same_index(tmp_path):
    # src/index.rs:772-792 construct_index_lowercase
    entries, _, _ = helpers.tricky_db(seed=1)
    lower = [(t, g, s.lower()) for t, g, s in entries]
    a, b = str(tmp_path / "u.idx"), str(tmp_path / "l.idx")
    M.MGIndex.build(entries, threads=2).write(a)
    M.MGIndex.build(lower, threads=2).write(b)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_fasta_builder_follows_header_grammar(tmp_path):
    entries, _, _ = helpers.tricky_db(seed=2)
    fa = tmp_path / "db.fasta"
    with open(fa, "w") as f:
        for tax, gi, seq in entries:
            f.write(f">{gi}-{tax} some description\n")
            s = seq.decode()
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
    a, b = str(tmp_path / "a.idx"), str(tmp_path / "b.idx")
    M.MGIndex.build_fasta(str(fa), threads=2).write(a)
    M.MGIndex.build(entries, threads=2).write(b)
    assert open(a, "rb").read() == open(b, "rb").read()
    for bad in (">12_34\nACGT\n", ">1-2-3\nACGT\n", ">x-2\nACGT\n", "ACGT\n"):  # util.rs:26-56
        p = tmp_path / "bad.fasta"
        p.write_text(bad)
        with pytest.raises(M.MtsvError) as e:
            M.MGIndex.build_fasta(str(p))
        assert e.value.code == _lib.E_FORMAT


def test_loader_fails_loudly_on_malformed_files(tmp_path):
    good = open(os.path.join(GOLD, "tiny.idx"), "rb").read()
    cases = {
        "truncated": good[:-5],
        "trailing": good + b"\0",
        "empty": b"",
        "huge_len": (1 << 60).to_bytes(8, "little") + good[8:],
        "bad_sentinel": good[:-1] + b"#",
    }
    bwt_at = 8 + 11 + 8 + 72 + 8
    flipped = bytearray(good)
    flipped[bwt_at] = ord("T") if flipped[bwt_at] != ord("T") else ord("A")  # bwt no longer matches less/occ
    cases["corrupt_bwt"] = bytes(flipped)
    for name, blob in cases.items():
        p = tmp_path / f"{name}.idx"
        p.write_bytes(blob)
        with pytest.raises(M.MtsvError) as e:
            M.MGIndex.load(str(p))
        assert e.value.code == _lib.E_FORMAT, name
    with pytest.raises(M.MtsvError) as e:
        M.MGIndex.load(str(tmp_path / "does_not_exist.idx"))
    assert e.value.code == _lib.E_IO

"""Block-parallel FASTA/FASTQ ingest of mtsv-binner (mtsv_tools_amd/csrc/fastx_ingest.hpp) against the
serial reader that mirrors bio::io::{fasta,fastq} as used at src/binner.rs:169-199.  `--parse-only`
prints record / base counts and checksums of ids, bases and read lengths; it needs no index or GPU."""
import os
import random
import subprocess

import pytest

BIN = os.path.join(os.path.dirname(__file__), "..", "mtsv_tools_amd", "bin", "mtsv-binner")


def parse_only(path, fastq, serial=False, block=None, threads=4, offset=0, batch=None):
    env = dict(os.environ, MTSV_HOST_THREADS=str(threads))
    if serial:
        env["MTSV_SERIAL_INGEST"] = "1"
    if block:
        env["MTSV_INGEST_BLOCK"] = str(block)
    cmd = [BIN, "--parse-only", "--fastq" if fastq else "--fasta", str(path), "--read-offset", str(offset)]
    if batch:
        cmd += ["--batch-reads", str(batch)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    return r.returncode, last


def rand_seq(rng, n):
    return "".join(rng.choice("ACGTN") for _ in range(n))


def fastq_text(rng, n, crlf=False, at_quality=True):
    nl = "\r\n" if crlf else "\n"
    out = []
    for i in range(n):
        L = rng.randint(1, 200)
        q = "".join(rng.choice("@+IJK#5") for _ in range(L))
        if at_quality and i % 3 == 0:
            q = "@" + q[1:]            # quality lines that look like headers
        if i % 5 == 0:
            q = "+" + q[1:]
        out.append(f"@r{i} desc {i}{nl}{rand_seq(rng, L)}{nl}+{nl}{q}{nl}")
    return "".join(out)


@pytest.mark.parametrize("crlf", [False, True])
def test_fastq_blocks_equal_serial(tmp_path, crlf):
    rng = random.Random(5)
    p = tmp_path / "a.fq"
    p.write_text(fastq_text(rng, 3000, crlf=crlf), newline="")
    want = parse_only(p, True, serial=True)
    assert want[0] == 0 and want[1].startswith("records=3000 ")
    for block in (1, 257, 4096, 100000, None):
        for threads in (1, 5):
            assert parse_only(p, True, block=block, threads=threads) == want
    # read offset and batch cutting
    w2 = parse_only(p, True, serial=True, offset=1234, batch=100)
    assert w2[1].startswith("records=1766 ")
    assert parse_only(p, True, block=4096, offset=1234, batch=100) == w2
    assert parse_only(p, True, block=4096, offset=5000) == parse_only(p, True, serial=True, offset=5000)


def test_fastq_no_trailing_newline_and_blank_tail(tmp_path):
    rng = random.Random(6)
    text = fastq_text(rng, 500)
    for name, t in (("nonl.fq", text[:-1]), ("blank.fq", text + "\n\n"), ("one.fq", "@x\nACGT\n+\nIIII")):
        p = tmp_path / name
        p.write_text(t)
        want = parse_only(p, True, serial=True)
        assert want[0] == 0
        assert parse_only(p, True, block=1000) == want


def test_wrapped_fastq_falls_back_to_the_serial_reader(tmp_path):
    rng = random.Random(7)
    recs = []
    for i in range(800):
        L = rng.randint(50, 150)
        s, q = rand_seq(rng, L), "".join(rng.choice("@+IJ") for _ in range(L))
        if i >= 400:  # second half: sequence and quality wrapped over several lines
            s = "\n".join(s[k:k + 40] for k in range(0, L, 40))
            q = "\n".join(q[k:k + 40] for k in range(0, L, 40))
        recs.append(f"@w{i}\n{s}\n+w{i}\n{q}\n")
    p = tmp_path / "w.fq"
    p.write_text("".join(recs))
    want = parse_only(p, True, serial=True)
    assert want[0] == 0 and want[1].startswith("records=800 ")
    for block in (500, 5000, None):
        assert parse_only(p, True, block=block) == want


def test_broken_fastq_reports_like_the_serial_reader(tmp_path):
    rng = random.Random(8)
    text = fastq_text(rng, 300)
    cases = {"trunc.fq": text[: len(text) // 2 + 7], "nohdr.fq": text.replace("@r150 ", "r150 ", 1),
             "qlen.fq": text + "@z\nACGT\n+\nII\n"}
    for name, t in cases.items():
        p = tmp_path / name
        p.write_text(t)
        want = parse_only(p, True, serial=True)
        got = parse_only(p, True, block=2000)
        assert got[0] == want[0], name
        if want[0] == 0:
            assert got == want
        else:
            assert want[0] == 12   # binner.rs:81-84


def test_fasta_blocks_equal_serial(tmp_path):
    rng = random.Random(9)
    recs = ["\n\n"]
    for i in range(1500):
        L = rng.randint(0, 400)
        s = rand_seq(rng, L)
        width = rng.choice((60, 70, 1000))
        body = "\n".join(s[k:k + width] for k in range(0, L, width))
        recs.append(f">s{i}\tsome text\n{body}\n" + ("\n" if i % 7 == 0 else ""))
    recs.append(">")  # empty id, no sequence, no newline
    p = tmp_path / "a.fa"
    p.write_text("".join(recs))
    want = parse_only(p, False, serial=True)
    assert want[0] == 0 and want[1].startswith("records=1501 ")
    for block in (1, 300, 10000, None):
        assert parse_only(p, False, block=block, threads=3) == want
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n>x\nACGT\n")
    assert parse_only(bad, False, serial=True)[0] == 12
    assert parse_only(bad, False, block=4)[0] == 12


def _gz(path, text, level=6, members=1):
    import gzip
    data = text.encode("latin-1")
    step = (len(data) + members - 1) // members or 1
    with open(path, "wb") as f:
        for i in range(0, max(len(data), 1), step):
            f.write(gzip.compress(data[i:i + step], level))


def parse_gz(path, fastq, serial=False, chunk=None, block=None, threads=4):
    env = dict(os.environ, MTSV_HOST_THREADS=str(threads))
    if serial:
        env["MTSV_SERIAL_GZIP"] = "1"
    if chunk:
        env["MTSV_PGZIP_CHUNK"] = str(chunk)
    if block:
        env["MTSV_INGEST_BLOCK"] = str(block)
    r = subprocess.run([BIN, "--parse-only", "--fastq" if fastq else "--fasta", str(path)], env=env, capture_output=True, text=True, timeout=120)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    return r.returncode, last


def test_gzip_input_parallel_inflate_equals_the_serial_stream(tmp_path):
    """binner.rs:21-33 opens .gz through one inflate stream; here the stream is inflated in parallel (pgzip.hpp:
    entry points found by search, windows resolved afterwards, CRC-32 of every member checked) and must hand the
    parsers exactly the bytes zlib would: compression levels, several members, small inflater chunks (many entry
    points, blocks cut inside them), CRLF text, FASTA, tiny and empty files."""
    rng = random.Random(10)
    text = fastq_text(rng, 20000)                      # ~4.5 MB
    (tmp_path / "a.fq").write_text(text)
    want = parse_only(tmp_path / "a.fq", True, serial=True)
    assert want[0] == 0 and want[1].startswith("records=20000 ")
    for level, members in ((1, 1), (6, 1), (9, 1), (6, 5), (0, 1)):
        p = tmp_path / f"a_{level}_{members}.fq.gz"
        _gz(p, text, level, members)
        assert parse_gz(p, True, serial=True) == want
        for chunk, block, threads in ((None, None, 4), (65536, 100000, 6), (70000, 4096, 3), (200000, None, 1)):
            assert parse_gz(p, True, chunk=chunk, block=block, threads=threads) == want, (level, members, chunk)
    crlf = fastq_text(rng, 3000, crlf=True)
    _gz(tmp_path / "crlf.fq.gz", crlf)
    (tmp_path / "crlf.fq").write_text(crlf, newline="")
    assert parse_gz(tmp_path / "crlf.fq.gz", True, chunk=65536, block=50000) == parse_only(tmp_path / "crlf.fq", True, serial=True)
    for name, t in (("tiny.fq.gz", "@x\nACGT\n+\nIIII\n"), ("nonl.fq.gz", "@x\nACGT\n+\nIIII"), ("empty.fq.gz", "")):
        _gz(tmp_path / name, t)
        assert parse_gz(tmp_path / name, True) == parse_gz(tmp_path / name, True, serial=True)
        assert parse_gz(tmp_path / name, True)[0] == 0
    # FASTA
    recs = []
    for i in range(4000):
        s = rand_seq(rng, rng.randint(0, 500))
        recs.append(f">s{i} text\n" + "\n".join(s[k:k + 70] for k in range(0, len(s), 70)) + "\n")
    fa = "".join(recs)
    _gz(tmp_path / "a.fa.gz", fa, 6, 3)
    (tmp_path / "a.fa").write_text(fa)
    assert parse_gz(tmp_path / "a.fa.gz", False, chunk=65536, block=30000) == parse_only(tmp_path / "a.fa", False, serial=True)


def test_gzip_block_types_and_code_shapes(tmp_path):
    """every deflate block type behind an entry point of the parallel inflater, and code sets of other shapes than
    gzip -6 makes: members compressed with Z_FIXED (fixed Huffman blocks), Z_HUFFMAN_ONLY (no matches, one unused
    distance tree), Z_RLE (distance 1 only: a single distance code), level 0 (stored), window bits 9 (short
    distances) -- between ordinary members, so that chunks start in one kind and run into the next"""
    import zlib
    rng = random.Random(12)
    text = fastq_text(rng, 24000, at_quality=False).encode()
    want_path = tmp_path / "s.fq"
    want_path.write_bytes(text)
    want = parse_only(want_path, True, serial=True)
    assert want[0] == 0
    cuts = [0]
    for k in range(1, 12):  # cut between records
        cuts.append(text.index(b"\n@r", len(text) * k // 12) + 1)
    cuts.append(len(text))
    shapes = [(6, zlib.Z_DEFAULT_STRATEGY, 15), (6, zlib.Z_FIXED, 15), (6, zlib.Z_DEFAULT_STRATEGY, 15), (6, zlib.Z_HUFFMAN_ONLY, 15),
              (9, zlib.Z_DEFAULT_STRATEGY, 15), (6, zlib.Z_RLE, 15), (1, zlib.Z_DEFAULT_STRATEGY, 15), (0, zlib.Z_DEFAULT_STRATEGY, 15),
              (6, zlib.Z_DEFAULT_STRATEGY, 9), (6, zlib.Z_FILTERED, 15), (6, zlib.Z_FIXED, 9), (6, zlib.Z_DEFAULT_STRATEGY, 15)]
    p = tmp_path / "s.fq.gz"
    with open(p, "wb") as f:
        for (level, strategy, wbits), a, b in zip(shapes, cuts, cuts[1:]):
            c = zlib.compressobj(level, zlib.DEFLATED, 16 + wbits, 9, strategy)
            f.write(c.compress(text[a:b]) + c.flush())
    assert parse_gz(p, True, serial=True) == want
    for chunk, threads in ((None, 4), (65536, 6), (150000, 3), (400000, 2)):
        assert parse_gz(p, True, chunk=chunk, threads=threads) == want, (chunk, threads)


def test_gzip_input_irregular_and_corrupt_files(tmp_path):
    """wrapped FASTQ inside a .gz falls back to the serial reader from the offset of the first block the strict
    parse rejects; truncated or damaged gzip data ends like it does with the serial stream (exit 12)"""
    rng = random.Random(11)
    recs = []
    for i in range(6000):
        L = rng.randint(50, 150)
        s, q = rand_seq(rng, L), "".join(rng.choice("@+IJ") for _ in range(L))
        if i >= 3000:
            s = "\n".join(s[k:k + 40] for k in range(0, L, 40))
            q = "\n".join(q[k:k + 40] for k in range(0, L, 40))
        recs.append(f"@w{i}\n{s}\n+w{i}\n{q}\n")
    text = "".join(recs)
    _gz(tmp_path / "w.fq.gz", text)
    (tmp_path / "w.fq").write_text(text)
    want = parse_only(tmp_path / "w.fq", True, serial=True)
    assert want[0] == 0 and want[1].startswith("records=6000 ")
    assert parse_gz(tmp_path / "w.fq.gz", True, chunk=65536, block=20000) == want
    assert parse_gz(tmp_path / "w.fq.gz", True) == want
    good = (tmp_path / "w.fq.gz").read_bytes()
    (tmp_path / "trunc.fq.gz").write_bytes(good[: len(good) // 2])
    bad = bytearray(good)
    for k in range(len(bad) // 3, len(bad) // 3 + 40):
        bad[k] ^= 0x5a
    (tmp_path / "bad.fq.gz").write_bytes(bytes(bad))
    # every deflate block intact, only the member's CRC-32 in the trailer wrong: the parallel inflater checks a member at its
    # end, after its earlier rounds have been handed on -- that must fail the run like the serial stream does, not resume
    crc = bytearray(good)
    crc[-8] ^= 0x01
    (tmp_path / "crc.fq.gz").write_bytes(bytes(crc))
    for name in ("trunc.fq.gz", "bad.fq.gz", "crc.fq.gz"):
        s = parse_gz(tmp_path / name, True, serial=True)
        g = parse_gz(tmp_path / name, True, chunk=65536)
        assert s[0] == 12 and g[0] == 12, (name, s, g)

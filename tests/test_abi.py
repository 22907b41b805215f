"""CPU tests of the product's C ABI: the library loads, exports every symbol include/mtsv_amd.h
declares, mirrors the reference's defaults and output grammar, and refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers
import mtsv_tools_amd as M
from mtsv_tools_amd import _lib
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mtsv_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mtsv_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(M.lib_path())
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mtsv_amd.h but not exported"
    assert set(names) == set(_lib.EXPORTS)


def test_struct_layouts_match_header():
    assert M.HIT_DTYPE.itemsize == 32
    assert ctypes.sizeof(M.Params) == 56


def test_default_params_are_the_cli_defaults():
    # src/bin/mtsv-binner.rs:63-94
    p = M.default_params()
    assert (p.edit_rate, p.seed_size, p.seed_interval, p.min_seed) == (0.13, 18, 15, 0.015)
    assert (p.max_hits, p.tune_max_hits, p.max_assignments, p.max_candidates) == (2000, 200, -1, -1)


def test_no_cpu_fallback(tmp_path):
    if M.device_count() > 0:
        pytest.skip("a GPU is present")
    ix = M.MGIndex.build([(1, 10, b"ACGTACGTAACCGGTTACGATCGATCGATCGTAGC" * 4)], threads=1)
    bases, off = helpers.reads_to_batch([b"ACGTACGTAACCGGTTACGATCGATC"])
    with pytest.raises(M.MtsvError) as e:
        ix.bin_batch(bases, off, device=0)
    assert e.value.code == _lib.E_DEVICE
    with pytest.raises(M.MtsvError) as e:
        ix.bin_batch(bases, off, device=-1)  # the survey's "-1 = CPU oracle" is deliberately not offered
    assert e.value.code == _lib.E_ARG
    # the multi-GPU entry points fail the same way (every device thread's error reaches the caller)
    with pytest.raises(M.MtsvError) as e:
        M.bin_batch_multi(ix, [0, 1], bases, off)
    assert e.value.code == _lib.E_DEVICE
    with pytest.raises(M.MtsvError) as e:
        M.bin_batch_chunks([ix, ix], [0, 0], bases, off)
    assert e.value.code == _lib.E_DEVICE
    with pytest.raises(M.MtsvError) as e:
        M.bin_batch_multi(ix, [0, -1], bases, off)
    assert e.value.code in (_lib.E_ARG, _lib.E_DEVICE)
    assert M.bin_batch_slice_reads(10) == 1024 and M.bin_batch_slice_reads(10**9) == 3 << 20


def test_format_results_equals_reference_vectors_and_oracle():
    hits = np.zeros(3, dtype=M.HIT_DTYPE)
    hits["read"] = 1
    hits["tax_id"] = [2, 2, 5]
    hits["gi"] = [10, 11, 12]
    hits["offset"] = [3, 8, 1]
    hits["edit"] = [7, 4, 9]
    assert M.format_results(hits, ["R0", "R1_1_0_0"]) == "R1_1_0_0:2=4,5=9\n"  # binner.rs:440-455
    hits = np.zeros(4, dtype=M.HIT_DTYPE)
    hits["tax_id"] = [2, 2, 2, 5]
    hits["gi"] = [10, 10, 11, 12]
    hits["offset"] = [3, 3, 8, 1]
    hits["edit"] = [7, 4, 6, 9]
    assert M.format_results(hits, ["R1_1_0_0"], True) == "R1_1_0_0:2-10-3=4,2-11-8=6,5-12-1=9\n"  # :457-472
    assert M.format_results(np.zeros(0, dtype=M.HIT_DTYPE), ["a", "b"]) == ""
    # random hit sets: product formatter == oracle formatter
    rng = np.random.default_rng(1)
    for long_fmt in (False, True):
        n = 200
        h = np.zeros(n, dtype=M.HIT_DTYPE)
        h["read"] = np.sort(rng.integers(0, 30, n))
        h["tax_id"] = rng.integers(0, 6, n)
        h["gi"] = rng.integers(0, 3, n)
        h["offset"] = rng.integers(0, 4, n)
        h["edit"] = rng.integers(0, 20, n)
        ids = [f"read{i} x" for i in range(30)]
        oh = np.zeros(n, dtype=O.HIT_DTYPE)
        for f in ("read", "tax_id", "gi", "offset", "edit"):
            oh[f] = h[f]
        want = "".join(O.format_line(ids[r], oh[oh["read"] == r], long_fmt) for r in range(30))
        assert M.format_results(h, ids, long_fmt) == want


def test_format_results_numbers_at_their_limits_and_many_lines():
    """the formatter writes its digits itself (binner.rs:326-379 formats with {}): every width of u32 / u64 including the
    largest values, zero, duplicates folded to the smallest edit, against Python's own formatting of the same rule"""
    rng = np.random.default_rng(3)
    n_reads = 5000
    edges32 = [0, 1, 9, 10, 99, 100, 999999999, 1000000000, 2**31 - 1, 2**31, 2**32 - 1]
    edges64 = [0, 9, 10, 2**32 - 1, 2**32, 10**18, 2**63, 2**64 - 1]
    rows = []
    for r in range(n_reads):
        for _ in range(int(rng.integers(0, 4))):
            tax = int(rng.choice(edges32)) if rng.random() < 0.5 else int(rng.integers(0, 2**32))
            gi = int(rng.choice(edges32)) if rng.random() < 0.5 else int(rng.integers(0, 2**32))
            off = int(rng.choice(np.array(edges64, dtype=np.uint64))) if rng.random() < 0.5 else int(rng.integers(0, 2**63))
            edit = int(rng.choice(edges32)) if rng.random() < 0.3 else int(rng.integers(0, 40))
            rows.append((r, tax, gi, edit, int(rng.integers(0, 2)), off))
            if rng.random() < 0.3:  # the same key again with another edit
                rows.append((r, tax, gi, int(rng.integers(0, 40)), 1, off))
    hits = np.zeros(len(rows), dtype=M.HIT_DTYPE)
    for k, (r, tax, gi, edit, strand, off) in enumerate(rows):
        hits[k] = (r, tax, gi, edit, strand, off)
    ids = [f"read/{i} x" if i % 7 else "" for i in range(n_reads)]
    for long_format in (False, True):
        want = []
        for r in range(n_reads):
            best = {}
            for (rr, tax, gi, edit, _, off) in rows:
                if rr != r:
                    continue
                key = (tax, gi, off) if long_format else (tax,)
                best[key] = min(best.get(key, edit), edit)
            if not best:
                continue
            items = sorted(best.items())
            body = ",".join((f"{k[0]}-{k[1]}-{k[2]}={e}" if long_format else f"{k[0]}={e}") for k, e in items)
            want.append(f"{ids[r]}:{body}\n")
        assert M.format_results(hits, ids, long_format) == "".join(want)


def test_format_results_rejects_unordered_hits():
    h = np.zeros(2, dtype=M.HIT_DTYPE)
    h["read"] = [1, 0]
    with pytest.raises(M.MtsvError):
        M.format_results(h, ["a", "b"])


def test_synthetic_workload_is_deterministic():
    a = M.MGIndex.synth(5, 4, 2, 3000, threads=2)
    b = M.MGIndex.synth(5, 4, 2, 3000, threads=5)
    ra, oa = M.synth_reads(a, 9, 500, 100)
    rb, ob = M.synth_reads(b, 9, 500, 100)
    assert np.array_equal(ra, rb) and np.array_equal(oa, ob)
    assert a.info()["n"] == 4 * 2 * 3000 + 1


def test_pack_bases_is_the_normalisation_of_the_reference_in_four_bits():
    """mtsv_pack_bases (the transfer format of mtsv_batch_run_host*): every byte value, every alignment of a chunk inside
    its segment, chunks chained through the shared byte, lengths on both sides of the vector and thread thresholds --
    against binner.rs:88-100 written out in numpy (A/a C/c G/g T/t -> 0..3, anything else -> 4)"""
    import numpy as np
    import mtsv_tools_amd as M

    lut = np.full(256, 4, dtype=np.uint8)
    for k, ch in enumerate("ACGT"):
        lut[ord(ch)] = lut[ord(ch.lower())] = k
    rng = np.random.default_rng(11)

    def image(codes, first):   # bytes [first / 2, (first + n + 1) / 2) of the segment's packed image
        seg = np.zeros(first + len(codes) + 2, dtype=np.uint8)
        seg[first:first + len(codes)] = codes
        b = seg[0::2][: (len(seg) + 1) // 2].copy()
        hi = seg[1::2]
        b[: len(hi)] |= hi << 4
        return b[first >> 1:(first + len(codes) + 1) >> 1]

    every = np.arange(256, dtype=np.uint8)
    got, last = M.pack_bases(every)
    assert last == lut[255] and np.array_equal(got, image(lut[every], 0))
    for n in (0, 1, 2, 3, 31, 32, 33, 63, 64, 65, 1000, 4097, (1 << 21) + 5, (3 << 20) + 1):
        src = rng.choice(np.frombuffer(b"ACGTNacgtnRY-*\x00\xff", dtype=np.uint8), size=n)
        for first in (0, 1, 6, 7):
            prev = int(rng.integers(0, 5))
            got, last = M.pack_bases(src, first_offset=first, prev_code=prev)
            want = image(lut[src], first)
            if first & 1 and n:
                want = want.copy()
                want[0] |= prev          # the base before the chunk shares its first byte
            assert np.array_equal(got, want), (n, first)
            assert last == (lut[src[-1]] if n else prev)
    # two chunks of one segment, cut at an odd offset: the second rewrites the shared byte with both nibbles
    src = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=2001)
    a, la = M.pack_bases(src[:777])
    b, _ = M.pack_bases(src[777:], first_offset=777, prev_code=la)
    whole, _ = M.pack_bases(src)
    assert np.array_equal(np.concatenate([a[:-1], b]), whole) and (a[-1] & 0xF) == (b[0] & 0xF)


def test_pack_pool_serves_concurrent_callers():
    """the packer's thread pool is shared by every workspace of the process (a worker thread per device, several lanes):
    three callers pack chunks of different sizes at once, every result checked"""
    import threading
    import numpy as np
    import mtsv_tools_amd as M

    lut = np.full(256, 4, dtype=np.uint8)
    for k, ch in enumerate("ACGT"):
        lut[ord(ch)] = lut[ord(ch.lower())] = k
    errors = []

    def caller(seed):
        r = np.random.default_rng(seed)
        try:
            for _ in range(25):
                n = int(r.integers(1 << 19, 5 << 19))
                src = r.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), size=n)
                got, _ = M.pack_bases(src)
                c = lut[src]
                if n & 1:
                    c = np.append(c, 0)
                assert np.array_equal(got, c[0::2] | (c[1::2] << 4))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=caller, args=(s,)) for s in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors and not any(t.is_alive() for t in th)

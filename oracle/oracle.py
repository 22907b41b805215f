"""ctypes binding of the CPU oracle (oracle/libmtsv_oracle.so) and of the reference's own
striped Smith-Waterman compiled from its C source (oracle/_ref/libssw_ref.so).

TEST INFRASTRUCTURE ONLY.  Import this module from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from mtsv_tools_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmtsv_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libssw_ref.so")


def build(force=False):
    """Compile the oracle (and oracle/_ref when the upstream checkout is present)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "mtsv_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libmtsv_oracle.so"])
    if force or not os.path.exists(_REF):
        subprocess.check_call(["make", "-C", _HERE, "ref"])


class Params(C.Structure):
    _fields_ = [("edit_rate", C.c_double), ("seed_size", C.c_uint64), ("seed_gap", C.c_uint64),
                ("min_seed", C.c_double), ("max_hits", C.c_uint64), ("tune_max_hits", C.c_uint64),
                ("max_candidates", C.c_int64), ("max_assignments", C.c_int64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("X", "S", "H", "W", "R", "Lsum", "n_sw", "n_edit", "n_cand", "n_seed")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Bin(C.Structure):
    _fields_ = [("gi", C.c_uint32), ("tax_id", C.c_uint32), ("start", C.c_uint64),
                ("end", C.c_uint64)]


HIT_DTYPE = np.dtype([("read", "<u8"), ("tax_id", "<u4"), ("gi", "<u4"), ("edit", "<u4"),
                      ("strand", "<u4"), ("offset", "<u8")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_last_error.restype = C.c_char_p
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_index_build.restype = C.c_void_p
        L.orc_index_build.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_uint64]
        L.orc_index_write.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_index_read.restype = C.c_void_p
        L.orc_index_read.argtypes = [C.c_char_p]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_occ_get.restype = C.c_uint64
        L.orc_occ_get.argtypes = [C.c_void_p, C.c_uint64, C.c_uint8]
        L.orc_backward_search.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
        L.orc_sa_get.restype = C.c_uint64
        L.orc_sa_get.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_min_edit_distance.restype = C.c_uint32
        L.orc_min_edit_distance.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        for fn in (L.orc_ssw_score, L.orc_ssw_byte, L.orc_ssw_word):
            fn.restype = C.c_uint32
            fn.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.orc_sw_exact.restype = C.c_uint32
        L.orc_sw_exact.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int, C.c_int]
        L.orc_candidate_indices.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(Bin), C.c_uint64,
                                            C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_matching_tax_ids.restype = C.c_int64
        L.orc_matching_tax_ids.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(Params),
                                           C.c_void_p, C.c_uint64, C.POINTER(Counters)]
        L.orc_bin_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                    C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_uint64), C.POINTER(Counters)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_format_line.restype = C.c_int64
        L.orc_format_line.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_char_p,
                                      C.c_uint64]
        L.orc_brute_find.restype = C.c_uint64
        L.orc_brute_find.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def default_params(**over):
    p = Params()
    lib().orc_default_params(C.byref(p))
    names = {f[0] for f in Params._fields_}
    for k, v in over.items():
        if k == "seed_interval":  # the product's name for the same CLI flag (--seed-interval)
            k = "seed_gap"
        if k not in names:
            raise AttributeError(f"no such parameter: {k}")
        if v is None:
            v = -1
        setattr(p, k, v)
    return p


class Index:
    """Owning handle of an orc_index."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        self.h = C.c_void_p(handle)

    @classmethod
    def build(cls, entries, occ_k=64, sa_s=32):
        """entries: iterable of (tax_id, gi, bytes) in database insertion order."""
        entries = list(entries)
        n = len(entries)
        tax = np.array([e[0] for e in entries], dtype=np.uint32)
        gi = np.array([e[1] for e in entries], dtype=np.uint32)
        bufs = [C.create_string_buffer(bytes(e[2]), len(e[2])) for e in entries]
        ptrs = (C.c_void_p * max(n, 1))(*[C.addressof(b) for b in bufs])
        lens = np.array([len(e[2]) for e in entries], dtype=np.uint64)
        return cls(lib().orc_index_build(n, tax.ctypes.data, gi.ctypes.data, ptrs,
                                         lens.ctypes.data, occ_k, sa_s))

    @classmethod
    def read(cls, path):
        return cls(lib().orc_index_read(os.fsencode(path)))

    def write(self, path):
        if lib().orc_index_write(self.h, os.fsencode(path)) != 0:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_index_free(self.h)
            self.h = None

    # -- primitives -------------------------------------------------------------------------
    def backward_search(self, pat):
        lo, hi = C.c_uint64(), C.c_uint64()
        ok = lib().orc_backward_search(self.h, pat, len(pat), C.byref(lo), C.byref(hi), None)
        return bool(ok), lo.value, hi.value

    def sa_get(self, row):
        return lib().orc_sa_get(self.h, row, None)

    def brute_find(self, pat, cap=1 << 16):
        out = np.zeros(cap, dtype=np.uint64)
        n = lib().orc_brute_find(self.h, pat, len(pat), out.ctypes.data, cap)
        return out[:min(n, cap)].copy()

    def matching_tax_ids(self, seq, params=None, counters=None):
        params = params or default_params()
        cap = 1 << 16
        buf = np.zeros(cap, dtype=HIT_DTYPE)
        n = lib().orc_matching_tax_ids(self.h, seq, len(seq), C.byref(params), buf.ctypes.data, cap,
                                       C.byref(counters) if counters is not None else None)
        if n < 0:
            raise RuntimeError("oracle hit buffer overflow")
        return buf[:n].copy()

    def bin_batch(self, bases, read_off, params=None, threads=1):
        """bases: uint8 array of concatenated reads, read_off: uint64[n+1].
        Returns (hits structured array ordered by (read, strand, rank), counters dict)."""
        params = params or default_params()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        n = len(read_off) - 1
        out = C.c_void_p()
        nh = C.c_uint64()
        ctr = Counters()
        rc = lib().orc_bin_batch(self.h, bases.ctypes.data, read_off.ctypes.data, n,
                                 C.byref(params), threads, C.byref(out), C.byref(nh), C.byref(ctr))
        try:
            if rc != 0:
                raise RuntimeError("oracle: " + lib().orc_last_error().decode())
            arr = np.frombuffer(C.string_at(out.value, nh.value * HIT_DTYPE.itemsize),
                                dtype=HIT_DTYPE).copy() if nh.value else np.zeros(0, HIT_DTYPE)
        finally:
            lib().orc_free(out)
        return arr, ctr.as_dict()


def min_edit_distance(p, t):
    return lib().orc_min_edit_distance(p, len(p), t, len(t))


def ssw_score(read, ref):
    return lib().orc_ssw_score(read, len(read), ref, len(ref))


def ssw_byte(read, ref):
    return lib().orc_ssw_byte(read, len(read), ref, len(ref))


def ssw_word(read, ref):
    return lib().orc_ssw_word(read, len(read), ref, len(ref))


def sw_exact(read, ref, go=1, ge=1):
    return lib().orc_sw_exact(read, len(read), ref, len(ref), go, ge)


def candidate_indices(site, qoff, bin_start, bin_end, read_len, ed):
    b = Bin(0, 1, bin_start, bin_end)
    s, e = C.c_uint64(), C.c_uint64()
    ok = lib().orc_candidate_indices(site, qoff, C.byref(b), read_len, ed, C.byref(s), C.byref(e))
    return (s.value, e.value) if ok else None


def format_line(read_id, hits, long_format=False):
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    cap = len(read_id) + 64 * (len(hits) + 1) + 16
    buf = C.create_string_buffer(cap)
    n = lib().orc_format_line(read_id.encode(), hits.ctypes.data, len(hits), int(long_format), buf,
                              cap)
    if n < 0:
        raise RuntimeError("format buffer too small")
    return buf.raw[:n].decode()


# ---- the reference's own SSW, compiled from ssw/src/ssw.c into oracle/_ref ------------------
_ref = None
_MAT = (C.c_int8 * 25)(*[1 if i // 5 == i % 5 else -1 for i in range(25)])  # ssw/src/lib.rs:11-16
_NUM = np.full(256, 4, dtype=np.int8)
for _i, _ch in enumerate(b"ACGT"):
    _NUM[_ch] = _i


class _RawAlign(C.Structure):  # ssw/src/lib.rs:118-130
    _fields_ = [("score1", C.c_uint16), ("score2", C.c_uint16), ("ref_begin1", C.c_int32),
                ("ref_end1", C.c_int32), ("read_begin1", C.c_int32), ("read_end1", C.c_int32),
                ("ref_end2", C.c_int32), ("cigar", C.c_void_p), ("cigar_len", C.c_int32)]


def ref_available():
    return os.path.exists(_REF)


def ref_lib():
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        R.ssw_init.restype = C.c_void_p
        R.ssw_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int8]
        R.ssw_align.restype = C.POINTER(_RawAlign)
        R.ssw_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8,
                                C.c_uint16, C.c_int32, C.c_int32]
        R.init_destroy.argtypes = [C.c_void_p]
        R.align_destroy.argtypes = [C.POINTER(_RawAlign)]
        _ref = R
    return _ref


def ref_ssw_scores(read, refs):
    """score1 of the reference's ssw_align for one read against many windows, called exactly as
    ssw/src/lib.rs:36-84 does (score_size 2, gap 1/1, flag 0, maskLen len/2)."""
    R = ref_lib()
    rnum = _NUM[np.frombuffer(read, dtype=np.uint8)].copy()
    prof = R.ssw_init(rnum.ctypes.data, len(read), _MAT, 5, 2)
    out = []
    for w in refs:
        wnum = _NUM[np.frombuffer(w, dtype=np.uint8)].copy()
        a = R.ssw_align(prof, wnum.ctypes.data, len(w), 1, 1, 0, 0, 0, len(read) // 2)
        out.append(int(a.contents.score1))
        R.align_destroy(a)
    R.init_destroy(prof)
    return out

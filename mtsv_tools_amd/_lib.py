"""ctypes binding of libmtsv_amd.so (include/mtsv_amd.h).  No algorithm lives here."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # MTSV_AMD_LIB: a variant build of the same library (tools/build_variant.sh, kernel experiments)
    return os.environ.get("MTSV_AMD_LIB") or os.path.join(_HERE, "libmtsv_amd.so")


class MtsvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mtsv error {code}: {msg}")
        self.code = code


E_ARG, E_IO, E_FORMAT, E_DEVICE, E_LIMIT, E_NOMEM = -1, -2, -3, -4, -5, -6


class Params(C.Structure):  # mtsv_params
    _fields_ = [("edit_rate", C.c_double), ("seed_size", C.c_uint32), ("seed_interval", C.c_uint32),
                ("min_seed", C.c_double), ("max_hits", C.c_uint64), ("tune_max_hits", C.c_uint64),
                ("max_assignments", C.c_int64), ("max_candidates", C.c_int64)]


class IndexInfo(C.Structure):  # mtsv_index_info_t
    _fields_ = [("n", C.c_uint64), ("n_bins", C.c_uint64), ("occ_k", C.c_uint32),
                ("sa_s", C.c_uint64), ("file_bytes", C.c_uint64), ("device_bytes", C.c_uint64),
                ("kmer_k", C.c_uint32), ("sa_full", C.c_uint32)]


N_STAGES = 8
STAGE_NAMES = ("search", "thin_scan", "expand", "locate", "coalesce", "verify", "gather", "total")


class BatchStats(C.Structure):  # mtsv_batch_stats
    _fields_ = [("stage_ms", C.c_float * N_STAGES), ("n_reads", C.c_uint64),
                ("n_seed_slots", C.c_uint64), ("n_seed_hits", C.c_uint64), ("lf_steps", C.c_uint64),
                ("n_candidates", C.c_uint64), ("n_verified", C.c_uint64),
                ("window_bytes", C.c_uint64), ("n_hits", C.c_uint64), ("n_passes", C.c_uint64),
                ("n_rounds", C.c_uint64), ("n_lanes", C.c_uint64), ("sw_cell_pairs", C.c_uint64), ("sw_prefilter_ms", C.c_float),
                ("sw_sweep_ms", C.c_float), ("n_sw_passed", C.c_uint64), ("sw_diag_ms", C.c_float), ("sw_bound_ms", C.c_float),
                ("edit_ms", C.c_float), ("myers_columns", C.c_uint64), ("n_sw_bound_refuted", C.c_uint64)]

    def as_dict(self):
        d = {n: (float(getattr(self, n)) if t is C.c_float else int(getattr(self, n))) for n, t in self._fields_[1:]}
        d["stage_ms"] = {STAGE_NAMES[i]: float(self.stage_ms[i]) for i in range(N_STAGES)}
        return d


# mtsv_hit: 8 + 4 + 4 + 4 + 1 + 3 pad + 8 = 32 bytes
HIT_DTYPE = np.dtype({"names": ["read", "tax_id", "gi", "edit", "strand", "offset"],
                      "formats": ["<u8", "<u4", "<u4", "<u4", "u1", "<u8"],
                      "offsets": [0, 8, 12, 16, 20, 24], "itemsize": 32})

EXPORTS = [
    "mtsv_last_error", "mtsv_version", "mtsv_params_default", "mtsv_device_count",
    "mtsv_index_load", "mtsv_index_build", "mtsv_index_build_fasta", "mtsv_index_write",
    "mtsv_index_info", "mtsv_index_free", "mtsv_set_build_device", "mtsv_index_to_device", "mtsv_bin_batch",
    "mtsv_hits_free", "mtsv_batch_create", "mtsv_batch_upload", "mtsv_batch_run", "mtsv_batch_run_host", "mtsv_batch_run_host_parts",
    "mtsv_batch_stats_get", "mtsv_batch_set_verify_mode", "mtsv_batch_download", "mtsv_batch_free", "mtsv_format_results",
    "mtsv_free", "mtsv_synth_index", "mtsv_synth_reads", "mtsv_bin_batch_workspace_reads",
    "mtsv_bin_batch_multi", "mtsv_bin_batch_chunks", "mtsv_set_default_verify_mode",
    "mtsv_host_alloc", "mtsv_host_free", "mtsv_host_register", "mtsv_host_unregister",
    "mtsv_batch_create_lanes", "mtsv_batch_reserve_host", "mtsv_pack_bases", "mtsv_host_pack_threads",
]

_lib = None


def lib():
    """Load libmtsv_amd.so; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise MtsvError(E_DEVICE, f"{p} is missing: run `make -C mtsv_tools_amd/csrc` "
                                      "(or __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(p)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        L.mtsv_last_error.restype = C.c_char_p
        L.mtsv_version.restype = C.c_char_p
        L.mtsv_params_default.argtypes = [C.POINTER(Params)]
        L.mtsv_index_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.mtsv_index_build.argtypes = [u64, vp, vp, vp, vp, u32, u64, i32, C.POINTER(vp)]
        L.mtsv_index_build_fasta.argtypes = [C.c_char_p, u32, u64, i32, C.POINTER(vp)]
        L.mtsv_index_write.argtypes = [vp, C.c_char_p]
        L.mtsv_index_info.argtypes = [vp, C.POINTER(IndexInfo)]
        L.mtsv_index_free.argtypes = [vp]
        L.mtsv_index_free.restype = None
        L.mtsv_index_to_device.argtypes = [vp, i32, u32]
        L.mtsv_bin_batch.argtypes = [vp, i32, vp, vp, u64, C.POINTER(Params), C.POINTER(vp),
                                     C.POINTER(u64)]
        L.mtsv_hits_free.argtypes = [vp]
        L.mtsv_hits_free.restype = None
        L.mtsv_bin_batch_multi.argtypes = [vp, vp, i32, vp, vp, u64, C.POINTER(Params), C.POINTER(vp), C.POINTER(u64)]
        L.mtsv_bin_batch_chunks.argtypes = [vp, vp, i32, vp, vp, u64, C.POINTER(Params), C.POINTER(vp), C.POINTER(u64)]
        L.mtsv_bin_batch_workspace_reads.argtypes = [u64]
        L.mtsv_bin_batch_workspace_reads.restype = u64
        L.mtsv_batch_create.argtypes = [vp, i32, u64, u64, u64, C.POINTER(vp)]
        L.mtsv_batch_create_lanes.argtypes = [vp, i32, u64, u64, u64, i32, C.POINTER(vp)]
        L.mtsv_batch_reserve_host.argtypes = [vp, u64, u64, u32]
        L.mtsv_pack_bases.argtypes = [vp, vp, u64, u64, C.c_uint8]
        L.mtsv_pack_bases.restype = C.c_uint8
        L.mtsv_batch_upload.argtypes = [vp, vp, vp, u64]
        L.mtsv_batch_run.argtypes = [vp, C.POINTER(Params)]
        L.mtsv_batch_run_host.argtypes = [vp, vp, vp, u64, C.POINTER(Params)]
        L.mtsv_batch_run_host_parts.argtypes = [vp, i32, vp, vp, vp, C.POINTER(Params)]
        L.mtsv_batch_set_verify_mode.argtypes = [vp, i32]
        L.mtsv_batch_stats_get.argtypes = [vp, C.POINTER(BatchStats)]
        L.mtsv_batch_download.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        L.mtsv_batch_free.argtypes = [vp]
        L.mtsv_batch_free.restype = None
        L.mtsv_format_results.argtypes = [vp, u64, C.c_char_p, vp, u64, i32, C.POINTER(vp),
                                          C.POINTER(u64)]
        L.mtsv_free.argtypes = [vp]
        L.mtsv_free.restype = None
        L.mtsv_synth_index.argtypes = [u64, u32, u32, u64, u32, u64, i32, C.POINTER(vp)]
        L.mtsv_synth_reads.argtypes = [vp, u64, u64, u32, vp, vp]
        L.mtsv_host_alloc.argtypes = [C.c_size_t]
        L.mtsv_host_alloc.restype = vp
        L.mtsv_host_free.argtypes = [vp]
        L.mtsv_host_free.restype = None
        L.mtsv_host_register.argtypes = [vp, C.c_size_t]
        L.mtsv_host_unregister.argtypes = [vp]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MtsvError(rc, lib().mtsv_last_error().decode(errors="replace"))


def version():
    return lib().mtsv_version().decode()


def device_count():
    return lib().mtsv_device_count()


def bin_batch_multi(index, devices, bases, read_off, params=None):
    """mtsv_bin_batch_multi: one index replicated on `devices`, reads in contiguous blocks (Mode A)"""
    params = params or default_params()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    dev = (C.c_int * len(devices))(*devices)
    out, n = C.c_void_p(), C.c_uint64()
    _check(lib().mtsv_bin_batch_multi(index.h, dev, len(devices), bases.ctypes.data, read_off.ctypes.data,
                                      len(read_off) - 1, C.byref(params), C.byref(out), C.byref(n)))
    return _hits_from(out, n.value)


def bin_batch_chunks(indexes, devices, bases, read_off, params=None):
    """mtsv_bin_batch_chunks: chunk k of the database on devices[k], hit lists merged per read (Mode B)"""
    params = params or default_params()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    dev = (C.c_int * len(devices))(*devices)
    hs = (C.c_void_p * len(indexes))(*[ix.h for ix in indexes])
    out, n = C.c_void_p(), C.c_uint64()
    _check(lib().mtsv_bin_batch_chunks(hs, dev, len(indexes), bases.ctypes.data, read_off.ctypes.data,
                                       len(read_off) - 1, C.byref(params), C.byref(out), C.byref(n)))
    return _hits_from(out, n.value)


class HostBuffer:
    """mtsv_host_alloc: page-locked host memory as a uint8 numpy array (`.array`); free with close()"""

    def __init__(self, nbytes):
        self.ptr = lib().mtsv_host_alloc(nbytes)
        if not self.ptr:
            raise MtsvError(E_DEVICE, lib().mtsv_last_error().decode(errors="replace"))
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(1, nbytes)).from_address(self.ptr))[:nbytes]

    def close(self):
        if self.ptr:
            self.array = None
            lib().mtsv_host_free(self.ptr)
            self.ptr = None


def pack_bases(bases, first_offset=0, prev_code=0):
    """mtsv_pack_bases: the transfer format of run_host (4-bit codes, two per byte) of a uint8 array of bases that starts
    at segment offset first_offset; returns (packed bytes, code of the last base)"""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    n = len(bases)
    out = np.zeros(((first_offset + n + 1) >> 1) - (first_offset >> 1) + 1, dtype=np.uint8)
    last = lib().mtsv_pack_bases(out.ctypes.data, bases.ctypes.data, first_offset, n, prev_code)
    return out[:-1], last


def host_register(arr):
    """page-lock the memory of a contiguous numpy array in place (mtsv_host_register); undo with host_unregister"""
    _check(lib().mtsv_host_register(arr.ctypes.data, arr.nbytes))


def host_unregister(arr):
    _check(lib().mtsv_host_unregister(arr.ctypes.data))


def bin_batch_slice_reads(n_reads):
    """reads per device workspace mtsv_bin_batch would create for a host batch of n_reads reads"""
    return int(lib().mtsv_bin_batch_workspace_reads(n_reads))


def set_build_device(device):
    """-1 = host suffix sort, >= 0 = GPU prefix doubling on that device (same index bytes)"""
    lib().mtsv_set_build_device.argtypes = [C.c_int]
    _check(lib().mtsv_set_build_device(device))


def default_params(**over):
    p = Params()
    lib().mtsv_params_default(C.byref(p))
    names = {f[0] for f in Params._fields_}
    for k, v in over.items():
        if k not in names:
            raise AttributeError(f"no such parameter: {k}")
        setattr(p, k, -1 if v is None else v)
    return p


def _hits_from(ptr, n):
    try:
        if n == 0:
            return np.zeros(0, dtype=HIT_DTYPE)
        raw = (C.c_ubyte * (n * HIT_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(raw, dtype=HIT_DTYPE).copy()   # one copy, then the C array is freed
    finally:
        lib().mtsv_hits_free(ptr)


class MGIndex:
    """Owning handle of an mtsv_index (host MG-index + per-device HBM layout)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        _check(lib().mtsv_index_load(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def build(cls, entries, occ_k=64, sa_s=32, threads=4):
        entries = list(entries)
        n = len(entries)
        tax = np.array([e[0] for e in entries], dtype=np.uint32)
        gi = np.array([e[1] for e in entries], dtype=np.uint32)
        bufs = [C.create_string_buffer(bytes(e[2]), max(len(e[2]), 1)) for e in entries]
        ptrs = (C.c_void_p * max(n, 1))(*[C.addressof(b) for b in bufs])
        lens = np.array([len(e[2]) for e in entries], dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().mtsv_index_build(n, tax.ctypes.data, gi.ctypes.data, ptrs, lens.ctypes.data,
                                      occ_k, sa_s, threads, C.byref(h)))
        return cls(h)

    @classmethod
    def build_fasta(cls, path, occ_k=64, sa_s=32, threads=4):
        h = C.c_void_p()
        _check(lib().mtsv_index_build_fasta(os.fsencode(path), occ_k, sa_s, threads, C.byref(h)))
        return cls(h)

    @classmethod
    def synth(cls, seed, n_taxa, gis_per_taxon, seq_len, occ_k=64, sa_s=32, threads=8):
        h = C.c_void_p()
        _check(lib().mtsv_synth_index(seed, n_taxa, gis_per_taxon, seq_len, occ_k, sa_s, threads,
                                      C.byref(h)))
        return cls(h)

    def write(self, path):
        _check(lib().mtsv_index_write(self.h, os.fsencode(path)))

    def info(self):
        i = IndexInfo()
        _check(lib().mtsv_index_info(self.h, C.byref(i)))
        return {n: int(getattr(i, n)) for n, _ in i._fields_}

    def to_device(self, device=0, flags=0):
        _check(lib().mtsv_index_to_device(self.h, device, flags))

    def bin_batch(self, bases, read_off, params=None, device=0):
        """mtsv_bin_batch: hits ordered by (read, strand, rank)."""
        params = params or default_params()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        out, n = C.c_void_p(), C.c_uint64()
        _check(lib().mtsv_bin_batch(self.h, device, bases.ctypes.data, read_off.ctypes.data,
                                    len(read_off) - 1, C.byref(params), C.byref(out), C.byref(n)))
        return _hits_from(out, n.value)

    def close(self):
        if self.h is not None and _lib is not None:
            _lib.mtsv_index_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """Owning handle of an mtsv_batch (HBM-resident read batch + workspace)."""

    def __init__(self, index, device, max_reads, max_bases, max_hits_ws=0, lanes=0):
        self.index = index
        self.h = C.c_void_p()
        _check(lib().mtsv_batch_create_lanes(index.h, device, max_reads, max_bases, max_hits_ws, lanes,
                                             C.byref(self.h)))

    def reserve_host(self, n_reads, n_bases, warm_read_len=0):
        """mtsv_batch_reserve_host: size what run_host would size on its first calls; warm_read_len > 0 also runs a
        small batch sampled from the index through every kernel"""
        _check(lib().mtsv_batch_reserve_host(self.h, n_reads, n_bases, warm_read_len))

    def upload(self, bases, read_off):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        _check(lib().mtsv_batch_upload(self.h, bases.ctypes.data, read_off.ctypes.data,
                                       len(read_off) - 1))

    def set_verify_mode(self, mode):
        _check(lib().mtsv_batch_set_verify_mode(self.h, mode))

    def run(self, params=None):
        params = params or default_params()
        _check(lib().mtsv_batch_run(self.h, C.byref(params)))

    def run_host(self, bases, read_off, params=None):
        """mtsv_batch_run_host: host buffers of any size, sliced and double-buffered on the device."""
        params = params or default_params()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        _check(lib().mtsv_batch_run_host(self.h, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1,
                                         C.byref(params)))

    def run_host_parts(self, parts, params=None):
        """parts: [(bases u8 array, read_off u64 array), ...]; the reads are numbered through the parts in order"""
        params = params or default_params()
        keep = [(np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(o, dtype=np.uint64)) for b, o in parts]
        k = len(keep)
        bp = (C.c_void_p * k)(*[b.ctypes.data for b, _ in keep])
        op = (C.c_void_p * k)(*[o.ctypes.data for _, o in keep])
        nr = (C.c_uint64 * k)(*[len(o) - 1 for _, o in keep])
        _check(lib().mtsv_batch_run_host_parts(self.h, k, bp, op, nr, C.byref(params)))

    def stats(self):
        s = BatchStats()
        _check(lib().mtsv_batch_stats_get(self.h, C.byref(s)))
        return s.as_dict()

    def download(self):
        out, n = C.c_void_p(), C.c_uint64()
        _check(lib().mtsv_batch_download(self.h, C.byref(out), C.byref(n)))
        return _hits_from(out, n.value)

    def close(self):
        if self.h is not None and _lib is not None:
            _lib.mtsv_batch_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_results(hits, read_ids, long_format=False):
    """write_assignments over a batch: returns the result lines as one str."""
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    blob = b"".join(i.encode() + b"\0" for i in read_ids)
    off = np.zeros(len(read_ids) + 1, dtype=np.uint64)
    np.cumsum([len(i.encode()) + 1 for i in read_ids], out=off[1:])
    out, n = C.c_void_p(), C.c_uint64()
    _check(lib().mtsv_format_results(hits.ctypes.data, len(hits), blob, off.ctypes.data,
                                     len(read_ids), int(long_format), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out.value, n.value).decode()
    finally:
        lib().mtsv_free(out)


def synth_reads(index, seed, n_reads, read_len):
    bases = np.empty(n_reads * read_len, dtype=np.uint8)
    off = np.empty(n_reads + 1, dtype=np.uint64)
    _check(lib().mtsv_synth_reads(index.h, seed, n_reads, read_len, bases.ctypes.data,
                                  off.ctypes.data))
    return bases, off

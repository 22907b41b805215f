"""mtsv_tools_amd -- MI355X-native mtsv-binner hot path.

The product is the C-ABI shared library ``libmtsv_amd.so`` (C++ host + hand-written HIP kernels
for gfx950, built in-tree from ``mtsv_tools_amd/csrc``; interface in ``include/mtsv_amd.h``).
This Python package is only a thin ctypes harness over that ABI for tests and ``bench.py``; it
holds no algorithm and has no CPU fallback: every call that needs the GPU raises ``MtsvError``
when the library or a device is missing.
"""
from ._lib import (  # noqa: F401
    HIT_DTYPE,
    HostBuffer,
    host_register,
    host_unregister,
    Batch,
    MGIndex,
    MtsvError,
    Params,
    bin_batch_chunks,
    bin_batch_multi,
    bin_batch_slice_reads,
    default_params,
    device_count,
    format_results,
    lib,
    lib_path,
    pack_bases,
    set_build_device,
    synth_reads,
    version,
)

DEV_DEFAULT = 0
DEV_SAMPLED_SA_ONLY = 1
DEV_NO_KMER_TABLE = 2

"""-m gpu parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bit-exact: every field of every hit, in order."""
import numpy as np
import pytest

import mtsv_tools_amd as M
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FIELDS = ("read", "tax_id", "gi", "edit", "strand", "offset")


def assert_same_hits(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for f in FIELDS:
        bad = np.nonzero(got[f] != want[f])[0]
        assert len(bad) == 0, (f, bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.fixture(scope="module")
def small_db(tmp_path_factory):
    ix = M.MGIndex.synth(seed=21, n_taxa=16, gis_per_taxon=4, seq_len=5000)
    p = str(tmp_path_factory.mktemp("idx") / "small.idx")
    ix.write(p)
    return ix, O.Index.read(p)


@pytest.mark.parametrize("flags", [M.DEV_SAMPLED_SA_ONLY | M.DEV_NO_KMER_TABLE, M.DEV_SAMPLED_SA_ONLY,
                                   M.DEV_NO_KMER_TABLE, M.DEV_DEFAULT])
@pytest.mark.parametrize("read_len", [100, 150])
def test_default_params_match_oracle(small_db, flags, read_len):
    ix, orc = small_db
    bases, off = M.synth_reads(ix, seed=5 + read_len, n_reads=3000, read_len=read_len)
    ix.to_device(0, flags)
    got = ix.bin_batch(bases, off, M.default_params(), device=0)
    want, _ = orc.bin_batch(bases, off, O.default_params(), threads=8)
    assert len(want) > 1000
    assert_same_hits(got, want)

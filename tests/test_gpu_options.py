"""Table-width options and the reference's `precompute` argument: every supported window width
must give the same bytes (the table layout, digit recoding and window count all depend on it)."""
import pytest

import golden_util as G
from kzg_ctypes import HIP_SO, Kzg
from test_gpu_commitment import rand_blob

pytestmark = pytest.mark.gpu


def _restore(api):
    for k, v in (("commit_wbits", 10), ("fk20_wbits", 0), ("proof_wbits", 8), ("direct_max", -1)):
        api.lib.ckzg_hip_set_option(k.encode(), v)


@pytest.mark.parametrize("wbits", [4, 5, 7, 8, 9, 12, 16])
def test_commitment_for_every_table_width(hip, wbits):
    api = Kzg(HIP_SO, "", precompute=0, options={"commit_wbits": wbits, "proof_wbits": 0})
    _restore(api)
    try:
        for name in G.case_names("blob_to_kzg_commitment"):
            got, exp = G.run_case(api, "blob_to_kzg_commitment", name)
            assert got == exp, (wbits, name)
        b = rand_blob(51, wbits)
        assert api.blob_to_kzg_commitment(b) == hip.blob_to_kzg_commitment(b)
        # every digit exactly +-2^(wbits-1) in some window: the edge of the signed recoding
        v = sum(1 << (wbits * w + wbits - 1) for w in range(0, 250 // wbits, 2))
        edge = (v.to_bytes(32, "big") + (v >> 1).to_bytes(32, "big")) * 2048
        assert api.blob_to_kzg_commitment(edge) == hip.blob_to_kzg_commitment(edge)
        if wbits == 16:  # 103 GB (8 windows of GLV half-scalars): really built at 16 bits, not silently narrowed
            api.lib.ckzg_hip_table_bytes.restype = __import__("ctypes").c_uint64
            assert api.lib.ckzg_hip_table_bytes(api.sp) >= 4096 * 8 * 32768 * 96
    finally:
        api.close()


@pytest.mark.parametrize("precompute,direct", [(9, 0), (4, 0), (0, 24)])
def test_cells_and_proofs_for_precompute_values(hip, precompute, direct):
    # precompute > 8 widens the FK20 table (reference: src/setup/setup.c:411-422 -> wbits);
    # proof_wbits 5 exercises a narrow table on the low-latency path
    api = Kzg(HIP_SO, "", precompute=precompute,
              options={"commit_wbits": 8, "direct_max": direct, "proof_wbits": 5 if direct else 0})
    _restore(api)
    try:
        assert api.s.wbits == precompute
        name = "compute_cells_and_kzg_proofs_case_valid_2"
        names = [n for n in G.case_names("compute_cells_and_kzg_proofs") if "valid" in n]
        got, exp = G.run_case(api, "compute_cells_and_kzg_proofs", names[2] if len(names) > 2 else names[0])
        assert got == exp
        b = rand_blob(52, precompute)
        assert api.compute_cells_and_kzg_proofs(b) == hip.compute_cells_and_kzg_proofs(b)
    finally:
        api.close()


def test_low_latency_proofs_with_the_widest_monomial_table(hip):
    # proof_wbits = 16: 103 GB table over the monomial points, 2 x 8 windows, int16 digits at their limit
    api = Kzg(HIP_SO, "", precompute=0, options={"commit_wbits": 8, "direct_max": 24, "proof_wbits": 16})
    _restore(api)
    try:
        import ctypes
        api.lib.ckzg_hip_table_wbits.restype = ctypes.c_int
        assert api.lib.ckzg_hip_table_wbits(api.sp, 2) == 16
        names = [n for n in G.case_names("compute_cells_and_kzg_proofs") if "valid" in n]
        for name in names[:3]:
            got, exp = G.run_case(api, "compute_cells_and_kzg_proofs", name)
            assert got == exp
        b = rand_blob(53, 16)
        assert api.compute_cells_and_kzg_proofs(b) == hip.compute_cells_and_kzg_proofs(b)
    finally:
        api.close()


def test_precompute_out_of_range_is_badargs():
    from kzg_ctypes import KzgError
    with pytest.raises(KzgError):
        Kzg(HIP_SO, "", precompute=16)

"""The reference's `precompute` = 10..15 (src/setup/setup.c:411-422 -> wbits; the reference's README benchmarks
exactly these, README.md:126-143) and explicit FK20 table widths up to 16 bits (206 GB): golden vectors, a random
blob against the oracle and a blob whose FK20 scalars sit on the edges of the GLV split and of the signed-window
recoding (fk20_edge.py), through the one-wave and the 16-lane forms of k_msm_small, plus a recover vector.
Reference: src/eip7594/fk20.c:222-247,231-239 (fixed-base MSM per column when precompute > 0),
src/setup/setup.c:291-323."""
import pytest

import golden_util as G
from kzg_ctypes import HIP_SO, Kzg
from test_gpu_commitment import rand_blob
from test_gpu_round2 import _cells_batch, _restore
from test_gpu_wide_tables import VALID_CP, VALID_REC, _wbits, edge_blob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precompute,fk20_wbits", [(10, 0), (12, 0), (13, 0), (15, 0), (0, 14), (0, 15), (0, 16)])
def test_fk20_table_widths(oracle, precompute, fk20_wbits):
    api = Kzg(HIP_SO, "", precompute=precompute,
              options={"commit_wbits": 8, "proof_wbits": 0, "direct_max": 0, "fk20_wbits": fk20_wbits})
    _restore(api)
    try:
        assert api.s.wbits == precompute
        built = _wbits(api)[1]
        # load_trusted_setup maps precompute > 8 to the FK20 width (capped at 13 without an explicit option:
        # device_ctx.hip: build_owner); an explicit fk20_wbits is taken as is
        assert built == (fk20_wbits if fk20_wbits else min(precompute, 13))
        for name in VALID_CP:
            got, exp = G.run_case(api, "compute_cells_and_kzg_proofs", name)
            assert got == exp, (precompute, fk20_wbits, name)
        blobs = [rand_blob(134, precompute + fk20_wbits), edge_blob(built, precompute)]
        exp = [oracle.compute_cells_and_kzg_proofs(b) for b in blobs]
        for b, e in zip(blobs, exp):
            got = api.compute_cells_and_kzg_proofs(b)   # one blob: the one-wave-per-vector form (LPV = 64)
            assert got[0] == e[0] and got[1] == e[1]
        # 33 blobs = 4224 vectors: the 16-lane form
        n = 33
        rc, _, proofs, st = _cells_batch(api, b"".join(blobs[i % 2] for i in range(n)), n, want_cells=False)
        assert rc == 0 and not any(st.raw)
        for i in range(n):
            assert proofs.raw[i * 6144:(i + 1) * 6144] == b"".join(exp[i % 2][1]), i
        # recover through the same table
        inp, rexp = G.get_case("recover_cells_and_kzg_proofs", VALID_REC[0])
        got = api.recover_cells_and_kzg_proofs(inp["cell_indices"], inp["cells"])
        assert got[0] == rexp[0] and got[1] == rexp[1]
    finally:
        api.close()

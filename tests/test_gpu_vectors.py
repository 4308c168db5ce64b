"""GPU parity proper: every consensus-spec vector through the product's C-ABI (libckzg_hip.so).
Same corpus and semantics as tests/test_oracle_vectors.py / bindings/python/tests.py of the
reference: null output <=> the call fails, otherwise byte-exact equality."""
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

CASES = [(fn, name) for fn in G.functions() for name in G.case_names(fn)]


@pytest.mark.parametrize("fn,name", CASES, ids=[c[1] for c in CASES])
def test_hip_matches_spec_vector(hip, fn, name):
    got, exp = G.run_case(hip, fn, name)
    assert got == exp

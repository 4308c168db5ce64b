"""Pins the CPU oracle (oracle/) against every consensus-spec vector the reference's tests hold
(/root/reference/tests, packed into tests/golden by tools/make_golden.py): 344 cases over the 10
public functions and the 2 Fiat-Shamir transcript functions.  Runs without a GPU."""
import pytest

import golden_util as G

CASES = [(fn, name) for fn in G.functions() for name in G.case_names(fn)]


def test_corpus_is_complete():
    counts = {fn: len(G.case_names(fn)) for fn in G.functions()}
    assert counts == {
        "blob_to_kzg_commitment": 11, "compute_blob_kzg_proof": 15, "compute_cells": 11,
        "compute_cells_and_kzg_proofs": 11, "compute_challenge": 9, "compute_kzg_proof": 52,
        "compute_verify_cell_kzg_proof_batch_challenge": 10, "recover_cells_and_kzg_proofs": 18,
        "verify_blob_kzg_proof": 29, "verify_blob_kzg_proof_batch": 24,
        "verify_cell_kzg_proof_batch": 32, "verify_kzg_proof": 122}


@pytest.mark.parametrize("fn,name", CASES, ids=[c[1] for c in CASES])
def test_oracle_matches_spec_vector(oracle, fn, name):
    got, exp = G.run_case(oracle, fn, name)
    assert got == exp

"""Loader for the packed consensus-spec vectors in tests/golden/ (made by tools/make_golden.py)
and the case runner shared by the oracle tests (CPU) and the HIP parity tests (GPU).  Semantics
follow bindings/python/tests.py:39-275 of the reference: ``output: null`` <=> the call must fail,
otherwise outputs must match byte for byte."""
import json
import os

from kzg_ctypes import KzgError

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def _load():
    if not _cache:
        with open(os.path.join(GOLDEN, "objects.json")) as f:
            _cache["index"] = json.load(f)
        with open(os.path.join(GOLDEN, "objects.bin"), "rb") as f:
            _cache["blob"] = f.read()
        with open(os.path.join(GOLDEN, "cases.json")) as f:
            _cache["cases"] = json.load(f)
    return _cache


def _unpack(v):
    c = _load()
    if isinstance(v, dict) and "$obj" in v:
        off, ln = c["index"][v["$obj"]]
        return c["blob"][off:off + ln]
    if isinstance(v, str) and v.startswith("0x"):
        try:
            return bytes.fromhex(v[2:])
        except ValueError:
            return None  # malformed on purpose -> the call must be rejected
    if isinstance(v, list):
        return [_unpack(x) for x in v]
    if isinstance(v, dict):
        return {k: _unpack(x) for k, x in v.items()}
    return v


def functions():
    return sorted(_load()["cases"].keys())


def case_names(fn):
    return sorted(_load()["cases"][fn].keys())


def get_case(fn, name):
    c = _load()["cases"][fn][name]
    return _unpack(c["input"]), _unpack(c["output"])


def _has_none(v):
    if v is None:
        return True
    if isinstance(v, list):
        return any(_has_none(x) for x in v)
    return False


def run_case(api, fn, name):
    """Run one vector against ``api`` (a kzg_ctypes.Kzg); returns (got, expected)."""
    inp, exp = get_case(fn, name)
    try:
        if any(_has_none(v) for v in inp.values()):
            raise KzgError("malformed hex input")
        if fn == "blob_to_kzg_commitment":
            got = api.blob_to_kzg_commitment(inp["blob"])
        elif fn == "compute_kzg_proof":
            got = list(api.compute_kzg_proof(inp["blob"], inp["z"]))
        elif fn == "compute_blob_kzg_proof":
            got = api.compute_blob_kzg_proof(inp["blob"], inp["commitment"])
        elif fn == "verify_kzg_proof":
            got = api.verify_kzg_proof(inp["commitment"], inp["z"], inp["y"], inp["proof"])
        elif fn == "verify_blob_kzg_proof":
            got = api.verify_blob_kzg_proof(inp["blob"], inp["commitment"], inp["proof"])
        elif fn == "verify_blob_kzg_proof_batch":
            got = api.verify_blob_kzg_proof_batch(inp["blobs"], inp["commitments"], inp["proofs"])
        elif fn == "compute_cells":
            got = api.compute_cells(inp["blob"])
        elif fn == "compute_cells_and_kzg_proofs":
            got = list(api.compute_cells_and_kzg_proofs(inp["blob"]))
        elif fn == "recover_cells_and_kzg_proofs":
            got = list(api.recover_cells_and_kzg_proofs(inp["cell_indices"], inp["cells"]))
        elif fn == "verify_cell_kzg_proof_batch":
            got = api.verify_cell_kzg_proof_batch(inp["commitments"], inp["cell_indices"],
                                                  inp["cells"], inp["proofs"])
        elif fn == "compute_challenge":
            got = api.compute_challenge(inp["blob"], inp["commitment"])
        elif fn == "compute_verify_cell_kzg_proof_batch_challenge":
            cells = [b"".join(c) if isinstance(c, list) else c for c in inp["cosets_evals"]]
            got = api.compute_verify_cell_kzg_proof_batch_challenge(
                inp["commitments"], inp["commitment_indices"], inp["cell_indices"], cells, inp["proofs"])
        else:
            raise NotImplementedError(fn)
    except KzgError:
        got = None
    return got, exp

"""Kernel resource gate (no GPU): the register / scratch budget of the kernels whose occupancy the design depends on,
read from the gfx950 code objects inside the PRODUCT's object files (the ones `make` linked into libckzg_hip.so, so the
Makefile's per-file flags such as -DCKZG_F28_ASM_BLOCKS are in).  Round 4 shipped k_msm_accumulate at 307 unified
VGPRs (one wave per SIMD instead of two) because nothing looked; this test is what looks."""
import importlib.util
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _tool():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def table():
    if os.environ.get("CKZG_HIP_SO"):
        pytest.skip("sanitizer / variant build: the budget is the product's")
    return _tool().collect()


def test_budgeted_kernels_exist_and_fit(table):
    budget = json.load(open(os.path.join(HERE, "kernel_budget.json")))
    budget.pop("_comment")
    bad = []
    for name, b in budget.items():
        assert name in table, "kernel %s is no longer in the product (rename it in tests/kernel_budget.json)" % name
        r = table[name]
        for key in ("vgpr", "agpr", "scratch", "lds"):
            if key in b and r[key] > b[key]:
                bad.append("%s: %s %d > budget %d" % (name, key, r[key], b[key]))
        if r["waves"] < b["min_waves"]:
            bad.append("%s: %d waves/SIMD < %d" % (name, r["waves"], b["min_waves"]))
    assert not bad, "\n".join(bad)


def test_object_files_are_the_linked_ones(table):
    """every kernel symbol of the object files is in libckzg_hip.so's device code too (same build)"""
    m = _tool()
    so = os.path.join(ROOT, "c-kzg-4844_amd", "libckzg_hip.so")
    linked = {n for (n, *_rest) in m.kernels_of(so)}
    assert linked, "no gfx950 code object found in libckzg_hip.so"
    objs = {k.split(":", 1)[1] for k in table}
    assert objs == linked, "object files and the linked library differ: %s" % sorted(objs ^ linked)[:8]
    # ... with the same register counts
    by_name = {n: (v, a, p) for (n, v, a, _s, _l, p) in m.kernels_of(so)}
    for k, r in table.items():
        n = k.split(":", 1)[1]
        assert by_name[n] == (r["vgpr"], r["agpr"], r["scratch"]), n


def test_waves_formula():
    m = _tool()
    assert m.waves(249) == 2 and m.waves(256) == 2 and m.waves(257) == 1 and m.waves(128) == 4 and m.waves(4) == 8

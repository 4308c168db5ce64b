"""The drop-in boundary from C: examples/commit.c is written against include/ckzg.h only and must
compile with a C compiler and link against libckzg_hip.so (CPU check); on the GPU box it runs
and its commitment must equal the oracle's."""
import os
import subprocess

import pytest

from conftest import ROOT
from kzg_ctypes import HIP_SO, TRUSTED_SETUP
from watchdog import run_watched

EXE = os.path.join(ROOT, "examples", "commit")


def _build():
    libdir = os.path.dirname(HIP_SO)
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "commit.c"), "-L", libdir, "-lckzg_hip",
           "-Wl,-rpath," + libdir, "-o", EXE]
    subprocess.check_call(cmd)


def test_c_program_compiles_and_links_against_the_library():
    if not os.path.exists(HIP_SO):
        pytest.fail("libckzg_hip.so not built")
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_program_runs_and_matches_oracle(oracle):
    _build()
    out = run_watched([EXE, TRUSTED_SETUP], timeout=240, name="c_link_example")
    assert out.returncode == 0, out.stdout + out.stderr
    blob = bytearray(131072)
    for i in range(4096):
        blob[32 * i + 31] = i & 0xff
        blob[32 * i + 30] = i >> 8
    expect = oracle.blob_to_kzg_commitment(bytes(blob)).hex()
    assert ("commitment " + expect) in out.stdout
    assert "verified=1" in out.stdout

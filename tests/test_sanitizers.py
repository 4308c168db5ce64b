"""The sanitizer passes of tools/run_sanitized.sh as tests (the reference's counterpart: src/Makefile:214-238).
They need the sanitizer builds (`make -C c-kzg-4844_amd sanitize`: ~3 min for the product, ~11 min for the host shim
under g++ -O1 -g with ASan+UBSan), which are not part of the ordinary build; a checkout without them skips.  And they are OPT-IN (CKZG_RUN_SANITIZERS=1):
built libraries travel to the driver's GPU box with the tree, and a pass of several minutes must never ride along in
the driver's time-limited run of the ordinary suite just because a sanitizer build was left behind."""
import os

import pytest

from watchdog import run_watched

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_LIB = os.path.join(ROOT, "c-kzg-4844_amd", "libckzg_hip_san.so")
SAN_SHIM = os.path.join(ROOT, "c-kzg-4844_amd", "csrc", "libhost_shim_san.so")
TSAN_LIB = os.path.join(ROOT, "c-kzg-4844_amd", "libckzg_hip_tsan.so")


def _run(mode):
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "CKZG_HIP_SO", "CKZG_SHIM_SO")}
    r = run_watched(["bash", os.path.join(ROOT, "tools", "run_sanitized.sh"), mode], cwd=ROOT, env=env, timeout=2400,
                    name="sanitized_" + mode)
    assert r.returncode == 0 and "sanitizers: clean" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])


OPT_IN = bool(os.environ.get("CKZG_RUN_SANITIZERS"))


@pytest.mark.timeout(2500)
@pytest.mark.skipif(not OPT_IN or not (os.path.exists(SAN_LIB) and os.path.exists(SAN_SHIM)),
                    reason="opt-in: CKZG_RUN_SANITIZERS=1 and make -C c-kzg-4844_amd sanitize")
def test_host_arithmetic_and_abi_under_asan_ubsan():
    _run("cpu")


@pytest.mark.gpu
@pytest.mark.timeout(2500)
@pytest.mark.skipif(not OPT_IN or not os.path.exists(SAN_LIB), reason="opt-in: CKZG_RUN_SANITIZERS=1 and make -C c-kzg-4844_amd sanitize")
def test_vectors_fuzz_and_verification_forms_under_asan_ubsan_on_the_gpu():
    _run("gpu")


# ThreadSanitizer watches a process whose GPU runtime is not instrumented and maps memory behind its back: what it
# reports can depend on where the kernel happens to place a mapping (tools/tsan.supp, api_common.hpp:
# CKZG_TSAN_NEW_MEMORY).  The pass is therefore an opt-in run (CKZG_RUN_TSAN=1; clean runs: profiles/r03_sanitize_tsan.log),
# not a gate of the ordinary GPU suite.
@pytest.mark.gpu
@pytest.mark.timeout(2500)
@pytest.mark.skipif(not os.path.exists(TSAN_LIB) or not os.environ.get("CKZG_RUN_TSAN"),
                    reason="opt-in: CKZG_RUN_TSAN=1 and make -C c-kzg-4844_amd tsan")
def test_concurrent_callers_and_background_threads_under_tsan_on_the_gpu():
    _run("tsan")

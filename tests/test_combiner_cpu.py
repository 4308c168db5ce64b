"""The combiner's state machine (csrc/combiner.hpp) without a GPU.

tests/native/combiner_stress.cpp drives the real header -- host functions stand in for the launches, malloc for the
page-locked buffers -- through five scenarios (plain, go-alone threshold, batch-buffer allocation failing, whole launches
failing, gathering), the burst / starvation checks, a launch whose runner does not come back (wait deadline) and the
schedule that stalled the round-5 driver run (an allocation that fails while the device goes idle) and checks that every caller gets the answer to ITS input.  Built plain and under ThreadSanitizer: the batch
state word, reference counts and copy counter are lock-free, and the GPU suite's TSan pass (tools/run_sanitized.sh tsan)
only sees the schedules a real device produces.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "combiner_stress.cpp")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build(tmp_path, name, extra):
    out = str(tmp_path / name)
    cmd = [HIPCC, "-O1", "-g", "-std=c++17", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC, "-o", out,
           "-lpthread"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("threads,calls", [(2, 1500), (5, 1000), (48, 300), (160, 60)])
def test_combiner_protocol(tmp_path, threads, calls):
    exe = _build(tmp_path, "cstress", [])
    r = subprocess.run([exe, str(threads), str(calls)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("scenario") and ln.endswith("wrong 0")) == 5, r.stdout
    # the two round-6 scenarios: a runner that does not come back fails the OTHER members at the wait deadline; a failed
    # batch-buffer allocation on a device that has gone idle meanwhile (the round-5 driver stall) no longer strands a caller
    assert "runner_never_comes_back:" in r.stdout and "(0 late), wrong 0, wrong afterwards 0" in r.stdout, r.stdout
    assert "wrong 0, open batches rescued 0, calls that gave up 0" in r.stdout, r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_combiner_protocol_tsan(tmp_path):
    exe = _build(tmp_path, "cstress_tsan", ["-fsanitize=thread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0")
    for threads, calls in ((4, 600), (48, 150)):
        r = subprocess.run([exe, str(threads), str(calls)], capture_output=True, text=True, timeout=600, env=env)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
        assert r.returncode == 0, r.stdout + r.stderr[-2000:]

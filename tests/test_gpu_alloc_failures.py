"""Allocation-failure injection through the C-ABI: the reference's error convention says C_KZG_MALLOC for a failed
allocation (src/common/ret.h:24-29, src/common/alloc.c:34-50) and its free_trusted_setup must cope with a
half-loaded struct (src/setup/setup.c:162-190, :497-504).  Here the allocations are hipMalloc / hipHostMalloc, so an
LD_PRELOAD interposer (tests/failalloc/failalloc.c) fails them -- and, in a second pass, the creation of streams and
events -- one by one under every entry point and under load_trusted_setup; tests/failalloc/driver.py holds the checks.

Each walk runs in a process of its own under tests/watchdog.py: a walk that does not come back within its deadline
fails ITS test with the stalled child's thread states, native backtraces and the library's own dump in the message,
and the rest of the suite goes on (round 5's single 1500-second child took the whole record down with it)."""
import json
import os
import subprocess
import sys

import pytest

from watchdog import run_watched

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu
DEADLINE = 280   # seconds per walk; the slowest (every allocation of every entry point, two passes) takes ~25 s


@pytest.fixture(scope="module")
def failalloc_so(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("failalloc") / "failalloc.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "failalloc", "failalloc.c"), "-ldl"])
    return so


def walk(so, sections):
    env = dict(os.environ, LD_PRELOAD=so, FAILALLOC_SO=so, FAILALLOC_SECTIONS=sections)
    r = run_watched([sys.executable, os.path.join(HERE, "failalloc", "driver.py")], env=env, timeout=DEADLINE,
                    name="failalloc_" + sections.replace(",", "+"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    rep = json.loads(lines[-1])
    assert r.returncode == 0 and rep["problems"] == [], (rep, r.stderr[-2000:])
    return rep


def test_every_allocation_of_every_entry_point_may_fail(failalloc_so):
    rep = walk(failalloc_so, "ops")
    host_only = {"verify_kzg_proof"}   # one pairing check on the host: allocates nothing on the device
    assert all(v["single"]["failures_injected"] >= 1 for n, v in rep["ops"].items() if n not in host_only), rep["ops"]


def test_every_allocation_of_load_trusted_setup_may_fail(failalloc_so):
    # the walk really reached its targets: a load allocates for every slot
    rep = walk(failalloc_so, "load")
    assert rep["load"]["ops"]["single"]["failures_injected"] >= 20


def test_streams_and_events_that_cannot_be_created(failalloc_so):
    rep = walk(failalloc_so, "ops_streams_events,load_streams_events")
    assert rep["load"]["ops_streams_events"]["single"]["failures_injected"] >= 40
    assert sum(v["single"]["failures_injected"] for v in rep["ops_streams_events"].values()) >= 10


def test_allocations_fail_under_a_fan_out_over_two_table_sets(failalloc_so):
    rep = walk(failalloc_so, "fan_out")
    assert rep["fan_out"]["failures_injected"] >= 40


def test_allocations_fail_under_coalesced_callers(failalloc_so):
    rep = walk(failalloc_so, "coalesced_callers")
    assert rep["coalesced_callers"]["levels_with_failures"] >= 10 and rep["coalesced_callers"]["batch_launches"] >= 10


def test_background_widening_runs_out_of_memory(failalloc_so):
    rep = walk(failalloc_so, "widening")
    assert rep["widening"]["ran"]

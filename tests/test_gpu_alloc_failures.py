"""Allocation-failure injection through the C-ABI: the reference's error convention says C_KZG_MALLOC for a failed
allocation (src/common/ret.h:24-29, src/common/alloc.c:34-50) and its free_trusted_setup must cope with a
half-loaded struct (src/setup/setup.c:162-190, :497-504).  Here the allocations are hipMalloc / hipHostMalloc, so an
LD_PRELOAD interposer (tests/failalloc/failalloc.c) fails them -- and, in a second pass, the creation of streams and
events -- one by one under every entry point and under load_trusted_setup; tests/failalloc/driver.py holds the checks."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_every_allocation_of_every_entry_point_may_fail(tmp_path):
    so = str(tmp_path / "failalloc.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "failalloc", "failalloc.c"), "-ldl"])
    env = dict(os.environ, LD_PRELOAD=so, FAILALLOC_SO=so)
    r = subprocess.run([sys.executable, os.path.join(HERE, "failalloc", "driver.py")], env=env, capture_output=True,
                       text=True, timeout=1500)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    rep = json.loads(lines[-1])
    assert r.returncode == 0 and rep["problems"] == [], (rep, r.stderr[-2000:])
    # the walks really reached their targets: a first call sets up its arena, a load allocates for every slot
    assert rep["load"]["ops"]["single"]["failures_injected"] >= 20
    assert rep["load"]["ops_streams_events"]["single"]["failures_injected"] >= 40
    assert rep["fan_out"]["failures_injected"] >= 40
    assert rep["coalesced_callers"]["levels_with_failures"] >= 10 and rep["coalesced_callers"]["batch_launches"] >= 10
    host_only = {"verify_kzg_proof"}   # one pairing check on the host: allocates nothing on the device
    assert all(v["single"]["failures_injected"] >= 1 for n, v in rep["ops"].items() if n not in host_only), rep["ops"]
    assert sum(v["single"]["failures_injected"] for v in rep["ops_streams_events"].values()) >= 10

"""Reproducer loop for stalls under injected allocation failures (round 5's driver run ended in one): runs the walks of
tests/failalloc/driver.py over and over, each in a child of its own under tests/watchdog.py, and prints one JSON line
per child: section, pass, seconds, return code.  A child that stalls is described (thread states, native backtraces,
ckzg_hip_debug_dump, Python tracebacks) in gpurun_out/stalls/ and the loop goes on.
  python tools/debug/failalloc_loop.py [passes] [deadline_s] [section,section,...]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from watchdog import Stall, run_watched  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
deadline = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sections = (sys.argv[3] if len(sys.argv) > 3 else
            "ops,load,ops_streams_events,load_streams_events,fan_out,coalesced_callers,widening").split(",")

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
so = os.path.join(tempfile.mkdtemp(), "failalloc.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "failalloc", "failalloc.c"), "-ldl"])
stalls = 0
for p in range(passes):
    for sec in sections:
        env = dict(os.environ, LD_PRELOAD=so, FAILALLOC_SO=so, FAILALLOC_SECTIONS=sec)
        t0 = time.time()
        row = {"section": sec, "pass": p}
        try:
            r = run_watched([sys.executable, os.path.join(ROOT, "tests", "failalloc", "driver.py")], env=env,
                            timeout=deadline, name="loop_%s_%d" % (sec, p))
            row["rc"] = r.returncode
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            row["problems"] = json.loads(lines[-1])["problems"] if lines else ["no report: " + r.stderr[-800:]]
        except Stall as e:
            stalls += 1
            row["rc"] = "STALL"
            row["last_progress"] = [ln for ln in str(e).splitlines() if ln.startswith("[failalloc]")][-3:]
        row["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(row), flush=True)
print(json.dumps({"stalls": stalls, "passes": passes}), flush=True)

"""Child processes of the test suite run under a deadline that FAILS the test instead of hanging the session, and a
child that stalls is made to describe itself first (the GPU box has no debugger):

  * per thread: name, scheduler state, wait channel and the system call it sits in (/proc/<pid>/task/<tid>/...);
  * per thread: its native backtrace (tests/native/stackdump.c, preloaded into the child; SIGUSR2 by tgkill), and
    the product's own view of its waits (ckzg_hip_debug_dump) if the loaded build has it;
  * the Python tracebacks of every thread (PYTHONFAULTHANDLER=1 + SIGABRT, which also ends the child).

The report goes into the assertion message and, when the repo has a gpurun_out/ (it has on a GPU box), into
gpurun_out/stalls/<name>.txt, which travels back.  The reference's runner has no counterpart: it is straight-line code
that cannot block (bindings/python/tests.py:39-275); this library runs pools, queues and device waits."""
import ctypes
import os
import signal
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_SYS_TGKILL = 234   # x86-64


class Stall(AssertionError):
    pass


class Result:
    def __init__(self, returncode, stdout, stderr):
        self.returncode, self.stdout, self.stderr = returncode, stdout, stderr


def stackdump_so():
    """tests/native/stackdump.c, built once per user and source version; None if it cannot be built (the deadline
    still holds, the report is then /proc + Python tracebacks only)"""
    src = os.path.join(HERE, "native", "stackdump.c")
    try:
        tag = "%d_%d" % (os.getuid(), int(os.stat(src).st_mtime))
        so = os.path.join(tempfile.gettempdir(), "ckzg_stackdump_%s.so" % tag)
        if not os.path.exists(so):
            tmp = so + ".%d" % os.getpid()
            subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", src, "-o", tmp, "-ldl"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.replace(tmp, so)
        return so
    except (OSError, subprocess.CalledProcessError):
        return None


def _read(path, limit=400):
    try:
        with open(path, "rb") as f:
            return f.read(limit).decode("utf-8", "replace").strip()
    except OSError as e:
        return "<%s>" % e.__class__.__name__


def describe_process(pid):
    """one line per thread of every process in pid's session (the child and whatever it started)"""
    lines = []
    pids = [pid]
    try:
        for d in os.listdir("/proc"):
            if d.isdigit() and int(d) != pid:
                st = _read("/proc/%s/stat" % d)
                # field 6 (after the parenthesised name) is the session id; the child was started as a session leader
                tail = st.rsplit(")", 1)[-1].split()
                if len(tail) > 3 and tail[3] == str(pid):
                    pids.append(int(d))
    except OSError:
        pass
    for p in pids:
        lines.append("process %d: %s" % (p, _read("/proc/%d/cmdline" % p).replace("\0", " ")))
        try:
            tids = sorted(int(t) for t in os.listdir("/proc/%d/task" % p))
        except OSError:
            continue
        for t in tids:
            base = "/proc/%d/task/%d/" % (p, t)
            stat = _read(base + "stat").rsplit(")", 1)[-1].split()
            lines.append("  thread %d %-16s state=%s wchan=%s syscall=%s" % (
                t, _read(base + "comm"), stat[0] if stat else "?", _read(base + "wchan"), _read(base + "syscall", 120)))
    return pids, "\n".join(lines)


def _signal_every_thread(pids, sig):
    libc = ctypes.CDLL(None, use_errno=True)
    for p in pids:
        try:
            for t in os.listdir("/proc/%d/task" % p):
                libc.syscall(_SYS_TGKILL, p, int(t), sig)
        except OSError:
            pass


def run_watched(cmd, *, timeout, name, env=None, cwd=None):
    """subprocess.run(capture_output=True, text=True) with a deadline that reports.  Raises Stall (an AssertionError)
    when the child does not finish within `timeout` seconds; returns a Result otherwise."""
    env = dict(os.environ if env is None else env)
    so = stackdump_so()
    if so:
        env["LD_PRELOAD"] = (env.get("LD_PRELOAD", "") + " " + so).strip()
    env.setdefault("PYTHONFAULTHANDLER", "1")
    with tempfile.TemporaryFile() as out, tempfile.TemporaryFile() as err:
        proc = subprocess.Popen(cmd, env=env, cwd=cwd, stdout=out, stderr=err, start_new_session=True)
        stalled = None
        try:
            proc.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            pids, threads = describe_process(proc.pid)
            _signal_every_thread(pids, signal.SIGUSR2)   # native backtraces (+ the library's own dump)
            time.sleep(2.0)
            try:
                os.kill(proc.pid, signal.SIGABRT)        # faulthandler: Python tracebacks of every thread
                proc.wait(timeout=5)
            except (OSError, subprocess.TimeoutExpired):
                pass
            stalled = threads
        finally:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.wait()
        out.seek(0)
        err.seek(0)
        stdout = out.read().decode("utf-8", "replace")
        stderr = err.read().decode("utf-8", "replace")
    if stalled is None:
        return Result(proc.returncode, stdout, stderr)
    report = ("%s: no result after %d s -- child killed.\n--- threads\n%s\n--- stdout (tail)\n%s\n--- stderr (tail)\n%s\n" %
              (name, timeout, stalled, stdout[-3000:], stderr[-20000:]))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        try:
            os.makedirs(os.path.join(out_dir, "stalls"), exist_ok=True)
            with open(os.path.join(out_dir, "stalls", "%s_%d.txt" % (name, int(time.time()))), "w") as f:
                f.write(report)
        except OSError:
            pass
    raise Stall(report)


if __name__ == "__main__":   # python tests/watchdog.py <timeout> <name> -- cmd ...
    t, nm = int(sys.argv[1]), sys.argv[2]
    try:
        r = run_watched(sys.argv[4:], timeout=t, name=nm)
    except Stall as e:
        sys.stderr.write(str(e))
        sys.exit(124)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr)
    sys.exit(r.returncode)

"""The library sizes its helper pools (challenge hashers, staging-copy helpers, point decompression at load) by this
process's SHARE of the host: CPUs of the affinity mask divided by the ranks of a
one-process-per-GPU launcher.  Eight ranks that each sized their pools by the machine would put 8 x 32 hashing threads
on the same cores.  No GPU needed: ckzg_hip_host_thread_budget is a host-only query."""
import os
import subprocess
import sys

from conftest import HIP_SO

CODE = ("import ctypes as C, os\n"
        "l = C.CDLL(%r)\n"
        "print(len(os.sched_getaffinity(0)), l.ckzg_hip_host_thread_budget())\n"
        "assert l.ckzg_hip_set_option(b'host_threads', 3) == 0\n"
        "print(l.ckzg_hip_host_thread_budget())\n"
        "assert l.ckzg_hip_set_option(b'host_threads', 0) == 0\n"
        "print(l.ckzg_hip_host_thread_budget())\n" % HIP_SO)


def _run(env_extra, affinity=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "LOCAL_WORLD_SIZE") and not k.startswith(("SLURM_", "PMI_", "OMPI_"))}
    env.update(env_extra)
    cmd = [sys.executable, "-c", CODE]
    if affinity is not None:
        cmd = ["taskset", "-c", affinity] + cmd
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = r.stdout.split("\n")
    cpus, auto = (int(x) for x in lines[0].split())
    return cpus, auto, int(lines[1]), int(lines[2])


def test_budget_is_the_affinity_mask_without_a_launcher():
    cpus, auto, forced, back = _run({})
    assert auto == cpus
    assert forced == 3 and back == auto


def test_budget_divides_by_the_ranks_on_the_host():
    cpus, one, _, _ = _run({})
    for world in (2, 8):
        _, auto, forced, back = _run({"WORLD_SIZE": str(world)})
        assert auto == max(1, one // world), (world, one, auto)
        assert forced == 3 and back == auto      # the option overrides, 0 gives the automatic value back
    # torchrun's LOCAL_WORLD_SIZE (ranks on THIS node) wins over WORLD_SIZE (ranks of the job)
    _, auto, _, _ = _run({"WORLD_SIZE": "64", "LOCAL_WORLD_SIZE": "2"})
    assert auto == max(1, one // 2)
    # ... and so do the node-local variables of the MPI launchers and Slurm
    for name in ("OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE"):
        _, auto, _, _ = _run({"WORLD_SIZE": "64", name: "4"})
        assert auto == max(1, one // 4), name


def test_job_wide_world_size_is_clamped_to_one_node():
    """WORLD_SIZE = 64 from mpirun / srun on 8 nodes (no LOCAL_WORLD_SIZE) must not leave each rank cpus / 64 threads:
    at most 8 ranks share a host (one process per GPU, 8 GPUs per node).  The clamp does not depend on the devices
    VISIBLE to the process -- a launcher that binds one GPU per rank would otherwise give every rank the whole machine
     -- and the query initialises no GPU runtime."""
    cpus, one, _, _ = _run({})
    for visible in (None, "0"):
        extra = {} if visible is None else {"ROCR_VISIBLE_DEVICES": visible, "HIP_VISIBLE_DEVICES": visible}
        _, auto, _, _ = _run(dict(extra, WORLD_SIZE="64"))
        assert auto == max(1, one // 8), (one, auto, visible)
        _, auto, _, _ = _run(dict(extra, WORLD_SIZE="4"))
        assert auto == max(1, one // 4), (one, auto, visible)
    # Slurm's compressed per-node lists and PMI's local size
    for name, val, ranks in (("SLURM_TASKS_PER_NODE", "8(x2)", 8), ("SLURM_STEP_TASKS_PER_NODE", "4,3", 4), ("PMI_LOCAL_SIZE", "2", 2)):
        _, auto, _, _ = _run({"WORLD_SIZE": "64", name: val})
        assert auto == max(1, one // ranks), (name, val, auto)
    # garbage is ignored, not atoi'ed
    for junk in ("", "abc", "-3", "0", "99999999999999999999", "8x"):
        _, auto, _, _ = _run({"WORLD_SIZE": junk})
        assert auto == one, junk


def test_budget_follows_a_restricted_affinity_mask():
    if len(os.sched_getaffinity(0)) < 2:
        return
    first = sorted(os.sched_getaffinity(0))[:2]
    cpus, auto, _, _ = _run({}, affinity=",".join(str(c) for c in first))
    assert cpus == 2 and auto <= 2

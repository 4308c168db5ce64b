"""The N>1 path on CPU: world_size 2, gloo backend.  Sharding + gather logic is device-agnostic;
here the per-shard compute is the CPU oracle (allowed in tests) so the distributed result can be
compared with the single-process one."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import hashlib, os, sys
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch, torch.distributed as dist
    from kzg_ctypes import _m as pkg
    import importlib.util
    spec = importlib.util.spec_from_file_location("multi_gpu", os.path.join(%(root)r, "c-kzg-4844_amd", "multi_gpu.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    from oracle_binding import OracleKzg

    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank = dist.get_rank()
    orc = OracleKzg(os.path.join(%(root)r, "oracle", "liboracle.so"))
    n = 5   # ragged on purpose: shards of 3 and 2
    def blob(i):
        return b"".join(b"\\x00" + hashlib.sha256(b"mg%%d|%%d" %% (i, j)).digest()[:31] for j in range(4096))
    blobs = [blob(i) for i in range(n)]
    def compute(lo, hi):
        rows = [list(orc.blob_to_kzg_commitment(blobs[i])) for i in range(lo, hi)]
        return torch.tensor(rows, dtype=torch.uint8)
    got = mg.sharded_map(compute, n, 48, torch.device("cpu"))
    commits = [bytes(got[i].tolist()) for i in range(n)]
    proofs = [orc.compute_blob_kzg_proof(blobs[i], commits[i]) for i in range(n)]
    def verify(lo, hi):
        try:
            return 0, orc.verify_blob_kzg_proof_batch(blobs[lo:hi], commits[lo:hi], proofs[lo:hi])
        except Exception:
            return 1, False
    ret, ok = mg.sharded_verify(verify, n, torch.device("cpu"))
    bad = list(proofs); bad[4] = proofs[0]
    def verify_bad(lo, hi):
        return 0, orc.verify_blob_kzg_proof_batch(blobs[lo:hi], commits[lo:hi], bad[lo:hi])
    ret2, ok2 = mg.sharded_verify(verify_bad, n, torch.device("cpu"))
    # the gather of the commitment bench (equal shards, results to rank 0 only)
    mine = torch.full((3, 48), 10 + rank, dtype=torch.uint8)
    at0 = mg.gather_to_rank0(mine)
    assert (at0 is None) == (rank != 0)
    if rank == 0:
        assert at0.shape == (6, 48) and at0[:3].eq(10).all() and at0[3:].eq(11).all()
    if rank == 0:
        single = [orc.blob_to_kzg_commitment(b) for b in blobs]
        assert commits == single, "sharded gather differs from single process"
        assert (ret, ok) == (0, True) and (ret2, ok2) == (0, False)
        print("MULTI_OK")
    dist.destroy_process_group()
''')


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert "MULTI_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_shard_bounds():
    import importlib.util
    spec = importlib.util.spec_from_file_location("multi_gpu", os.path.join(ROOT, "c-kzg-4844_amd", "multi_gpu.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    assert mg.shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert mg.shard_bounds(4096, 8) == [(512 * i, 512 * (i + 1)) for i in range(8)]
    assert mg.shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    assert mg.shard_bounds(0, 2) == [(0, 0), (0, 0)]


def test_bench_gpus_flag_is_honoured_or_fails_loudly():
    """`python bench.py --gpus N` without a launcher must spawn N ranks itself -- and must refuse, not
    silently run on one device, when fewer than N devices are visible (here: none)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 but only" in (r.stderr + r.stdout)

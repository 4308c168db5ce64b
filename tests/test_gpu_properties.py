"""GPU parity beyond the golden vectors: seeded random inputs against the CPU oracle for every
entry point, edge cases (evaluation point inside the domain, all-zero blobs, duplicated and
unsorted cells, tampered proofs), and size-independent properties at the BASELINE batch size."""
import ctypes as C
import hashlib
import random

import pytest

from test_gpu_commitment import R, _batch, rand_blob

pytestmark = pytest.mark.gpu


def _fr_bytes(v):
    return (v % R).to_bytes(32, "big")


def _domain_point(oracle, i):
    # brp_roots_of_unity[i] as canonical bytes, read out of the oracle's settings
    arr = (C.c_uint8 * (8192 * 32)).from_address(oracle.s.brp_roots_of_unity)
    out = C.create_string_buffer(32)
    oracle.lib.ofr_to_bytes(out, bytes(arr[32 * i:32 * i + 32]))
    return out.raw


def test_compute_kzg_proof_random_and_in_domain(hip, oracle):
    b = rand_blob(41, 0)
    rnd = random.Random(41)
    zs = [_fr_bytes(rnd.randrange(R)), _fr_bytes(0), _fr_bytes(1), _domain_point(oracle, 0),
          _domain_point(oracle, 1), _domain_point(oracle, 2111), _domain_point(oracle, 4095)]
    for z in zs:
        assert hip.compute_kzg_proof(b, z) == oracle.compute_kzg_proof(b, z)


def test_blob_proof_roundtrip_and_tamper(hip, oracle):
    b = rand_blob(42, 0)
    c = hip.blob_to_kzg_commitment(b)
    p = hip.compute_blob_kzg_proof(b, c)
    assert p == oracle.compute_blob_kzg_proof(b, c)
    assert hip.verify_blob_kzg_proof(b, c, p) is True
    other = hip.blob_to_kzg_commitment(rand_blob(42, 1))
    assert hip.verify_blob_kzg_proof(b, other, p) is False
    assert hip.verify_blob_kzg_proof(b, c, other) is False
    z = _fr_bytes(12345)
    proof, y = hip.compute_kzg_proof(b, z)
    assert hip.verify_kzg_proof(c, z, y, proof) is True
    assert hip.verify_kzg_proof(c, z, _fr_bytes(int.from_bytes(y, "big") + 1), proof) is False


@pytest.mark.parametrize("n", [2, 3, 5, 6, 70, 513])
def test_verify_blob_batch_sizes_cover_host_and_gpu_paths(hip, n):
    # n <= 5 keeps the scalar multiplications on the host, larger n uses k_validate_g1 / k_lincomb;
    # large batches also hash the Fiat-Shamir challenges on the GPU (k_sha256_challenges); the
    # threshold depends on the host CPU, so n = 513 is run both ways
    base = [rand_blob(43, i) for i in range(5)]
    cs = [hip.blob_to_kzg_commitment(b) for b in base]
    ps = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(base, cs)]
    blobs = [base[i % 5] for i in range(n)]
    C_ = [cs[i % 5] for i in range(n)]
    P = [ps[i % 5] for i in range(n)]
    assert hip.verify_blob_kzg_proof_batch(blobs, C_, P) is True
    if n > 500:
        hip.lib.ckzg_hip_set_option(b"gpu_sha_min", 1)      # force the GPU hash ...
        try:
            assert hip.verify_blob_kzg_proof_batch(blobs, C_, P) is True
            hip.lib.ckzg_hip_set_option(b"gpu_sha_min", 1 << 20)  # ... and the host hash
            assert hip.verify_blob_kzg_proof_batch(blobs, C_, P) is True
        finally:
            hip.lib.ckzg_hip_set_option(b"gpu_sha_min", 0)
    P[n - 1] = ps[n % 5]  # the proof of a different blob
    assert hip.verify_blob_kzg_proof_batch(blobs, C_, P) is False
    # a commitment that is on the curve but outside the subgroup must be rejected as BADARGS;
    # x = 0, y = 2 has order 3 (compressed: 0x80.. with the sign chosen for y = 2)
    from kzg_ctypes import KzgError
    bad = bytes([0x80]) + bytes(47)
    with pytest.raises(KzgError):
        hip.verify_blob_kzg_proof_batch(blobs, [bad] + C_[1:], P)


def test_zero_blob_everywhere(hip, oracle):
    z = bytes(131072)
    assert hip.blob_to_kzg_commitment(z) == bytes([0xc0]) + bytes(47)
    cells, proofs = hip.compute_cells_and_kzg_proofs(z)
    assert all(c == bytes(2048) for c in cells)
    assert all(p == bytes([0xc0]) + bytes(47) for p in proofs)
    assert hip.verify_cell_kzg_proof_batch([bytes([0xc0]) + bytes(47)] * 4, [0, 5, 77, 127], cells[:4], proofs[:4]) is True


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_recover_random_patterns(hip, seed):
    rnd = random.Random(seed)
    b = rand_blob(44, seed)
    cells, proofs = hip.compute_cells_and_kzg_proofs(b)
    keep = sorted(rnd.sample(range(128), rnd.choice([64, 65, 100, 127])))
    rc, rp = hip.recover_cells_and_kzg_proofs(keep, [cells[i] for i in keep])
    assert rc == cells and rp == proofs


def test_verify_cells_multi_blob_duplicates_unsorted(hip, oracle):
    rnd = random.Random(45)
    blobs = [rand_blob(45, i) for i in range(3)]
    cps = [hip.compute_cells_and_kzg_proofs(b) for b in blobs]
    cms = [hip.blob_to_kzg_commitment(b) for b in blobs]
    picks = [(rnd.randrange(3), rnd.randrange(128)) for _ in range(40)]
    picks += picks[:5]  # duplicates
    rnd.shuffle(picks)
    commitments = [cms[b] for b, _ in picks]
    idx = [k for _, k in picks]
    cells = [cps[b][0][k] for b, k in picks]
    proofs = [cps[b][1][k] for b, k in picks]
    assert hip.verify_cell_kzg_proof_batch(commitments, idx, cells, proofs) is True
    assert oracle.verify_cell_kzg_proof_batch(commitments, idx, cells, proofs) is True
    bad = bytearray(cells[7])
    bad[31] ^= 1
    cells2 = list(cells)
    cells2[7] = bytes(bad)
    assert hip.verify_cell_kzg_proof_batch(commitments, idx, cells2, proofs) is False
    proofs2 = list(proofs)
    proofs2[3] = proofs[4] if proofs[4] != proofs[3] else proofs[5]
    assert hip.verify_cell_kzg_proof_batch(commitments, idx, cells, proofs2) is False


def test_full_batch_checksum_property(hip, oracle):
    # BASELINE configs[1] size: 1024 blobs in one call.  Linearity gives a size-independent
    # check: the sum of all 1024 commitments equals the commitment of the element-wise sum blob.
    n = 1024
    base = [rand_blob(46, i) for i in range(16)]
    blobs = [base[(i * 7 + i // 16) % 16] for i in range(n)]
    counts = [0] * 16
    for i in range(n):
        counts[(i * 7 + i // 16) % 16] += 1
    ret, outs, status = _batch(hip, blobs)
    assert ret == 0 and not any(status)
    sum_blob = bytearray()
    for j in range(4096):
        v = sum(counts[k] * int.from_bytes(base[k][32 * j:32 * j + 32], "big") for k in range(16)) % R
        sum_blob += v.to_bytes(32, "big")
    expect = oracle.blob_to_kzg_commitment(bytes(sum_blob))
    o = oracle.lib
    acc = C.create_string_buffer(144)
    aff, pt = C.create_string_buffer(96), C.create_string_buffer(144)
    for c in outs:
        assert o.og1_uncompress(aff, c) == 0
        o.og1_from_affine(pt, aff)
        o.og1_add(acc, acc, pt)
    got = C.create_string_buffer(48)
    o.og1_compress(got, acc)
    assert got.raw == expect
    # and every distinct blob's commitment agrees with the single-call API
    for k in range(16):
        i = next(i for i in range(n) if (i * 7 + i // 16) % 16 == k)
        assert outs[i] == hip.blob_to_kzg_commitment(base[k])


def test_concurrent_callers_share_one_settings(hip):
    # the reference allows concurrent readers of one loaded KZGSettings (Rust marks it Send+Sync,
    # bindings/rust/src/bindings/mod.rs:910-913; the Go benchmark fans out goroutines,
    # bindings/go/main_test.go:953-971): calls from several threads must give the same bytes
    import threading
    blobs = [rand_blob(47, i) for i in range(6)]
    expect_c = [hip.blob_to_kzg_commitment(b) for b in blobs]
    expect_p = hip.compute_cells_and_kzg_proofs(blobs[0])
    errors = []

    def worker(tid):
        try:
            for k in range(4):
                i = (tid + k) % 6
                if hip.blob_to_kzg_commitment(blobs[i]) != expect_c[i]:
                    errors.append(("commit", tid, i))
            if tid % 2 == 0 and hip.compute_cells_and_kzg_proofs(blobs[0]) != expect_p:
                errors.append(("cells", tid))
        except Exception as e:  # noqa: BLE001
            errors.append(("exc", tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def _proof_batch(hip, blobs, commitments):
    n = len(blobs)
    out = C.create_string_buffer(48 * n)
    status = C.create_string_buffer(n)
    f = hip.lib.ckzg_hip_compute_blob_kzg_proof_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_void_p]
    ret = f(out, status, b"".join(blobs), b"".join(commitments), n, C.addressof(hip.s))
    return ret, [out.raw[48 * i:48 * i + 48] for i in range(n)], list(status.raw)


@pytest.mark.parametrize("n", [1, 5, 20])
def test_blob_proof_batch_matches_single_calls_and_oracle(hip, oracle, n):
    # GPU evaluation + quotient kernel (k_eval_barycentric<true>) + 4096-term MSM per blob
    blobs = [rand_blob(48, i) for i in range(n)]
    cms = [hip.blob_to_kzg_commitment(b) for b in blobs]
    ret, proofs, status = _proof_batch(hip, blobs, cms)
    assert ret == 0 and not any(status)
    for i in range(n):
        assert proofs[i] == hip.compute_blob_kzg_proof(blobs[i], cms[i])
    assert proofs[0] == oracle.compute_blob_kzg_proof(blobs[0], cms[0])
    assert hip.verify_blob_kzg_proof_batch(blobs, cms, proofs) is True


def test_blob_proof_batch_flags_bad_inputs(hip):
    blobs = [rand_blob(49, i) for i in range(12)]
    cms = [hip.blob_to_kzg_commitment(b) for b in blobs]
    bad_blob = bytearray(blobs[4])
    bad_blob[64:96] = b"\xff" * 32
    blobs[4] = bytes(bad_blob)
    cms[9] = bytes([0x80]) + bytes(47)  # on the curve, outside the subgroup
    ret, proofs, status = _proof_batch(hip, blobs, cms)
    assert ret == 1
    assert [i for i, s in enumerate(status) if s] == [4, 9]
    for i in (0, 3, 11):
        assert proofs[i] == hip.compute_blob_kzg_proof(blobs[i], cms[i])

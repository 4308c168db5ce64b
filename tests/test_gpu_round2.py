"""GPU tests for the round-2 work: slot pools (concurrent callers overlap), the C-ABI multi-device
fan-out (exercised on one GPU through two table replicas), the GPU-hash/small-batch interaction, the
mirrored x_ext_fft_columns against the oracle's init_fk20_multi_settings, and scalars that sit on the
edges of the GLV split / signed-window recoding of the fixed-base tables."""
import ctypes as C
import threading
import time

import pytest

from conftest import ORACLE_SO
from kzg_ctypes import HIP_SO, Kzg
from test_gpu_commitment import R, _batch, rand_blob

pytestmark = pytest.mark.gpu

LAMBDA = 0xd201000000010000 ** 2 - 1


def _restore(api):
    for k, v in (("commit_wbits", 10), ("fk20_wbits", 0), ("proof_wbits", 8), ("direct_max", -1),
                 ("replicas", 1), ("streams", 8), ("devices", 0), ("gpu_sha_min", 0), ("async_tables", 0)):
        api.lib.ckzg_hip_set_option(k.encode(), v)


# ---------------------------------------------------------------------------------------------
# GPU challenge hashing must never see the small-batch path's unset buffers
# ---------------------------------------------------------------------------------------------

def test_small_verify_batches_with_gpu_sha_forced(hip, oracle):
    blobs = [rand_blob(71, i) for i in range(6)]
    cm = [hip.blob_to_kzg_commitment(b) for b in blobs]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    assert pr[0] == oracle.compute_blob_kzg_proof(blobs[0], cm[0])
    hip.lib.ckzg_hip_set_option(b"gpu_sha_min", 1)
    try:
        # an earlier call leaves other commitments in the slot's arena: a stale read would bind z to them
        assert hip.verify_blob_kzg_proof_batch(blobs[::-1], cm[::-1], pr[::-1])
        for n in range(1, 7):
            assert hip.verify_blob_kzg_proof_batch(blobs[:n], cm[:n], pr[:n]), n
            if n >= 2:
                wrong = cm[:n]
                wrong[n - 1] = cm[0]  # a valid point, but not this blob's commitment
                assert not hip.verify_blob_kzg_proof_batch(blobs[:n], wrong, pr[:n]), n
        assert hip.verify_blob_kzg_proof(blobs[3], cm[3], pr[3])
        assert not hip.verify_blob_kzg_proof(blobs[3], cm[2], pr[3])
    finally:
        hip.lib.ckzg_hip_set_option(b"gpu_sha_min", 0)


# ---------------------------------------------------------------------------------------------
# f4: the host mirror of x_ext_fft_columns equals the oracle's (setup.c:238-330)
# ---------------------------------------------------------------------------------------------

def test_x_ext_fft_columns_match_oracle(hip, oracle):
    o = C.CDLL(ORACLE_SO)
    o.og1_equal.restype = C.c_bool
    o.og1_equal.argtypes = [C.c_void_p, C.c_void_p]
    cols_h = C.cast(hip.s.x_ext_fft_columns, C.POINTER(C.c_void_p))
    cols_o = C.cast(oracle.s.x_ext_fft_columns, C.POINTER(C.c_void_p))
    bad = []
    for j in range(128):
        for i in range(64):
            if not o.og1_equal(cols_h[j] + 144 * i, cols_o[j] + 144 * i):
                bad.append((j, i))
    assert not bad, bad[:8]
    # and the other arrays a binding may read: setup points and roots, byte for byte where the
    # representation is canonical (Fr Montgomery limbs), projectively for G1
    assert C.string_at(hip.s.roots_of_unity, 8193 * 32) == C.string_at(oracle.s.roots_of_unity, 8193 * 32)
    assert C.string_at(hip.s.brp_roots_of_unity, 8192 * 32) == C.string_at(oracle.s.brp_roots_of_unity, 8192 * 32)
    assert C.string_at(hip.s.reverse_roots_of_unity, 8193 * 32) == C.string_at(oracle.s.reverse_roots_of_unity, 8193 * 32)
    for i in (0, 1, 2047, 4095):
        assert o.og1_equal(hip.s.g1_values_monomial + 144 * i, oracle.s.g1_values_monomial + 144 * i)
        assert o.og1_equal(hip.s.g1_values_lagrange_brp + 144 * i, oracle.s.g1_values_lagrange_brp + 144 * i)


# ---------------------------------------------------------------------------------------------
# GLV tables: scalars on the edges of the split and of the signed windows
# ---------------------------------------------------------------------------------------------

def _edge_scalars(wbits):
    half = 1 << (wbits - 1)
    twin = 127 // wbits + 1
    vals = [0, 1, R - 1, LAMBDA, LAMBDA + 1, LAMBDA - 1, LAMBDA // 2, LAMBDA // 2 + 1, R - LAMBDA,
            (LAMBDA + 1) * (LAMBDA // 2), (LAMBDA + 1) * (LAMBDA // 2 + 1) % R, LAMBDA * LAMBDA % R]
    # half-scalars whose windows are all exactly +-half (the recoding's carry rule), in both halves and signs
    m = sum(half << (wbits * w) for w in range(0, twin - 1, 2))
    m2 = sum((half - 1) << (wbits * w) for w in range(twin - 1))
    for a in (m, m2, m >> 1):
        for b in (m, m2, 1, 0):
            for sa in (1, -1):
                for sb in (1, -1):
                    vals.append((sa * a + LAMBDA * sb * b) % R)
    return vals


@pytest.mark.parametrize("wbits", [4, 8, 13, 16])
def test_glv_edge_scalars_vs_oracle(oracle, wbits):
    api = Kzg(HIP_SO, "", precompute=0, options={"commit_wbits": wbits, "proof_wbits": 0})
    _restore(api)
    try:
        vals = _edge_scalars(wbits)
        blob = b"".join(vals[j % len(vals)].to_bytes(32, "big") for j in range(4096))
        assert api.blob_to_kzg_commitment(blob) == oracle.blob_to_kzg_commitment(blob)
        # a non-canonical element must be flagged, and must not read outside the table while doing so
        bad = bytearray(blob)
        bad[32 * 77:32 * 78] = (R + 5).to_bytes(32, "big")
        bad[32 * 78:32 * 79] = b"\xff" * 32
        ret, _, st = _batch(api, [blob, bytes(bad), blob])
        assert ret == 1 and st == [0, 1, 0]
    finally:
        api.close()


# ---------------------------------------------------------------------------------------------
# N1: concurrent callers of one KZGSettings overlap on the GPU
# ---------------------------------------------------------------------------------------------

def test_eight_threads_of_single_blob_commitments_overlap(hip):
    blobs = [rand_blob(81, i) for i in range(8)]
    expect = [hip.blob_to_kzg_commitment(b) for b in blobs]
    per_thread = 60

    def run(nthreads):
        errs = []

        def work(t):
            for k in range(per_thread):
                i = (t + k) % 8
                if hip.blob_to_kzg_commitment(blobs[i]) != expect[i]:
                    errs.append((t, k))

        th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        assert not errs, errs[:4]
        return nthreads * per_thread / dt

    run(8)  # every slot allocates its scratch once
    r1, r8 = run(1), run(8)
    print("single-blob commitments/s: 1 thread %.0f, 8 threads %.0f (x%.2f)" % (r1, r8, r8 / r1))
    assert r8 > 2.0 * r1, (r1, r8)


def test_mixed_concurrent_calls_are_correct(hip, oracle):
    blobs = [rand_blob(82, i) for i in range(4)]
    cm = [hip.blob_to_kzg_commitment(b) for b in blobs]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, cm)]
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in blobs]
    assert cp[1] == oracle.compute_cells_and_kzg_proofs(blobs[1])
    errs = []

    def work(t):
        try:
            for k in range(3):
                i = (t + k) % 4
                kind = (t + k) % 5
                if kind == 0:
                    ok = hip.blob_to_kzg_commitment(blobs[i]) == cm[i]
                elif kind == 1:
                    ok = hip.compute_cells_and_kzg_proofs(blobs[i]) == cp[i]
                elif kind == 2:
                    ok = hip.verify_blob_kzg_proof_batch(blobs, cm, pr)
                elif kind == 3:
                    keep = list(range(0, 128, 2))
                    ok = hip.recover_cells_and_kzg_proofs(keep, [cp[i][0][c] for c in keep]) == cp[i]
                else:
                    cols = list(range(16 * t % 128, 16 * t % 128 + 16))
                    ok = hip.verify_cell_kzg_proof_batch([cm[i]] * 16, cols, [cp[i][0][c] for c in cols],
                                                         [cp[i][1][c] for c in cols])
                if not ok:
                    errs.append((t, k, kind))
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(12)]  # more threads than slots: some wait
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs


# ---------------------------------------------------------------------------------------------
# C-ABI multi-device fan-out, on one GPU through two replicas of the tables
# ---------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def hip2():
    api = Kzg(HIP_SO, "", precompute=0, options={"replicas": 2, "commit_wbits": 8, "proof_wbits": 6})
    _restore(api)
    api.lib.ckzg_hip_num_devices.restype = C.c_int
    assert api.lib.ckzg_hip_num_devices(api.sp) == 2
    yield api
    api.close()


def test_fan_out_commitments_and_status(hip, hip2):
    base = [rand_blob(91, i) for i in range(5)]
    single = [hip.blob_to_kzg_commitment(b) for b in base]
    n = 203  # ragged halves: 102 + 101
    blobs = [base[i % 5] for i in range(n)]
    bad = bytearray(base[0])
    bad[32 * 9:32 * 10] = R.to_bytes(32, "big")
    blobs[150] = bytes(bad)  # lands in the second shard
    ret, outs, st = _batch(hip2, blobs)
    assert ret == 1
    assert [i for i, v in enumerate(st) if v] == [150]
    for i in range(n):
        if i != 150:
            assert outs[i] == single[i % 5], i
    # below the per-device minimum the batch stays on one device
    ret, outs, st = _batch(hip2, blobs[:7])
    assert ret == 0 and outs == [single[i % 5] for i in range(7)]


def test_fan_out_verify_blob_batch(hip, hip2):
    base = [rand_blob(92, i) for i in range(4)]
    cm = [hip.blob_to_kzg_commitment(b) for b in base]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(base, cm)]
    n = 600
    bl, cc, pp = [base[i % 4] for i in range(n)], [cm[i % 4] for i in range(n)], [pr[i % 4] for i in range(n)]
    assert hip2.verify_blob_kzg_proof_batch(bl, cc, pp)
    for pos in (7, 433):  # a wrong proof in either shard sinks the batch
        p2 = list(pp)
        p2[pos] = pr[(pos + 1) % 4]
        assert not hip2.verify_blob_kzg_proof_batch(bl, cc, p2), pos
    c2 = list(cc)
    c2[555] = b"\x00" * 48  # not a valid encoding: BADARGS from the shard that holds it
    with pytest.raises(Exception):
        hip2.verify_blob_kzg_proof_batch(bl, c2, pp)


def test_fan_out_cells_proofs_recover_and_blob_proofs(hip, hip2):
    base = [rand_blob(93, i) for i in range(3)]
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in base]
    n = 70
    f = hip2.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    cells = C.create_string_buffer(n * 128 * 2048)
    proofs = C.create_string_buffer(n * 128 * 48)
    st = C.create_string_buffer(n)
    assert f(cells, proofs, st, b"".join(base[i % 3] for i in range(n)), C.c_uint64(n), hip2.sp) == 0
    craw, praw = cells.raw, proofs.raw
    for i in (0, 1, 34, 35, 36, 69):
        assert craw[i * 262144:(i + 1) * 262144] == b"".join(cp[i % 3][0]), i
        assert praw[i * 6144:(i + 1) * 6144] == b"".join(cp[i % 3][1]), i
    keep = list(range(64, 128))
    nb = 40
    rc, rp = hip2.recover_cells_and_kzg_proofs_batch(keep, [[cp[b % 3][0][c] for c in keep] for b in range(nb)])
    for b in (0, 19, 20, 39):
        assert rc[b] == cp[b % 3][0] and rp[b] == cp[b % 3][1], b
    cm = [hip.blob_to_kzg_commitment(b) for b in base]
    pr = [hip.compute_blob_kzg_proof(b, c) for b, c in zip(base, cm)]
    g = hip2.lib.ckzg_hip_compute_blob_kzg_proof_batch
    g.restype = C.c_int
    m = 130
    out = C.create_string_buffer(48 * m)
    assert g(out, None, b"".join(base[i % 3] for i in range(m)), b"".join(cm[i % 3] for i in range(m)),
             C.c_uint64(m), hip2.sp) == 0
    assert all(out.raw[48 * i:48 * i + 48] == pr[i % 3] for i in range(m))


def test_fan_out_verify_cell_batch(hip, hip2):
    base = [rand_blob(94, i) for i in range(2)]
    cm = [hip.blob_to_kzg_commitment(b) for b in base]
    cp = [hip.compute_cells_and_kzg_proofs(b) for b in base]
    n = 4300  # two shards of 2150 cells
    rows = [(i // 128) % 2 for i in range(n)]
    cols = [i % 128 for i in range(n)]
    args = ([cm[r] for r in rows], cols, [cp[r][0][c] for r, c in zip(rows, cols)],
            [cp[r][1][c] for r, c in zip(rows, cols)])
    assert hip2.verify_cell_kzg_proof_batch(*args)
    pr2 = list(args[3])
    pr2[4000] = cp[0][1][(cols[4000] + 1) % 128]
    assert not hip2.verify_cell_kzg_proof_batch(args[0], args[1], args[2], pr2)


def test_single_calls_spread_over_the_pools(hip, hip2):
    b = rand_blob(95, 0)
    exp = hip.blob_to_kzg_commitment(b)
    for _ in range(5):  # round robin: both table replicas serve single calls
        assert hip2.blob_to_kzg_commitment(b) == exp
    assert hip2.compute_cells_and_kzg_proofs(b) == hip.compute_cells_and_kzg_proofs(b)


def _visible_gpus():
    """through the library's own runtime (ckzg_hip_device_count).  NOT through torch: torch brings a second copy of the HIP
    runtime (torch/lib/libamdhip64.so), a process can only have ONE runtime that owns the GPU, and importing torch into
    this pytest process made later dlopen("libamdhip64.so") calls of the suite resolve to that dead second copy."""
    lib = C.CDLL(HIP_SO)
    lib.ckzg_hip_device_count.restype = C.c_int
    return int(lib.ckzg_hip_device_count())


def test_two_real_devices(hip):
    if _visible_gpus() < 2:
        pytest.skip("needs two visible GPUs")
    api = Kzg(HIP_SO, "", precompute=0, options={"devices": 3, "commit_wbits": 8})
    _restore(api)
    try:
        api.lib.ckzg_hip_num_devices.restype = C.c_int
        assert api.lib.ckzg_hip_num_devices(api.sp) == 2
        base = [rand_blob(96, i) for i in range(4)]
        single = [hip.blob_to_kzg_commitment(b) for b in base]
        ret, outs, st = _batch(api, [base[i % 4] for i in range(256)])
        assert ret == 0 and outs == [single[i % 4] for i in range(256)]
    finally:
        api.close()


# ---------------------------------------------------------------------------------------------
# pipelined host-pointer batches (staged input, OutPipe-drained output)
# ---------------------------------------------------------------------------------------------

def _cells_batch(api, blobs_bytes, n, want_cells=True, want_proofs=True):
    f = api.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]
    cells = C.create_string_buffer(n * 128 * 2048) if want_cells else None
    proofs = C.create_string_buffer(n * 128 * 48) if want_proofs else None
    st = C.create_string_buffer(n)
    rc = f(cells, proofs, st, blobs_bytes, n, C.addressof(api.s))
    return rc, cells, proofs, st


@pytest.mark.parametrize("n", [65, 300, 4200])
def test_pipelined_cells_and_proofs_batch(hip_fk20, oracle, n):
    # 65: first size of the throughput path; 300: a ragged second sub-chunk; 4200: three chunks, so the
    # third reuses the first one's output buffers after they have been drained
    base = [rand_blob(97, i) for i in range(3)]
    cp = [hip_fk20.compute_cells_and_kzg_proofs(b) for b in base]
    assert cp[0] == oracle.compute_cells_and_kzg_proofs(base[0])
    order = [(7 * i + i // 5) % 3 for i in range(n)]
    bad_at = n - 3
    blobs = [base[k] for k in order]
    bad = bytearray(base[0])
    bad[32 * 4095:32 * 4096] = (R + 1).to_bytes(32, "big")
    blobs[bad_at] = bytes(bad)
    rc, cells, proofs, st = _cells_batch(hip_fk20, b"".join(blobs), n)
    assert rc == 1
    assert [i for i, v in enumerate(st.raw) if v] == [bad_at]
    craw, praw = memoryview(cells).cast("B"), memoryview(proofs).cast("B")
    exp_c = [b"".join(c[0]) for c in cp]
    exp_p = [b"".join(c[1]) for c in cp]
    check = range(n) if n <= 300 else list(range(0, n, 97)) + [255, 256, 2047, 2048, 2049, 4095, 4096, 4097, n - 1]
    for i in check:
        if i == bad_at:
            continue
        assert craw[i * 262144:(i + 1) * 262144] == exp_c[order[i]], i
        assert praw[i * 6144:(i + 1) * 6144] == exp_p[order[i]], i
    if n == 300:
        rc, cells2, none_p, _ = _cells_batch(hip_fk20, b"".join(blobs[:bad_at]), bad_at, True, False)
        assert rc == 0 and none_p is None and cells2.raw == cells.raw[:bad_at * 262144]
        rc, none_c, proofs2, _ = _cells_batch(hip_fk20, b"".join(blobs[:bad_at]), bad_at, False, True)
        assert rc == 0 and none_c is None and proofs2.raw == proofs.raw[:bad_at * 6144]


def test_pipelined_recover_batch_over_several_chunks(hip_fk20):
    base = [rand_blob(98, i) for i in range(3)]
    cp = [hip_fk20.compute_cells_and_kzg_proofs(b) for b in base]
    keep = list(range(0, 128, 2))
    nb = 1100  # chunks of 512: the third chunk reuses the first one's buffers
    rows = [[cp[b % 3][0][c] for c in keep] for b in range(nb)]
    rc, rp = hip_fk20.recover_cells_and_kzg_proofs_batch(keep, rows)
    for b in list(range(0, nb, 53)) + [511, 512, 1023, 1024, nb - 1]:
        assert rc[b] == cp[b % 3][0] and rp[b] == cp[b % 3][1], b


# ---------------------------------------------------------------------------------------------
# N2: variable-base sums -- bucket kernels and per-term ladders against the oracle's g1_lincomb_fast
# (src/common/lincomb.c:65-123; the reference's own check is Pippenger == naive, src/test/tests.c:929-946)
# ---------------------------------------------------------------------------------------------

def _lincomb(hip, pts, scalars_mont, algo):
    f = hip.lib.ckzg_hip_g1_lincomb
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_int, C.c_void_p]
    out = C.create_string_buffer(144)
    rc = f(out, b"".join(pts), b"".join(scalars_mont), len(pts), algo, C.addressof(hip.s))
    return rc, out


@pytest.fixture(scope="module")
def hip_buckets(hip):
    """The bucket kernels of pippenger.hip are part of the product since round 6 (ckzg_hip_g1_lincomb, algo = 2); rounds
    2-5 kept them in a second library and the product refused algo = 2."""
    return hip


def test_product_build_runs_the_bucket_kernels_on_request(hip):
    # algo = 2 on the identity with the zero scalar: the empty bucket set, through every bucket kernel
    rc, out = _lincomb(hip, [bytes(144)], [bytes(32)], 2)
    assert rc == 0 and out.raw[96:144] == bytes(48)      # Z = 0: the identity


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 700, 5000])
def test_g1_lincomb_buckets_and_ladders_vs_oracle(hip, hip_buckets, n):
    import random
    o = C.CDLL(ORACLE_SO)
    o.og1_equal.restype = C.c_bool
    o.og1_is_inf.restype = C.c_bool
    o.okzg_g1_lincomb_fast.restype = C.c_int
    rnd = random.Random(1000 + n)
    gen = C.create_string_buffer(144)
    aff = C.create_string_buffer(96)
    # the generator through its compressed form (draft-irtf-cfrg-pairing-friendly-curves 4.2.1)
    g48 = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    assert o.og1_uncompress(aff, g48) == 0
    o.og1_from_affine(gen, aff)

    def mul(p, k):
        r = C.create_string_buffer(144)
        kk = (C.c_uint64 * 4)(*[(k >> (64 * i)) & (2 ** 64 - 1) for i in range(4)])
        o.og1_mul_raw(r, p, kk, 255)
        return r.raw

    def fr_mont(k):
        raw = (C.c_uint64 * 4)(*[(k >> (64 * i)) & (2 ** 64 - 1) for i in range(4)])
        out = C.create_string_buffer(32)
        o.ofr_from_raw(out, raw)
        return out.raw

    distinct = [mul(gen, rnd.randrange(1, R)) for _ in range(min(n, 40))]
    inf = bytes(144)
    pts, ks = [], []
    for i in range(n):
        sel = rnd.random()
        if sel < 0.05:
            pts.append(inf)                                  # the identity as an input point
        elif sel < 0.15 and i >= 2:
            neg = C.create_string_buffer(144)
            o.og1_neg(neg, pts[i - 1])
            pts.append(neg.raw)                               # P, -P next to each other
        else:
            pts.append(distinct[rnd.randrange(len(distinct))])   # many duplicates
        k = rnd.choice([0, 1, R - 1, LAMBDA, LAMBDA + 1]) if rnd.random() < 0.1 else rnd.randrange(R)
        if sel >= 0.05 and sel < 0.15 and i >= 2 and rnd.random() < 0.5:
            k = ks[-1]                                        # same scalar on P and -P: the pair cancels
        ks.append(k)
    if n >= 2:
        ks[1] = ks[0]
        pts[1] = pts[0]                                       # an exact duplicate term: a doubling inside a bucket
    sm = [fr_mont(k) for k in ks]
    exp = C.create_string_buffer(144)  # all zero = the identity (Z = 0): the value of the empty sum
    if n:
        assert o.okzg_g1_lincomb_fast(exp, b"".join(pts), b"".join(sm), n) == 0
    for algo in (1, 2, 3, 4, 0):  # ladders by size, buckets, one-lane ladders, four-lane (DPP quad) ladders, default
        rc, got = _lincomb(hip_buckets if algo == 2 else hip, pts, sm, algo)   # ... and served by the buckets build
        assert rc == 0, (n, algo)
        assert o.og1_equal(got, exp), (n, algo)
    if n == 700:
        # a point of the curve outside the prime-order subgroup is refused (the kernels use the endomorphism)
        bad48 = None
        x = 5
        P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
        while bad48 is None:
            y2 = (x ** 3 + 4) % P_MOD
            y = pow(y2, (P_MOD + 1) // 4, P_MOD)
            if y * y % P_MOD == y2:
                enc = bytearray(x.to_bytes(48, "big"))
                enc[0] |= 0x80 | (0x20 if y > P_MOD - y else 0)
                a2 = C.create_string_buffer(96)
                if o.og1_uncompress(a2, bytes(enc)) == 0:
                    j = C.create_string_buffer(144)
                    o.og1_from_affine(j, a2)
                    o.og1_in_subgroup.restype = C.c_bool
                    if not o.og1_in_subgroup(j):
                        bad48 = j.raw
            x += 1
        for lib, algo in ((hip_buckets, 2), (hip, 1)):
            rc, _ = _lincomb(lib, [bad48] + pts[1:], sm, algo)
            assert rc == 1


def test_pinned_caller_memory_is_read_in_place(hip):
    """A page-locked host buffer (here a pinned torch tensor) is DMA'd from directly, chunk by chunk, instead
    of being staged: same commitments, same per-blob status."""
    rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")  # the runtime the library itself is linked against
    base = [rand_blob(99, i) for i in range(3)]
    single = [hip.blob_to_kzg_commitment(b) for b in base]
    n = 700   # several geometric chunks (64, 192, 444)
    blobs = [base[i % 3] for i in range(n)]
    bad = bytearray(base[1])
    bad[0:32] = R.to_bytes(32, "big")
    blobs[650] = bytes(bad)
    raw = b"".join(blobs)
    pinned = C.c_void_p()
    assert rt.hipHostMalloc(C.byref(pinned), C.c_size_t(len(raw)), C.c_uint(0)) == 0
    C.memmove(pinned, raw, len(raw))

    class _T:  # minimal stand-in for the tensor the calls below address
        @staticmethod
        def data_ptr():
            return pinned.value
    t = _T()
    f = hip.lib.ckzg_hip_blob_to_kzg_commitment_batch
    f.restype = C.c_int
    out = C.create_string_buffer(48 * n)
    st = C.create_string_buffer(n)
    rc = f(out, st, C.cast(t.data_ptr(), C.c_char_p), C.c_uint64(n), hip.sp)
    assert rc == 1 and [i for i, v in enumerate(st.raw) if v] == [650]
    assert all(out.raw[48 * i:48 * i + 48] == single[i % 3] for i in range(n) if i != 650)
    # cells + proofs from pinned memory
    g = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch
    g.restype = C.c_int
    m = 70
    cells = C.create_string_buffer(m * 128 * 2048)
    proofs = C.create_string_buffer(m * 128 * 48)
    assert g(cells, proofs, None, C.cast(t.data_ptr(), C.c_char_p), C.c_uint64(m), hip.sp) == 0
    exp = hip.compute_cells_and_kzg_proofs(base[2])
    assert cells.raw[2 * 262144:3 * 262144] == b"".join(exp[0]) and proofs.raw[68 * 6144:69 * 6144] == b"".join(exp[1])
    assert rt.hipHostFree(pinned) == 0


def test_device_pointer_batches_match_the_host_pointer_calls(hip):
    """The *_batch_device entry points (what bench.py times) on buffers from hipMalloc: commitments, and cells +
    proofs for 520 blobs -- 66,560 small MSMs in one launch, the size from which k_msm_small runs 8 lanes per
    vector -- against the host-pointer calls on the same blobs, with one non-canonical blob flagged (in d_status
    only: the device-pointer forms do not copy a verdict back, include/ckzg_hip.h)."""
    rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    H2D, D2H = 1, 2
    base = [rand_blob(123, i) for i in range(4)]
    n = 520
    blobs = [base[(i * 5 + i // 7) % 4] for i in range(n)]
    bad = bytearray(base[0])
    bad[64:96] = R.to_bytes(32, "big")
    blobs[333] = bytes(bad)
    raw = b"".join(blobs)

    def dmalloc(nbytes):
        q = C.c_void_p()
        assert rt.hipMalloc(C.byref(q), nbytes) == 0
        return q

    d_blobs, d_out, d_st = dmalloc(len(raw)), dmalloc(48 * n), dmalloc(n)
    d_cells, d_proofs = dmalloc(n * 262144), dmalloc(n * 6144)
    try:
        assert rt.hipMemcpy(d_blobs, C.cast(C.c_char_p(raw), C.c_void_p), len(raw), H2D) == 0
        f = hip.lib.ckzg_hip_blob_to_kzg_commitment_batch_device
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        rc = f(d_out, d_st, d_blobs, n, hip.sp)
        out, st = C.create_string_buffer(48 * n), C.create_string_buffer(n)
        assert rt.hipMemcpy(out, d_out, 48 * n, D2H) == 0 and rt.hipMemcpy(st, d_st, n, D2H) == 0
        assert rc == 0 and [i for i, v in enumerate(st.raw) if v] == [333]
        single = [hip.blob_to_kzg_commitment(b) for b in base]
        assert all(out.raw[48 * i:48 * i + 48] == single[(i * 5 + i // 7) % 4] for i in range(n) if i != 333)

        g = hip.lib.ckzg_hip_compute_cells_and_kzg_proofs_batch_device
        g.restype = C.c_int
        g.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        rc = g(d_cells, d_proofs, d_st, d_blobs, n, hip.sp)
        assert rt.hipMemcpy(st, d_st, n, D2H) == 0
        assert rc == 0 and [i for i, v in enumerate(st.raw) if v] == [333]
        proofs = C.create_string_buffer(n * 6144)
        assert rt.hipMemcpy(proofs, d_proofs, n * 6144, D2H) == 0
        exp = [hip.compute_cells_and_kzg_proofs(b) for b in base]
        for i in range(n):
            if i != 333:
                assert proofs.raw[6144 * i:6144 * (i + 1)] == b"".join(exp[(i * 5 + i // 7) % 4][1]), i
        cells = C.create_string_buffer(262144)
        for i in (0, 332, 334, n - 1):
            assert rt.hipMemcpy(cells, C.c_void_p(d_cells.value + 262144 * i), 262144, D2H) == 0
            assert cells.raw == b"".join(exp[(i * 5 + i // 7) % 4][0]), i
    finally:
        for q in (d_blobs, d_out, d_st, d_cells, d_proofs):
            rt.hipFree(q)


def _run_bench(extra_env, args):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    from watchdog import run_watched
    r = run_watched([sys.executable, os.path.join(root, "bench.py")] + args, env=env, timeout=280, name="bench_child")
    assert r.returncode == 0, r.stderr[-2000:]
    # stdout: ONE compact line (the driver's contract, < 4 KB); the full record is one stderr line
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(out) == 1 and len(out[0]) < 4000, (len(out), [len(x) for x in out])
    compact = json.loads(out[0])
    full = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith('{"bench_full_record"')][-1])["bench_full_record"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline"):
        assert compact[k] == full[k] or k == "roofline", k
    assert compact["roofline"]["frac"] == full["roofline"]["frac"]
    return full


def test_bench_two_ranks_control_flow_on_one_gpu():
    """`python bench.py --gpus 2` spawns two ranks itself (torch.distributed.run); here both ranks share the one
    GPU over gloo, which exercises the barrier / MAX-over-ranks timing, the whole-job host-pointer leg and the
    C-ABI fan-out leg (two table replicas standing in for two devices)."""
    line = _run_bench({"CKZG_BENCH_ONE_GPU": "1", "CKZG_BENCH_BACKEND": "gloo"},
                      ["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["parity_spot_check_vs_oracle"] is True
    assert line["value"] > 0 and line["host_pointer"]["value"] > 0
    assert line["c_abi_fan_out"]["devices"] == 2 and line["c_abi_fan_out"]["rc"] == 0


@pytest.mark.skipif("__import__('subprocess').run(['bash', '-c', 'rocm-smi --showid | grep -c \"Device ID\"'], capture_output=True, text=True).stdout.strip() in ('', '0', '1')",
                    reason="needs two visible GPUs")
def test_bench_two_ranks_on_two_gpus_over_rccl():
    line = _run_bench({}, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["parity_spot_check_vs_oracle"] is True
    assert line["c_abi_fan_out"]["devices"] == 2 and line["c_abi_fan_out"]["rc"] == 0

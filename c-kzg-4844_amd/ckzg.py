"""Python host-side mirror of the reference's binding surface (bindings/python/ckzg_wrap.c) over
the C-ABI of libckzg_hip.so (include/ckzg.h, include/ckzg_hip.h).

``Kzg`` is a plain ctypes binding for any shared library that speaks the ckzg.h calling
convention with an optional symbol prefix; the product is ``Kzg(HIP_SO)``.  Length checks happen
here (a wrong-sized blob/commitment/cell is an error before the C call, ckzg_wrap.c:48-73), a
non-zero C_KZG_RET raises ``KzgError``, outputs are returned as ``bytes``.  There is no CPU
fallback: if libckzg_hip.so is missing or no GPU is visible, construction fails.
"""
import ctypes as C
import os

BYTES_PER_BLOB = 131072
BYTES_PER_CELL = 2048
CELLS_PER_EXT_BLOB = 128
PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
# CKZG_HIP_SO: another build of the same library (tools/build_variant.sh), for A/B measurements
HIP_SO = os.path.abspath(os.environ["CKZG_HIP_SO"]) if os.environ.get("CKZG_HIP_SO") else os.path.join(PKG, "libckzg_hip.so")
TRUSTED_SETUP = os.path.join(PKG, "data", "trusted_setup.txt")


class KzgError(Exception):
    pass


class KZGSettings(C.Structure):
    # src/setup/settings.h:27-79 -- 8 pointers + 2 size_t = 80 bytes
    _fields_ = [(n, C.c_void_p) for n in (
        "roots_of_unity", "brp_roots_of_unity", "reverse_roots_of_unity", "g1_values_monomial",
        "g1_values_lagrange_brp", "g2_values_monomial", "x_ext_fft_columns", "tables")] + [
        ("wbits", C.c_size_t), ("scratch_size", C.c_size_t)]


def _check(cond, what):
    if not cond:
        raise KzgError("bad length: " + what)


class Kzg:
    def __init__(self, libpath=HIP_SO, prefix="", precompute=0, setup_path=TRUSTED_SETUP, options=None):
        if not os.path.exists(libpath):
            raise KzgError("%s is not built (run __graft_entry__.build())" % libpath)
        self.lib = C.CDLL(libpath)
        for k, v in (options or {}).items():
            f = self.lib.ckzg_hip_set_option
            f.restype = C.c_int
            f.argtypes = [C.c_char_p, C.c_int64]
            if f(k.encode(), v) != 0:
                raise KzgError("bad option %s=%r" % (k, v))
        self.prefix = prefix
        self.s = KZGSettings()
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        fp = libc.fopen(setup_path.encode(), b"r")
        if not fp:
            raise KzgError("cannot open " + setup_path)
        f = self._fn("load_trusted_setup_file")
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        ret = f(C.byref(self.s), fp, precompute)
        libc.fclose(fp)
        if ret != 0:
            raise KzgError("load_trusted_setup_file -> %d" % ret)
        self._loaded = True

    def _fn(self, name):
        f = getattr(self.lib, self.prefix + name)
        f.restype = C.c_int
        return f

    def close(self):
        if getattr(self, "_loaded", False):
            f = self._fn("free_trusted_setup")
            f.restype = None
            f.argtypes = [C.c_void_p]
            f(C.byref(self.s))
            self._loaded = False

    def _call(self, name, *args):
        ret = self._fn(name)(*args)
        if ret != 0:
            raise KzgError("%s -> C_KZG_RET %d" % (name, ret))

    @property
    def sp(self):
        return C.byref(self.s)

    # ---- EIP-4844 (src/eip4844/eip4844.h:43-81) ----
    def blob_to_kzg_commitment(self, blob):
        _check(len(blob) == BYTES_PER_BLOB, "blob")
        out = C.create_string_buffer(48)
        self._call("blob_to_kzg_commitment", out, bytes(blob), self.sp)
        return out.raw

    def compute_kzg_proof(self, blob, z):
        _check(len(blob) == BYTES_PER_BLOB, "blob")
        _check(len(z) == 32, "z")
        proof, y = C.create_string_buffer(48), C.create_string_buffer(32)
        self._call("compute_kzg_proof", proof, y, bytes(blob), bytes(z), self.sp)
        return proof.raw, y.raw

    def compute_blob_kzg_proof(self, blob, commitment):
        _check(len(blob) == BYTES_PER_BLOB, "blob")
        _check(len(commitment) == 48, "commitment")
        out = C.create_string_buffer(48)
        self._call("compute_blob_kzg_proof", out, bytes(blob), bytes(commitment), self.sp)
        return out.raw

    def verify_kzg_proof(self, commitment, z, y, proof):
        _check(len(commitment) == 48 and len(proof) == 48, "commitment/proof")
        _check(len(z) == 32 and len(y) == 32, "z/y")
        ok = C.c_bool(False)
        self._call("verify_kzg_proof", C.byref(ok), bytes(commitment), bytes(z), bytes(y),
                   bytes(proof), self.sp)
        return ok.value

    def verify_blob_kzg_proof(self, blob, commitment, proof):
        _check(len(blob) == BYTES_PER_BLOB, "blob")
        _check(len(commitment) == 48 and len(proof) == 48, "commitment/proof")
        ok = C.c_bool(False)
        self._call("verify_blob_kzg_proof", C.byref(ok), bytes(blob), bytes(commitment),
                   bytes(proof), self.sp)
        return ok.value

    def verify_blob_kzg_proof_batch(self, blobs, commitments, proofs):
        n = len(blobs)
        _check(len(commitments) == n and len(proofs) == n, "list lengths")
        for b in blobs:
            _check(len(b) == BYTES_PER_BLOB, "blob")
        for c in list(commitments) + list(proofs):
            _check(len(c) == 48, "commitment/proof")
        ok = C.c_bool(False)
        self._call("verify_blob_kzg_proof_batch", C.byref(ok), b"".join(blobs),
                   b"".join(commitments), b"".join(proofs), C.c_uint64(n), self.sp)
        return ok.value

    # ---- EIP-7594 (src/eip7594/eip7594.h:35-57) ----
    def compute_cells_and_kzg_proofs(self, blob, want_cells=True, want_proofs=True):
        _check(len(blob) == BYTES_PER_BLOB, "blob")
        cells = C.create_string_buffer(CELLS_PER_EXT_BLOB * BYTES_PER_CELL) if want_cells else None
        proofs = C.create_string_buffer(CELLS_PER_EXT_BLOB * 48) if want_proofs else None
        self._call("compute_cells_and_kzg_proofs", cells, proofs, bytes(blob), self.sp)
        craw = cells.raw if want_cells else b""   # .raw copies the whole buffer: take it once
        praw = proofs.raw if want_proofs else b""
        cl = [craw[i * BYTES_PER_CELL:(i + 1) * BYTES_PER_CELL]
              for i in range(CELLS_PER_EXT_BLOB)] if want_cells else None
        pl = [praw[i * 48:(i + 1) * 48] for i in range(CELLS_PER_EXT_BLOB)] if want_proofs else None
        return cl, pl

    def compute_cells(self, blob):
        # not a C symbol: bindings call compute_cells_and_kzg_proofs(cells, NULL, ..)
        # (bindings/go/main.go:411-431)
        return self.compute_cells_and_kzg_proofs(blob, True, False)[0]

    def recover_cells_and_kzg_proofs(self, cell_indices, cells):
        _check(len(cell_indices) == len(cells), "list lengths")
        for c in cells:
            _check(len(c) == BYTES_PER_CELL, "cell")
        n = len(cells)
        idx = (C.c_uint64 * max(n, 1))(*cell_indices)
        rc = C.create_string_buffer(CELLS_PER_EXT_BLOB * BYTES_PER_CELL)
        rp = C.create_string_buffer(CELLS_PER_EXT_BLOB * 48)
        self._call("recover_cells_and_kzg_proofs", rc, rp, idx, b"".join(cells), C.c_uint64(n), self.sp)
        craw, praw = rc.raw, rp.raw
        return ([craw[i * BYTES_PER_CELL:(i + 1) * BYTES_PER_CELL] for i in range(CELLS_PER_EXT_BLOB)],
                [praw[i * 48:(i + 1) * 48] for i in range(CELLS_PER_EXT_BLOB)])

    def recover_cells_and_kzg_proofs_batch(self, cell_indices, rows, want_cells=True, want_proofs=True):
        """rows: one list of cells per blob, every row holding the columns `cell_indices` (additive
        API ckzg_hip_recover_cells_and_kzg_proofs_batch).  Returns ([cells128 per row], [proofs128 per row])."""
        nb, nc = len(rows), len(cell_indices)
        for r in rows:
            _check(len(r) == nc, "list lengths")
            for c in r:
                _check(len(c) == BYTES_PER_CELL, "cell")
        idx = (C.c_uint64 * max(nc, 1))(*cell_indices)
        rc = C.create_string_buffer(max(nb, 1) * CELLS_PER_EXT_BLOB * BYTES_PER_CELL) if want_cells else None
        rp = C.create_string_buffer(max(nb, 1) * CELLS_PER_EXT_BLOB * 48) if want_proofs else None
        self._call("ckzg_hip_recover_cells_and_kzg_proofs_batch", rc, rp, None, idx,
                   b"".join(b"".join(r) for r in rows), C.c_uint64(nc), C.c_uint64(nb), self.sp)
        cs = BYTES_PER_CELL * CELLS_PER_EXT_BLOB
        craw = rc.raw if want_cells else b""      # .raw copies the whole buffer: take it once
        praw = rp.raw if want_proofs else b""
        out_c = [[craw[b * cs + i * BYTES_PER_CELL: b * cs + (i + 1) * BYTES_PER_CELL]
                  for i in range(CELLS_PER_EXT_BLOB)] for b in range(nb)] if want_cells else None
        out_p = [[praw[(b * CELLS_PER_EXT_BLOB + i) * 48:(b * CELLS_PER_EXT_BLOB + i + 1) * 48]
                  for i in range(CELLS_PER_EXT_BLOB)] for b in range(nb)] if want_proofs else None
        return out_c, out_p

    def verify_cell_kzg_proof_batch(self, commitments, cell_indices, cells, proofs):
        n = len(cells)
        _check(len(commitments) == n and len(cell_indices) == n and len(proofs) == n, "list lengths")
        for c in cells:
            _check(len(c) == BYTES_PER_CELL, "cell")
        for c in list(commitments) + list(proofs):
            _check(len(c) == 48, "commitment/proof")
        idx = (C.c_uint64 * max(n, 1))(*cell_indices)
        ok = C.c_bool(False)
        self._call("verify_cell_kzg_proof_batch", C.byref(ok), b"".join(commitments), idx,
                   b"".join(cells), b"".join(proofs), C.c_uint64(n), self.sp)
        return ok.value

    # ---- test-exposed internals (src/eip4844/eip4844.h:84, src/eip7594/eip7594.h:59-68) ----
    def compute_challenge(self, blob, commitment):
        _check(len(blob) == BYTES_PER_BLOB, "blob")
        _check(len(commitment) == 48, "commitment")
        g1 = C.create_string_buffer(144)
        self._call("bytes_to_kzg_commitment", g1, bytes(commitment))
        fr = C.create_string_buffer(32)
        f = getattr(self.lib, self.prefix + "compute_challenge")
        f.restype = None
        f(fr, bytes(blob), g1)
        out = C.create_string_buffer(32)
        g = getattr(self.lib, self.prefix + "bytes_from_bls_field")
        g.restype = None
        g(out, fr)
        return out.raw

    def compute_verify_cell_kzg_proof_batch_challenge(self, commitments, commitment_indices, cell_indices,
                                                      cells, proofs):
        n = len(cells)
        _check(len(commitment_indices) == n and len(cell_indices) == n and len(proofs) == n, "list lengths")
        for c in cells:
            _check(len(c) == BYTES_PER_CELL, "cell")
        for c in list(commitments) + list(proofs):
            _check(len(c) == 48, "commitment/proof")
        fr = C.create_string_buffer(32)
        ci = (C.c_uint64 * max(n, 1))(*commitment_indices)
        xi = (C.c_uint64 * max(n, 1))(*cell_indices)
        self._call("compute_verify_cell_kzg_proof_batch_challenge", fr, b"".join(commitments),
                   C.c_uint64(len(commitments)), ci, xi, b"".join(cells), b"".join(proofs), C.c_uint64(n))
        return self._fr_to_bytes(fr)

    def _fr_to_bytes(self, fr):
        out = C.create_string_buffer(32)
        g = getattr(self.lib, self.prefix + "bytes_from_bls_field")
        g.restype = None
        g(out, fr)
        return out.raw

"""The oracle speaks the same calling convention as ckzg.h with an ``okzg_`` prefix, except for
the two test-exposed internals whose helper symbols are named differently (og1_*/ofr_*)."""
import ctypes as C

from kzg_ctypes import Kzg, KzgError


class OracleKzg(Kzg):
    def __init__(self, libpath, precompute=0, **kw):
        super().__init__(libpath, "okzg_", precompute=precompute, **kw)

    def _fr_to_bytes(self, fr):
        out = C.create_string_buffer(32)
        self.lib.ofr_to_bytes(out, fr)
        return out.raw

    def compute_challenge(self, blob, commitment):
        if len(blob) != 131072 or len(commitment) != 48:
            raise KzgError("bad length")
        aff = C.create_string_buffer(96)
        if self.lib.og1_uncompress(aff, bytes(commitment)) != 0:
            raise KzgError("bad commitment")
        g1 = C.create_string_buffer(144)
        self.lib.og1_from_affine(g1, aff)
        fr = C.create_string_buffer(32)
        f = self.lib.okzg_compute_challenge
        f.restype = None
        f(fr, bytes(blob), g1)
        return self._fr_to_bytes(fr)

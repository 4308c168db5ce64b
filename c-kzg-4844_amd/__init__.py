"""c-kzg-4844_amd: MI355X-native G1-MSM / Fr-NTT hot path behind the c-kzg-4844 C-ABI.

The directory name contains '-', so import it by path (see __graft_entry__.load_package()).
"""
from .ckzg import Kzg, KzgError, KZGSettings, HIP_SO, TRUSTED_SETUP  # noqa: F401
from . import fanout  # noqa: F401

"""MI355X-native Plonky2 proving backend for the 0xPARC/plonky2-aes gadget circuits.

The directory name carries a hyphen (it mirrors the reference's name), so import it with
`importlib` -- see `load_package()` in `__graft_entry__.py` -- under the module name `plonky2_aes_amd`.
"""
from .api import (  # noqa: F401
    AesGcmTarget,
    CircuitBuilder,
    CircuitData,
    ECGFP5SecretKey,
    P2Error,
    PartialWitness,
    PoseidonEncryptTarget,
    ProveError,
    ecgfp5,
    lib,
    lib_path,
    native,
    poseidon_native,
)

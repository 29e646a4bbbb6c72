"""Arch plugin resolved by `config.network == 'RefVSR'` (cf. models/SRNet.py:20-21 in the reference)."""
from refvsr_amd.model import Network  # noqa: F401

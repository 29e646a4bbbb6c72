"""Arch plugin resolved by `config.network == 'RefVSR_IR'` (cf. models/SRNet.py:20-21, models/archs/RefVSR_IR.py)."""
from refvsr_amd.engine_ir import EngineIR, WeightsIR
from refvsr_amd.model import Network as _Network


class Network(_Network):
    """models/archs/RefVSR_IR.py:Network (inference path) on the HIP engine."""
    _engine_cls = EngineIR
    _weights_cls = WeightsIR
    _family = 'RefVSR_IR'

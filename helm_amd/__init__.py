"""helm_amd — MI355X-native encrypted-circuit evaluator (HELM hot path).

Python mirror of the reference's interface for the gates-mode hot path
(reference src/circuit.rs, src/gates.rs, src/verilog_parser.rs) over two
in-tree native libraries; see include/helm_hip.h for the drop-in C ABI.
"""
from . import _native  # noqa: F401  (fails loudly if the native libraries are missing)
from .engine import ClientKey, ServerKey, DeviceWires, Program, named_params  # noqa: F401
from ._native import HelmError, Params  # noqa: F401

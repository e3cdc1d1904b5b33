"""helm_amd — MI355X-native encrypted-circuit evaluator (HELM hot path).

Python mirror of the reference's interface for the gates-mode hot path
(reference src/circuit.rs, src/gates.rs, src/verilog_parser.rs) over two
in-tree native libraries; see include/helm_hip.h for the drop-in C ABI.
"""
from . import _native  # noqa: F401  (fails loudly if the native libraries are missing)
from .engine import ClientKey, ServerKey, DeviceWires, Program, named_params  # noqa: F401
from ._native import HelmError, Params, SiParams  # noqa: F401
from .shortint import SiClientKey, SiServerKey, SiWires, si_named_params  # noqa: F401
from .wopbs import WopClientKey, WopServerKey, wop_named_params  # noqa: F401
from ._native import WopParams  # noqa: F401
from . import verilog_parser, circuit, gates, netlists, comm  # noqa: F401,E402
from .circuit import (Circuit, GateCircuit, EvalCircuit, EncWireMap, LutCircuit, ArithCircuit,  # noqa: F401,E402
                      SiEncWireMap)
from .gates import PtxtType, GateType, Gate  # noqa: F401,E402


def gen_keys(name="boolean_default", seed=None, device=0):
    """tfhe::boolean::gen_keys() (reference src/bin/helm.rs:241) -> (client_key, server_key)."""
    ck = ClientKey.generate(name, seed)
    return ck, ServerKey(ck, device=device)


def gen_keys_shortint(name="shortint_m2c2", seed=None, device=0):
    """tfhe::shortint::gen_keys(PARAM_...) (reference src/bin/helm.rs:301) -> (client_key, server_key)."""
    ck = SiClientKey.generate(name, seed)
    return ck, SiServerKey(ck, device=device)


def gen_keys_wopbs(client_key, server_key, name="wopbs_m2c2", seed=None):
    """tfhe::shortint::wopbs::WopbsKey::new_wopbs_key(cks, sks, params) next to a shortint key pair
    (what reference src/gates.rs:787-815 takes as wk_si / wk) -> (wop_client_key, wop_server_key)."""
    wk = WopClientKey.generate(client_key, name, seed)
    return wk, WopServerKey(server_key, wk)

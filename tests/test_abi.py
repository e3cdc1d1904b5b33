"""The C-ABI libraries load and export every symbol include/*.h declares; on a GPU-less
box the engine refuses to create a context (no CPU fallback) and argument validation
works without touching a device.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helm_amd
from helm_amd import _native as nv
from helm_amd import _host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix, exclude=None):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))
    return [n for n in names if not (exclude and n.startswith(exclude))]


@pytest.mark.parametrize("header,prefix,lib,table,exclude", [
    ("helm_hip.h", "helm_hip_", nv.hip, nv.HIP_API, None),
    ("helm_client.h", "helm_client_", nv.host, nv.CLIENT_API, None),
    ("helm_client.h", "helm_si_client_", nv.host, nv.SI_CLIENT_API, None),
    ("helm_shortint.h", "helm_si_", nv.hip, nv.SI_API, None),
    ("helm_host.h", "helm_host_", nv.host, _host.HOST_API, None),
    ("helm_client.h", "helm_keys_", nv.host, nv.KEYS_API, None),
    ("helm_wopbs.h", "helm_wop_", nv.hip, nv.WOP_API, "helm_wop_client_"),
    ("helm_wopbs.h", "helm_wop_client_", nv.host, nv.WOP_CLIENT_API, None),
    ("helm_comm.h", "helm_comm_", nv.hip, nv.COMM_API, None),
])
def test_every_declared_symbol_is_exported_and_bound(header, prefix, lib, table, exclude):
    names = _declared(header, prefix, exclude)
    assert len(names) >= 4
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"
        assert n in table, f"{n} declared in include/{header} but not bound in the Python layer"
    for n in table:
        assert n in names, f"{n} bound but not declared in include/{header}"


def test_no_torch_types_in_the_abi():
    for h in ("helm_hip.h", "helm_shortint.h", "helm_client.h", "helm_host.h", "helm_wopbs.h", "helm_comm.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        assert "torch" not in text.lower() and "at::" not in text and "#include <hip" not in text


def test_product_package_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "helm_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liborc" not in src, f
                assert "tfhe_oracle" not in src.replace("oracle/tfhe_oracle.c header", ""), f
                assert "shortint_oracle" not in src and "orc64_" not in src, f
                assert "wopbs_oracle" not in src and "orcw_" not in src, f


def test_parameter_validation_needs_no_device():
    p, _, _ = helm_amd.named_params("boolean_default")
    h = nv.vp()
    bad = helm_amd.Params.from_buffer_copy(p)
    bad.torus_bits = 64
    assert nv.hip.helm_hip_ctx_create(0, C.byref(bad), C.byref(h)) == -1
    assert b"torus_bits" in nv.hip.helm_hip_last_error()
    bad = helm_amd.Params.from_buffer_copy(p)
    bad.N = 2048
    assert nv.hip.helm_hip_ctx_create(0, C.byref(bad), C.byref(h)) == -1
    bad = helm_amd.Params.from_buffer_copy(p)
    bad.pbs_logB = 8  # (k+1) l N B/2 2^31 would exceed the NTT prime's exact range
    assert nv.hip.helm_hip_ctx_create(0, C.byref(bad), C.byref(h)) == -1
    assert b"capacity" in nv.hip.helm_hip_last_error()
    assert nv.hip.helm_hip_ctx_create(0, None, C.byref(h)) == -1


def test_shortint_parameter_validation_needs_no_device():
    p, _, _ = helm_amd.si_named_params("shortint_m2c2")
    assert p.as_tuple() == (742, 1, 2048, 1, 23, 5, 3, 4, 4, 0)
    mb, _, _ = helm_amd.si_named_params("shortint_m2c2_multibit3")  # the arithmetic-mode set of helm.rs:83
    assert mb.as_tuple() == (888, 1, 2048, 1, 21, 3, 4, 4, 4, 3)
    h = nv.vp()
    for field, value, msg in (("k", 2, b"unsupported"), ("N", 4096, b"unsupported"), ("pbs_logB", 31, b"decomposition"),
                              ("message_modulus", 3, b"power of two"), ("ks_logB", 9, b"keyswitch"),
                              ("grouping_factor", 4, b"grouping_factor"), ("grouping_factor", 3, b"grouping_factor")):
        bad = helm_amd.SiParams.from_buffer_copy(p)
        setattr(bad, field, value)
        assert nv.hip.helm_si_ctx_create(0, C.byref(bad), C.byref(h)) == -1, field
        assert msg in nv.hip.helm_hip_last_error(), (field, nv.hip.helm_hip_last_error())
    bad = helm_amd.SiParams.from_buffer_copy(p)
    bad.pbs_l, bad.pbs_logB = 1, 24  # 2 * 2048 * 2^23 * 2^63 = 2^98 exceeds the two-prime CRT range (2^97.5); 25 and up: "decomposition"
    assert nv.hip.helm_si_ctx_create(0, C.byref(bad), C.byref(h)) == -1
    assert b"capacity" in nv.hip.helm_hip_last_error()


def test_shortint_client_roundtrip():
    for name in ("si_toy_512", "shortint_m2c2"):
        ck = helm_amd.SiClientKey.generate(name, seed=4)
        vals = np.arange(ck.t, dtype=np.uint64)
        ct = ck.encrypt(vals)
        assert ct.shape == (ck.t, ck.dim + 1)
        assert np.array_equal(ck.decrypt_message_and_carry(ct), vals)
        assert np.array_equal(ck.decrypt(ct), vals % ck.params.message_modulus)
        assert ck.bsk.size == ck.params.n * ck.params.pbs_l * 4 * ck.params.N
        assert ck.ksk.size == ck.dim * ck.params.ks_l * (ck.params.n + 1)
    # multi-bit key: 2^g GGSWs per group of g mask words (include/helm_shortint.h)
    ck = helm_amd.SiClientKey.generate("si_toy_2048_mb3", seed=4)
    assert ck.bsk.size == (ck.params.n // 3) * 8 * ck.params.pbs_l * 4 * ck.params.N


def test_communicator_binds_rccl_lazily_and_validates_without_a_device(have_gpu):
    """include/helm_comm.h: libhelm_hip.so has no link-time dependency on librccl (dlopen at first use: a process that
    already holds PyTorch's copy keeps it), and argument validation needs no device."""
    import subprocess
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "helm_amd", "csrc", "libhelm_hip.so")],
                            capture_output=True, text=True).stdout
    assert "librccl" not in needed
    h = nv.vp()
    ident = np.zeros(128, dtype=np.uint8)
    assert nv.hip.helm_comm_create(0, None, 0, 1, C.byref(h)) == -1
    assert nv.hip.helm_comm_create(0, nv.as_u8p(ident), 2, 2, C.byref(h)) == -1     # rank outside the world
    assert b"rank" in nv.hip.helm_hip_last_error()
    assert nv.hip.helm_comm_destroy(None) == 0
    assert nv.hip.helm_comm_all_gather(None, None, None, 16, None) == -1
    assert nv.hip.helm_comm_available() in (0, 1)
    if not have_gpu and nv.hip.helm_comm_available():
        assert nv.hip.helm_comm_create(0, nv.as_u8p(ident), 0, 1, C.byref(h)) == -2   # HELM_ERR_NO_DEVICE: no fallback
    # the transport form (a host-supplied all-gather instead of RCCL) validates the same way and needs no RCCL at all
    cb = nv.COMM_ALL_GATHER_FN(lambda *_: 0)
    assert nv.hip.helm_comm_create_with_transport(0, 0, 1, nv.COMM_ALL_GATHER_FN(0), None, C.byref(h)) == -1   # no callback
    assert nv.hip.helm_comm_create_with_transport(0, 3, 2, cb, None, C.byref(h)) == -1                       # rank outside the world
    if not have_gpu:
        assert nv.hip.helm_comm_create_with_transport(0, 0, 1, cb, None, C.byref(h)) == -2


def test_no_cpu_fallback(have_gpu):
    if have_gpu:
        pytest.skip("GPU present: covered by the gpu tests")
    ck = helm_amd.ClientKey.generate("toy", seed=1)
    with pytest.raises(helm_amd.HelmError, match="no HIP device|no CPU fallback|-2"):
        helm_amd.ServerKey(ck)
    with pytest.raises(helm_amd.HelmError, match="no HIP device|no CPU fallback|-2"):
        helm_amd.SiServerKey(helm_amd.SiClientKey.generate("si_toy_512", seed=1))


def test_client_roundtrip_and_noise():
    for name in ("toy", "boolean_default"):
        ck = helm_amd.ClientKey.generate(name, seed=9)
        bits = np.random.default_rng(0).integers(0, 2, size=64).astype(bool)
        ct = ck.encrypt(bits)
        assert ct.shape == (64, ck.params.n + 1)
        assert np.array_equal(ck.decrypt(ct), bits)
        ph = ck.phase(ct).astype(np.int64)
        want = np.where(bits, 1 << 29, 7 << 29)
        err = (ph - want + 2**31) % 2**32 - 2**31
        _, lwe_std, _ = helm_amd.named_params(name)
        assert np.abs(err).max() < 8 * lwe_std * 2**32 + 2
        assert ck.bsk.size == ck.params.n * ck.params.pbs_l * (ck.params.k + 1) ** 2 * ck.params.N
        assert ck.ksk.size == ck.params.k * ck.params.N * ck.params.ks_l * (ck.params.n + 1)


def test_named_params_match_the_reference():
    p, lwe, glwe = helm_amd.named_params("helm_cuda")  # reference src/bin/helm.rs:141-146
    assert (p.n, p.k, p.N, p.pbs_l, p.pbs_logB, p.ks_l, p.ks_logB) == (512, 1, 1024, 3, 7, 8, 2)
    assert lwe == glwe == 0.00000002980232238769531
    with pytest.raises(helm_amd.HelmError):
        helm_amd.named_params("nope")

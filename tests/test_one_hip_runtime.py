"""One HIP runtime per process, whatever the import order (the boundary of reference src/circuit.rs:35-58 is reached
through this loader).

libhelm_hip.so NEEDs `libamdhip64.so.7` (RUNPATH: the ROCm installation); PyTorch's libtorch_hip.so NEEDs
`libamdhip64.so` and ships its own copy.  `import helm_amd; import torch` used to map BOTH, and a torch stream handed to
helm_hip_set_stream then crossed from one runtime into the other (round 5: `std::bad_variant_access`, core dumped).
helm_amd/_native.py now binds the process to one copy before libhelm_hip.so is opened; where a handle of the caller's
crosses the ABI the library checks (helm_hip_runtime_copies) and fails with both paths instead of aborting.

Every case runs in a FRESH interpreter: what is mapped depends on everything the process has imported before."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, os, sys
sys.path.insert(0, {root!r})
order = sys.argv[1]
def mapped():
    out = []
    for line in open("/proc/self/maps"):
        f = line.split(None, 5)
        if len(f) == 6 and os.path.basename(f[5].strip()).startswith("libamdhip64.so"):
            p = os.path.realpath(f[5].strip())
            if p not in out:
                out.append(p)
    return out
res = {{}}
if order == "helm_first":
    import helm_amd
    res["after_first"] = mapped()
    import torch
elif order == "torch_first":
    import torch
    res["after_first"] = mapped()
    import helm_amd
else:
    import helm_amd
    res["after_first"] = mapped()
from helm_amd import _native as nv
import ctypes as C
res["mapped"] = mapped()
buf = C.create_string_buffer(4096)
res["native_count"] = nv.hip.helm_hip_runtime_copies(buf, len(buf))
res["native_paths"] = buf.value.decode().split("\n")
res["bound_to"] = nv.hip_runtime
try:
    nv.require_one_hip_runtime("test")
    res["python_guard"] = None
except nv.HelmError as e:
    res["python_guard"] = str(e)
# a handle-crossing entry point that needs no device to refuse: the host's all-gather is handed our device pointers
cb = nv.COMM_ALL_GATHER_FN(lambda *a: 0)
h = nv.vp()
res["transport_rc"] = nv.hip.helm_comm_create_with_transport(0, 0, 1, cb, None, C.byref(h))
res["transport_err"] = nv.hip.helm_hip_last_error().decode()
print("RESULT " + json.dumps(res))
'''


def _child(order, env=None):
    e = dict(os.environ)
    e.pop("HELM_HIP_RUNTIME", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), order], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _torch_runtime():
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None:
        return None
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    return os.path.realpath(p) if os.path.exists(p) else None


@pytest.mark.parametrize("order", ["helm_first", "torch_first"])
def test_exactly_one_hip_runtime_in_both_import_orders(order):
    res = _child(order)
    assert len(res["after_first"]) == 1, res        # (helm_amd alone already decides for the copy torch will look for)
    assert len(res["mapped"]) == 1, res
    assert res["mapped"] == res["after_first"]       # the second import reused the first one's runtime
    assert res["native_count"] == 1 and res["native_paths"] == res["mapped"]
    assert res["python_guard"] is None
    tr = _torch_runtime()
    if tr is not None:
        assert res["mapped"] == [tr]                 # both orders end on the SAME copy: the wheel's
        assert res["bound_to"] == tr
    # one runtime: the transport form is not refused for that reason (without a GPU it fails on the device instead)
    assert res["transport_rc"] in (0, -2, -3) and "HIP runtimes" not in res["transport_err"]


def test_helm_alone_binds_to_the_copy_torch_would_load():
    res = _child("helm_only")
    assert len(res["mapped"]) == 1
    tr = _torch_runtime()
    if tr is not None:
        assert res["mapped"] == [tr]


@pytest.mark.skipif(_torch_runtime() is None, reason="needs PyTorch's bundled HIP runtime to build the two-runtime process")
def test_two_runtimes_are_refused_with_both_paths_never_an_abort():
    # HELM_HIP_RUNTIME=system keeps the RUNPATH's copy; torch then brings its own: the process round 5 aborted in
    res = _child("helm_first", env={"HELM_HIP_RUNTIME": "system"})
    if len(res["mapped"]) < 2:
        pytest.skip("the ROCm installation's libamdhip64 and the wheel's are the same file here")
    assert res["native_count"] == 2 and sorted(res["native_paths"]) == sorted(res["mapped"])
    for p in res["mapped"]:
        assert p in res["python_guard"]              # HelmError names every copy
        assert p in res["transport_err"]             # ... and so does the library's own check
    assert res["transport_rc"] == -4                 # HELM_ERR_STATE, before any device pointer was handed out
    assert "One HIP runtime per process" in res["python_guard"]


def test_explicit_runtime_path_is_honoured_and_a_wrong_one_is_an_import_error():
    tr = _torch_runtime()
    if tr is not None:
        res = _child("helm_only", env={"HELM_HIP_RUNTIME": tr})
        assert res["mapped"] == [tr] and res["bound_to"] == tr
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); import helm_amd"], capture_output=True, text=True,
                       env=dict(os.environ, HELM_HIP_RUNTIME="/nonexistent/libamdhip64.so"), timeout=600)
    assert r.returncode != 0 and "HELM_HIP_RUNTIME" in r.stderr

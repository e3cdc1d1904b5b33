"""The C stand-in for the Rust shim (rust/helm-hip cannot be compiled here: no rustc): tests/c/shim_sequence.c
makes the shim's calls in the shim's order against the two shared libraries.  CPU: it compiles and links against
include/*.h with nothing but the C ABI, and without a GPU it fails loudly at helm_hip_ctx_create (no fallback).
GPU: it runs the 2-bit adder known answer of reference tests/circuit_test.rs:17-45 at the full parameter set."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "helm_amd", "csrc")


def _build(tmp_path, name="shim_sequence"):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe, "-L", CSRC, "-lhelm_host", "-lhelm_hip",
                           "-lpthread", f"-Wl,-rpath,{CSRC}"])
    return exe


def test_shim_sequence_builds_against_the_c_abi_and_refuses_to_run_without_a_gpu(tmp_path):
    from helm_amd import _native
    exe = _build(tmp_path)
    if _native.hip.helm_hip_device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe, "toy"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "helm_hip_ctx_create failed" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("params", ["boolean_default", "helm_cuda"])
def test_shim_sequence_on_the_gpu(tmp_path, params):
    exe = _build(tmp_path)
    r = subprocess.run([exe, params], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_shortint_shim_sequence_builds_and_refuses_to_run_without_a_gpu(tmp_path):
    """tests/c/shim_sequence_si.c = the calls of rust/helm-hip's HipLutCircuit and HipArithCircuit (reference
    src/circuit.rs:969-1111, 1113-1483) over include/helm_shortint.h + include/helm_host.h."""
    from helm_amd import _native
    exe = _build(tmp_path, "shim_sequence_si")
    if _native.hip.helm_hip_device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe, "si_toy_1024"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "helm_si_ctx_create failed" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("params", ["shortint_m2c2", "shortint_m2c2_multibit3"])
def test_shortint_shim_sequence_on_the_gpu(tmp_path, params):
    """8-bit LUT-3-1 adder + READY latch, then the FheUint16 known answers of tests/gates_test.rs:127-310, through the C
    ABI alone, under the LUT-mode test set and the reference's arithmetic-mode set (helm.rs:83)."""
    exe = _build(tmp_path, "shim_sequence_si")
    r = subprocess.run([exe, params], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_communicator_sequence_builds_and_refuses_to_run_without_a_gpu(tmp_path):
    """tests/c/shim_sequence_comm.c = the calls of rust/helm-hip/src/multi_gpu.rs (HipComm, shard_over, the sharded
    evaluate_encrypted) over include/helm_comm.h + include/helm_hip.h, in a process without Python or torch."""
    from helm_amd import _native
    exe = _build(tmp_path, "shim_sequence_comm")
    if _native.hip.helm_hip_device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe, "toy"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "helm_hip_ctx_create failed" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("params", ["boolean_default", "toy_k2"])
def test_communicator_sequence_on_the_gpu(tmp_path, params):
    """A process that holds only libhelm_hip.so and libhelm_host.so creates the library's RCCL communicator (RCCL comes
    from the loader's search path / /opt/rocm: nothing has loaded one before), packs the 2-bit adder for the world size
    and sends every launch through ncclAllGather inside the library: same wire table as helm_hip_program_run, the
    reference's known answer (tests/circuit_test.rs:17-45) decrypts."""
    exe = _build(tmp_path, "shim_sequence_comm")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe, params], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ok:" in r.stdout, r.stdout + r.stderr


def test_rank_thread_sequence_builds_and_refuses_to_run_without_a_gpu(tmp_path):
    """tests/c/shim_sequence_threads.c: a C host that drives its ranks from pthreads over helm_comm_create_in_process."""
    from helm_amd import _native
    exe = _build(tmp_path, "shim_sequence_threads")
    if _native.hip.helm_hip_device_count() > 0:
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([exe, "toy"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "helm_hip_ctx_create failed" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("params,world,overlap", [("toy_k2", 8, 0), ("toy_k2", 5, 1), ("boolean_default", 3, 0)])
def test_rank_thread_sequence_on_the_gpu(tmp_path, params, world, overlap):
    """Eight (five, three) rank threads of ONE C process, one engine context each: helm_hip_program_run_sharded_comm over the
    library's in-process group - launches of 1..4 gates cut over up to eight ranks (most chunks empty and padded) - leaves
    every rank the wire table of helm_hip_program_run; the reference's known answer (tests/circuit_test.rs:17-45) decrypts."""
    exe = _build(tmp_path, "shim_sequence_threads")
    r = subprocess.run([exe, params, str(world), str(overlap)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr

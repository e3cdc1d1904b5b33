"""bench.py's N > 1 contract: `python3 bench.py --gpus N` starts its own workers from a parent that has not touched
the GPU, prints ONE JSON line whatever happens, and a failing / hung worker group shows as a non-zero exit code with
the error in the line.  The sharded unit is the netlist level, reference
src/circuit.rs:531."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


def test_launcher_imports_nothing_that_touches_the_gpu():
    """The launcher path must stay on the standard library: no torch / helm_amd / numpy at module level."""
    import ast
    tree = ast.parse(open(BENCH).read())
    top = set()
    for node in tree.body:
        if isinstance(node, ast.Import):
            top |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom):
            top.add(node.module.split(".")[0])
    assert top <= {"argparse", "json", "os", "sys", "time"}, top
    # and inside launch_workers only standard-library modules
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "launch_workers")
    inner = {a.name.split(".")[0] for n in ast.walk(fn) if isinstance(n, ast.Import) for a in n.names}
    assert inner <= {"signal", "socket", "subprocess", "tempfile", "threading"}, inner
    assert "execv" not in open(BENCH).read().replace("never an exec", "")


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="failure path: needs a box without a GPU")
def test_failing_workers_give_one_line_with_error_and_nonzero_rc():
    rc, lines, err = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--blocks", "1"])
    assert rc != 0
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["metric"].startswith("encrypted gate-bootstraps/sec")
    assert line["n_gpus"] == 2 and line["value"] is None
    assert "error" in line and "rank" in line["error"]


def test_hung_worker_group_is_stopped_and_reported():
    """A group that outlives --launch-timeout is stopped (its own process group only) and reported: rc 3, one line."""
    rc, lines, err = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--blocks", "1", "--launch-timeout", "0.2"])
    assert rc == 3
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert "still running" in line["error"] and line["value"] is None


@pytest.mark.gpu
def test_rehearsal_two_ranks_on_one_gpu_reports_every_named_result():
    """HELM_BENCH_REHEARSE=1: both ranks on cuda:0, the library's communicator over a host transport - the whole N > 1 path of bench.py, launcher
    included: sharded_weak (the headline: the per-level shard at fixed work per GPU), strong and weak each under its own name, rc 0."""
    rc, lines, err = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "4", "--side-steps", "1"],
                               {"HELM_BENCH_REHEARSE": "1"}, timeout=1100)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert "error" not in line, line.get("error")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["config"]["blocks_total"] == 8 and "ONE job of 8 blocks (4 per GPU)" in line["config"]["workload"]
    # the data path is the real run's (the library's communicator: helm_hip_program_run_sharded_comm), carried by a host
    # transport in the rehearsal and labelled as such
    rr = line["rccl_ranks"]
    assert rr["world_size"] == 2 and rr["rccl_version"] == 0 and "HOST TRANSPORT" in rr["communicator"]
    assert rr["one_process_per_gpu"] is False and rr["collectives_issued_by_rank_0"] > 0
    assert line["config"]["sharded_launches"] > 0 and line["config"]["exchanged_MB_per_step"] > 0
    for kind in ("strong", "weak", "sharded_weak"):
        assert line[kind]["value"] > 0, kind
    assert line["sharded_weak"]["value"] == line["value"] and line["sharded_weak"]["steps"] == 2 and line["strong"]["steps"] == 1
    assert line["weak"]["exchanged_MB_per_step"] == 0 and line["sharded_weak"]["sharded_launches"] > 0
    assert line["sharded_weak"]["bootstraps_per_step"] == 2 * line["strong"]["bootstraps_per_step"]


@pytest.mark.gpu
def test_rccl_unavailable_on_one_rank_fails_loudly_unless_the_host_fallback_is_allowed():
    """A rank that cannot get its RCCL communicator (here: an injected pre-check failure on rank 1; nobody enters
    ncclCommInitRank) must not turn into a run over gloo that looks like a result: value null, the error, rc != 0 -
    every rank leaves, nobody hangs.  --allow-host-fallback: the run goes through over the host transport, rc 0, and the
    line says what it is."""
    env = {"HELM_BENCH_REHEARSE": "1", "HELM_BENCH_INJECT_COMM_FAILURE": "precheck:1"}
    args = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--blocks", "1", "--no-side-legs", "--leg-timeout", "300"]
    rc, lines, err = run_bench(args, env, timeout=600)
    assert rc != 0, err[-2000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["value"] is None and "RCCL communicator could not be created" in line["error"] and "injected failure: precheck" in line["error"]
    rc, lines, err = run_bench(args + ["--allow-host-fallback"], env, timeout=900)
    assert rc == 0, err[-3000:]
    line = json.loads(lines[0])
    assert line["value"] > 0 and "injected failure: precheck" in line["rccl_error"] and line["not_a_measurement_of_the_rccl_path"] is True
    assert line["rccl_ranks"]["rccl_version"] == 0 and "--allow-host-fallback" in line["rccl_ranks"]["communicator"]


@pytest.mark.gpu
def test_rehearsal_with_rank_threads_reports_every_named_result():
    """HELM_BENCH_REHEARSE=threads: the ranks are threads of ONE process (the box allows six GPU processes, the scaling run
    has eight ranks - profiles/r05/rehearse_n8.json is this command at --gpus 8): the library's in-process communicator
    (helm_comm_create_in_process) under the real run's programs, sharded passes, three named runs and decryption checks."""
    rc, lines, err = run_bench(["--gpus", "3", "--steps", "1", "--warmup", "0", "--blocks", "3", "--side-steps", "1", "--scaling", "strong"],
                               {"HELM_BENCH_REHEARSE": "threads"}, timeout=900)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert "error" not in line, line.get("error")
    assert line["n_gpus"] == 3 and line["scaling"] == "strong" and line["value"] == line["strong"]["value"] > 0   # --scaling strong: the fixed job is the headline
    rr = line["rccl_ranks"]
    assert rr["world_size"] == 3 and rr["rccl_version"] == 0 and "rank threads" in rr["communicator"] and rr["one_process_per_gpu"] is False
    assert "THREADS" in line["data"]
    for kind in ("strong", "weak", "sharded_weak"):
        assert line[kind]["value"] > 0 and "every rank" in line[kind]["decrypt_check"], kind
    assert line["sharded_weak"]["bootstraps_per_step"] == 3 * line["strong"]["bootstraps_per_step"]

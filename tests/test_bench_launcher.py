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
    # the record explains itself: which run the headline is, the N = 1 value of the same job shape measured inside this run
    # (rank 0 alone, 4 blocks unsharded) on every row, the plan that was printed before the runs started, the rank layout
    assert line["headline_kind"] == "sharded_weak"
    assert line["solo"]["bootstraps_per_step"] == line["strong"]["bootstraps_per_step"] and line["solo"]["exchanged_MB_per_step"] == 0
    for kind in ("strong", "weak", "sharded_weak"):
        assert line[kind]["n1_same_job"]["value"] == line["n1_same_job_value"] == line["solo"]["value"] > 0
        assert line[kind]["linear_would_be"] == 2
        assert abs(line[kind]["speedup_over_n1"] - line[kind]["value"] / line["solo"]["value"]) < 1e-3
    assert set(line["planned_wall_s"]) == {"sharded_weak", "solo", "strong", "weak"} and line["wall_s_total"] > 0
    assert "plan: sharded_weak (headline)" in err and "plan:" in err.split("plan: sharded_weak (headline)")[1]
    assert [d["rank"] for d in rr["devices"]] == [0, 1] and "rehearsal" in rr["checked"]


def test_rank_layout_is_checked_not_assumed():
    """bench.check_rank_devices: an N > 1 number needs N ranks on N different devices under an RCCL communicator of N ranks;
    anything else is a list of problems (the worker raises it: value null, rc != 0)."""
    sys.path.insert(0, ROOT)
    import bench
    def rec(rank, pci, world=2, comm_rank=None, rccl=22705):
        return {"rank": rank, "local_rank": rank, "pci": pci, "uuid": "u" + pci, "comm_world": world,
                "comm_rank": rank if comm_rank is None else comm_rank, "rccl_version": rccl}
    good = [rec(0, "0000:05:00"), rec(1, "0000:15:00")]
    assert bench.check_rank_devices(good, 2, 2) == []
    three = [rec(0, "0000:05:00"), rec(1, "0000:15:00"), rec(2, "0000:05:00")]
    assert any("ranks 0 and 2 share one device" in p for p in bench.check_rank_devices(three, 3, 3))
    same_index = [rec(0, "0000:05:00"), dict(rec(1, "0000:15:00"), local_rank=0)]
    assert any("both use device index 0" in p for p in bench.check_rank_devices(same_index, 2, 2))
    # a platform that reports ONE identity for every device says nothing: distinct device indices stand, no false alarm
    assert bench.check_rank_devices([rec(0, "0000:00:00"), rec(1, "0000:00:00")], 2, 2) == []
    assert bench.check_rank_devices([rec(0, "0000:05:00"), dict(rec(1, "0000:05:00"), local_rank=0)], 2, 2, rehearsal=True) == []
    assert any("world size 2 != --gpus 8" in p for p in bench.check_rank_devices(good, 2, 8))
    assert any("reports world size 1" in p for p in bench.check_rank_devices([rec(0, "a", world=1), rec(1, "b")], 2, 2))
    assert any("reports rank 0" in p for p in bench.check_rank_devices([rec(0, "a"), rec(1, "b", comm_rank=0)], 2, 2))
    assert any("not RCCL's" in p for p in bench.check_rank_devices([rec(0, "a"), rec(1, "b", rccl=0)], 2, 2))
    assert bench.check_rank_devices([rec(0, "a", rccl=0), rec(1, "b", rccl=0)], 2, 2, host_fallback=True) == []
    assert any("1 of 2 ranks" in p for p in bench.check_rank_devices([rec(0, "a"), None], 2, 2))


def test_the_plan_of_an_eight_gpu_run_fits_the_budget():
    """The default N = 8 run (driver: --steps 20 --warmup 1 at most) is planned under --wall-budget = 450 s with the figures
    one GPU holds; the arithmetic of Bench.plan without a GPU."""
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "8"])
    assert args.wall_budget == 450.0
    b = bench.Bench.__new__(bench.Bench)
    b.args, b.world, b.pbs_per_block = args, 8, 32464
    head = b.plan("sharded_weak", 2, 1, 134e3, 0.05)
    side = [b.plan(k, args.side_steps, 1, 134e3, 0.05) for k in ("solo", "strong", "weak")]
    assert 7.5 < head["s_per_step"] < 8.1 and head["wall_s"] + sum(p["wall_s"] for p in side) + 60 < args.wall_budget
    assert side[1]["s_per_step"] < 1.5   # the fixed job cut eight ways


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
    assert line["headline_kind"] == "strong" and line["strong"]["n1_same_job"]["value"] == line["solo"]["value"] > 0
    assert "must NOT be decided from rank-thread figures" in rr["in_process_exchange"]

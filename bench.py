#!/usr/bin/env python3
"""bench.py — encrypted gate-bootstraps/sec on the AES-128 gates-mode netlist.

A step = one full evaluation of the (generated, stand-in) AES-128 netlist over a batch of `--blocks` independent
input blocks, inputs already encrypted and resident in HBM.  The level schedule of the batch is launch-packed
(helm_amd/csrc/host/level_pack.cpp): the blocks share no wires, so launches hold whole lockstep rounds and only
the drain at the end of a pass is partial.

    python3 bench.py --gpus N --steps K --warmup W

With N > 1 and no RANK in the environment this process is only a launcher: it has not touched the GPU (it imports
nothing beyond the standard library), starts `python -m torch.distributed.run --nproc-per-node N ... bench.py` as a
CHILD process (never an exec), relays its ONE JSON line and leaves with its exit code.  Started under
torch.distributed.run directly (RANK set), the process is a worker.

One line, every result named:
  value / scaling "weak"     = `sharded_weak`: the north star's per-level shard at FIXED WORK PER GPU - ONE job of
                             N x `--blocks` blocks (at N = 1 the 32-block job), every packed launch split into N chunks by
                             bootstrap weight, keys and wire table replicated, the launch's output ciphertexts all-gathered
                             with ncclAllGather INSIDE libhelm_hip.so (include/helm_comm.h:
                             helm_hip_program_run_sharded_comm - no torch in the data path); EXACTLY K timed steps
  strong                     the same shard of a FIXED job of `--blocks` blocks whatever N (a 32-block level holds about
                             5,000 bootstraps: at N = 8 a rank's chunk is two thirds of a lockstep round - the netlist's
                             width, not the fabric, bounds it; modelled 0.72 of linear)
  weak                       independent blocks: every GPU evaluates its own `--blocks` blocks, no data-path collective
  rccl_ranks                 what the library's own RCCL communicator reports (ncclCommCount / ncclCommUserRank), the RCCL
                             version, and what carried the control plane (barrier, maximum over the ranks, the unique id)
`--scaling strong` / `--scaling weak` make one of the other two the headline instead; all three are always in the line
under their own names (the runs that are not the headline with `--side-steps` timed steps).

A worker that fails - an exception, a wrong decryption, a collective that does not come back within
`--leg-timeout` seconds - ends with a non-zero exit code and its error in the line; nothing is retried.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC: without this RCCL's peer mappings fail with `hipIpcGetMemHandle: invalid
# argument`.  Exported on the boxes already; kept here so that a launcher with a scrubbed environment still gets it (it has to be
# in place before the HIP runtime initialises, i.e. before the workers import torch)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_CLOCK_GHZ = 2.4          # MI355X engine clock the nominal peaks are quoted at
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md
# measured fabric traffic of the dominant kernels (tools/pmc_traffic.sh: separate --pmc passes); the file is
# rewritten by every round's profiling run, the per-round copies live under profiles/rNN/
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "pmc_traffic.json")
# SQ_INSTS_VALU / SQ_WAVES of the lockstep k_pbs (tools/pmc_issue.py over a --pmc pass of tools/prof_pbs.py)
PMC_ISSUE = os.path.join(ROOT, "profiles", "pmc_issue.json")
METRIC = "encrypted gate-bootstraps/sec on AES-128 gates-mode netlist"
# gate-bootstraps per second ONE GPU held in the last round's driver runs (BENCH_r05.json; profiles/r05/bench_n1_helm_cuda.json):
# only the prior of the printed plan of an N > 1 run - replaced by this run's own headline as soon as it is measured
PRIOR_GPU_RATE = {"boolean_default": 134e3, "helm_cuda": 133e3}
RESULT_FILE_ENV = "HELM_BENCH_RESULT_FILE"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=32, help="AES blocks of the fixed job (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=["sharded_weak", "strong", "weak"], default="sharded_weak",
                    help="which N > 1 run is the headline `value` (default: the per-level shard at fixed work per GPU)")
    ap.add_argument("--params", default="boolean_default")
    ap.add_argument("--no-pack", action="store_true", help="level-synchronous launches (the round-1 schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the LUT-mode / arithmetic-mode side measurements")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration wall-clock lines (BASELINE.json configs 1-3)")
    ap.add_argument("--no-side-legs", action="store_true", help="N > 1: only the headline run")
    ap.add_argument("--overlap", action="store_true",
                    help="sharded runs: the library's overlapped exchange (helm_hip_program_run_sharded_comm, overlap = 1): a launch's "
                         "all-gather + scatter on the engine's exchange stream while the launches that do not need its outputs run; "
                         "off until a SCALE run has measured it")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="N > 1: if the RCCL communicator cannot be created on every rank, carry the exchange over the control plane "
                         "(gloo through host memory) instead of failing; the line then says so - such a number is NOT a measurement "
                         "of the RCCL path.  Without this flag the run ends with value null, the error and a non-zero exit code")
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1: run the headline through the sharded path anyway - a world-size-1 RCCL communicator inside the library, "
                         "every launch stage -> ncclAllGather -> scatter (what the exchange machinery itself costs; rccl_ranks is filled)")
    ap.add_argument("--side-steps", type=int, default=3, help="N > 1: timed steps of the runs that are not the headline")
    ap.add_argument("--wall-budget", type=float, default=450.0,
                    help="N > 1: seconds the whole run should stay under; the planned wall time of every run is printed (stderr, "
                         "and `planned_wall_s` in the line) before it starts, and the runs that are NOT the headline are cut to one "
                         "timed step - then dropped - if the plan does not fit (the headline's K steps are never touched)")
    ap.add_argument("--leg-timeout", type=float, default=900.0,
                    help="N > 1: seconds one run may take before it counts as hung (the line is printed with the error, rc 3)")
    ap.add_argument("--launch-timeout", type=float, default=3300.0, help="launcher: seconds before the worker group is stopped")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU baseline: stop after the level that passes this time")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline threads (0 = the cores this process may use)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launcher (N > 1, started as `python3 bench.py --gpus N`): standard library only, never touches the GPU
# ----------------------------------------------------------------------------------------------------------------
def launch_workers(args, argv):
    import signal
    import socket
    import subprocess
    import tempfile
    import threading

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    fd, result_file = tempfile.mkstemp(prefix="helm_bench_", suffix=".json")
    os.close(fd)
    env = dict(os.environ)
    env[RESULT_FILE_ENV] = result_file
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    lines = []

    def relay():
        for ln in proc.stdout:
            ln = ln.rstrip("\n")
            try:
                ok = isinstance(json.loads(ln), dict) and ln.startswith('{"metric"')
            except ValueError:
                ok = False
            if ok:
                lines.append(ln)
            elif ln:
                print(ln, file=sys.stderr)  # stdout carries the ONE line only
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    error = None
    try:
        rc = proc.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        error = f"launcher: worker group still running after {args.launch_timeout:.0f} s, stopped"
        for sig in (signal.SIGTERM, signal.SIGKILL):  # exactly the process group started above
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=20)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = 3
    t.join(timeout=10)
    if lines:
        line = lines[-1]
        if rc != 0 and "error" not in json.loads(line):
            line = json.dumps(dict(json.loads(line), error=error or f"worker group left with exit code {rc}"))
    else:  # no line from rank 0: whatever it had checkpointed, with the error
        partial = {}
        try:
            with open(result_file) as f:
                partial = json.load(f)
        except (OSError, ValueError):
            pass
        partial.setdefault("metric", METRIC)
        partial.setdefault("value", None)
        partial.setdefault("n_gpus", args.gpus)
        partial["error"] = error or partial.get("error") or f"worker group left with exit code {rc} and no result line"
        line = json.dumps(partial)
        rc = rc or 1
    try:
        os.unlink(result_file)
    except OSError:
        pass
    print(line)
    sys.stdout.flush()
    return rc


# ----------------------------------------------------------------------------------------------------------------
# worker
# ----------------------------------------------------------------------------------------------------------------
def _load_profile(path):
    """-> (the committed counter summary, {"file", "sha256_16", "round"}) or (None, None): figures that come from a profile
    are labelled as such in the line - this run did not measure them."""
    import hashlib
    if not os.path.exists(path):
        return None, None
    with open(path, "rb") as f:
        raw = f.read()
    data = json.loads(raw)
    return data, {"file": os.path.relpath(path, ROOT), "sha256_16": hashlib.sha256(raw).hexdigest()[:16],
                  "copy_of": data.get("copy_of"), "copy_committed_in": data.get("copy_committed_in"), "measured_by_this_run": False}


def build_program_arrays(circuit, wire_names, blocks):
    """Tile the one-block level schedule over `blocks` copies of the wire table."""
    import numpy as np
    from helm_amd.distributed import level_arrays
    index = {w: i for i, w in enumerate(wire_names)}
    ops, i0, i1, i2, out, off = level_arrays(circuit, index)
    nw = len(wire_names)
    n_levels = len(off) - 1
    T = lambda a: [np.concatenate([np.where(a[off[l]:off[l + 1]] >= 0, a[off[l]:off[l + 1]] + b * nw, -1)
                                   for b in range(blocks)]) for l in range(n_levels)]
    opsT = [np.tile(ops[off[l]:off[l + 1]], blocks) for l in range(n_levels)]
    cat = lambda parts: np.concatenate(parts).astype(np.int32)
    new_off = np.concatenate([[0], np.cumsum([len(x) for x in opsT])]).astype(np.int64)
    return cat(opsT), cat(T(i0)), cat(T(i1)), cat(T(i2)), cat(T(out)), new_off, index


def make_program(sk, circuit, wire_names, blocks, quantum, pack=True, overlap_split=0):
    """-> (Program, launches, levels, None): the batch's level schedule, launch-packed to `quantum` bootstraps.
    overlap_split > 0: every launch cut into sub-launches of at most that many bootstraps, for the overlapped sharded
    schedule (the library computes the launch dependencies itself: helm_hip_program_run_sharded_comm, overlap = 1)."""
    import helm_amd
    from helm_amd.distributed import pack_levels, split_launches
    ops, i0, i1, i2, out, off, _ = build_program_arrays(circuit, wire_names, blocks)
    levels = len(off) - 1
    if pack:  # cost-aware: launches narrower than a round take the engine's most efficient width (wide / duo / lockstep)
        ops, i0, i1, i2, out, off, _ = pack_levels(ops, i0, i1, i2, out, off, quantum, quarter_cost=sk.launch_costs())
    if overlap_split > 0:
        off = split_launches(ops, off, overlap_split)
    return helm_amd.Program(sk, ops, i0, i1, i2, out, off), len(off) - 1, levels, None


def upload_inputs(ck, wires, index, nw, keys_pt, first_block=0):
    import numpy as np
    in_rows, in_bits = [], []
    for b, (key, pt) in enumerate(keys_pt):
        kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
        for i in range(128):
            in_rows += [(first_block + b) * nw + index[f"key[{i}]"], (first_block + b) * nw + index[f"pt[{i}]"]]
            in_bits += [(kv >> i) & 1, (pv >> i) & 1]
    wires.upload(np.array(in_rows, np.int32), ck.encrypt(np.array(in_bits, dtype=bool)))


class WrongResult(RuntimeError):
    pass


class PowerSampler:
    """Socket power, shader clock and temperature of ONE GPU read from its hwmon files (amdgpu: power1_input / power1_average in
    microwatts, power1_cap, freq1_input = sclk in Hz, temp*_input in millidegrees) every `period` seconds on a host thread
    while the timed steps run - no subprocess, no HIP call, nothing on the GPU.  The device is found by its PCI address.
    Answers VERDICT r05 item 5: is the clock the kernels hold (2.1 - 2.4 of 2.4 GHz nominal) a power cap or not."""

    def __init__(self, pci, period=0.25):
        import glob
        import threading
        self.period, self.samples, self.dir, self.cap_w = period, [], None, None
        for dev in glob.glob("/sys/class/drm/card*/device") if pci else []:
            try:
                if os.path.basename(os.path.realpath(dev)).lower().startswith(pci.lower()):
                    mons = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
                    if mons:
                        self.dir = mons[0]
                        break
            except OSError:
                continue
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _loop(self):
        while not self._stop.is_set():
            pw = self._read("power1_input")
            if pw is None:
                pw = self._read("power1_average")
            self.samples.append((time.perf_counter(), pw, self._read("freq1_input"), self._read("temp2_input") or self._read("temp1_input")))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.dir:
            self.cap_w = (self._read("power1_cap") or 0) / 1e6 or None
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.dir:
            self._thread.join(timeout=2)
        return False

    def summary(self):
        if not self.dir or len(self.samples) < 2:
            return {"available": False, "why": "no readable hwmon directory for the device" if not self.dir else "run too short for two samples"}
        pw = [x[1] / 1e6 for x in self.samples if x[1]]
        fr = [x[2] / 1e6 for x in self.samples if x[2]]
        tp = [x[3] / 1e3 for x in self.samples if x[3]]
        res = {"available": True, "samples": len(self.samples), "period_s": self.period, "source": os.path.join(self.dir, "{power1_input,freq1_input,temp*_input}"),
               "socket_power_w": {"mean": round(sum(pw) / len(pw), 1), "max": round(max(pw), 1), "min": round(min(pw), 1)} if pw else None,
               "power_cap_w": self.cap_w,
               "sclk_mhz_hwmon": {"mean": round(sum(fr) / len(fr), 1), "max": round(max(fr), 1), "min": round(min(fr), 1)} if fr else None,
               "temperature_c": {"mean": round(sum(tp) / len(tp), 1), "max": round(max(tp), 1)} if tp else None}
        if pw and self.cap_w:
            frac = sum(pw) / len(pw) / self.cap_w
            res["mean_power_over_cap"] = round(frac, 3)
            # the package's power management holds the socket a few per cent under its limit and pays with the clock: at or
            # above 0.9 of the cap the held clock IS the power limit's doing, and frac_at_held_clock is the kernel's figure
            res["at_power_limit"] = bool(frac >= 0.9)
        return res


def check_outputs(ck, wires, index, nw, keys_pt, what):
    import numpy as np
    from helm_amd.netlists import aes128_reference_encrypt
    out_rows = np.array([b * nw + index[f"ct[{i}]"] for b in range(len(keys_pt)) for i in range(128)], np.int32)
    dec = ck.decrypt(wires.download(out_rows)).reshape(len(keys_pt), 128)
    for b, (key, pt) in enumerate(keys_pt):
        got = sum(int(dec[b, i]) << i for i in range(128)).to_bytes(16, "big")
        if got != aes128_reference_encrypt(key, pt):
            raise WrongResult(f"{what}: decrypted AES output of block {b} is WRONG")


class RcclUnavailable(RuntimeError):
    pass


class RankLayoutError(RuntimeError):
    pass


def check_rank_devices(devices, world, n_gpus, rehearsal=False, host_fallback=False):
    """What must hold before an N > 1 number means anything: as many ranks as --gpus, a communicator that reports that
    world size and this rank on every rank, RCCL carrying it (unless the run is a labelled rehearsal / host fallback), and
    no two ranks on the same device.  -> list of problems (empty: fine).  Pure function of the gathered records
    (tests/test_bench_launcher.py feeds it by hand)."""
    problems = []
    if world != n_gpus:
        problems.append(f"world size {world} != --gpus {n_gpus}")
    if len(devices) != world or any(d is None for d in devices):
        return problems + [f"{sum(d is not None for d in devices)} of {world} ranks reported their device"]
    for d in devices:
        if d["comm_world"] != world:
            problems.append(f"rank {d['rank']}: its communicator reports world size {d['comm_world']}, not {world}")
        if d["comm_rank"] != d["rank"]:
            problems.append(f"rank {d['rank']}: its communicator reports rank {d['comm_rank']}")
        if not d["rccl_version"] and not (rehearsal or host_fallback):
            problems.append(f"rank {d['rank']}: the communicator is not RCCL's")
    if not rehearsal:
        # two ranks on one device index is always wrong; the PCI address / uuid torch reports must differ as well - unless the
        # platform reports the SAME identity for every device (then it says nothing, and the distinct indices stand)
        by_index = {}
        for d in devices:
            if d["local_rank"] in by_index:
                problems.append(f"ranks {by_index[d['local_rank']]} and {d['rank']} both use device index {d['local_rank']}")
            by_index.setdefault(d["local_rank"], d["rank"])
        keys = [(d["pci"], d["uuid"]) for d in devices]
        if not (world > 1 and len(set(keys)) == 1 and len(by_index) == world):
            seen = {}
            for d, key in zip(devices, keys):
                if key in seen:
                    problems.append(f"ranks {seen[key]} and {d['rank']} share one device ({d['pci']}): one process per GPU needs "
                                    f"{world} different devices")
                seen.setdefault(key, d["rank"])
    return problems


def _injected_failure():
    """HELM_BENCH_INJECT_COMM_FAILURE=<step>:<rank> (id | precheck | create): the handshake's failure paths, for
    tests/test_bench_launcher.py; never set otherwise."""
    v = os.environ.get("HELM_BENCH_INJECT_COMM_FAILURE")
    if not v:
        return None
    step, rank = v.split(":")
    return step, int(rank)


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; stdout carries the ONE JSON line only."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class Bench:
    """State shared by the runs of one worker."""

    def __init__(self, args, thread_rank=None):
        import numpy as np
        import torch
        import torch.distributed as dist
        import helm_amd
        from helm_amd import Circuit, verilog_parser
        from helm_amd.netlists import aes128
        self.args, self.np, self.torch, self.dist = args, np, torch, dist
        self.t_start = time.time()
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.threads = thread_rank is not None
        if self.threads:
            # HELM_BENCH_REHEARSE=threads: the ranks are THREADS of this one process (a one-GPU box allows six GPU
            # processes, the scaling run has eight ranks); control plane and transport are in-process
            self.dist = dist = thread_rank
            self.rank, self.world, local_rank = dist.get_rank(), dist.get_world_size(), 0
        # rehearsal of the N > 1 path on a one-GPU box: HELM_BENCH_REHEARSE=1 puts every rank on cuda:0 and
        # carries the collectives over gloo (RCCL needs one GPU per rank); never used for reported numbers
        self.rehearse = os.environ.get("HELM_BENCH_REHEARSE") in ("1", "threads")
        if self.rehearse:
            local_rank = 0
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        try:  # the PCI address of this rank's GPU, for the power / clock sampler (rank threads: asked once, one at a time)
            with (thread_rank.lock if thread_rank is not None else __import__("contextlib").nullcontext()):
                pr = torch.cuda.get_device_properties(local_rank)
            self.pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:  # noqa: BLE001 - the sampler then reports itself unavailable
            self.pci = None
        self.comm, self.comm_error = None, None
        if self.threads:
            self.comm = dist.comm      # helm_amd.comm.Comm.in_process_group: device-to-device copies between the rank threads
        elif self.world > 1:
            from helm_amd import comm as hc
            # control plane (barrier, maximum over the ranks, the communicator's unique id): torch.distributed over gloo
            dist.init_process_group("gloo")
            if self.rehearse and _injected_failure() is None:
                # the same code path as the real run (helm_hip_program_run_sharded_comm, helm_si_set_exchange_comm), the
                # library's communicator carried by a host transport because the ranks share the GPU
                self.comm = hc.Comm.over_torch_dist(dist, local_rank)
            else:
                # data path: the library's own RCCL communicator, one rank per GPU (include/helm_comm.h).  Comm.agree: id
                # from rank 0 (or its error) broadcast unconditionally, local pre-checks agreed on BEFORE anybody enters
                # ncclCommInitRank (which has no timeout), the outcome agreed on afterwards - every rank gets the same answer
                with _StdoutToStderr():
                    self.comm, self.comm_error = hc.Comm.agree(dist, local_rank, _inject=_injected_failure())
                if self.comm is None:
                    if not args.allow_host_fallback:
                        raise RcclUnavailable(f"the RCCL communicator could not be created on every rank ({self.comm_error}); "
                                              "--allow-host-fallback carries the exchange over gloo instead (not a measurement of the RCCL path)")
                    self.comm = hc.Comm.over_torch_dist(dist, local_rank)
        elif args.force_comm:
            from helm_amd import comm as hc
            with _StdoutToStderr():
                self.comm = hc.Comm.single(local_rank)
        self.devices = None
        if self.world > 1 and not self.threads:
            # one process per GPU means N DIFFERENT devices and a communicator of N ranks: checked, not assumed - every rank
            # learns every rank's device (PCI address + uuid as torch reports them) and what its own communicator says, and
            # every rank raises the same error (value null, rc != 0) if two ranks share a device or the world is not --gpus
            pr = torch.cuda.get_device_properties(local_rank)
            mine = {"rank": self.rank, "local_rank": local_rank, "pci": f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}",
                    "uuid": str(getattr(pr, "uuid", "")), "comm_world": self.comm.info()["world_size"],
                    "comm_rank": self.comm.info()["rank"], "rccl_version": self.comm.info()["rccl_version"]}
            self.devices = [None] * self.world
            dist.all_gather_object(self.devices, mine)
            problems = check_rank_devices(self.devices, self.world, args.gpus, rehearsal=self.rehearse,
                                          host_fallback=bool(self.comm_error) and args.allow_host_fallback)
            if problems:
                raise RankLayoutError("; ".join(problems))
        # keys (identical on every rank: same deterministic benchmark seed) and engine
        t0 = time.time()
        self.ck = helm_amd.ClientKey.generate(args.params, seed=1)
        with self.setup_lock():
            self.sk = helm_amd.ServerKey(self.ck, device=local_rank)
            if not self.threads:   # (rank threads keep the context's own stream: torch's current stream is the same null stream in every thread)
                self.sk.set_stream(torch.cuda.current_stream().cuda_stream)
            self.sk.sync()
        self.t_keys = time.time() - t0
        self.quantum = self.sk.launch_quantum()
        gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
        self.circuit = Circuit(gates, inputs, outputs, dffs)
        self.n_gates = len(gates)
        self.circuit.sort_circuit()
        self.circuit.compute_levels()
        self.wire_names = list(inputs) + sorted(wire_set)
        self.nw = len(self.wire_names)
        self.index = {w: i for i, w in enumerate(self.wire_names)}
        from helm_amd.distributed import gate_pbs, level_arrays
        self.pbs_per_block = int(gate_pbs(level_arrays(self.circuit, self.index)[0]).sum())

    def agree_max(self, values):
        """The maximum over the ranks of each value: identical on every rank afterwards (plans every rank must reach the same
        way are made from such figures only)."""
        out = []
        for v in values:
            t = self.torch.tensor([float(v)], dtype=self.torch.float64)
            if self.world > 1:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            out.append(float(t.item()))
        return out

    def plan(self, kind, steps, warmup, gpu_rate, setup_s_per_block=None):
        """Planned wall time of one run, from the rate one GPU holds (bootstraps per second: the last round's 134 k before
        anything is measured, this run's own headline afterwards) and the measured set-up time per block of a wire table
        (encryption and upload of the inputs, launch packing, the decryption check)."""
        a, w = self.args, self.world
        per_gpu_blocks = a.blocks / w if kind == "strong" else a.blocks
        # the fixed job cut N ways fills N GPUs by about three quarters (chunks of two thirds of a lockstep round: DESIGN 7)
        step_s = per_gpu_blocks * self.pbs_per_block / gpu_rate / (0.76 if kind == "strong" and w > 1 else 1.0)
        table_blocks = a.blocks * (w if kind == "sharded_weak" else 1)
        setup_s = 2.0 + table_blocks * (0.25 if setup_s_per_block is None else setup_s_per_block)
        return {"steps": steps, "warmup": warmup, "s_per_step": round(step_s, 2), "setup_and_check_s": round(setup_s, 1),
                "wall_s": round((steps + warmup) * step_s + setup_s, 1)}

    def setup_lock(self):
        """Rank THREADS of one process plan, allocate, upload and download one at a time (eight ranks' set-up at once would only
        compete for the same host cores and the one GPU's copy engines); the passes themselves run concurrently.  Processes:
        no lock."""
        import contextlib
        return self.dist.lock if self.threads else contextlib.nullcontext()

    def sync_all(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, kind, steps, warmup, keep=False):
        """One measured run.  kind: "strong" (fixed job of --blocks blocks, every launch sharded over the ranks),
        "weak" (this rank's own --blocks blocks, nothing exchanged), "sharded_weak" (one job of world x --blocks
        blocks, every launch sharded).  W untimed steps, then exactly `steps` timed ones between barrier +
        synchronize on both sides; the elapsed time is the maximum over the ranks."""
        from helm_amd.distributed import GpuLevelExecutor, ShardedRunner
        a, np, torch, dist = self.args, self.np, self.torch, self.dist
        # "solo" (N > 1 only): the N = 1 value of the SAME job shape, measured in this very run - rank 0 alone evaluates the
        # `--blocks` job unsharded (what `python3 bench.py --gpus 1` times) while every other rank idles at the barrier, so
        # that each N > 1 row carries its own baseline and a scaling efficiency is computable from ONE record
        solo = kind == "solo"
        if solo and self.rank != 0:
            dist.barrier()
            return None
        sync = torch.cuda.synchronize if solo else self.sync_all
        t_run0 = time.perf_counter()
        sharded = kind not in ("weak", "solo") and (self.world > 1 or self.comm is not None)
        blocks = a.blocks * (self.world if kind == "sharded_weak" else 1)   # blocks in this rank's wire table
        # --overlap: launches cut to one lockstep round per rank (the gates of a launch are independent, so any cut is
        # valid): the sub-launches of one packed launch do not depend on each other, the exchange of one travels while the
        # next one's bootstraps run
        with self.setup_lock():
            prog, launches, levels, _ = make_program(self.sk, self.circuit, self.wire_names, blocks,
                                                     self.quantum * (self.world if sharded else 1), pack=not a.no_pack,
                                                     overlap_split=self.quantum * self.world if (sharded and a.overlap) else 0)
        overlapped = bool(sharded and a.overlap and prog.overlap_applies())
        rng = np.random.default_rng(0x48454C4D + (0 if sharded else self.rank))
        keys_pt = [(bytes(rng.integers(0, 256, 16, dtype=np.uint8)), bytes(rng.integers(0, 256, 16, dtype=np.uint8)))
                   for _ in range(blocks)]
        if self.rank == 0 or sharded:
            keys_pt[0] = (bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff"))  # FIPS-197 C.1
        # synthetic inputs: encrypted on the host, uploaded once: resident in HBM before the timed region
        with self.setup_lock():
            wires = self.sk.wires(self.nw * blocks)
            upload_inputs(self.ck, wires, self.index, self.nw, keys_pt)
            runner = ShardedRunner(GpuLevelExecutor(prog, wires), self.rank, self.world if sharded else 1,
                                   dist if sharded else None, time_collective=sharded,
                                   comm=self.comm if sharded else None, overlap=overlapped)
            self.sk.sync()
        sync()
        t_setup = time.perf_counter() - t_run0
        for _ in range(warmup):
            runner.run()
        sync()
        runner.collective_ms(reset=True)
        self.sk.timing_enable(True)
        self.sk.timing(reset=True)
        with PowerSampler(self.pci) as power:
            t0 = time.perf_counter()
            for _ in range(steps):
                runner.run()
            sync()
            elapsed = time.perf_counter() - t0
        tm = self.sk.timing(reset=True)
        self.sk.timing_enable(False)
        clock_ghz = self.sk.kernel_clock_ghz()
        if self.world > 1 and not solo:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # correctness of what was timed: every block of this rank's table decrypts to AES(key, pt)
        t_chk0 = time.perf_counter()
        with self.setup_lock():
            check_outputs(self.ck, wires, self.index, self.nw, keys_pt, f"{kind} run, rank {self.rank}")
        t_setup += time.perf_counter() - t_chk0
        pbs = prog.total_pbs()
        job_pbs = pbs * (self.world if kind == "weak" else 1)
        r = {"kind": kind, "elapsed": elapsed, "steps": steps, "warmup": warmup, "job_pbs": int(job_pbs),
             "value": job_pbs * steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
             "blocks_per_table": blocks, "blocks_total": blocks * (self.world if kind == "weak" else 1),
             "launches": launches, "levels": levels, "sharded_launches": len(runner.sharded_levels),
             "exchanged_MB_per_step": runner.exchanged_bytes_per_pass() / 1e6,
             # in-library communicator: the engine's events around every ncclAllGather; torch path: events around the calls
             "collective_ms_per_step": ((tm.exchange_ms if self.comm is not None else runner.collective_ms(reset=True)) / steps
                                        if sharded else 0.0),
             "collectives_per_step": (int(tm.exchange_count) // steps if (sharded and self.comm is not None)
                                      else len(runner.sharded_levels) if sharded else 0),
             "overlapped": overlapped, "setup_s": t_setup, "power": power.summary(),
             "tm": tm, "clock_ghz": clock_ghz}
        if keep:
            r.update(prog=prog, wires=wires, keys_pt=keys_pt)
        else:
            with self.setup_lock():
                prog.destroy()
                wires.free()
        if solo:
            dist.barrier()   # releases the ranks that idled
        elif self.threads:
            dist.barrier()   # nobody frees while another rank still runs
        return r

    def describe(self, r):
        """The named summary of a run that is not the headline."""
        w = self.world
        what = {"strong": f"fixed job of {r['blocks_total']} AES-128 block(s), every packed launch sharded over {w} GPU(s), "
                          "output ciphertexts all-gathered",
                "weak": f"{self.args.blocks} independent AES-128 block(s) per GPU ({r['blocks_total']} in total), keys replicated, "
                        "no data-path collective",
                "sharded_weak": f"ONE job of {r['blocks_total']} AES-128 blocks ({self.args.blocks} per GPU), every packed launch "
                                f"sharded over {w} GPU(s), output ciphertexts all-gathered",
                "solo": f"the N = 1 job measured inside this N = {w} run: rank 0 ALONE evaluates {r['blocks_total']} AES-128 block(s) "
                        "unsharded (what `bench.py --gpus 1` times), the other ranks idle at a barrier"}[r["kind"]]
        return {"workload": what, "scaling": {"strong": "strong", "solo": "none (one GPU)"}.get(r["kind"], "weak"),
                "value": round(r["value"], 1), "unit": "gate-bootstraps/s", "steps": r["steps"], "warmup": r["warmup"],
                "ms_per_step": round(r["ms_per_step"], 3), "bootstraps_per_step": r["job_pbs"],
                "launches_per_step": r["launches"], "sharded_launches": r["sharded_launches"],
                "exchanged_MB_per_step": round(r["exchanged_MB_per_step"], 2),
                "collective_ms_per_step": round(r["collective_ms_per_step"], 3),
                "collectives_per_step": r["collectives_per_step"],
                "exchange_overlapped_with_next_launch": r["overlapped"],
                "decrypt_check": "all blocks == software AES on every rank"}


_emit_state = {"done": False}


def checkpoint(result):
    """Rank 0: what is known so far, for the launcher to print if this process does not live to print it."""
    path = os.environ.get(RESULT_FILE_ENV)
    if path:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(result, f)
        os.replace(tmp, path)


def emit(result):
    """Rank 0 prints THE line, once."""
    if _emit_state["done"]:
        return
    _emit_state["done"] = True
    checkpoint(result)
    print(json.dumps(result))
    sys.stdout.flush()


def watched(world, rank, timeout, result, what, fn):
    """Run one leg under a watchdog: a collective that never comes back must not pass for a finished run.
    On timeout rank 0 prints the line with what it has and the error, and EVERY rank leaves with exit code 3."""
    import threading
    if world == 1:
        return fn()
    finished = threading.Event()

    def watchdog():
        if not finished.wait(timeout):
            if rank == 0:
                result["error"] = f"{what}: no answer within {timeout:.0f} s (hung collective?)"
                emit(result)
            sys.stderr.write(f"[bench rank {rank}] {what}: timed out, leaving with exit code 3\n")
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        return fn()
    finally:
        finished.set()


def guarded(bench, result, what, fn):
    return watched(bench.world, bench.rank, bench.args.leg_timeout, result, what, fn)


class ThreadDist:
    """What bench.py uses of torch.distributed, for ranks that are threads of one process (HELM_BENCH_REHEARSE=threads)."""

    class ReduceOp:
        MAX, MIN = "max", "min"

    def __init__(self, rank, shared):
        self.rank, self.shared = rank, shared
        self.comm = shared["comms"][rank]
        self.lock = shared["lock"]

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return len(self.shared["slots"])

    def get_backend(self):
        return "threads"

    def barrier(self):
        self.shared["barrier"].wait()

    def all_reduce(self, t, op="max"):
        s = self.shared
        s["slots"][self.rank] = float(t.item())
        s["barrier"].wait()
        v = (max if op == "max" else min)(s["slots"])
        s["barrier"].wait()
        t.fill_(v)

    def destroy_process_group(self):
        pass


def thread_workers(args):
    """HELM_BENCH_REHEARSE=threads python3 bench.py --gpus N: N rank threads in THIS process, all on cuda:0, each with its own
    engine context; the library's communicator over helm_amd.comm.Comm.in_process_group.  Everything else - programs packed
    for N ranks, helm_hip_program_run_sharded_comm, the three runs, the decryption checks on every rank - is the real run's."""
    import threading
    from helm_amd.comm import Comm
    n = args.gpus
    shared = {"slots": [0.0] * n, "barrier": threading.Barrier(n, timeout=args.leg_timeout), "comms": Comm.in_process_group([0] * n),
              "lock": threading.Lock()}
    rcs = [1] * n

    def main(r):
        try:
            rcs[r] = worker(args, ThreadDist(r, shared))
        except BaseException:  # noqa: BLE001
            import traceback
            traceback.print_exc()
        finally:
            if rcs[r] != 0:
                shared["barrier"].abort()
                shared["comms"][r].abort_group()
    ts = [threading.Thread(target=main, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return max(rcs)


def worker(args, thread_rank=None):
    import signal
    world0 = thread_rank.get_world_size() if thread_rank else int(os.environ.get("WORLD_SIZE", "1"))
    result = {"metric": METRIC, "value": None, "unit": "gate-bootstraps/s", "n_gpus": world0,
              "steps": args.steps, "warmup": args.warmup}
    rank = thread_rank.get_rank() if thread_rank else int(os.environ.get("RANK", "0"))
    if rank == 0 and thread_rank is None:
        def on_term(signum, _frame):  # torch.distributed.run stops the group when another rank dies
            result.setdefault("error", f"rank 0 stopped by signal {signum} (another rank failed?)")
            emit(result)
            os._exit(128 + signum)
        signal.signal(signal.SIGTERM, on_term)
    rc = 0
    bench = None
    try:
        # the set-up has collectives of its own (process group, the communicator's handshake, ncclCommInitRank)
        bench = watched(result["n_gpus"], rank, args.leg_timeout, result, "set-up (process group, communicator, keys)", lambda: Bench(args, thread_rank))
        fill_result(bench, result)
    except BaseException as e:  # incl. SystemExit / KeyboardInterrupt: the line carries the error, rc != 0
        import traceback
        traceback.print_exc()
        result["error"] = f"rank {rank}: {e!r}"
        rc = 1
    if rank == 0:
        emit(result)
    if bench is not None and bench.world > 1 and rc == 0 and thread_rank is None:
        if bench.comm is not None:
            bench.comm.destroy()
        bench.dist.destroy_process_group()
    return rc


def fill_result(bench, result):
    args, np = bench.args, bench.np
    world, rank = bench.world, bench.rank
    p = bench.ck.params
    head_kind = args.scaling if world > 1 else "strong"
    say = (lambda msg: (sys.stderr.write("[bench] " + msg + "\n"), sys.stderr.flush())) if rank == 0 else (lambda msg: None)
    planned = {}
    if world > 1:
        # before anything is measured: the rate one GPU held in the last round's driver run (BENCH_r05.json, boolean_default)
        planned[head_kind] = bench.plan(head_kind, args.steps, args.warmup, PRIOR_GPU_RATE.get(args.params, 130e3))
        say(f"plan: {head_kind} (headline) {args.warmup} + {args.steps} steps x ~{planned[head_kind]['s_per_step']} s + set-up and check "
            f"~{planned[head_kind]['setup_and_check_s']} s = ~{planned[head_kind]['wall_s']} s; budget for the whole run {args.wall_budget:.0f} s")
    head = guarded(bench, result, f"{head_kind} run", lambda: bench.run(head_kind, args.steps, args.warmup, keep=True))
    tm, clock_ghz, quantum = head["tm"], head["clock_ghz"], bench.quantum
    sharded = head_kind != "weak" and world > 1

    # ---- N > 1: which other runs follow, decided the same way on every rank from figures every rank holds ----
    legs, dropped = [], []
    if world > 1 and not args.no_side_legs:
        used, setup_s = guarded(bench, result, "plan of the remaining runs",
                                lambda: bench.agree_max([time.time() - bench.t_start, head["setup_s"]]))
        gpu_rate = head["value"] / world / (0.76 if head_kind == "strong" else 1.0)
        per_block = setup_s / head["blocks_per_table"]
        planned[head_kind]["measured_wall_s"] = round(head["elapsed"] * (1 + args.warmup / max(1, args.steps)) + setup_s, 1)
        # the N = 1 baseline first (it is what makes the record self-explanatory), then the two other shapes
        for steps_each in (args.side_steps, 1):
            legs = [(k, steps_each, 1) for k in ["solo"] + side_kinds(head_kind)]
            if used + sum(bench.plan(k, st, wu, gpu_rate, per_block)["wall_s"] for k, st, wu in legs) <= args.wall_budget:
                break
        while len(legs) > 1 and used + sum(bench.plan(k, st, wu, gpu_rate, per_block)["wall_s"] for k, st, wu in legs) > args.wall_budget:
            dropped.append(legs.pop()[0])
        for k, st, wu in legs:
            planned[k] = bench.plan(k, st, wu, gpu_rate, per_block)
        say(f"plan: {used:.0f} s used so far (headline measured: {head['ms_per_step'] / 1e3:.2f} s per step); then " +
            ", ".join(f"{k} {wu} + {st} steps = ~{planned[k]['wall_s']} s" for k, st, wu in legs) +
            f"; total ~{used + sum(planned[k]['wall_s'] for k, _, _ in legs):.0f} s of {args.wall_budget:.0f} s" +
            (f"; dropped to fit: {', '.join(dropped)}" if dropped else ""))
    solo = None
    if legs and legs[0][0] == "solo":
        _, st, wu = legs.pop(0)
        solo = guarded(bench, result, "solo run (the N = 1 job inside this run)", lambda: bench.run("solo", st, wu))

    # ---- roofline of the dominant kernel (lockstep build of k_pbs), HIP events on its own stream ----
    K1 = p.k + 1
    logN = int(np.log2(p.N))
    bsk_bytes = p.n * p.pbs_l * K1 * K1 * p.N * 8
    io_bytes = 2 * (p.n + 1) * 4 + (p.k * p.N + 1) * 4      # two input LWEs read, one big LWE written
    if tm.pbs_main_launches > 0:
        dom_kernel, n_launch = "k_pbs<PbsCfg<..., NB = 4>> (lockstep build)", tm.pbs_main_launches
        avg_pbs_per_launch, avg_launch_s = tm.pbs_main_count / n_launch, tm.pbs_main_ms / n_launch * 1e-3
    else:
        dom_kernel, n_launch = "k_pbs (all builds)", max(1, tm.pbs_launches)
        avg_pbs_per_launch, avg_launch_s = tm.pbs_count / n_launch, tm.pbs_ms / n_launch * 1e-3
    algo_bytes = bsk_bytes + avg_pbs_per_launch * io_bytes
    # ALGORITHMIC arithmetic of one bootstrap, SURVEY.md 8(d): n steps x [((k+1) l + (k+1)) transforms of
    # N/2 log2 N butterflies + (k+1)^2 l N multiply-accumulates]; priced at 8 fp64 lane-operations per butterfly
    # (6-operation exact modular multiplication + add + sub) and 7 per multiply-accumulate (FMA = 1 operation)
    bfly = p.n * (K1 * p.pbs_l + K1) * (p.N // 2) * logN
    macs = p.n * K1 * K1 * p.pbs_l * p.N
    algo_ops = bfly * 8 + macs * 7
    survey_ops = algo_ops
    # What the algorithm needs on this engine since round 4, and what `achieved` / `frac` are priced with since round 5: in
    # the fields p = b^4 + 1 the first two stages of every forward transform on decomposition digits are ONE radix-4
    # butterfly of 10 plain operations per four values (no modular reduction) instead of two stages of N/2 butterflies x 8.
    # Pricing the kernel with operations it no longer executes (the survey's count, kept as `frac_survey_priced`) would
    # overstate its fp64 utilisation.
    if bench.sk.short_root_stages() == 2:
        algo_ops = survey_ops - p.n * (K1 * p.pbs_l) * (2 * (p.N // 2) * 8 - (p.N // 4) * 10)
    n_cus = quantum // 4
    peak_tops = n_cus * 64 * PEAK_CLOCK_GHZ * 1e9 / 1e12
    achieved_tops = algo_ops * avg_pbs_per_launch / avg_launch_s / 1e12
    survey_tops = survey_ops * avg_pbs_per_launch / avg_launch_s / 1e12
    # counters of the committed profile (separate rocprofv3 --pmc passes; this run did not collect them): fabric traffic of
    # the lockstep launches, and the vector instructions a wave issues
    traffic, traffic_from = _load_profile(PMC_TRAFFIC)
    issue, issue_from = _load_profile(PMC_ISSUE)
    traffic_bytes = None
    if traffic:
        # bytes per bootstrap measured at the launch size of the committed profile, scaled to this run's average launch
        traffic_bytes = int(traffic["bytes_per_launch"] * avg_pbs_per_launch / traffic["bootstraps_per_launch"])
    valu_issue = None
    if issue and tm.pbs_main_launches > 0 and issue.get("params", args.params) == args.params:
        # lane-instructions the kernel issues per bootstrap (every wave64 vector instruction occupies its SIMD for four
        # cycles = 64 lane-slots) against the chip's issue capacity = the same 256 x 64 lanes per cycle the fp64 peak counts
        lane_insts = issue["valu_insts_per_wave"] * issue["waves_per_bootstrap"] * 64
        issued_tops = lane_insts * avg_pbs_per_launch / avg_launch_s / 1e12
        valu_issue = {"frac": round(issued_tops / peak_tops, 4),
                      "frac_at_held_clock": round(issued_tops / (peak_tops * clock_ghz / PEAK_CLOCK_GHZ), 4) if clock_ghz else None,
                      "valu_insts_per_wave_step": round(issue["valu_insts_per_wave"] / p.n, 1),
                      "lane_insts_per_bootstrap": int(lane_insts),
                      "over_algorithmic": round(lane_insts / algo_ops, 3),
                      "what": "SQ_INSTS_VALU of the lockstep k_pbs x 64 lanes x this run's bootstraps per second over the chip's "
                              "vector issue capacity (256 CUs x 4 SIMDs x 16 lanes per cycle): how full the issue slots are; "
                              "the rest of the distance to the peak is instruction overhead (`over_algorithmic`) and clock",
                      "from_profile": issue_from}

    if rank != 0:
        # the other ranks only take part in the remaining runs
        for kind, st, wu in legs:
            guarded(bench, result, f"{kind} run", lambda k=kind, st=st, wu=wu: bench.run(k, st, wu))
        return

    def with_baseline(row):
        """Every N > 1 row carries the N = 1 value of the same job shape (--blocks blocks on one GPU, unsharded: per-GPU work
        of `weak` / `sharded_weak`, the whole job of `strong`), measured by rank 0 inside this run: value / (N x that) is the
        row's distance to linear, from this one record."""
        if solo is not None:
            row["n1_same_job"] = {"value": round(solo["value"], 1), "ms_per_step": round(solo["ms_per_step"], 3), "steps": solo["steps"],
                                  "blocks": solo["blocks_total"], "measured": "in this run, rank 0 alone, the other ranks idle"}
            row["speedup_over_n1"] = round(row["value"] / solo["value"], 4)
            row["linear_would_be"] = world
        return row

    result.update({
        "value": round(head["value"], 1),
        "ms_per_step": round(head["ms_per_step"], 3),
        "higher_is_better": True,
        # which of the named runs `value` is; at N = 1 there is only the fixed `--blocks` job on one GPU (the N = 1 point of
        # every one of the three curves)
        "headline_kind": head_kind if world > 1 else "fixed_job_single_gpu",
        # N = 1: the `--blocks` job.  N > 1: per-GPU work fixed (sharded_weak: ONE job of N x `--blocks` blocks, every launch
        # sharded; weak: independent blocks) unless --scaling strong (the `--blocks` job whatever N)
        "scaling": "strong" if args.scaling == "strong" else "weak",
        "vs_baseline": None,
        "dtype": "u32 torus (exact NTT in f64 FMA over a 49-bit prime)",
        "data": "synthetic: generated AES-128 netlist (stand-in for HELM's), seeded random keys/plaintexts, "
                "fresh encryptions resident in HBM" + (" [REHEARSAL: all ranks on one GPU, " + ("rank THREADS of one process, in-process transport]" if bench.threads else "gloo]") if bench.rehearse else ""),
        "config": {
            "workload": (f"AES-128 gates-mode netlist, ONE job of {head['blocks_total']} blocks ({args.blocks} per GPU), every launch "
                         f"sharded over {world} GPUs" if head_kind == "sharded_weak" else
                         f"AES-128 gates-mode netlist, fixed job of {head['blocks_total']} block(s)"
                         + (f", every launch sharded over {world} GPUs" if sharded else "") if head_kind != "weak" else
                         f"AES-128 gates-mode netlist, {args.blocks} independent block(s) per GPU")
                        + (", launch-packed" if not args.no_pack else ", level-synchronous"),
            "params": args.params, "n": p.n, "k": p.k, "N": p.N, "pbs_l": p.pbs_l, "pbs_logB": p.pbs_logB,
            "ks_l": p.ks_l, "ks_logB": p.ks_logB,
            "levels": head["levels"], "launches_per_step": head["launches"], "bootstraps_per_step": head["job_pbs"],
            "blocks_total": head["blocks_total"],
            "value_is": f"a batch of {head['blocks_total']} independent AES-128 evaluations run together (the blocks share no wires, so launches "
                        "hold whole lockstep rounds); ONE evaluation, what helm.rs:256-262 runs, is `single_block` with its own roofline",
            "parallelism": ("single GPU" if world == 1 else
                            f"launch-shard x{world} + all-gather of launch outputs " +
                            ("inside the library (ncclAllGather, RCCL)" if bench.comm.info()["rccl_version"] else
                             "through the library's communicator over a HOST TRANSPORT (" + ("in-process copies" if bench.threads else "gloo through host memory") + "; RCCL not used: " +
                             ("rehearsal on one GPU" if bench.rehearse and not bench.comm_error else "--allow-host-fallback") + ")") +
                            (", exchange overlapped with independent launches" if head["overlapped"] else "") +
                            ", keys and wire table replicated" if sharded else
                            f"block-parallel x{world}: independent blocks per GPU, keys replicated, no data-path collective"),
            "sharded_launches": head["sharded_launches"],
            "exchanged_MB_per_step": round(head["exchanged_MB_per_step"], 2),
            "collective_ms_per_step": round(head["collective_ms_per_step"], 3),
        },
        "wall_s_per_step": round(head["elapsed"] / head["steps"], 4),
        # all gates of the netlist (NOT included, which costs no bootstrap) over the same wall-clock
        "netlist_gates_per_s": round(bench.n_gates * head["blocks_total"] * head["steps"] / head["elapsed"], 1),
        "decrypt_check": "all blocks == software AES (FIPS-197 C.1 vector in block 0)",
        "kernel_ms_per_step": {"k_pbs": round(tm.pbs_ms / args.steps, 3),
                               "k_pbs_lockstep_build": round(tm.pbs_main_ms / args.steps, 3),
                               "k_pbs_other_builds_share": round(1.0 - tm.pbs_main_ms / max(tm.pbs_ms, 1e-9), 4),
                               "k_keyswitch": round(tm.ks_ms / args.steps, 3),
                               "k_linear": round(tm.linear_ms / args.steps, 3)},
        "roofline": {
            "kernel": dom_kernel,
            # the binding resource: fp64 vector issue - the key is served by L2 / Infinity Cache
            "bound": "fp64_valu",
            "achieved": round(achieved_tops, 2), "peak": round(peak_tops, 2), "unit": "T fp64 lane-op/s (FMA = 1)",
            "frac": round(achieved_tops / peak_tops, 4),
            "peak_assumes": f"{n_cus} CUs x 64 lanes x {PEAK_CLOCK_GHZ} GHz, one fp64 operation per lane and cycle",
            "held_clock_ghz": round(clock_ghz, 3) if clock_ghz else None,
            # socket power and the driver's own clock reading over the timed steps (rank 0's GPU); a held clock
            # below nominal with the power far under its cap is the part's own frequency management, not a power limit
            "power_and_clock_over_the_timed_steps": head["power"],
            "frac_at_held_clock": round(achieved_tops / (peak_tops * clock_ghz / PEAK_CLOCK_GHZ), 4) if clock_ghz else None,
            "algorithmic_lane_ops_per_bootstrap": int(algo_ops),
            "frac_survey_priced": round(survey_tops / peak_tops, 4),
            "survey_lane_ops_per_bootstrap": int(survey_ops),
            "note_on_the_count": ("`achieved` and `frac` price what the algorithm needs on this engine: SURVEY 8(d)'s butterflies x 8 and "
                                  "multiply-accumulates x 7 lane-operations, MINUS the two leading stages of every forward transform on digits, "
                                  "which the b^4 + 1 fields do as one radix-4 butterfly of 10 plain operations per four values (round 4). "
                                  "`frac_survey_priced` is the figure of rounds 1-4 (the survey's count unchanged): it credits operations the "
                                  "kernel no longer executes"),
            "valu_issue": valu_issue,
            # "achieved modmul/s against a measured integer-multiply micro-benchmark peak" - the micro-benchmark is
            # tools/ubench_modmul.hip (profiles/r01_ubench_modmul.txt: one exact modular butterfly on the fp64 pipe, mulmod + add +
            # sub, 33.6 SIMD-cycles per wave = 4,678 G butterflies/s on 256 CUs at 2.4 GHz; a 64-bit integer Goldilocks butterfly
            # 137 cycles, two 31-bit Shoup primes 64); a multiply-accumulate of the external product is priced as one modmul too
            "modmul": {"achieved_G_per_s": round((bfly + macs) * avg_pbs_per_launch / avg_launch_s / 1e9, 1),
                       "ubench_peak_G_per_s": 4677.8, "unit": "exact modular multiplications (butterflies + multiply-accumulates) per second",
                       "frac": round((bfly + macs) * avg_pbs_per_launch / avg_launch_s / 1e9 / 4677.8, 4),
                       "peak_from": "profiles/r01_ubench_modmul.txt (fp_bfly, tools/ubench_modmul.hip)",
                       "note": "above the butterfly micro-benchmark is possible: multiply-accumulates sum lazily (one reduction per "
                               "column value) and the short-root stages multiply by small integers"},
            "avg_launch_ms": round(avg_launch_s * 1e3, 4), "avg_bootstraps_per_launch": round(avg_pbs_per_launch, 1),
            "traffic": traffic_bytes,
            "traffic_over_algorithmic": round(traffic_bytes / algo_bytes, 2) if traffic_bytes else None,
            "traffic_from_profile": traffic_from,
            "traffic_source": (traffic or {}).get("source"),
            "hbm": {"achieved": round(algo_bytes / avg_launch_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(algo_bytes / avg_launch_s / 1e9 / HBM_PEAK_GBS, 5),
                    "algorithmic_bytes_per_launch": int(algo_bytes),
                    "note": "BSK (shared by every ciphertext of a launch) + per-bootstrap LWE rows; not the binding resource"},
        },
        "setup_s": {"keygen_upload": round(bench.t_keys, 2)},
    })
    if world > 1 or bench.comm is not None:
        info = bench.comm.info()   # what RCCL itself reports for the library's communicator
        over_rccl = bool(info["rccl_version"])
        result["rccl_ranks"] = {"world_size": info["world_size"], "rank_of_this_line": info["rank"], "rccl_version": info["rccl_version"],
                                "communicator": ("helm_comm: ncclCommInitRank / ncclAllGather inside libhelm_hip.so (include/helm_comm.h)"
                                                 if over_rccl else
                                                 "helm_comm over a HOST TRANSPORT (" + ("device-to-device copies between rank threads of one process" if bench.threads
                                                                                    else "gloo through host memory") + "), RCCL NOT used: " +
                                                 ("rehearsal on one GPU" if bench.rehearse and not bench.comm_error else f"--allow-host-fallback after: {bench.comm_error}")),
                                "collectives_issued_by_rank_0": bench.comm.stats()["collectives"],
                                "control_plane": ("in-process (threading.Barrier between the rank threads)" if bench.threads else
                                                  f"torch.distributed {bench.dist.get_backend()} (barrier, max over ranks, unique id)"
                                                  if world > 1 else "none (one process)"),
                                "one_process_per_gpu": not bench.rehearse}
        if bench.threads:
            # (advisor, round 5) helm_comm's in-process all-gather synchronises the rank's stream and waits at two barriers ON
            # THE RANK'S HOST THREAD: with --overlap the exchange stream's collective stalls the thread that would feed the next
            # launch, so this rehearsal serialises more than RCCL does and cannot show what the overlap is worth
            result["rccl_ranks"]["in_process_exchange"] = ("blocks the rank's host thread (stream sync + two barriers per collective): "
                                                           "overlap on / off must NOT be decided from rank-thread figures")
        if bench.comm_error:
            result["rccl_error"] = bench.comm_error
            result["not_a_measurement_of_the_rccl_path"] = True
        if bench.devices:
            result["rccl_ranks"]["devices"] = [{"rank": d["rank"], "local_rank": d["local_rank"], "pci": d["pci"], "uuid": d["uuid"]}
                                               for d in bench.devices]
            result["rccl_ranks"]["checked"] = ("world size == --gpus, every communicator reports that world and its rank, RCCL carries it, "
                                               "no two ranks on one device - or the run ends with value null and rc != 0" if not bench.rehearse
                                               else "world size and communicator ranks (rehearsal: the ranks share one device by design)")
        result[head_kind] = with_baseline(bench.describe(head))   # the headline under its own name as well
    if world > 1:
        if solo is not None:
            result["solo"] = bench.describe(solo)
            result["n1_same_job_value"] = round(solo["value"], 1)
        result["planned_wall_s"] = planned
        if dropped:
            result["dropped_for_wall_budget"] = dropped
    checkpoint(result)

    # ---- N > 1: the runs that are not the headline, each under its own name -----------------------------
    for kind, st, wu in legs:
        r = guarded(bench, result, f"{kind} run", lambda k=kind, st=st, wu=wu: bench.run(k, st, wu))
        result[kind] = with_baseline(bench.describe(r))
        checkpoint(result)
    if world > 1:
        result["wall_s_total"] = round(time.time() - bench.t_start, 1)

    # ---- wall-clock of ONE AES-128 evaluation (latency; levels are 80-256 gates wide, so the
    #      GPU is far from full: this is the n-step blind-rotation chain, 207 levels deep) ------
    if world == 1:
        prog1, launches1, _, _ = make_program(bench.sk, bench.circuit, bench.wire_names, 1, quantum)
        prog1.run(head["wires"])
        bench.sync_all()
        t0 = time.perf_counter()
        prog1.run(head["wires"])
        bench.sync_all()
        t1 = time.perf_counter() - t0
        # once more with the engine's events on (outside wall_s): the launches of ONE circuit are 80 - 256 bootstraps wide, so
        # the kernel is k_pbs_wide (one bootstrap per CU on all four SIMDs) and the figure of merit is the latency of the
        # n-step chain, not the chip's throughput
        bench.sk.timing_enable(True)
        bench.sk.timing(reset=True)
        prog1.run(head["wires"])
        bench.sync_all()
        tm1 = bench.sk.timing(reset=True)
        bench.sk.timing_enable(False)
        n_l = max(1, tm1.pbs_launches)
        avg_ms, avg_w = tm1.pbs_ms / n_l, tm1.pbs_count / n_l
        ach1 = algo_ops * tm1.pbs_count / (tm1.pbs_ms * 1e-3) / 1e12
        occ = avg_w / n_cus
        cyc = avg_ms * 1e-3 * (clock_ghz or PEAK_CLOCK_GHZ) * 1e9 / p.n
        result["single_block"] = {
            "wall_s": round(t1, 4), "gate_bootstraps_per_s": round(prog1.total_pbs() / t1, 1),
            "bootstraps": int(prog1.total_pbs()), "launches": int(launches1),
            "what": "ONE AES-128 evaluation - what the reference binary runs (helm.rs:256-262); `value` above is a batch of "
                    f"{head['blocks_total']} such blocks evaluated together",
            "roofline": {
                "kernel": "k_pbs_wide<WideCfg<...>> ((k+1) l = 9 waves per bootstrap, one bootstrap per CU)" if p.N == 512 else "k_pbs_wide",
                "bound": "fp64_valu (latency of the n-step chain: a launch of <= one bootstrap per CU costs one chain whatever its width)",
                "achieved": round(ach1, 2), "peak": round(peak_tops, 2), "unit": "T fp64 lane-op/s (FMA = 1)",
                "frac": round(ach1 / peak_tops, 4),
                "occupancy": round(occ, 3),
                "frac_of_the_busy_cus": round(ach1 / (peak_tops * max(occ, 1e-9)), 4),
                "avg_launch_ms": round(avg_ms, 4), "avg_bootstraps_per_launch": round(avg_w, 1), "launches_timed": int(n_l),
                "cycles_per_step": round(cyc), "cycles_per_step_if_the_four_simds_were_perfectly_packed": 5800,
                "packed_bound_is": "5,778 vector instructions per bootstrap-step (9 waves x 642: profiles/r04/pmc_and_stats_summary.txt) x 4 cycles / 4 SIMDs",
                "cycles_per_step_clock": "the clock k_pbs held in the timed region (k_pbs_wide has no probe of its own)" if clock_ghz else "nominal",
                "algorithmic_lane_ops_per_bootstrap": int(algo_ops),
                "issue_slot_fraction": "0.61 (VALU active 0.272 per wave x 2.25 waves per SIMD: profiles/r04/pmc_and_stats_summary.txt)",
                "traffic": None,
            }}
        prog1.destroy()
        checkpoint(result)

    # ---- every BASELINE.json configuration as a wall-clock line (each one circuit, outside the timed region) ----
    if world == 1 and not args.no_configs:
        try:
            result["configs"] = baseline_configs(bench, result)
        except Exception as e:  # the headline line must survive a failure here
            result["configs"] = {"error": repr(e)}
        checkpoint(result)

    # ---- CPU baseline: the oracle (a port, not tfhe-rs) on this box's host cores --------
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, bench.ck, bench.circuit, bench.wire_names, bench.index, bench.nw,
                                              head["wires"], head["keys_pt"], gpu_value=head["value"],
                                              gpu_single_block=result["single_block"]["gate_bootstraps_per_s"])
        checkpoint(result)
    # ---- the other two modes of the reference, same GPU, outside the timed region: 3-input LUT
    #      gates (BASELINE config 3's primitive) and the chi-squared u32 netlist (config 5) ------
    if world == 1 and not args.no_other_modes:
        try:
            result["other_modes"] = other_modes(bench.local_rank)
        except Exception as e:  # the headline line must survive a failure here
            result["other_modes"] = {"error": repr(e)}
        om, cfg = result["other_modes"], result.get("configs")
        if isinstance(cfg, dict) and "error" not in om:
            cfg["3_lut_adder_8bit"] = om.pop("lut_adder_8bit", None)
            am = om["arith_mode_multibit3"]
            cfg["5_chi_squared_u32"] = {"what": "config 5: chi_squared_arith.v, arithmetic mode u32 under the reference's own multi-bit set (helm.rs:83); "
                                                "details under other_modes.arith_mode_multibit3 (classical set: other_modes.arith_mode)",
                                        "wall_s": am["wall_s"], "bootstraps": am["bootstraps"], "rounds_in_a_row": am["rounds_in_a_row"],
                                        "decrypt_ok": am["decrypt_ok"], "wall_s_classical_set": om["arith_mode"]["wall_s"]}


def _one_gates_circuit(bench, path, seed):
    """One gates-mode circuit through the evaluator API (GateCircuit: encrypt_inputs -> evaluate_encrypted, what
    helm.rs:242-262 runs): wall-clock of the evaluation, every OUTPUT wire checked against the plaintext evaluator."""
    from helm_amd import Circuit, GateCircuit, PtxtType, verilog_parser
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_file(path, False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    rng = np.random.default_rng(seed)
    vals = {w: PtxtType.Bool(int(rng.integers(0, 2))) for w in inputs}
    ptxt = {w: PtxtType.None_() for w in wire_set}
    ptxt.update(vals)
    ptxt = c.evaluate(ptxt)
    gc = GateCircuit(bench.ck, bench.sk, c)
    enc = gc.encrypt_inputs(wire_set, vals)
    gc.evaluate_encrypted(enc, 1, "bool")          # warm-up: plans and uploads the program
    bench.sk.sync()
    t0 = time.perf_counter()
    enc = gc.evaluate_encrypted(enc, 2, "bool")    # a new cycle: the same-cycle memo must not answer
    bench.sk.sync()
    dt = time.perf_counter() - t0
    ok = all(bench.ck.decrypt(enc[w]) == bool(ptxt[w].value) for w in outputs)
    pbs = gc.pbs_per_cycle()
    return {"netlist": os.path.relpath(path, ROOT), "gates": len(gates), "levels": len(c.level_map()), "bootstraps": int(pbs),
            "wall_s": round(dt, 5), "gate_bootstraps_per_s": round(pbs / dt, 1), "decrypt_ok": bool(ok)}


def baseline_configs(bench, result):
    """BASELINE.json `configs`, each as ONE circuit evaluation with its wall-clock (config 4 is the headline and
    `single_block`, config 5 `other_modes.arith_mode`; config 3 is filled in by other_modes(), which owns the 64-bit keys)."""
    nets = os.path.join(ROOT, "tests", "netlists")
    res = {}
    c1 = _one_gates_circuit(bench, os.path.join(nets, "2-bit-adder.v"), 1)
    c1["what"] = "config 1: 2-bit adder, gates mode (the reference's own CPU-runnable case, here on the GPU path; K-1's netlist)"
    res["1_2bit_adder_gates"] = c1
    c2 = _one_gates_circuit(bench, os.path.join(nets, "alu-c880-class.v"), 0x48454C4D)
    c2["what"] = ("config 2: c880-CLASS STAND-IN (a generated 8-bit ALU netlist of ISCAS'85 c880's size; c880.v itself is not "
                  "obtainable offline), gates mode, one circuit: its levels are narrower than one bootstrap per CU, so the wall-clock "
                  "is levels x the latency of one bootstrap chain (k_pbs_wide)")
    res["2_c880_class_gates"] = c2
    res["4_aes128_gates"] = {"see": "value (32-block batch) and single_block (ONE evaluation)",
                             "wall_s": result.get("single_block", {}).get("wall_s"),
                             "gate_bootstraps_per_s": result.get("single_block", {}).get("gate_bootstraps_per_s"),
                             "decrypt_ok": True}
    return res


def side_kinds(head_kind):
    return [k for k in ("strong", "weak", "sharded_weak") if k != head_kind]


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    threads = os.environ.get("HELM_BENCH_REHEARSE") == "threads" and args.gpus > 1
    if "RANK" not in os.environ and args.gpus > 1 and not threads:
        sys.exit(launch_workers(args, argv))
    globals()["np"] = __import__("numpy")  # workers only: the launcher stays on the standard library
    if threads:
        sys.exit(thread_workers(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    sys.exit(worker(args))


def granted_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (the 1-GPU box shows
    every logical CPU in the mask but grants a share)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, ck, circuit, wire_names, index, nw, wires, keys_pt, cpu_blocks=8, gpu_value=None, gpu_single_block=None):
    """The oracle's SIMD route (oracle/fp_route.inc: exact fp64-FMA NTT, one gate per SIMD lane, OpenMP over
    groups of gates like rayon over a level) on the first levels of the same netlist, `cpu_blocks` blocks of the
    GPU's batch, on the cores this process is granted; the GPU's ciphertexts of those levels must be bit-identical
    (the oracle is the checker here, never the product)."""
    import oracle
    p = ck.params
    cpu_blocks = min(cpu_blocks, len(keys_pt))
    orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=False, use_fp=True)
    o_ops, o_i0, o_i1, o_i2, o_out, o_off, _ = build_program_arrays(circuit, wire_names, cpu_blocks)
    host = np.zeros((nw * cpu_blocks, p.n + 1), dtype=np.uint32)
    rows = np.array([b * nw + index[f"{w}[{i}]"] for b in range(cpu_blocks) for w in ("key", "pt") for i in range(128)], np.int32)
    host[rows] = wires.download(rows)  # the very ciphertexts the GPU evaluated (blocks 0..cpu_blocks-1)
    granted = granted_cores()
    threads = max(1, args.cpu_threads or granted)
    n_pbs, L = 0, 0
    t0 = time.perf_counter()
    while L < len(o_off) - 1 and (L < 1 or time.perf_counter() - t0 < args.cpu_seconds):
        s = slice(o_off[L], o_off[L + 1])
        orc.eval_level_fp(host, o_ops[s], o_i0[s], o_i1[s], o_i2[s], o_out[s], nthreads=threads)
        n_pbs += int(np.sum(o_ops[s] != oracle.NOT))
        L += 1
    cpu_s = time.perf_counter() - t0
    same = bool(np.array_equal(wires.download(o_out[:o_off[L]]), host[o_out[:o_off[L]]]))
    if not same:
        raise SystemExit("GPU ciphertexts differ from the CPU oracle on the sampled levels")
    value = n_pbs / cpu_s
    share = f"{granted} of {os.cpu_count()} logical CPUs (affinity mask capped by the cgroup quota)"
    res = {
        # the portable figures first: one thread's rate does not depend on the share of the host this run was granted
        "per_core": round(value / threads, 2), "ms_per_gate_per_thread": round(cpu_s * threads / n_pbs * 1e3, 2),
        "value": round(value, 2), "unit": "gate-bootstraps/s", "cores": threads, "kind": "port",
        "host_share": share,
        "sample": f"first {L} level(s) of the same AES-128 netlist, {cpu_blocks} block(s) of the GPU's batch ({n_pbs} gate-bootstraps, "
                  f"{cpu_s:.1f} s); {oracle.ntt_route_name()}; prime {orc.fp_prime():#x}; OpenMP over groups of gates of a level; NOT tfhe-rs",
        "gpu_ciphertexts_bit_identical_on_sample": same,
    }
    if gpu_value:
        res["gpu_over_cpu"] = {"batched": round(gpu_value / value, 1),
                               "one_aes_block": round(gpu_single_block / value, 1) if gpu_single_block else None,
                               "against": f"{threads} threads on {share}: the ratio scales with the share, the per-core figures do not",
                               "gpu_over_one_core": round(gpu_value / (value / threads), 0)}
    return res


def si_algo_ops(n, k, N, pbs_l, g=1):
    """ALGORITHMIC fp64 lane-operations of one 64-bit-torus bootstrap: two CRT fields; per blind-rotation step
    ((k+1) l forward + (k+1) inverse) transforms of N/2 log2 N butterflies (x 8: exact 6-operation modular
    multiplication + add + sub) and (k+1)^2 l N multiply-accumulates (x 7); n steps, or n/g group steps for the
    multi-bit rotation, whose key of a step is the sum over the 2^g subsets of monomial x GGSW:
    2^g (k+1)^2 l N more multiply-accumulates per field."""
    K1, logN = k + 1, N.bit_length() - 1
    bfly = (K1 * pbs_l + K1) * (N // 2) * logN * 2
    macs = K1 * K1 * pbs_l * N * 2
    steps = n
    if g > 1:
        macs += (1 << g) * K1 * K1 * pbs_l * N * 2
        steps = n // g
    return steps * (bfly * 8 + macs * 7)


def si_roofline(kernel, algo_ops, bootstraps, kernel_ms, launches, n_cus, bsk_bytes, io_bytes):
    """`roofline` of a 64-bit-torus bootstrap kernel from the engine's own HIP events (helm_si_get_timing)."""
    peak = n_cus * 64 * PEAK_CLOCK_GHZ * 1e9 / 1e12
    ach = algo_ops * bootstraps / (kernel_ms * 1e-3) / 1e12
    algo_bytes = bsk_bytes * launches + bootstraps * io_bytes
    return {"kernel": kernel, "bound": "fp64_valu", "achieved": round(ach, 2), "peak": round(peak, 2),
            "unit": "T fp64 lane-op/s (FMA = 1)", "frac": round(ach / peak, 4),
            "algorithmic_lane_ops_per_bootstrap": int(algo_ops), "bootstraps": int(bootstraps),
            "kernel_ms": round(kernel_ms, 3), "launches": int(launches),
            "traffic": None,
            "hbm": {"achieved": round(algo_bytes / (kernel_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(algo_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "algorithmic_bytes": int(algo_bytes), "note": "key once per launch + rows and tables; not the binding resource"}}


def other_modes(device):
    """LUT mode under PARAM_MESSAGE_2_CARRY_2 (classical blind rotation) and arithmetic mode under both that set
    and the reference's own PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3 (helm.rs:83, multi-bit blind rotation)."""
    res = _other_modes_set(device, "shortint_m2c2", "PARAM_MESSAGE_2_CARRY_2_KS_PBS [dimensions recalled; the reference binary's "
                           "LUT mode names PARAM_MESSAGE_1_CARRY_1 (helm.rs:301), which cannot hold 3-input LUT indices]")
    mb = _other_modes_set(device, "shortint_m2c2_multibit3", "dimensions of PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS "
                          "(helm.rs:83) [recalled; LWE noise extrapolated, not tfhe's value: approximate set]")
    res["lut_mode_multibit3"], res["arith_mode_multibit3"] = mb["lut_mode"], mb["arith_mode"]
    res["lut_mode_m1c1"] = _lut_m1c1_leg(device)
    res["wide_lut_wopbs"] = _wide_lut_leg(device)
    return res


def _lut_m1c1_leg(device):
    """LUT mode as the reference BINARY configures it (helm.rs:301: PARAM_MESSAGE_1_CARRY_1_KS_PBS, k = 3, N = 512): 2,048
    independent 2-input LUT gates (XOR; four rounds of two ciphertexts per CU, as the 1,024 of the m2c2 leg are four of one), the bivariate form of gates::lut() (gates.rs:761-764), on k_pbs64k."""
    import helm_amd
    import torch
    ck, sk = helm_amd.gen_keys_shortint("shortint_m1c1", seed=1, device=device)
    B = 2048
    bits = np.random.default_rng(0).integers(0, 2, size=2 * B).astype(np.uint64)
    w = sk.wires(3 * B)
    w.upload(np.arange(2 * B), ck.encrypt(bits))
    in_idx = np.arange(2 * B, dtype=np.int32).reshape(2, B).T.copy()
    ar, tb, out = np.full(B, 2, np.int32), np.full(B, 0x6, np.uint64), np.arange(2 * B, 3 * B, dtype=np.int32)
    w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    sk.timing_enable(True)
    sk.timing(reset=True)
    t0 = time.perf_counter()
    w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    dt = time.perf_counter() - t0
    tm = sk.timing(reset=True)
    ok = bool(np.array_equal(ck.decrypt(w.download(out)), bits[:B] ^ bits[B:]))
    p = ck.params
    K1 = p.k + 1
    n_cus = torch.cuda.get_device_properties(device).multi_processor_count
    res = {"workload": f"{B} independent 2-input LUT gates (keyswitch + programmable bootstrap), PARAM_MESSAGE_1_CARRY_1_KS_PBS - the set "
                       "the reference binary installs for LUT mode (helm.rs:301) [dimensions recalled; GLWE noise interpolated: approximate set]",
           "params": {"n": p.n, "k": p.k, "N": p.N, "pbs_l": p.pbs_l, "pbs_logB": p.pbs_logB, "ks_l": p.ks_l, "ks_logB": p.ks_logB},
           "luts_per_s": round(B / dt, 1), "decrypt_ok": ok,
           # round 6: the CRT pair follows the loaded key (helm_si_field_bits): 46 = 2736^4 + 1, 2872^4 + 1 for a generated key
           "crt_pair_bits": sk.field_bits(),
           "kernel_ms": {"k_pbs64k": round(tm.pbs_ms, 3), "keyswitch": round(tm.ks_ms, 3), "linear": round(tm.linear_ms, 3)},
           "roofline": si_roofline("k_pbs64k<Pbs64kCfg<9, 3>, " + ("FpJ, FpJ2>" if sk.field_bits() == 46 else "FpG, FpG2>"), si_algo_ops(p.n, p.k, p.N, p.pbs_l), tm.pbs_count, tm.pbs_ms,
                                   tm.pbs_launches, n_cus, p.n * p.pbs_l * K1 * K1 * p.N * 8 * 2,
                                   (p.n + 1) * 8 + (p.k * p.N + 1) * 8 + p.N * 8)}
    sk.close()
    return res


def _wide_lut_leg(device):
    """Wide LUT gates through the WoP-PBS path (high_precision_lut, reference src/gates.rs:787-815; never called by
    the reference): 256 six-input gates under the LUT-mode encoding the reference names (message_modulus =
    carry_modulus = 2, helm.rs:301), two bits extracted per block as tfhe's degree bookkeeping gives after the
    cleaning bootstrap, and one bit per block (enough for LUT mode's one-bit wires)."""
    import helm_amd
    from helm_amd import wopbs
    from helm_amd.shortint import si_named_params
    sp, sa, sb = si_named_params("shortint_m2c2")
    wp, wa, wb = wopbs.wop_named_params("wopbs_m1c1")
    sp.message_modulus, sp.carry_modulus = wp.message_modulus, wp.carry_modulus
    ck = helm_amd.SiClientKey(sp, sa, sb, seed=1)
    wk = wopbs.WopClientKey(ck, wp, wa, wb, seed=2)
    sk = helm_amd.SiServerKey(ck, device=device)
    wsk = wopbs.WopServerKey(sk, wk)
    rng = np.random.default_rng(0)
    import torch
    n_cus = torch.cuda.get_device_properties(device).multi_processor_count
    G, m = 256, 6
    truth = rng.integers(0, 2, size=1 << m, dtype=np.uint64)
    xs = rng.integers(0, 1 << m, size=G)
    bits_in = np.array([[(x >> (m - 1 - q)) & 1 for q in range(m)] for x in xs], dtype=np.uint64)
    w = sk.wires(G * (m + 1))
    w.upload(np.arange(G * m), ck.encrypt(bits_in.reshape(-1)))
    in_idx = np.arange(G * m, dtype=np.int32).reshape(G, m)
    out_idx = np.arange(G * m, G * (m + 1), dtype=np.int32)
    res = {"workload": f"{G} independent {m}-input LUT gates through bit extraction, circuit bootstrap and vertical packing; "
                       "PBS side: dimensions of PARAM_MESSAGE_2_CARRY_2_KS_PBS with message_modulus = carry_modulus = 2, "
                       "WoP side: WOPBS_PARAM_MESSAGE_1_CARRY_1_KS_PBS [dimensions recalled]"}
    for b in (2, 1):
        wsk.eval_luts(w, in_idx, truth, out_idx, bits_per_block=b)
        sk.sync()
        wsk.timing(reset=True)
        t0 = time.perf_counter()
        wsk.eval_luts(w, in_idx, truth, out_idx, bits_per_block=b)
        sk.sync()
        dt = time.perf_counter() - t0
        t = wsk.timing()
        ok = [int(v) for v in ck.decrypt_message_and_carry(w.download(out_idx))] == [int(truth[x]) for x in xs]
        res[f"bits_per_block_{b}"] = {"gates_per_s": round(G / dt, 1), "wall_s": round(dt, 4), "bootstraps": t["bootstraps"],
                                      "bootstraps_per_s": round(t["bootstraps"] / dt, 1), "decrypt_ok": ok,
                                      "stage_ms": {k[:-3]: round(v, 2) for k, v in t.items() if k.endswith("_ms")}}
        # the stage that is the path: G x (m b) x cbs_l WoP-side bootstraps (N = 2048, two levels) on k_pbs64s
        n_cbs = G * m * b * wp.cbs_l
        K1 = wp.k + 1
        res[f"bits_per_block_{b}"]["roofline"] = si_roofline(
            "k_pbs64s<Pbs64sCfg<11, 2>> (circuit-bootstrap stage)", si_algo_ops(wp.n, wp.k, wp.N, wp.pbs_l), n_cbs,
            t["cbs_pbs_ms"], max(1, -(-n_cbs // 4096)), n_cus, wp.n * wp.pbs_l * K1 * K1 * wp.N * 8 * 2,
            (wp.n + 1) * 8 + (wp.k * wp.N + 1) * 8 + wp.N * 8)
    wsk.close()
    sk.close()
    return res


def _lut_adder_leg(ck, sk):
    """BASELINE config 3: 8-bit-adder-lut-3-1.v (the reference's LUT test, circuit_test.rs:266-311) as ONE circuit through
    LutCircuit, wall-clock of the evaluation; every wire must decrypt to the plaintext evaluator's value."""
    from helm_amd import Circuit, EvalCircuit, LutCircuit, PtxtType, verilog_parser
    gs, ws, ins, outs, d, _, _ = verilog_parser.read_verilog_file(os.path.join(ROOT, "tests", "netlists", "8-bit-adder-lut-3-1.v"), False)
    c = Circuit(gs, ins, outs, d)
    c.sort_circuit()
    c.compute_levels()
    a, b, cin = 0xB7, 0x6E, 1
    inputs = {f"a[{i}]": PtxtType.Bool((a >> i) & 1) for i in range(8)}
    inputs.update({f"b[{i}]": PtxtType.Bool((b >> i) & 1) for i in range(8)})
    inputs["cin"] = PtxtType.Bool(cin)
    ptxt = c.evaluate(c.initialize_wire_map(ws, inputs, "bool"))
    lc = LutCircuit(ck, sk, c)
    enc = EvalCircuit.encrypt_inputs(lc, ws, inputs)
    EvalCircuit.evaluate_encrypted(lc, enc, 1, "bool")
    sk.sync()
    t0 = time.perf_counter()
    out = EvalCircuit.evaluate_encrypted(lc, enc, 2, "bool")
    sk.sync()
    dt = time.perf_counter() - t0
    ok = all(ck.decrypt(out[w]) == int(bool(v)) for w, v in ptxt.items())
    return {"what": "config 3: 8-bit-adder-lut-3-1.v, LUT mode (3-input look-ups = keyswitch + programmable bootstrap on the 64-bit torus), "
                    "one circuit: a carry chain of 8 levels x 2 look-ups, so the wall-clock is 8 x the latency of one bootstrap (k_pbs64s)",
            "netlist": "tests/netlists/8-bit-adder-lut-3-1.v", "luts": int(lc.pbs_per_cycle()), "levels": len(c.level_map()),
            "wall_s": round(dt, 5), "luts_per_s": round(lc.pbs_per_cycle() / dt, 1), "decrypt_ok": bool(ok)}


def _other_modes_set(device, set_name, tfhe_name):
    import helm_amd
    from helm_amd import ArithCircuit, Circuit, PtxtType, verilog_parser
    ck, sk = helm_amd.gen_keys_shortint(set_name, seed=1, device=device)
    B = 1024
    bits = np.random.default_rng(0).integers(0, 2, size=3 * B).astype(np.uint64)
    w = sk.wires(4 * B)
    w.upload(np.arange(3 * B), ck.encrypt(bits))
    in_idx = np.arange(3 * B, dtype=np.int32).reshape(3, B).T.copy()
    ar, tb, out = np.full(B, 3, np.int32), np.full(B, 0xE8, np.uint64), np.arange(3 * B, 4 * B, dtype=np.int32)
    w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    sk.timing_enable(True)
    sk.timing(reset=True)
    t0 = time.perf_counter()
    w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    dt = time.perf_counter() - t0
    tm = sk.timing(reset=True)
    sk.timing_enable(False)
    ok = bool(np.array_equal(ck.decrypt(w.download(out)), (bits[:B] + bits[B:2 * B] + bits[2 * B:]) >= 2))
    p = ck.params
    import torch
    n_cus = torch.cuda.get_device_properties(device).multi_processor_count
    g = max(1, p.grouping_factor)
    K1 = p.k + 1
    bsk_bytes = (p.n // g) * (1 << g if g > 1 else 1) * p.pbs_l * K1 * K1 * p.N * 8 * 2
    io_bytes = (p.n + 1) * 8 + (p.k * p.N + 1) * 8 + p.N * 8     # small LWE in, big LWE out, test polynomial
    res = {"lut_mode": {"workload": f"{B} independent 3-input LUT gates (keyswitch + programmable bootstrap), " + tfhe_name,
                        "luts_per_s": round(B / dt, 1), "decrypt_ok": ok,
                        "kernel_ms": {"k_pbs64s": round(tm.pbs_ms, 3), "keyswitch": round(tm.ks_ms, 3), "linear": round(tm.linear_ms, 3)},
                        "roofline": si_roofline("k_pbs64s" + (f" (multi-bit, g = {g})" if g > 1 else ""),
                                                si_algo_ops(p.n, p.k, p.N, p.pbs_l, g), tm.pbs_count, tm.pbs_ms,
                                                tm.pbs_launches, n_cus, bsk_bytes, io_bytes)}}
    if set_name == "shortint_m2c2":
        res["lut_adder_8bit"] = _lut_adder_leg(ck, sk)
    gs, ws, i, o, d, _, _ = verilog_parser.read_verilog_file(os.path.join(ROOT, "tests", "netlists", "chi_squared_arith.v"), True)
    c = Circuit(gs, i, o, d)
    c.sort_circuit()
    c.compute_levels()
    ac = ArithCircuit(ck, sk, c)
    enc = ac.encrypt_inputs(ws, {"N0": PtxtType.U32(2), "N1": PtxtType.U32(7), "N2": PtxtType.U32(9)})
    want = {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250}
    # the default evaluation: sub-circuits that share no wire (alpha's and the betas') run concurrently as chains on the
    # server key's context, their look-up rounds merged (helm::RoundMerger)
    ac.evaluate_encrypted(enc, 1, "u32")
    t0 = time.perf_counter()
    outl = ac.evaluate_encrypted(enc, 2, "u32")  # a new cycle each time: the same-cycle memo (gates.rs:307-312) must not answer
    dtl = time.perf_counter() - t0
    decl = {k: int(v.value) for k, v in ac.decrypt_outputs(outl, True).items()}
    res["arith_mode"] = {"workload": "chi_squared_arith.v, u32 (16 radix blocks per integer), " + tfhe_name, "wall_s": round(dtl, 4),
                         "bootstraps": ac.pbs_per_cycle(), "rounds_in_a_row": ac.pbs_rounds_per_cycle(),
                         "evaluation": "the default: two independent sub-circuits as chains on one context, their look-up rounds merged into launches of at most one ciphertext per CU (rounds_in_a_row = launches)", "decrypt_ok": decl == want}
    # the same evaluation once more with the engine's events on (outside wall_s): what the bootstrap kernel achieves when
    # its launches are as wide as the circuit allows, not as wide as the chip
    sk.timing_enable(True)
    sk.timing(reset=True)
    ac.evaluate_encrypted(enc, 5, "u32")
    tm = sk.timing(reset=True)
    sk.timing_enable(False)
    rf = si_roofline("k_pbs64s" + (f" (multi-bit, g = {g})" if g > 1 else ""), si_algo_ops(p.n, p.k, p.N, p.pbs_l, g), tm.pbs_count,
                     tm.pbs_ms, tm.pbs_launches, n_cus, bsk_bytes, io_bytes)
    rf["note"] = (f"{tm.pbs_count} bootstraps in {tm.pbs_launches} launches on {n_cus} CUs that hold one ciphertext each: the circuit's "
                  "width, not the kernel, sets this fraction (a launch costs one bootstrap's time whatever it holds)")
    rf["occupancy"] = round(tm.pbs_count / max(1, tm.pbs_launches * n_cus), 3)
    res["arith_mode"]["roofline"] = rf
    # level by level, as the reference joins every level (circuit.rs:1321): identical ciphertexts, more rounds in a row
    ac.set_lanes(1)
    ac.evaluate_encrypted(enc, 3, "u32")
    t0 = time.perf_counter()
    outm = ac.evaluate_encrypted(enc, 4, "u32")
    dt = time.perf_counter() - t0
    dec = {k: int(v.value) for k, v in ac.decrypt_outputs(outm, True).items()}
    res["arith_mode"]["level_by_level"] = {"wall_s": round(dt, 4), "batched_rounds": ac.pbs_rounds_per_cycle(),
                                           "bit_identical_to_lanes": all(np.array_equal(outm[k], outl[k]) for k in outm.keys()),
                                           "decrypt_ok": dec == want}
    sk.close()
    return res


if __name__ == "__main__":
    main()

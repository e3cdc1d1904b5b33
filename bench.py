#!/usr/bin/env python3
"""bench.py — encrypted gate-bootstraps/sec on the AES-128 gates-mode netlist.

A step = one full evaluation of the (generated, stand-in) AES-128 netlist over a batch of independent input
blocks, inputs already encrypted and resident in HBM.  The level schedule of the batch is launch-packed
(helm_amd/csrc/host/level_pack.cpp): the blocks share no wires, so launches hold whole lockstep rounds and only
the drain at the end of a pass is partial.

  --scaling weak   (default)  every GPU evaluates its own `--blocks` blocks: the partition the workload offers
                   (independent blocks), no data-path collective; N GPUs = N x blocks.
  --scaling strong            `--blocks` blocks in total, every launch sharded across the N GPUs, keys and wire
                   table replicated, the launch's output ciphertexts all-gathered over RCCL
                   (helm_amd/distributed.py) - the north star's per-level shard, the mode that shortens ONE job.
With N > 1 the weak run also times a short strong-scaling pass (outside the timed region) and reports it under
"strong_scaling", so one driver invocation exercises RCCL.

Contract: python bench.py --gpus N --steps K --warmup W   (N>1 under torch.distributed.run)
prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_CLOCK_GHZ = 2.4          # MI355X engine clock the nominal peaks are quoted at
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r02", "pmc_traffic.json")  # tools/pmc_traffic.sh, separate --pmc passes


def build_program_arrays(circuit, wire_names, blocks):
    """Tile the one-block level schedule over `blocks` copies of the wire table."""
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


def make_program(sk, circuit, wire_names, blocks, quantum, pack=True):
    """-> (Program, launches, levels): the batch's level schedule, launch-packed to `quantum` bootstraps."""
    import helm_amd
    from helm_amd.distributed import pack_levels
    ops, i0, i1, i2, out, off, _ = build_program_arrays(circuit, wire_names, blocks)
    levels = len(off) - 1
    if pack:
        ops, i0, i1, i2, out, off, _ = pack_levels(ops, i0, i1, i2, out, off, quantum)
    return helm_amd.Program(sk, ops, i0, i1, i2, out, off), len(off) - 1, levels


def upload_inputs(ck, wires, index, nw, keys_pt, first_block=0):
    in_rows, in_bits = [], []
    for b, (key, pt) in enumerate(keys_pt):
        kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
        for i in range(128):
            in_rows += [(first_block + b) * nw + index[f"key[{i}]"], (first_block + b) * nw + index[f"pt[{i}]"]]
            in_bits += [(kv >> i) & 1, (pv >> i) & 1]
    wires.upload(np.array(in_rows, np.int32), ck.encrypt(np.array(in_bits, dtype=bool)))


def check_outputs(ck, wires, index, nw, keys_pt, what):
    from helm_amd.netlists import aes128_reference_encrypt
    out_rows = np.array([b * nw + index[f"ct[{i}]"] for b in range(len(keys_pt)) for i in range(128)], np.int32)
    dec = ck.decrypt(wires.download(out_rows)).reshape(len(keys_pt), 128)
    for b, (key, pt) in enumerate(keys_pt):
        got = sum(int(dec[b, i]) << i for i in range(128)).to_bytes(16, "big")
        if got != aes128_reference_encrypt(key, pt):
            raise SystemExit(f"{what}: decrypted AES output of block {b} is WRONG")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=32, help="AES blocks per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--params", default="boolean_default")
    ap.add_argument("--no-pack", action="store_true", help="level-synchronous launches (the round-1 schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the LUT-mode / arithmetic-mode side measurements")
    ap.add_argument("--strong-leg-timeout", type=float, default=240.0,
                    help="N > 1: seconds after which the extra strong-scaling pass is abandoned and the line printed without it")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1, weak: skip the short strong-scaling pass")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="CPU baseline: stop after the level that passes this time")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline threads (0 = the cores this process may use)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch
    import torch.distributed as dist
    import helm_amd
    from helm_amd import Circuit, verilog_parser
    from helm_amd.distributed import GpuLevelExecutor, ShardedRunner
    from helm_amd.netlists import aes128

    # rehearsal of the N > 1 path on a one-GPU box: HELM_BENCH_REHEARSE=1 puts every rank on cuda:0 and
    # carries the collectives over gloo (RCCL needs one GPU per rank); never used for reported numbers
    rehearse = os.environ.get("HELM_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- keys (identical on every rank: same deterministic benchmark seed) and engine ---
    t0 = time.time()
    ck = helm_amd.ClientKey.generate(args.params, seed=1)
    sk = helm_amd.ServerKey(ck, device=local_rank)
    sk.set_stream(torch.cuda.current_stream().cuda_stream)
    p = ck.params
    t_keys = time.time() - t0
    quantum = sk.launch_quantum()

    # ---- netlist -> launch schedule over this rank's batch -----------------------------
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
    circuit = Circuit(gates, inputs, outputs, dffs)
    circuit.sort_circuit()
    circuit.compute_levels()
    wire_names = list(inputs) + sorted(wire_set)
    nw = len(wire_names)
    index = {w: i for i, w in enumerate(wire_names)}
    strong = args.scaling == "strong" and world > 1
    my_blocks = args.blocks                         # blocks in this rank's wire table
    total_blocks = args.blocks if (strong or world == 1) else args.blocks * world
    prog, launches, levels = make_program(sk, circuit, wire_names, my_blocks, quantum * (world if strong else 1),
                                          pack=not args.no_pack)
    pbs_per_pass = prog.total_pbs()                 # of this rank's table (strong: the whole job)

    # ---- synthetic inputs: seeded random key / plaintext per block, encrypted on the host,
    #      uploaded once: resident in HBM before the timed region ------------------------
    rng = np.random.default_rng(0x48454C4D + (0 if strong else rank))
    keys_pt = [(bytes(rng.integers(0, 256, 16, dtype=np.uint8)), bytes(rng.integers(0, 256, 16, dtype=np.uint8)))
               for _ in range(my_blocks)]
    if rank == 0 or strong:
        keys_pt[0] = (bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff"))  # FIPS-197 C.1
    wires = sk.wires(nw * my_blocks)
    upload_inputs(ck, wires, index, nw, keys_pt)

    runner = ShardedRunner(GpuLevelExecutor(prog, wires), rank, world if strong else 1, dist if strong else None)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.run()
    sync_all()
    sk.timing_enable(True)
    sk.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.run()
    sync_all()
    elapsed = time.perf_counter() - t0
    tm = sk.timing(reset=True)
    sk.timing_enable(False)
    clock_ghz = sk.kernel_clock_ghz()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was timed: every block decrypts to AES(key, pt) ------------
    check_outputs(ck, wires, index, nw, keys_pt, f"rank {rank}")

    # ---- N > 1, weak: a short strong-scaling pass on a fixed job (outside the timed region), AFTER rank 0 has the
    #      headline line ready.  It must never cost that line: an exception is caught on every rank, and a collective
    #      that hangs is cut off by a watchdog (rank 0 then prints the line without this leg and every rank exits).
    want_strong_leg = world > 1 and not strong and not args.no_strong_leg

    def guarded_strong_leg(on_timeout):
        import threading
        finished = threading.Event()

        def watchdog():
            if not finished.wait(args.strong_leg_timeout):
                on_timeout()
                sys.stdout.flush()
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            # a fixed job of 2 x --blocks blocks: its launches hold >= 8 x 1,024 ready bootstraps, so that up to 8 ranks
            # each get whole lockstep rounds (a 32-block job leaves 640 per rank per launch at N = 8)
            return strong_scaling_leg(sk, ck, circuit, wire_names, index, nw, 2 * args.blocks, quantum, rank, world, dist, torch)
        except Exception as e:
            return {"error": repr(e)}
        finally:
            finished.set()

    if rank != 0:
        if world > 1:
            if want_strong_leg:
                guarded_strong_leg(lambda: None)
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    job_pbs = pbs_per_pass if (strong or world == 1) else pbs_per_pass * world
    value = job_pbs * args.steps / elapsed

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
    n_cus = quantum // 4
    peak_tops = n_cus * 64 * PEAK_CLOCK_GHZ * 1e9 / 1e12
    achieved_tops = algo_ops * avg_pbs_per_launch / avg_launch_s / 1e12
    traffic = None
    if os.path.exists(PMC_TRAFFIC):
        with open(PMC_TRAFFIC) as f:
            traffic = json.load(f)
    traffic_bytes = None
    if traffic:
        # bytes per bootstrap measured at the launch size of the committed profile, scaled to this run's average launch
        traffic_bytes = int(traffic["bytes_per_launch"] * avg_pbs_per_launch / traffic["bootstraps_per_launch"])

    result = {
        "metric": "encrypted gate-bootstraps/sec on AES-128 gates-mode netlist",
        "value": round(value, 1),
        "unit": "gate-bootstraps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u32 torus (exact NTT in f64 FMA over a 49-bit prime)",
        "data": "synthetic: generated AES-128 netlist (stand-in for HELM's), seeded random keys/plaintexts, "
                "fresh encryptions resident in HBM" + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearse else ""),
        "config": {
            "workload": f"AES-128 gates-mode netlist, {my_blocks} block(s) " + ("in total, every launch sharded" if strong else "per GPU")
                        + (", launch-packed" if not args.no_pack else ", level-synchronous"),
            "params": args.params, "n": p.n, "k": p.k, "N": p.N, "pbs_l": p.pbs_l, "pbs_logB": p.pbs_logB,
            "ks_l": p.ks_l, "ks_logB": p.ks_logB,
            "levels": levels, "launches_per_step": launches, "bootstraps_per_step": int(job_pbs), "blocks_total": total_blocks,
            "parallelism": ("single GPU" if world == 1 else
                            f"launch-shard x{world} + all-gather of launch outputs (RCCL)" if strong else
                            f"block-parallel x{world}: independent blocks per GPU, keys replicated, no data-path collective"),
            "sharded_launches": len(runner.sharded_levels),
            "exchanged_MB_per_step": round(runner.exchanged_bytes_per_pass() / 1e6, 2),
        },
        "wall_s_per_step": round(elapsed / args.steps, 4),
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
            "frac_at_held_clock": round(achieved_tops / (peak_tops * clock_ghz / PEAK_CLOCK_GHZ), 4) if clock_ghz else None,
            "algorithmic_lane_ops_per_bootstrap": int(algo_ops),
            "avg_launch_ms": round(avg_launch_s * 1e3, 4), "avg_bootstraps_per_launch": round(avg_pbs_per_launch, 1),
            "traffic": traffic_bytes,
            "traffic_over_algorithmic": round(traffic_bytes / algo_bytes, 2) if traffic_bytes else None,
            "traffic_source": (traffic or {}).get("source"),
            "hbm": {"achieved": round(algo_bytes / avg_launch_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(algo_bytes / avg_launch_s / 1e9 / HBM_PEAK_GBS, 5),
                    "algorithmic_bytes_per_launch": int(algo_bytes),
                    "note": "BSK (shared by every ciphertext of a launch) + per-bootstrap LWE rows; not the binding resource"},
        },
        "setup_s": {"keygen_upload": round(t_keys, 2)},
    }

    # ---- wall-clock of ONE AES-128 evaluation (latency; levels are 80-256 gates wide, so the
    #      GPU is far from full: this is the n-step blind-rotation chain, 207 levels deep) ------
    if world == 1:
        prog1, _, _ = make_program(sk, circuit, wire_names, 1, quantum)
        prog1.run(wires)
        sync_all()
        t0 = time.perf_counter()
        prog1.run(wires)
        sync_all()
        t1 = time.perf_counter() - t0
        result["single_block"] = {"wall_s": round(t1, 4), "gate_bootstraps_per_s": round(prog1.total_pbs() / t1, 1),
                                  "bootstraps": int(prog1.total_pbs())}
        prog1.destroy()

    # ---- CPU baseline: the oracle (a port, not tfhe-rs) on this box's host cores --------
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, ck, circuit, wire_names, index, nw, wires, keys_pt)
    # ---- the other two modes of the reference, same GPU, outside the timed region: 3-input LUT
    #      gates (BASELINE config 3's primitive) and the chi-squared u32 netlist (config 5) ------
    if world == 1 and not args.no_other_modes:
        try:
            result["other_modes"] = other_modes(local_rank)
        except Exception as e:  # the headline line must survive a failure here
            result["other_modes"] = {"error": repr(e)}
    if want_strong_leg:
        result["strong_scaling"] = guarded_strong_leg(
            lambda: print(json.dumps(dict(result, strong_scaling={"error": f"no answer within {args.strong_leg_timeout} s"}))))
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def strong_scaling_leg(sk, ck, circuit, wire_names, index, nw, blocks, quantum, rank, world, dist, torch, steps=2):
    """A fixed job of `blocks` AES blocks, every packed launch sharded across the ranks (helm_amd/distributed.py):
    the regime launch-sharding exists for.  Same inputs on every rank (replicated wire table)."""
    from helm_amd.distributed import GpuLevelExecutor, ShardedRunner
    prog, launches, _ = make_program(sk, circuit, wire_names, blocks, quantum * world)
    rng = np.random.default_rng(0x57A0)
    keys_pt = [(bytes(rng.integers(0, 256, 16, dtype=np.uint8)), bytes(rng.integers(0, 256, 16, dtype=np.uint8)))
               for _ in range(blocks)]
    wires = sk.wires(nw * blocks)
    upload_inputs(ck, wires, index, nw, keys_pt)
    runner = ShardedRunner(GpuLevelExecutor(prog, wires), rank, world, dist, time_collective=True)
    runner.run()
    dist.barrier()
    torch.cuda.synchronize()
    runner.collective_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.run()
    dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    check_outputs(ck, wires, index, nw, keys_pt, f"strong-scaling leg, rank {rank}")
    res = {"workload": f"{blocks} AES-128 blocks in total, every launch sharded over {world} GPUs", "steps": steps,
           "ms_per_step": round(el / steps * 1e3, 3), "value": round(prog.total_pbs() * steps / el, 1),
           "unit": "gate-bootstraps/s", "launches_per_step": launches, "sharded_launches": len(runner.sharded_levels),
           "exchanged_MB_per_step": round(runner.exchanged_bytes_per_pass() / 1e6, 2),
           "collective_ms_per_step": round(runner.collective_ms(reset=True) / steps, 3),
           "decrypt_check": "all blocks == software AES on every rank"}
    prog.destroy()
    wires.free()
    return res


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


def cpu_baseline(args, ck, circuit, wire_names, index, nw, wires, keys_pt, cpu_blocks=8):
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
    return {
        "value": round(n_pbs / cpu_s, 2), "unit": "gate-bootstraps/s", "cores": threads, "kind": "port",
        "per_core": round(n_pbs / cpu_s / threads, 2), "ms_per_gate_per_thread": round(cpu_s * threads / n_pbs * 1e3, 2),
        "sample": f"first {L} level(s) of the same AES-128 netlist, {cpu_blocks} block(s) of the GPU's batch ({n_pbs} gate-bootstraps, "
                  f"{cpu_s:.1f} s); {oracle.ntt_route_name()}; prime {orc.fp_prime():#x}; OpenMP over groups of gates of a level; NOT tfhe-rs",
        "gpu_ciphertexts_bit_identical_on_sample": same,
        "host": f"{os.cpu_count()} logical CPUs, {granted} granted to this process (affinity mask capped by the cgroup quota)",
    }


def other_modes(device):
    """LUT mode under PARAM_MESSAGE_2_CARRY_2 (classical blind rotation) and arithmetic mode under both that set
    and the reference's own PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3 (helm.rs:83, multi-bit blind rotation)."""
    res = _other_modes_set(device, "shortint_m2c2", "PARAM_MESSAGE_2_CARRY_2_KS_PBS [dimensions recalled; the reference binary's "
                           "LUT mode names PARAM_MESSAGE_1_CARRY_1 (helm.rs:301), which cannot hold 3-input LUT indices]")
    mb = _other_modes_set(device, "shortint_m2c2_multibit3", "dimensions of PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS "
                          "(helm.rs:83) [recalled; LWE noise extrapolated, not tfhe's value: approximate set]")
    res["lut_mode_multibit3"], res["arith_mode_multibit3"] = mb["lut_mode"], mb["arith_mode"]
    res["wide_lut_wopbs"] = _wide_lut_leg(device)
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
    wsk.close()
    sk.close()
    return res


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
    t0 = time.perf_counter()
    w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(ck.decrypt(w.download(out)), (bits[:B] + bits[B:2 * B] + bits[2 * B:]) >= 2))
    res = {"lut_mode": {"workload": f"{B} independent 3-input LUT gates (keyswitch + programmable bootstrap), " + tfhe_name,
                        "luts_per_s": round(B / dt, 1), "decrypt_ok": ok}}
    g, ws, i, o, d, _, _ = verilog_parser.read_verilog_file(os.path.join(ROOT, "tests", "netlists", "chi_squared_arith.v"), True)
    c = Circuit(g, i, o, d)
    c.sort_circuit()
    c.compute_levels()
    ac = ArithCircuit(ck, sk, c)
    enc = ac.encrypt_inputs(ws, {"N0": PtxtType.U32(2), "N1": PtxtType.U32(7), "N2": PtxtType.U32(9)})
    ac.evaluate_encrypted(enc, 1, "u32")
    t0 = time.perf_counter()
    outm = ac.evaluate_encrypted(enc, 1, "u32")
    dt = time.perf_counter() - t0
    dec = {k: int(v.value) for k, v in ac.decrypt_outputs(outm, True).items()}
    res["arith_mode"] = {"workload": "chi_squared_arith.v, u32 (16 radix blocks per integer), " + tfhe_name, "wall_s": round(dt, 4),
                         "bootstraps": ac.pbs_per_cycle(), "batched_rounds": ac.pbs_rounds_per_cycle(),
                         "decrypt_ok": dec == {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250}}
    # the same evaluation with the two sub-circuits that share no wire (alpha's and the betas') on two lanes
    # (helm_si_ctx_fork): concurrent instead of level by level, identical ciphertexts
    ac.set_lanes(2)
    ac.evaluate_encrypted(enc, 1, "u32")
    t0 = time.perf_counter()
    outl = ac.evaluate_encrypted(enc, 1, "u32")
    dtl = time.perf_counter() - t0
    decl = {k: int(v.value) for k, v in ac.decrypt_outputs(outl, True).items()}
    res["arith_mode"]["two_lanes"] = {"wall_s": round(dtl, 4), "rounds_in_a_row": ac.pbs_rounds_per_cycle(),
                                      "bit_identical_to_level_by_level": all(np.array_equal(outm[k], outl[k]) for k in outm.keys()),
                                      "decrypt_ok": decl == dec}
    sk.close()
    return res


if __name__ == "__main__":
    main()

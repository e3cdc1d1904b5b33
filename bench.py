#!/usr/bin/env python3
"""bench.py — encrypted gate-bootstraps/sec on the AES-128 gates-mode netlist.

A step = one full level-by-level evaluation of the (generated, stand-in) AES-128 netlist
over a batch of `--blocks` independent input blocks per GPU, inputs already encrypted
and resident in HBM.  With N > 1 ranks the batch is N x blocks and every level wider
than one GPU wave is sharded across the ranks, the level's output ciphertexts
all-gathered over RCCL (helm_amd/distributed.py); keys and wire table are replicated.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=32, help="AES blocks per GPU evaluated together")
    ap.add_argument("--params", default="boolean_default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the LUT-mode / arithmetic-mode side measurements")
    ap.add_argument("--cpu-levels", type=int, default=8, help="at most this many netlist levels (1 block) on the CPU oracle")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="stop the CPU sample after this much time")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="OpenMP threads of the CPU sample (0 = every logical CPU); measured on the 1-GPU box, whose CPU "
                         "share is 16 cores: 16 -> 192, 32 -> 212, 64 -> 213, 128 -> 160, 256 -> 123 gates/s")
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
    from helm_amd.netlists import aes128, aes128_reference_encrypt

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

    # ---- keys (identical on every rank: same seed) and engine --------------------------
    t0 = time.time()
    ck = helm_amd.ClientKey.generate(args.params, seed=1)
    sk = helm_amd.ServerKey(ck, device=local_rank)
    sk.set_stream(torch.cuda.current_stream().cuda_stream)
    p = ck.params
    t_keys = time.time() - t0

    # ---- netlist -> level schedule over the whole batch --------------------------------
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
    circuit = Circuit(gates, inputs, outputs, dffs)
    circuit.sort_circuit()
    circuit.compute_levels()
    wire_names = list(inputs) + sorted(wire_set)
    total_blocks = args.blocks * world
    ops, i0, i1, i2, out, off, index = build_program_arrays(circuit, wire_names, total_blocks)
    nw = len(wire_names)
    prog = helm_amd.Program(sk, ops, i0, i1, i2, out, off)
    pbs_per_step = prog.total_pbs()
    widths = np.diff(off)

    # ---- synthetic inputs: seeded random key / plaintext per block, encrypted on the host,
    #      uploaded once: resident in HBM before the timed region ------------------------
    rng = np.random.default_rng(0x48454C4D)
    keys_pt = [(bytes(rng.integers(0, 256, 16, dtype=np.uint8)), bytes(rng.integers(0, 256, 16, dtype=np.uint8)))
               for _ in range(total_blocks)]
    keys_pt[0] = (bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff"))  # FIPS-197 C.1
    wires = sk.wires(nw * total_blocks)
    in_rows, in_bits = [], []
    for b, (key, pt) in enumerate(keys_pt):
        kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
        for i in range(128):
            in_rows += [b * nw + index[f"key[{i}]"], b * nw + index[f"pt[{i}]"]]
            in_bits += [(kv >> i) & 1, (pv >> i) & 1]
    wires.upload(np.array(in_rows, np.int32), ck.encrypt(np.array(in_bits, dtype=bool)))

    runner = ShardedRunner(GpuLevelExecutor(prog, wires), rank, world, dist if world > 1 else None)

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
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was timed: every block decrypts to AES(key, pt) ------------
    ok = True
    out_rows = np.array([b * nw + index[f"ct[{i}]"] for b in range(total_blocks) for i in range(128)], np.int32)
    dec = ck.decrypt(wires.download(out_rows)).reshape(total_blocks, 128)
    for b, (key, pt) in enumerate(keys_pt):
        got = sum(int(dec[b, i]) << i for i in range(128)).to_bytes(16, "big")
        ok &= got == aes128_reference_encrypt(key, pt)
    if not ok:
        raise SystemExit(f"rank {rank}: decrypted AES outputs are WRONG")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = pbs_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernel (k_pbs), from HIP events on its own stream -----
    K1 = p.k + 1
    bsk_bytes = p.n * p.pbs_l * K1 * K1 * p.N * 8
    io_bytes = 2 * (p.n + 1) * 4 + (p.k * p.N + 1) * 4
    # the dominant kernel is the lockstep build of k_pbs, which takes the full rounds (4 bootstraps per
    # CU) of every level; a level's remainder goes to the throughput / wide builds (part of pbs_ms)
    if tm.pbs_main_launches > 0:
        dom_kernel, launches = "k_pbs<PbsCfg<..., NB = 4>> (lockstep build)", tm.pbs_main_launches
        avg_pbs_per_launch = tm.pbs_main_count / launches
        avg_launch_s = tm.pbs_main_ms / launches * 1e-3
    else:
        dom_kernel, launches = "k_pbs (all builds)", max(1, tm.pbs_launches)
        avg_pbs_per_launch = tm.pbs_count / launches
        avg_launch_s = tm.pbs_ms / launches * 1e-3
    algo_bytes = bsk_bytes + avg_pbs_per_launch * io_bytes
    achieved_gbs = algo_bytes / avg_launch_s / 1e9
    # fp64 lane-operations of one bootstrap (DESIGN.md "k_pbs"): per CMUX step and wave, forward
    # transforms (8 per butterfly), pointwise multiply-accumulate (6 + 1 per word), inverse
    # transform (8 per butterfly + 4 recentrings of 3) and the hand-over sums; the 49-bit field
    # needs no recentring inside forward transforms and products
    logN = int(np.log2(p.N))
    E = p.N // 64
    bfly = E // 2 * logN
    lazy = p.N == 512  # field chosen by helm_hip_ctx_create for boolean_default
    dp_fwd = p.pbs_l * (bfly * 8 + (0 if lazy else 2 * E * 3))
    dp_mac = K1 * E * (p.pbs_l * 7 - 1) + (0 if lazy else K1 * E * 3)
    dp_inv = bfly * 8 + 4 * E * 3 + E * p.k
    dp_ops_per_pbs = p.n * K1 * 64 * (dp_fwd + dp_mac + dp_inv)
    fp64_tops = dp_ops_per_pbs * tm.pbs_count / (tm.pbs_ms * 1e-3) / 1e12

    result = {
        "metric": "encrypted gate-bootstraps/sec on AES-128 gates-mode netlist",
        "value": round(value, 1),
        "unit": "gate-bootstraps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 torus (exact NTT in f64 FMA over a 49-bit prime)",
        "data": "synthetic: generated AES-128 netlist (stand-in for HELM's), seeded random keys/plaintexts, "
                "fresh encryptions resident in HBM" + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearse else ""),
        "config": {
            "workload": f"AES-128 gates-mode netlist, {args.blocks} block(s) per GPU evaluated level-synchronously",
            "params": args.params, "n": p.n, "k": p.k, "N": p.N, "pbs_l": p.pbs_l, "pbs_logB": p.pbs_logB,
            "ks_l": p.ks_l, "ks_logB": p.ks_logB,
            "netlist_gates_per_block": int(widths.sum() // total_blocks), "levels": int(len(widths)),
            "bootstraps_per_step": int(pbs_per_step), "blocks_total": total_blocks,
            "parallelism": f"level-shard x{world} + all-gather of level outputs" if world > 1 else "single GPU",
            "sharded_levels": len(runner.sharded_levels),
            "exchanged_MB_per_step": round(runner.exchanged_bytes_per_pass() / 1e6, 2),
        },
        "wall_s_per_step": round(elapsed / args.steps, 4),
        "netlist_gates_per_s": round(float(widths.sum()) * args.steps / elapsed, 1),  # every gate, NOT / BUF included
        "decrypt_check": "all blocks == software AES (FIPS-197 C.1 vector in block 0)",
        "kernel_ms_per_step": {"k_pbs": round(tm.pbs_ms / args.steps, 3),
                               "k_pbs_lockstep_build": round(tm.pbs_main_ms / args.steps, 3), "k_keyswitch": round(tm.ks_ms / args.steps, 3),
                               "k_linear": round(tm.linear_ms / args.steps, 3)},
        "roofline": {
            "kernel": dom_kernel, "bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved_gbs / 8000.0, 5), "traffic": None,
            "traffic_measured_separately": "rocprofv3 --pmc FETCH_SIZE on a 1,024-bootstrap launch of this kernel "
                                           "(profiles/r01/pmc_fetch_size.txt): 0.63 GB of fabric-side reads per launch "
                                           "(Infinity-Cache hits included; 7x the algorithmic 90 MB), 4.1 MB written",
            "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(avg_launch_s * 1e3, 4),
            "avg_bootstraps_per_launch": round(avg_pbs_per_launch, 1),
            "note": "the bootstrapping key (80 MB) is shared by every ciphertext of a launch and stays in "
                    "L2/Infinity Cache, so HBM is not the binding resource; the kernel is bound by fp64 VALU "
                    "issue (see fp64_valu)",
        },
        "fp64_valu": {"achieved": round(fp64_tops, 2), "peak": 39.3, "unit": "T lane-op/s (FMA = 1)",
                      "frac": round(fp64_tops / 39.3, 4), "dp_lane_ops_per_bootstrap": int(dp_ops_per_pbs)},
        "setup_s": {"keygen_upload": round(t_keys, 2)},
    }

    # ---- wall-clock of ONE AES-128 evaluation (latency; levels are 80-256 gates wide, so the
    #      GPU is far from full: this is the n-step blind-rotation chain, 207 levels deep) ------
    if world == 1:
        o1 = build_program_arrays(circuit, wire_names, 1)
        prog1 = helm_amd.Program(sk, *o1[:6])
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
        import oracle
        orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk)
        one = build_program_arrays(circuit, wire_names, 1)
        o_ops, o_i0, o_i1, o_i2, o_out, o_off, _ = one
        host = np.zeros((nw, p.n + 1), dtype=np.uint32)
        key, pt = keys_pt[0]
        kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
        bits = np.array([(kv >> i) & 1 for i in range(128)] + [(pv >> i) & 1 for i in range(128)], dtype=bool)
        rows = [index[f"key[{i}]"] for i in range(128)] + [index[f"pt[{i}]"] for i in range(128)]
        del bits  # the oracle must see the very ciphertexts the GPU evaluated
        host[rows] = wires.download(np.array(rows, np.int32))
        ncpu = os.cpu_count() or 1
        threads = max(1, min(args.cpu_threads or ncpu, ncpu))
        n_pbs, L = 0, 0
        t0 = time.perf_counter()
        while L < min(args.cpu_levels, len(o_off) - 1) and (L < 1 or time.perf_counter() - t0 < args.cpu_seconds):
            s = slice(o_off[L], o_off[L + 1])
            orc.eval_level(host, o_ops[s], o_i0[s], o_i1[s], o_i2[s], o_out[s], nthreads=threads)
            n_pbs += int(np.sum(o_ops[s] != oracle.NOT))
            L += 1
        cpu_s = time.perf_counter() - t0
        gpu_rows = wires.download(o_out[:o_off[L]])
        same = bool(np.array_equal(gpu_rows, host[o_out[:o_off[L]]]))
        result["cpu_baseline"] = {
            "value": round(n_pbs / cpu_s, 2), "unit": "gate-bootstraps/s", "cores": threads, "kind": "port",
            "sample": f"first {L} level(s) of the same AES-128 netlist, 1 block ({n_pbs} gate-bootstraps, "
                      f"{cpu_s:.1f} s); scalar C restatement with a Goldilocks NTT, OpenMP over gates; NOT tfhe-rs",
            "gpu_ciphertexts_bit_identical_on_sample": same,
            "host": f"{os.cpu_count()} logical CPUs",
        }
        if not same:
            raise SystemExit("GPU ciphertexts differ from the CPU oracle on the sampled levels")
    # ---- the other two modes of the reference, same GPU, outside the timed region: 3-input LUT
    #      gates (BASELINE config 3's primitive) and the chi-squared u32 netlist (config 5) ------
    if world == 1 and not args.no_other_modes:
        try:
            result["other_modes"] = other_modes(local_rank)
        except Exception as e:  # the headline line must survive a failure here
            result["other_modes"] = {"error": repr(e)}
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def other_modes(device):
    """LUT mode under PARAM_MESSAGE_2_CARRY_2 (classical blind rotation) and arithmetic mode under both that set
    and the reference's own PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3 (helm.rs:83, multi-bit blind rotation)."""
    res = _other_modes_set(device, "shortint_m2c2", "PARAM_MESSAGE_2_CARRY_2_KS_PBS")
    mb = _other_modes_set(device, "shortint_m2c2_multibit3", "PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS")
    res["lut_mode_multibit3"], res["arith_mode_multibit3"] = mb["lut_mode"], mb["arith_mode"]
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
    sk.close()
    return res


if __name__ == "__main__":
    main()

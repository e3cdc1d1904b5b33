#!/usr/bin/env python3
"""A MODEL, not a measurement: what the fixed-job (strong-scaling) run of bench.py should cost at N = 1, 2, 4, 8 from the
launch lists alone.  The job's level schedule is launch-packed for N ranks (quantum = N x 4 x CUs; with `--costed` the
packer also knows what a launch of 1/4, 2/4, 3/4 of a round costs and picks the best width, helm_host_pack_levels_costed),
every launch cut into N contiguous chunks; a rank's chunk of c bootstraps is priced with the kernel table measured on one
MI355X (`--table r03|r04`; r04: profiles/r04/microbench.jsonl: whole lockstep rounds of 1,024 at 8.67 ms, a remainder of
<= 256 on the wide build 3.55 ms, <= 512 on k_pbs_duo 5.50 ms (r03: throughput build 7.4 ms), <= 768 in a partial lockstep
round 7.50 ms (r03: 8.5)); launches of <= 256 bootstraps are computed on every rank.  The exchange is priced at bytes /
50 GB/s + 40 us per all-gather.  Runs on the CPU (host library only).
usage: predict_strong_scaling.py [--blocks 32] [--table r04] [--costed] [--weak] [--json]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helm_amd import Circuit, verilog_parser  # noqa: E402
from helm_amd.distributed import gate_pbs, level_arrays, pack_levels, shard_bounds  # noqa: E402
from helm_amd.netlists import aes128  # noqa: E402

TABLES = {  # ms for <= 1/4, 2/4, 3/4, 4/4 of a round of 4 x CUs bootstraps
    "r03": (3.6, 7.4, 8.5, 8.5),
    "r04": (3.55, 5.50, 7.50, 8.67),  # profiles/r04/microbench.jsonl (one box: 256 / 512 / 768 / 1,024)
    "r04b": (3.26, 5.40, 7.13, 8.25),  # with k_pbs_trio and the twiddle registers (profiles/r04/trio_experiments.txt, one box)
    "r04c": (3.33, 5.20, 6.65, 7.74),  # final build of round 4: + the short-root field (profiles/r04/microbench.jsonl)
    "r05": (3.33, 5.34, 6.89, 8.32),   # round 5, same kernels on the round's profiling box (profiles/r05/microbench.jsonl)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--table", default="r04c", choices=sorted(TABLES))
    ap.add_argument("--costed", action="store_true", help="cost-aware launch packing (helm_host_pack_levels_costed)")
    ap.add_argument("--by-count", action="store_true", help="cut launches by gate count (rounds 1-4) instead of by bootstrap weight")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--weak", action="store_true", help="fixed work per GPU: the job at N ranks is N x --blocks blocks (bench.py's "
                                                        "headline since round 5, `sharded_weak`) instead of --blocks whatever N")
    a = ap.parse_args()
    if a.weak:
        return weak(a)
    run(a, a.blocks, (1, 2, 4, 8))


def weak(a):
    rows = []
    for n in (1, 2, 4, 8):
        rows.append(run(a, a.blocks * n, (n,), quiet=True)[0])
    base = rows[0]["k_gate_bootstraps_per_s"]
    for r in rows:
        r["of_linear"] = round(r["k_gate_bootstraps_per_s"] / base / r["n_gpus"], 3)
        if not a.json:
            print(f"N = {r['n_gpus']}: job of {a.blocks * r['n_gpus']:3d} blocks, {r['launches']:4d} launches, {r['ms_per_step']:8.1f} ms per step "
                  f"(exchange {r['exchange_ms']:6.1f} ms) -> {r['k_gate_bootstraps_per_s']:8.1f} k gate-bootstraps/s, {r['of_linear']:5.2f} of linear")
    if a.json:
        print(json.dumps({"model": "launch lists priced with the kernel table; NOT a measurement", "blocks_per_gpu": a.blocks,
                          "scaling": "weak (ONE job of N x blocks, every launch sharded)",
                          "table_ms": dict(zip(("le_256", "le_512", "le_768", "le_1024"), TABLES[a.table])),
                          "costed_packing": a.costed, "rows": rows}))


def run(a, blocks, worlds, quiet=False):
    CUS = 256
    T = TABLES[a.table]
    quarter_cost = [t / T[3] for t in T]
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    names = list(inputs) + sorted(wire_set)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(c, index)
    nw, nl = len(names), len(off) - 1
    tile = lambda x: np.concatenate([np.concatenate([np.where(x[off[l]:off[l + 1]] >= 0, x[off[l]:off[l + 1]] + b * nw, -1)
                                                     for b in range(blocks)]) for l in range(nl)]).astype(np.int32)
    opsT = np.concatenate([np.tile(ops[off[l]:off[l + 1]], blocks) for l in range(nl)]).astype(np.int32)
    offT = (off * blocks).astype(np.int64)
    arrs = (opsT, tile(i0), tile(i1), tile(i2), tile(out))

    def chunk_ms(cnt):
        full, rem = divmod(cnt, 4 * CUS)
        tail = 0.0 if rem == 0 else T[min(3, (rem - 1) // CUS)]
        return full * T[3] + tail

    base, rows = None, []
    for n in worlds:
        p_ops, _, _, _, _, p_off, _ = pack_levels(*arrs, offT, 4 * CUS * n, quarter_cost if a.costed else None)
        w = gate_pbs(p_ops)
        cs = np.concatenate([[0], np.cumsum(w)])
        total_ms, xchg_ms, launches = 0.0, 0.0, len(p_off) - 1
        for l in range(launches):
            lo, hi = int(p_off[l]), int(p_off[l + 1])
            pbs = int(cs[hi] - cs[lo])
            if n == 1 or pbs <= CUS:
                total_ms += chunk_ms(pbs)
                continue
            if a.by_count:
                gates_per_rank = -(-(hi - lo) // n)
                # the heaviest chunk: contiguous gates, bootstraps by the running count
                worst = max(int(cs[min(hi, lo + (r + 1) * gates_per_rank)] - cs[min(hi, lo + r * gates_per_rank)]) for r in range(n))
            else:  # the engine's cut since round 5 (helm_amd/csrc/shard_rule.h): by bootstrap weight
                b, gates_per_rank = shard_bounds(p_ops[lo:hi], n)
                worst = max(int(cs[lo + b[r + 1]] - cs[lo + b[r]]) for r in range(n))
            total_ms += chunk_ms(worst)
            xchg_ms += gates_per_rank * n * 2892 / 50e9 * 1e3 + 0.04
        step = total_ms + xchg_ms
        if base is None:
            base = step
        rows.append({"n_gpus": n, "launches": launches, "ms_per_step": round(step, 1), "exchange_ms": round(xchg_ms, 1),
                     "k_gate_bootstraps_per_s": round(int(cs[-1]) / step, 1), "of_linear": round(base / step / n, 3)})
        if not a.json and not quiet:
            print(f"N = {n}: {launches:4d} launches, {step:8.1f} ms per step (exchange {xchg_ms:6.1f} ms) -> "
                  f"{int(cs[-1]) / step:8.1f} k gate-bootstraps/s, {base / step / n:5.2f} of linear")
    if quiet:
        return rows
    if a.json:
        print(json.dumps({"model": "launch lists priced with the kernel table; NOT a measurement", "blocks": blocks,
                          "table_ms": dict(zip(("le_256", "le_512", "le_768", "le_1024"), T)), "costed_packing": a.costed,
                          "rows": rows}))


if __name__ == "__main__":
    main()

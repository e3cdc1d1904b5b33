#!/usr/bin/env python3
"""A MODEL, not a measurement: what the fixed-job (strong-scaling) run of bench.py should cost at N = 1, 2, 4, 8 from the
launch lists alone.  The job's level schedule is launch-packed for N ranks (quantum = N x 4 x CUs), every launch cut into
N contiguous chunks; a rank's chunk of c bootstraps is priced with the kernel table measured on one MI355X (DESIGN.md
4.2): whole lockstep rounds of 1,024 at 8.5 ms, a remainder of <= 256 on the wide build (3.6 ms), <= 512 on the
throughput build (7.4 ms), else one more lockstep round; launches of <= 256 bootstraps are computed on every rank.
The exchange is priced at bytes / 50 GB/s + 40 us per all-gather.  Runs on the CPU (host library only).
usage: predict_strong_scaling.py [blocks = 32]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from helm_amd import Circuit, verilog_parser
from helm_amd.distributed import gate_pbs, level_arrays, pack_levels
from helm_amd.netlists import aes128

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CUS, ROUND_MS = 256, 8.5
gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
c = Circuit(gates, inputs, outputs, dffs)
c.sort_circuit()
c.compute_levels()
names = list(inputs) + sorted(wire_set)
index = {w: i for i, w in enumerate(names)}
ops, i0, i1, i2, out, off = level_arrays(c, index)
nw, nl = len(names), len(off) - 1
tile = lambda a: np.concatenate([np.concatenate([np.where(a[off[l]:off[l + 1]] >= 0, a[off[l]:off[l + 1]] + b * nw, -1)
                                                 for b in range(blocks)]) for l in range(nl)]).astype(np.int32)
opsT = np.concatenate([np.tile(ops[off[l]:off[l + 1]], blocks) for l in range(nl)]).astype(np.int32)
offT = (off * blocks).astype(np.int64)
arrs = (opsT, tile(i0), tile(i1), tile(i2), tile(out))


def chunk_ms(cnt):
    full, rem = divmod(cnt, 4 * CUS)
    tail = 0.0 if rem == 0 else 3.6 if rem <= CUS else 7.4 if rem <= 2 * CUS else ROUND_MS
    return full * ROUND_MS + tail


base = None
for n in (1, 2, 4, 8):
    p_ops, _, _, _, _, p_off, _ = pack_levels(*arrs, offT, 4 * CUS * n)
    w = gate_pbs(p_ops)
    cs = np.concatenate([[0], np.cumsum(w)])
    total_ms, xchg_ms, launches = 0.0, 0.0, len(p_off) - 1
    for l in range(launches):
        a, b = int(p_off[l]), int(p_off[l + 1])
        pbs = int(cs[b] - cs[a])
        if n == 1 or pbs <= CUS:
            total_ms += chunk_ms(pbs)
            continue
        gates_per_rank = -(-(b - a) // n)
        # the heaviest chunk: contiguous gates, bootstraps by the running count
        worst = max(int(cs[min(b, a + (r + 1) * gates_per_rank)] - cs[min(b, a + r * gates_per_rank)]) for r in range(n))
        total_ms += chunk_ms(worst)
        xchg_ms += gates_per_rank * n * 2892 / 50e9 * 1e3 + 0.04
    if base is None:
        base = total_ms
    step = total_ms + xchg_ms
    print(f"N = {n}: {launches:4d} launches, bootstraps {step:8.1f} ms per step (exchange {xchg_ms:6.1f} ms) -> {int(cs[-1]) / step:8.1f} k gate-bootstraps/s, "
          f"{base / step / n:5.2f} of linear")

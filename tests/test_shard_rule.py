"""The cut of a launch into one chunk per GPU (helm_amd/csrc/shard_rule.h): by bootstrap weight - binary gate 1, MUX 2,
NOT / BUF / DFF / constants 0 - so that a sharded launch gives every rank the same number of bootstraps
whatever the mix of the level (the reference balances the level loop dynamically: src/circuit.rs:531 `par_iter_mut`).
The C rule (exported through libhelm_host.so; the engine uses the same header) and its Python mirror must agree."""
import ctypes as C

import numpy as np
import pytest

from helm_amd import _host as H
from helm_amd.distributed import gate_pbs, shard_bounds

AND, DFF, MUX, NAND, NOT, XOR, BUF, ONE, ZERO = 0, 1, 3, 4, 6, 9, 10, 11, 12


def c_bounds(op, world):
    op = np.ascontiguousarray(op, dtype=np.int32)
    b = np.zeros(world + 1, dtype=np.int64)
    rows = H.host.helm_host_shard_bounds(op.ctypes.data_as(C.POINTER(C.c_int32)), len(op), world,
                                         b.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rows >= 0
    return b, int(rows)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_python_mirror_equals_the_c_rule_and_balances_bootstraps(world):
    rng = np.random.default_rng(world)
    for trial in range(200):
        n = int(rng.integers(0, 400))
        op = rng.choice([AND, DFF, MUX, NAND, NOT, XOR, BUF, ONE, ZERO], size=n,
                        p=[.2, .05, .15, .1, .2, .15, .05, .05, .05]).astype(np.int32)
        if trial % 5 == 0 and n:  # free gates clustered at one end: what a cut by gate count gets wrong
            op = np.concatenate([np.full(n, NOT), op]).astype(np.int32)
        b, rows = c_bounds(op, world)
        pb, prows = shard_bounds(op, world)
        assert np.array_equal(b, pb) and rows == prows
        assert b[0] == 0 and b[-1] == len(op) and np.all(np.diff(b) >= 0) and rows == int(np.max(np.diff(b)))
        w = gate_pbs(op)
        total = int(w.sum())
        per_rank = [int(w[b[r]:b[r + 1]].sum()) for r in range(world)]
        assert sum(per_rank) == total
        if total:
            # every rank within one gate (a MUX: 2) of total / world
            assert max(per_rank) <= total / world + 2 and min(per_rank) >= total / world - 2


def test_fewer_gates_than_ranks_and_launches_without_bootstraps():
    b, rows = c_bounds([XOR, XOR, XOR], 8)
    assert rows == 1 and list(np.diff(b)) == [1, 0, 1, 0, 0, 1, 0, 0]
    b, rows = c_bounds([NOT] * 10, 4)        # no bootstrap at all: by gate count
    assert list(b) == [0, 3, 6, 9, 10] and rows == 3
    b, rows = c_bounds([], 3)
    assert list(b) == [0, 0, 0, 0] and rows == 0
    b, rows = c_bounds([MUX], 2)             # one gate: one rank has it
    assert list(np.diff(b)) == [1, 0]
    # a cut by count would give rank 0 four free gates and rank 1 four bootstraps
    b, rows = c_bounds([NOT] * 4 + [XOR] * 4, 2)
    assert list(b) == [0, 6, 8] and rows == 6
    assert H.host.helm_host_shard_bounds(None, 3, 0, None) == -1

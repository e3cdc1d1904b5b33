"""world_size-2 (and 3) runs of the level-sharding driver on CPU: torch.distributed with
the gloo backend, the level executor played by the CPU oracle (test infrastructure), the
driver (helm_amd/distributed.py: chunking, all-gather, scatter, replicate-vs-shard policy)
being the code under test.  The GPU executor's own chunk/scatter arithmetic is covered by
tests/test_gpu_circuits.py::test_sharded_levels_equal_unsharded."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helm_amd
import oracle
from helm_amd import Circuit, verilog_parser
from helm_amd.distributed import ShardedRunner, level_arrays, shard_bounds

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleExecutor:
    """Same interface as helm_amd.distributed.GpuLevelExecutor, CPU oracle inside."""

    def __init__(self, orc, arrays, wires):
        self.orc, self.wires = orc, wires
        self.ops, self.i0, self.i1, self.i2, self.out, self.off = arrays
        self.n_levels = len(self.off) - 1
        self.row_words = wires.shape[1]

    def level_count(self, l):
        return int(self.off[l + 1] - self.off[l])

    def level_pbs(self, l):
        s = slice(self.off[l], self.off[l + 1])
        return int(np.sum(~np.isin(self.ops[s], [oracle.NOT, oracle.BUF, oracle.DFF])))

    def new_buffer(self, rows):
        return torch.zeros((rows, self.row_words), dtype=torch.int32)

    def run_level(self, l):
        s = slice(self.off[l], self.off[l + 1])
        self.orc.eval_level(self.wires, self.ops[s], self.i0[s], self.i1[s], self.i2[s], self.out[s], nthreads=1)

    def _bounds(self, l, world):
        """The engine's cut: by bootstrap weight (helm_amd/csrc/shard_rule.h, mirrored by distributed.shard_bounds)."""
        return shard_bounds(self.ops[self.off[l]:self.off[l + 1]], world)

    def chunk_rows(self, l, world):
        return self._bounds(l, world)[1]

    def run_level_shard(self, l, rank, world, staging):
        b, _ = self._bounds(l, world)
        g0, g1 = int(b[rank]), int(b[rank + 1])
        staging.zero_()
        if g1 > g0:
            s = slice(self.off[l] + g0, self.off[l] + g1)
            tmp = self.wires.copy()
            self.orc.eval_level(tmp, self.ops[s], self.i0[s], self.i1[s], self.i2[s], self.out[s], nthreads=1)
            staging[:g1 - g0] = torch.from_numpy(tmp[self.out[s]].view(np.int32))

    def scatter_level(self, l, world, gathered):
        b, rows_per_rank = self._bounds(l, world)
        rows = gathered.numpy().view(np.uint32)
        for r in range(world):  # rank r's slot: its chunk's outputs in gate order, then padding
            g0, g1 = int(b[r]), int(b[r + 1])
            self.wires[self.out[self.off[l] + g0:self.off[l] + g1]] = rows[r * rows_per_rank:r * rows_per_rank + (g1 - g0)]


def _setup():
    ck = helm_amd.ClientKey.generate("toy", seed=5)
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_file(
        os.path.join(HERE, "netlists", "8-bit-adder.v"), False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    names = list(inputs) + sorted(wire_set)
    index = {w: i for i, w in enumerate(names)}
    arrays = level_arrays(c, index)
    rng = np.random.default_rng(9)
    bits = rng.integers(0, 2, size=len(inputs)).astype(bool)
    wires = np.zeros((len(names), ck.params.n + 1), dtype=np.uint32)
    wires[:len(inputs)] = ck.encrypt(bits)
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk)
    return ck, orc, arrays, wires, index, dict(zip(inputs, bits))


def _worker(rank, world, port, replicate_below, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ck, orc, arrays, wires, index, bits = _setup()
    ex = OracleExecutor(orc, arrays, wires)
    # world 1 shards only when asked to (force): every launch then still goes stage -> all-gather -> scatter - what
    # tests/test_gpu_rccl_world1.py runs over real RCCL on the one-GPU box
    runner = ShardedRunner(ex, rank, world, dist, replicate_below=replicate_below, force=world == 1)
    assert runner.active
    runner.run()
    q.put((rank, wires.copy(), len(runner.sharded_levels), runner.exchanged_bytes_per_pass()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,replicate_below", [(2, 0), (3, 0), (2, 2), (1, 0)])  # world 1: the forced single-rank form
def test_sharded_runner_matches_single_process(world, replicate_below):
    ck, orc, arrays, wires, index, bits = _setup()
    single = wires.copy()
    ShardedRunner(OracleExecutor(orc, arrays, single)).run()
    a = sum(int(bits[f"a[{i}]"]) << i for i in range(8))
    b = sum(int(bits[f"b[{i}]"]) << i for i in range(8))
    got = sum(int(ck.decrypt(single[index[f"sum[{i}]"]])) << i for i in range(8)) + \
        (int(ck.decrypt(single[index["cout"]])) << 8)
    assert got == a + b + int(bits["cin"])

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, replicate_below, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, w, n_sharded, nbytes in results:
        assert np.array_equal(w, single), f"rank {rank} diverged from the single-process result"
        if replicate_below == 0:
            assert n_sharded == len(arrays[5]) - 1 and nbytes > 0
        else:
            # levels with <= 2 bootstraps are computed redundantly, the others sharded
            assert 0 < n_sharded < len(arrays[5]) - 1


def test_split_launches_and_dependencies():
    """The two host helpers of the overlapped sharded schedule: sub-launches never exceed the cap by more than one MUX,
    keep every gate in order, and a launch's dependency is the last earlier launch that writes one of its inputs."""
    from helm_amd.distributed import gate_pbs, launch_dependencies, split_launches
    rng = np.random.default_rng(5)
    n = 5000
    op = rng.choice([0, 3, 4, 6, 9], size=n).astype(np.int32)        # AND, MUX, NAND, NOT, XOR
    off = np.array([0, 1200, 1300, 4000, n], dtype=np.int64)
    new = split_launches(op, off, 256)
    assert new[0] == 0 and new[-1] == n and np.all(np.diff(new) > 0) and set(off) <= set(new.tolist())
    w = gate_pbs(op)
    assert max(int(w[a:b].sum()) for a, b in zip(new[:-1], new[1:])) <= 257
    # a chain: gate g reads the output of gate g - 700 (if any): dependency = the launch holding that gate
    out = np.arange(100, 100 + n)
    i0 = np.where(np.arange(n) >= 700, out - 700, np.arange(n) % 100).astype(np.int64)
    none = np.full(n, -1)
    deps = launch_dependencies(i0, none, none, out, new, 100 + n)
    launch_of = np.searchsorted(new, np.arange(n), side="right") - 1
    for l, (a, b) in enumerate(zip(new[:-1], new[1:])):
        want = max([int(launch_of[g - 700]) for g in range(a, b) if g >= 700], default=-1)
        assert deps[l] == want and deps[l] < l
    with pytest.raises(ValueError, match="written more than once"):
        launch_dependencies(i0, none, none, np.zeros(n, dtype=np.int64), new, 100 + n)

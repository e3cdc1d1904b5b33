"""Two ranks, ONE GPU: the multi-GPU driver (helm_amd/distributed.py) with the real GPU level
executor (helm_hip_program_run_level_shard / _scatter_level) on both ranks and torch.distributed's
gloo backend carrying the all-gather of the device staging buffers.  RCCL needs one GPU per rank
(the driver runs that at round end); this covers everything else of the N > 1 path on the one-GPU
box: sharded levels, replicated levels, staging / scatter on device memory, and that every rank
ends with the wire table a single-process evaluation produces."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, blocks, result_dir, backend="gloo", overlap=False):
    import helm_amd
    from helm_amd import Circuit, verilog_parser
    from helm_amd.distributed import GpuLevelExecutor, ShardedRunner, launch_dependencies, level_arrays, split_launches
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = rank if backend == "nccl" else 0  # RCCL needs one GPU per rank; gloo shares cuda:0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ck = helm_amd.ClientKey.generate("toy_k2", seed=5)
    sk = helm_amd.ServerKey(ck, device=dev)
    # no sk.set_stream() here: ShardedRunner binds the engine to torch's current stream itself (an engine
    # left on its own stream would run the all-gather unordered with the shard kernels)
    # the AES netlist: 207 levels, several hundred gates wide with `blocks` copies - launches long enough
    # that a collective not ordered behind the engine's kernels reads stale staging rows (this test
    # caught exactly that: helm_hip_set_stream(NULL) used to mean "the context's own stream")
    from helm_amd.netlists import aes128
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
    deps = None
    t0, t1, t2, tout = tile(i0), tile(i1), tile(i2), tile(out)
    in_library = overlap == "library"
    comm = None
    if overlap == "comm":
        # the library's communicator over a host transport (helm_comm_create_with_transport): launch loop, in-place slot
        # of rank 1, gather and scatter all inside libhelm_hip.so, exactly the path bench.py takes over RCCL
        from helm_amd.comm import Comm
        comm = Comm.over_torch_dist(dist, dev)
        assert comm.info() == {"rank": rank, "world_size": world, "device": dev, "rccl_version": 0}
        assert comm.all_reduce(float(rank + 1), "sum") == world * (world + 1) / 2 and comm.all_reduce(float(rank), "max") == world - 1
    overlap = overlap is True
    if overlap:
        # the overlapped schedule: levels cut into sub-launches of <= 300 bootstraps, each launch's all-gather and scatter
        # on a side stream while the next sub-launch's bootstraps run; a launch waits only for the launch it depends on
        offT = split_launches(opsT, offT, 300)
        deps = launch_dependencies(t0, t1, t2, tout, offT, nw * blocks)
        assert len(offT) - 1 > nl and max(deps) >= 0
    prog = helm_amd.Program(sk, opsT, t0, t1, t2, tout, offT)
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2, size=(blocks, len(inputs))).astype(bool)
    wires = sk.wires(nw * blocks)
    rows = np.concatenate([b * nw + np.arange(len(inputs)) for b in range(blocks)]).astype(np.int32)
    wires.upload(rows, ck.encrypt(bits.reshape(-1)))
    runner = ShardedRunner(GpuLevelExecutor(prog, wires), rank, world, dist, depends_on=deps,
                           replicate_below=64 if overlap else 256, in_library=in_library, comm=comm)
    assert runner.in_library == in_library
    runner.run()
    if comm is not None:
        assert comm.stats()["collectives"] == len(runner.sharded_levels) + 2  # (+ the two all-reduces above)
    if overlap:
        runner.run()  # a second pass right behind the first: the ring of staging pairs and the events are reused
    torch.cuda.synchronize()
    dist.barrier()
    got = wires.download()
    # single-process reference on a fresh table (same ciphertexts: the client RNG is seeded)
    ref = sk.wires(nw * blocks)
    ref.upload(rows, got[rows])
    prog.run(ref)
    sk.sync()
    np.save(os.path.join(result_dir, f"rank{rank}.npy"), np.array([
        int(np.array_equal(got, ref.download())), len(runner.sharded_levels), nl], dtype=np.int64))
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 4, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        same, sharded, nl = np.load(tmp_path / f"rank{r}.npy")
        assert same == 1, f"rank {r}: sharded evaluation differs from the single-process one"
        assert 0 < sharded <= nl


def test_two_ranks_overlapped_exchange(tmp_path):
    """The overlapped schedule (ShardedRunner(depends_on=...)): all-gather + scatter of a launch on a side stream while
    the next launch computes, launches waiting only for the launch they depend on - the same wire table, bit for bit, as
    the single-process evaluation (two passes back to back)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 4, str(tmp_path), "gloo", True), nprocs=2, join=True)
    for r in range(2):
        same, sharded, nl = np.load(tmp_path / f"rank{r}.npy")
        assert same == 1, f"rank {r}: overlapped sharded evaluation differs from the single-process one"
        assert sharded > nl  # sub-launches: more sharded launches than levels


def test_two_ranks_through_the_c_abi_pass(tmp_path):
    """helm_hip_program_run_sharded: the launch loop of the sharded pass inside libhelm_hip.so, calling back for the
    all-gather only (what a Rust host binds: INTEGRATION.md, Multi-GPU) - the same wire table as one process."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 4, str(tmp_path), "gloo", "library"), nprocs=2, join=True)
    for r in range(2):
        same, sharded, nl = np.load(tmp_path / f"rank{r}.npy")
        assert same == 1, f"rank {r}: the in-library sharded pass differs from the single-process one"
        assert 0 < sharded <= nl


def test_two_ranks_through_the_library_communicator(tmp_path):
    """helm_hip_program_run_sharded_comm at world size 2: the communicator is the library's (include/helm_comm.h), its
    all-gather carried by a host transport because both ranks share one GPU (RCCL wants one per rank).  Everything
    bench.py's N > 1 run does except RCCL itself - rank 1's slot offset, padded chunks, scatter - against one process."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 4, str(tmp_path), "gloo", "comm"), nprocs=2, join=True)
    for r in range(2):
        same, sharded, nl = np.load(tmp_path / f"rank{r}.npy")
        assert same == 1, f"rank {r}: the pass through the library's communicator differs from the single-process one"
        assert 0 < sharded <= nl


def test_three_ranks_through_the_library_communicator(tmp_path):
    """The same with THREE ranks on the one GPU: launch sizes that three does not divide - padded last chunks, rank 2's
    slot behind two others - through helm_hip_program_run_sharded_comm."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(3, port, 4, str(tmp_path), "gloo", "comm"), nprocs=3, join=True)
    for r in range(3):
        same, sharded, nl = np.load(tmp_path / f"rank{r}.npy")
        assert same == 1, f"rank {r}: the three-rank pass through the library's communicator differs from the single-process one"
        assert 0 < sharded <= nl


def test_two_ranks_over_rccl(tmp_path):
    """The same run over the real `nccl` backend (RCCL over xGMI), one GPU per rank.  Needs two GPUs: on a one-GPU
    box this is reported as an expected failure - RCCL NOT exercised - rather than passing silently."""
    if torch.cuda.device_count() < 2:
        pytest.xfail("RCCL not exercised: this box has one GPU (the all-gather ran over gloo only)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 4, str(tmp_path), "nccl"), nprocs=2, join=True)
    for r in range(2):
        same, sharded, nl = np.load(tmp_path / f"rank{r}.npy")
        assert same == 1, f"rank {r}: evaluation sharded over RCCL differs from the single-process one"
        assert 0 < sharded <= nl

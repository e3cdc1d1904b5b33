"""RCCL carries the sharded path on ONE GPU: world size 1, real collectives.

A one-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on one device), but a world-size-1
communicator is a real communicator: ncclCommInitRank, ncclAllGather on the engine's stream, the kernels
RCCL launches.  With `replicate_below = 0` EVERY launch that bootstraps goes stage -> all-gather -> scatter,
so the first multi-GPU run is not the first time RCCL meets this code (reference unit of sharding:
src/circuit.rs:531, the gates of a level).

  * the library's own communicator (include/helm_comm.h): helm_hip_program_run_sharded_comm and
    helm_si_set_exchange_comm - launch loop and ncclAllGather inside libhelm_hip.so, no torch in the data path;
  * torch.distributed's `nccl` backend through ShardedRunner (level by level, and through the C-ABI pass
    with the all-gather as callback), and through helm_si_set_exchange.

Every variant must leave the wire table of helm_hip_program_run / the unsharded LUT level, bit for bit.
One worker process does all of it (one RCCL initialisation each)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _aes_program(helm_amd, sk, blocks):
    from helm_amd import Circuit, verilog_parser
    from helm_amd.distributed import level_arrays
    from helm_amd.netlists import aes128
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    names = list(inputs) + sorted(wire_set)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(c, index)
    nw, nl = len(names), len(off) - 1
    # the first 24 levels are enough: every level shape (wide XOR layers, the S-box's AND layers, NOT-only levels)
    nl = min(nl, 24)
    tile = lambda a: np.concatenate([np.concatenate([np.where(a[off[l]:off[l + 1]] >= 0, a[off[l]:off[l + 1]] + b * nw, -1)
                                                     for b in range(blocks)]) for l in range(nl)]).astype(np.int32)
    opsT = np.concatenate([np.tile(ops[off[l]:off[l + 1]], blocks) for l in range(nl)]).astype(np.int32)
    offT = (off[:nl + 1] * blocks).astype(np.int64)
    arrays = (opsT, tile(i0), tile(i1), tile(i2), tile(out), offT)
    prog = helm_amd.Program(sk, *arrays)
    rows = np.concatenate([b * nw + np.arange(len(inputs)) for b in range(blocks)]).astype(np.int32)
    return prog, nw * blocks, rows, len(inputs), arrays


def _worker(_rank, port, result_path):
    import torch.distributed as dist
    import helm_amd
    from helm_amd import comm as hc
    from helm_amd.distributed import GpuLevelExecutor, ShardedRunner
    res = {}
    torch.cuda.set_device(0)
    ck = helm_amd.ClientKey.generate("toy_k2", seed=5)
    sk = helm_amd.ServerKey(ck, device=0)
    blocks = 3
    prog, n_rows, in_rows, n_in, prog_arrays = _aes_program(helm_amd, sk, blocks)
    bits = np.random.default_rng(3).integers(0, 2, size=blocks * n_in).astype(bool)
    enc = ck.encrypt(bits)
    ref = sk.wires(n_rows)
    ref.upload(in_rows, enc)
    prog.run(ref)
    sk.sync()
    want = ref.download()
    n_pbs_levels = sum(1 for l in range(prog.n_levels) if prog.level_pbs(l) > 0)
    res["levels_with_bootstraps"] = n_pbs_levels

    def fresh():
        w = sk.wires(n_rows)
        w.upload(in_rows, enc)
        return w

    # ---- (1) the library's own communicator: launch loop + ncclAllGather inside libhelm_hip.so --------------------
    assert hc.available()
    c = hc.Comm.single(0)
    res["comm_info"] = c.info()
    w = fresh()
    sk.timing_enable(True)
    sk.timing(reset=True)
    runner = ShardedRunner(GpuLevelExecutor(prog, w), 0, 1, comm=c, replicate_below=0)
    runner.run()
    runner.run()  # the gather buffer and the shard tables are reused
    sk.sync()
    tm = sk.timing(reset=True)
    sk.timing_enable(False)
    res["comm_same"] = bool(np.array_equal(w.download(), want))
    res["comm_sharded_levels"] = len(runner.sharded_levels)
    res["comm_stats"] = c.stats()
    res["comm_exchange_count"] = int(tm.exchange_count)
    res["comm_exchange_ms"] = float(tm.exchange_ms)
    res["comm_exchange_bytes"] = int(tm.exchange_bytes)
    res["comm_bytes_per_pass"] = int(runner.exchanged_bytes_per_pass())
    # ---- (1a) the overlapped exchange inside the library: the all-gather + scatter of a launch on the engine's exchange
    #      stream, later launches waiting (events) only for the launch they depend on; the launches cut into sub-launches
    #      so that consecutive ones ARE independent; two passes back to back (ring buffers and events reused) ------------
    from helm_amd.distributed import split_launches
    cut = split_launches(prog_arrays[0], prog_arrays[5], 40)
    prog_cut = helm_amd.Program(sk, *prog_arrays[:5], cut)
    res["overlap_applies"] = prog_cut.overlap_applies()
    res["overlap_launches"] = [int(prog.n_levels), int(prog_cut.n_levels)]
    w = fresh()
    before = c.stats()["collectives"]
    sk.timing_enable(True)
    sk.timing(reset=True)
    runner = ShardedRunner(GpuLevelExecutor(prog_cut, w), 0, 1, comm=c, replicate_below=0, overlap=True)
    runner.run()
    runner.run()
    sk.sync()
    tmo = sk.timing(reset=True)
    sk.timing_enable(False)
    res["overlap_same"] = bool(np.array_equal(w.download(), want))
    res["overlap_collectives"] = c.stats()["collectives"] - before
    res["overlap_sharded_levels"] = len(runner.sharded_levels)
    res["overlap_exchange_count"] = int(tmo.exchange_count)
    res["overlap_exchange_ms"] = float(tmo.exchange_ms)
    prog_cut.destroy()
    res["comm_allreduce_max"] = c.all_reduce(41.5, "max")
    res["comm_allreduce_sum"] = c.all_reduce(2.25, "sum")
    c.barrier()

    # ---- (1b) the evaluator API: GateCircuit.shard_over (helm_host_gate_circuit_shard_over) ---------------------------
    from helm_amd import Circuit, GateCircuit, PtxtType, verilog_parser
    here = os.path.dirname(os.path.abspath(__file__))
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_file(os.path.join(here, "netlists", "alu-c880-class.v"), False)
    circ = Circuit(gates, inputs, outputs, dffs)
    circ.sort_circuit()
    circ.compute_levels()
    gc = GateCircuit(ck, sk, circ)
    vals = {w: PtxtType.Bool(bool(v)) for w, v in zip(inputs, np.random.default_rng(8).integers(0, 2, len(inputs)))}
    enc_in = gc.encrypt_inputs(wire_set, vals)
    one_gpu = gc.evaluate_encrypted(enc_in, 1, "bool")
    gc.shard_over(c, 0)
    before = c.stats()["collectives"]
    sharded = gc.evaluate_encrypted(enc_in, 2, "bool")
    res["gc_collectives"] = c.stats()["collectives"] - before
    res["gc_same"] = all(np.array_equal(one_gpu[w], sharded[w]) for w in one_gpu.keys())
    res["gc_log"] = gc.log()
    gc.shard_over(None)
    res["gc_back"] = all(np.array_equal(one_gpu[w], gc.evaluate_encrypted(enc_in, 3, "bool")[w]) for w in list(one_gpu.keys())[:8])

    # ---- (2) the 64-bit-torus engine through the same communicator: helm_si_set_exchange_comm ---------------------
    sck = helm_amd.SiClientKey.generate("si_toy_1024", seed=1)
    ssk = helm_amd.SiServerKey(sck, device=0)
    B = 40
    sbits = np.random.default_rng(4).integers(0, 2, size=3 * B).astype(np.uint64)
    senc = sck.encrypt(sbits)
    in_idx = np.arange(3 * B, dtype=np.int32).reshape(3, B).T.copy()
    ar, tb, out = np.full(B, 3, np.int32), np.full(B, 0x96, np.uint64), np.arange(3 * B, 4 * B, dtype=np.int32)
    sw = ssk.wires(4 * B)
    sw.upload(np.arange(3 * B), senc)
    sw.eval_lut_level(ar, in_idx, tb, out)
    ssk.sync()
    swant = sw.download()
    res["si_decrypt_ok"] = bool(np.array_equal(sck.decrypt(swant[3 * B:]), sbits[:B] ^ sbits[B:2 * B] ^ sbits[2 * B:]))
    ssk.set_exchange_comm(c, min_batch=1, capacity_rows=16)  # 40 look-ups in three rounds of the gather buffer
    sw2 = ssk.wires(4 * B)
    sw2.upload(np.arange(3 * B), senc)
    sw2.eval_lut_level(ar, in_idx, tb, out)
    ssk.sync()
    res["si_comm_same"] = bool(np.array_equal(sw2.download(), swant))
    res["si_comm_stats"] = list(ssk.exchange_stats())
    ssk.set_exchange_comm(None)
    res["comm_stats_end"] = c.stats()

    # ---- (3) torch.distributed's nccl backend, world size 1, through ShardedRunner --------------------------------
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res["torch_backend"] = dist.get_backend()
    for name, kw in (("torch_levels", {}), ("torch_library_pass", {"in_library": True})):
        w = fresh()
        runner = ShardedRunner(GpuLevelExecutor(prog, w), 0, 1, dist, replicate_below=0, force=True, time_collective=True, **kw)
        assert runner.in_library == bool(kw)
        runner.run()
        torch.cuda.synchronize()
        res[name + "_same"] = bool(np.array_equal(w.download(), want))
        res[name + "_sharded_levels"] = len(runner.sharded_levels)
    ssk.set_exchange(dist, 0, 1, min_batch=1, capacity_rows=16, force=True)
    sw3 = ssk.wires(4 * B)
    sw3.upload(np.arange(3 * B), senc)
    sw3.eval_lut_level(ar, in_idx, tb, out)
    ssk.sync()
    res["si_torch_same"] = bool(np.array_equal(sw3.download(), swant))
    res["si_torch_stats"] = list(ssk.exchange_stats())
    ssk.set_exchange(dist, 0, 1)
    # the way bench.py creates the communicator: rank 0 draws the id, torch.distributed carries it to the other ranks
    c2 = hc.Comm.from_torch_dist(dist, 0)
    res["comm_from_dist"] = c2.info()
    res["comm_from_dist_max"] = c2.all_reduce(7.0, "max")
    c2.destroy()
    dist.destroy_process_group()
    c.destroy()
    ssk.close()
    sk.close()
    with open(result_path, "w") as f:
        json.dump(res, f)


@pytest.fixture(scope="module")
def world1(tmp_path_factory):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    path = str(tmp_path_factory.mktemp("rccl") / "world1.json")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_worker, args=(port, path), nprocs=1, join=True)
    with open(path) as f:
        return json.load(f)


def test_in_library_communicator_carries_every_launch(world1):
    r = world1
    assert r["comm_info"]["world_size"] == 1 and r["comm_info"]["rank"] == 0 and r["comm_info"]["device"] == 0
    assert r["comm_info"]["rccl_version"] > 20000          # what ncclGetVersion reports
    assert r["comm_same"], "sharded pass through the library's RCCL communicator differs from helm_hip_program_run"
    # replicate_below = 0: every launch that bootstraps was exchanged, twice (two passes)
    assert r["comm_sharded_levels"] == r["levels_with_bootstraps"] > 0
    assert r["comm_exchange_count"] == 2 * r["comm_sharded_levels"]
    assert r["comm_exchange_bytes"] == 2 * r["comm_bytes_per_pass"] > 0
    assert r["comm_stats"]["collectives"] == r["comm_exchange_count"]  # (read before the later parts issue more)
    assert r["comm_exchange_ms"] > 0.0
    assert r["comm_allreduce_max"] == 41.5 and r["comm_allreduce_sum"] == 2.25
    # the overlapped exchange (run_sharded_comm, overlap = 1) over real RCCL: same table, every sub-launch exchanged
    assert r["overlap_applies"] and r["overlap_launches"][1] > r["overlap_launches"][0]
    assert r["overlap_same"], "overlapped sharded pass over RCCL differs from helm_hip_program_run"
    assert r["overlap_collectives"] == r["overlap_exchange_count"] == 2 * r["overlap_sharded_levels"] > 2 * r["comm_sharded_levels"]
    assert r["overlap_exchange_ms"] > 0.0
    # the evaluator API on top of it: GateCircuit.shard_over
    assert r["gc_same"] and r["gc_back"] and r["gc_collectives"] > 0
    assert "sharded over 1 rank(s)" in r["gc_log"]


def test_shortint_engine_through_the_library_communicator(world1):
    r = world1
    assert r["si_decrypt_ok"]
    assert r["si_comm_same"], "LUT level sharded through helm_si_set_exchange_comm differs from the unsharded one"
    batches, rows = r["si_comm_stats"]
    assert batches == 3 and rows == 40       # 40 look-ups through a 16-row gather buffer
    assert r["comm_stats_end"]["collectives"] >= r["comm_stats"]["collectives"] + 3


def test_torch_nccl_backend_world_size_one(world1):
    r = world1
    assert r["torch_backend"] == "nccl"
    for name in ("torch_levels", "torch_library_pass"):
        assert r[name + "_same"], f"{name}: sharded pass over torch.distributed's nccl backend differs from helm_hip_program_run"
        assert r[name + "_sharded_levels"] == r["levels_with_bootstraps"]
    assert r["si_torch_same"]
    assert r["si_torch_stats"][0] == 3
    assert r["comm_from_dist"]["world_size"] == 1 and r["comm_from_dist_max"] == 7.0

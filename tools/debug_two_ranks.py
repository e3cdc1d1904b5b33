"""Debug: 2 ranks on one GPU (gloo), AES netlist, compare level by level with a single-process table."""
import os, sys, socket
import numpy as np
import torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, ".")

def worker(rank, world, port, params, blocks):
    import helm_amd
    from helm_amd import Circuit, verilog_parser
    from helm_amd.distributed import GpuLevelExecutor, ShardedRunner
    from helm_amd.netlists import aes128
    sys.path.insert(0, ".")
    import bench
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ck = helm_amd.ClientKey.generate(params, seed=1)
    sk = helm_amd.ServerKey(ck, device=0)
    sk.set_stream(torch.cuda.current_stream().cuda_stream)
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
    c = Circuit(gates, inputs, outputs, dffs); c.sort_circuit(); c.compute_levels()
    names = list(inputs) + sorted(wire_set)
    ops, i0, i1, i2, out, off, index = bench.build_program_arrays(c, names, blocks)
    nw = len(names)
    prog = helm_amd.Program(sk, ops, i0, i1, i2, out, off)
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2, size=blocks * len(inputs)).astype(bool)
    rows = np.concatenate([b * nw + np.arange(len(inputs)) for b in range(blocks)]).astype(np.int32)
    cts = ck.encrypt(bits)
    wires = sk.wires(nw * blocks); wires.upload(rows, cts)
    ref = sk.wires(nw * blocks); ref.upload(rows, cts)
    runner = ShardedRunner(GpuLevelExecutor(prog, wires), rank, world, dist)
    ex = runner.ex
    bad = None
    for l in range(ex.n_levels):
        if l not in runner._sharded:
            ex.run_level(l)
        else:
            r = -(-ex.level_count(l) // world)
            stage = runner._stage[:r]; gathered = runner._gather[:r * world]
            ex.run_level_shard(l, rank, world, stage)
            dist.all_gather_into_tensor(gathered, stage)
            ex.scatter_level(l, world, gathered)
        prog.run(ref, l, l + 1)
        torch.cuda.synchronize()
        o = out[off[l]:off[l + 1]]
        a, b = wires.download(o), ref.download(o)
        if not np.array_equal(a, b):
            badrows = np.nonzero((a != b).any(axis=1))[0]
            print(f"rank {rank}: level {l} differs: {len(badrows)}/{len(o)} rows, first {badrows[:6]}, sharded={l in runner._sharded}, "
                  f"count={ex.level_count(l)} pbs={ex.level_pbs(l)}", flush=True)
            bad = l
            break
    if bad is None:
        print(f"rank {rank}: all {ex.n_levels} levels identical", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    params = sys.argv[1] if len(sys.argv) > 1 else "toy_k2"
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, params, blocks), nprocs=2, join=True)

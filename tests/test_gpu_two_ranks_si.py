"""Two ranks, ONE GPU, LUT mode and arithmetic mode: helm_si_set_exchange() shards every bootstrap
batch of the 64-bit-torus engine over the ranks ("LUT and arithmetic modes shard the
same way at PBS-batch granularity"); gloo carries the all-gather of the device staging rows here,
RCCL on a multi-GPU node.  Checked: every wire of the sharded evaluation is bit-identical to the
single-GPU evaluation of the same input ciphertexts, batches really were sharded (also in several
rounds: the staging capacity is smaller than the widest batch), and the outputs decrypt correctly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NET = os.path.join(HERE, "netlists")


def _circuit(path, is_arith):
    from helm_amd import Circuit, verilog_parser
    gates_set, wire_set, input_wires, output_wires, dffs, _, _ = verilog_parser.read_verilog_file(path, is_arith)
    c = Circuit(gates_set, input_wires, output_wires, dffs)
    c.sort_circuit()
    c.compute_levels()
    return c, wire_set


def _both_ways(circ, sk, wire_set, inputs, ptxt_type, blocks, rank, world, comm=None):
    """(wires of the sharded run, wires of the single-GPU run on the same inputs, sharded batches)"""
    from helm_amd import SiEncWireMap
    if comm is not None:
        sk.set_exchange_comm(comm, min_batch=2, capacity_rows=24)
    else:
        sk.set_exchange(dist, rank, world, min_batch=2, capacity_rows=24)
    enc_in = circ.encrypt_inputs(wire_set, inputs)
    saved = {w: np.array(enc_in[w], copy=True) for w in enc_in.keys()}
    out = circ.evaluate_encrypted(enc_in, 1, ptxt_type)
    torch.cuda.synchronize()
    sharded = {w: np.array(out[w], copy=True) for w in out.keys()}
    batches, rows = sk.exchange_stats()
    dist.barrier()
    if comm is not None:
        sk.set_exchange_comm(None)
    else:
        sk.set_exchange(dist, rank, 1)
    again = SiEncWireMap(sk, blocks=blocks)
    for w, ct in saved.items():
        again[w] = ct
    out1 = circ.evaluate_encrypted(again, 2, ptxt_type)  # a new cycle: arithmetic circuits memoise per cycle (gates.rs:307-312)
    single = {w: np.array(out1[w], copy=True) for w in out1.keys()}
    return sharded, single, batches, out


def _worker(rank, world, port, result_dir, set_name, through_comm=False):
    import helm_amd
    from helm_amd import ArithCircuit, LutCircuit, PtxtType, verilog_parser
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ck, sk = helm_amd.gen_keys_shortint(set_name, seed=1)  # same seed: same keys and encryptions on every rank
    res = []
    comm = None
    if through_comm:
        # helm_si_set_exchange_comm at world size 2: the library's communicator over a host transport (both ranks share the
        # GPU, RCCL wants one each) - rank 1's slot in the engine-owned gather buffer, several exchange rounds per batch
        from helm_amd.comm import Comm
        comm = Comm.over_torch_dist(dist, 0)

    # LUT mode: the 8-bit adder of 3-input LUTs (BASELINE config 3)
    circuit, wire_set = _circuit(f"{NET}/8-bit-adder-lut-3-1.v", False)
    a, b, cin = 0xB7, 0x6E, 1
    inputs = {f"a[{i}]": PtxtType.Bool((a >> i) & 1) for i in range(8)}
    inputs.update({f"b[{i}]": PtxtType.Bool((b >> i) & 1) for i in range(8)})
    inputs["cin"] = PtxtType.Bool(cin)
    lc = LutCircuit(ck, sk, circuit)
    sharded, single, batches, out = _both_ways(lc, sk, wire_set, inputs, "bool", 1, rank, world, comm)
    same = set(sharded) == set(single) and all(np.array_equal(sharded[w], single[w]) for w in single)
    dec = lc.decrypt_outputs(out, True)
    total = sum(dec[f"sum[{i}]"].value << i for i in range(8)) + (dec["cout"].value << 8)
    res += [int(same), batches, int(total == a + b + cin)]

    # arithmetic mode: chi-squared on u32 (BASELINE config 5), batches of up to several hundred look-ups
    circuit, wire_set = _circuit(f"{NET}/chi_squared_arith.v", True)
    inputs = verilog_parser.read_input_wires(os.path.join(HERE, "golden", "chi_squared_arith_1.inputs.csv"), "u32")
    ac = ArithCircuit(ck, sk, circuit)
    sharded, single, batches, out = _both_ways(ac, sk, wire_set, inputs, "u32", 16, rank, world, comm)
    same = set(sharded) == set(single) and all(np.array_equal(sharded[w], single[w]) for w in single)
    dec = {k: v.value for k, v in ac.decrypt_outputs(out, True).items()}
    res += [int(same), batches, int(dec == {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250})]
    np.save(os.path.join(result_dir, f"rank{rank}.npy"), np.array(res, dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()
    sk.close()


@pytest.mark.parametrize("set_name", ["shortint_m2c2", "shortint_m2c2_multibit3"])  # classical / multi-bit rotation
def test_lut_and_arith_modes_two_ranks_on_one_gpu(tmp_path, set_name):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), set_name), nprocs=2, join=True)
    for r in range(2):
        lut_same, lut_batches, lut_ok, ar_same, ar_batches, ar_ok = np.load(tmp_path / f"rank{r}.npy")
        assert lut_same == 1, f"rank {r}: sharded LUT evaluation differs from the single-GPU one"
        assert ar_same == 1, f"rank {r}: sharded arithmetic evaluation differs from the single-GPU one"
        assert lut_ok == 1 and ar_ok == 1
        # the adder: 2 look-ups per level; chi-squared: 39 rounds, the wide ones in several exchanges of 48 rows
        assert lut_batches >= 4 and ar_batches > 39, (lut_batches, ar_batches)


def test_lut_and_arith_modes_two_ranks_through_the_library_communicator(tmp_path):
    """The same through helm_si_set_exchange_comm (what bench.py's other_modes and a Rust host use over RCCL)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), "shortint_m2c2", True), nprocs=2, join=True)
    for r in range(2):
        lut_same, lut_batches, lut_ok, ar_same, ar_batches, ar_ok = np.load(tmp_path / f"rank{r}.npy")
        assert lut_same == 1 and ar_same == 1, f"rank {r}: sharded evaluation differs from the single-GPU one"
        assert lut_ok == 1 and ar_ok == 1
        assert lut_batches >= 4 and ar_batches > 39, (lut_batches, ar_batches)

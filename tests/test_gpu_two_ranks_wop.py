"""Two ranks, ONE GPU, wide LUT gates: with helm_si_set_exchange() on the PBS-side context helm_wop_eval_luts splits
the gates of a batch over the ranks (every stage of a gate stays on its rank; keys replicated), the result rows are
all-gathered (gloo here, RCCL on a multi-GPU node) and scattered into every rank's table.  Checked: bit-identical to
the unsharded evaluation of the same ciphertexts, the batches really were sharded, also a batch with fewer gates than
ranks * capacity and one rank left without work."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, result_dir):
    import helm_amd
    from helm_amd import wopbs
    from helm_amd.shortint import si_named_params
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sp, a, b = si_named_params("si_toy_512")
    wp, c, d = wopbs.wop_named_params("wop_toy_512")
    sp.message_modulus = sp.carry_modulus = wp.message_modulus = wp.carry_modulus = 2
    ck = helm_amd.SiClientKey(sp, a, b, seed=3)  # same seeds: same keys and encryptions on every rank
    wk = wopbs.WopClientKey(ck, wp, c, d, seed=4)
    sk = helm_amd.SiServerKey(ck)
    wsk = wopbs.WopServerKey(sk, wk)
    rng = np.random.default_rng(0)
    res = []
    for count, m in ((11, 4), (1, 3)):  # 11 gates: rounds of 2 x 3 rows; 1 gate: rank 1 has nothing to do
        truth = rng.integers(0, 2, size=(count, 1 << m), dtype=np.uint64)
        xs = rng.integers(0, 1 << m, size=count)
        bits_in = np.array([[(x >> (m - 1 - q)) & 1 for q in range(m)] for x in xs], dtype=np.uint64)
        cts = ck.encrypt(bits_in.reshape(-1))
        in_idx = np.arange(count * m, dtype=np.int32).reshape(count, m)
        out_idx = np.arange(count * m, count * (m + 1), dtype=np.int32)
        got = []
        for shard in (True, False):
            sk.set_exchange(dist, rank, world if shard else 1, min_batch=1, capacity_rows=3)
            w = sk.wires(count * (m + 1))
            w.upload(np.arange(count * m), cts)
            wsk.eval_luts(w, in_idx, truth, out_idx, bits_per_block=1)
            got.append(w.download(out_idx))
            if shard:
                batches, _ = sk.exchange_stats()
            dist.barrier()
        same = np.array_equal(got[0], got[1])
        ok = [int(v) for v in ck.decrypt_message_and_carry(got[0])] == [int(truth[g, xs[g]]) for g in range(count)]
        res += [int(same), int(ok), batches]
    np.save(os.path.join(result_dir, f"rank{rank}.npy"), np.array(res, dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()
    wsk.close()
    sk.close()


def test_wide_lut_gates_two_ranks_on_one_gpu(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        same_a, ok_a, batches_a, same_b, ok_b, batches_b = np.load(tmp_path / f"rank{r}.npy")
        assert same_a == 1 and same_b == 1, f"rank {r}: sharded wide-LUT evaluation differs from the single-GPU one"
        assert ok_a == 1 and ok_b == 1
        assert batches_a == 2 and batches_b == 1, (batches_a, batches_b)  # 11 gates in rounds of 2 x 3 rows; one gate: one round

"""The communicator handshake (helm_amd.comm.Comm.agree) must end on EVERY rank whatever fails on one of them:
ncclCommInitRank blocks until the whole world has entered it and has no timeout, so a rank-local failure before it must
reach the others over the control plane before anybody goes in (the reference has no multi-GPU code; the sharded unit is
the level of src/circuit.rs:531).  gloo, world 3, the three device-touching calls replaced by hooks; a failure is injected
at each step on each kind of rank and every rank must come back with the same answer within the test's time limit."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeComm:
    destroyed = False

    def __init__(self, device, uid, rank, world):
        assert uid.shape == (128,) and uid[0] == 42
        self.rank = rank

    def destroy(self):
        self.destroyed = True


def _worker(rank, world, port, inject, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helm_amd.comm import Comm
    made = []

    def create(device, uid, r, w):
        c = _FakeComm(device, uid, r, w)
        made.append(c)
        return c
    hooks = {"unique_id": lambda: np.full(128, 42, dtype=np.uint8), "precheck": lambda d: None, "create": create}
    comm, err = Comm.agree(dist, 0, _inject=inject, _hooks=hooks)
    q.put((rank, comm is not None, err, [c.destroyed for c in made]))
    dist.barrier()  # the control plane is still usable afterwards (the fallback / the error exit both need it)
    dist.destroy_process_group()


def _run(world, inject):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, inject, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_every_rank_gets_the_communicator_when_nothing_fails():
    res = _run(3, None)
    assert all(ok and err is None and destroyed == [False] for _, ok, err, destroyed in res)


@pytest.mark.parametrize("inject", [("id", 0), ("precheck", 0), ("precheck", 2), ("create", 1)])
def test_a_failure_on_one_rank_reaches_every_rank(inject):
    res = _run(3, inject)
    step = {"id": "unique id", "precheck": "precheck", "create": "create"}[inject[0]]
    for rank, ok, err, destroyed in res:
        assert not ok and err and f"injected failure: {step}" in err, (rank, err)
        # nobody is left holding a communicator its peers do not have
        assert all(destroyed), (rank, destroyed)
    if inject[0] != "create":
        assert all(destroyed == [] for _, _, _, destroyed in res)  # nobody entered ncclCommInitRank


def test_precheck_fails_without_a_device():
    """On this CPU box the real helm_comm_precheck must fail (no device or no RCCL) with a message, not crash."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    from helm_amd import comm
    from helm_amd._native import HelmError
    with pytest.raises(HelmError):
        comm.precheck(0)

"""The library's own RCCL communicator (include/helm_comm.h): one per process, one process per GPU.

The sharded unit is the level of reference src/circuit.rs:531 (the reference has no multi-GPU path);
with a `Comm` the all-gather of a launch's output ciphertexts runs inside libhelm_hip.so
(`Program.run_sharded_comm`, `SiServerKey.set_exchange_comm`) - no torch in the data path.

The 128-byte unique id has to reach every rank by some control plane; `Comm.from_torch_dist`
uses whatever torch.distributed backend the caller has initialised (gloo is enough), and
`Comm.single` is the world-size-1 communicator (every collective still goes through RCCL).
"""
import ctypes as C

import numpy as np

from . import _native as nv
from ._native import hip, hip_check

ID_BYTES = 128


def available():
    """True when an RCCL library could be bound in this process (never touches a device)."""
    return bool(hip.helm_comm_available())


def precheck(device):
    """What helm_comm_create needs from this process alone (RCCL bound, the device exists), without entering a
    collective; raises with the library's message otherwise (include/helm_comm.h: helm_comm_precheck)."""
    hip_check(hip.helm_comm_precheck(int(device)))


def unique_id():
    buf = np.zeros(ID_BYTES, dtype=np.uint8)
    hip_check(hip.helm_comm_get_unique_id(nv.as_u8p(buf)))
    return buf


class Comm:
    def __init__(self, device, uid, rank, world):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        assert uid.shape == (ID_BYTES,)
        h = nv.vp()
        hip_check(hip.helm_comm_create(int(device), nv.as_u8p(uid), int(rank), int(world), C.byref(h)))
        self._h = h
        self.device = int(device)

    @classmethod
    def single(cls, device=0):
        return cls(device, unique_id(), 0, 1)

    @classmethod
    def from_torch_dist(cls, dist, device):
        """Rank 0 draws the id, torch.distributed (any backend) carries it to the others."""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [unique_id().tobytes() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(device, np.frombuffer(box[0], dtype=np.uint8).copy(), rank, world)

    @classmethod
    def agree(cls, dist, device, _inject=None, _hooks=None):
        """The RCCL communicator over torch.distributed's control plane, in three steps that EVERY rank finishes whatever
        fails on one of them - ncclCommInitRank has no timeout, so nobody may enter it unless everybody will:
          1. rank 0 draws the unique id inside a try and broadcasts (id | error) unconditionally;
          2. every rank runs its local pre-checks (helm_comm_precheck: RCCL bound, device usable) and the ranks exchange
             the outcomes; one failure anywhere and nobody goes on;
          3. ncclCommInitRank (helm_comm_create), the outcomes exchanged once more; a rank that has a communicator while
             another has none destroys it.
        -> (Comm, None) on every rank, or (None, reason) on every rank.  `_inject` = (step, rank) makes that rank fail
        at that step ("id", "precheck", "create"); `_hooks` replaces the three device-touching calls (unique_id, precheck,
        create): the CPU tests of the handshake (tests/test_comm_handshake.py)."""
        rank, world = dist.get_rank(), dist.get_world_size()
        hooks = {"unique_id": unique_id, "precheck": precheck, "create": cls}
        hooks.update(_hooks or {})

        def fails(step):
            return _inject is not None and _inject == (step, rank)

        box = [None]
        if rank == 0:
            try:
                if fails("id"):
                    raise RuntimeError("injected failure: unique id")
                box = [("id", hooks["unique_id"]().tobytes())]
            except Exception as e:  # noqa: BLE001 - travels to the other ranks instead of leaving them in the broadcast
                box = [("error", f"rank 0 could not draw the communicator's id: {e!r}")]
        dist.broadcast_object_list(box, src=0)
        kind, payload = box[0]
        err = payload if kind == "error" else None
        if err is None:
            try:
                if fails("precheck"):
                    raise RuntimeError("injected failure: precheck")
                hooks["precheck"](device)
            except Exception as e:  # noqa: BLE001
                err = f"rank {rank}: {e!r}"
        outcomes = [None] * world
        dist.all_gather_object(outcomes, err)
        bad = [o for o in outcomes if o]
        if bad:
            return None, "; ".join(sorted(set(bad)))
        comm = None
        try:
            if fails("create"):
                raise RuntimeError("injected failure: create")
            comm = hooks["create"](device, np.frombuffer(payload, dtype=np.uint8).copy(), rank, world)
        except Exception as e:  # noqa: BLE001
            err = f"rank {rank}: {e!r}"
        dist.all_gather_object(outcomes, err)
        bad = [o for o in outcomes if o]
        if bad:
            if comm is not None:
                comm.destroy()
            return None, "; ".join(sorted(set(bad)))
        return comm, None

    @classmethod
    def with_transport(cls, device, rank, world, all_gather):
        """A communicator over a transport the host brings (helm_comm_create_with_transport):
        `all_gather(send_ptr, recv_ptr, bytes_per_rank, stream_ptr)` gathers every rank's bytes into recv_ptr in rank
        order, ordered behind / ahead of the work on the HIP stream `stream_ptr`; an exception fails the collective."""
        self = cls.__new__(cls)

        def trampoline(_user, send, recv, nbytes, stream):
            try:
                all_gather(int(send or 0), int(recv or 0), int(nbytes), int(stream or 0))
                return 0
            except Exception:  # noqa: BLE001 - reported through the status code
                import traceback
                traceback.print_exc()
                return 1

        self._cb = nv.COMM_ALL_GATHER_FN(trampoline)  # kept alive as long as the communicator
        h = nv.vp()
        hip_check(hip.helm_comm_create_with_transport(int(device), int(rank), int(world), self._cb, None, C.byref(h)))
        self._h = h
        self.device = int(device)
        return self

    @classmethod
    def over_torch_dist(cls, dist, device):
        """The transport form over ANY torch.distributed backend, staged through host memory (gloo is enough): for
        hosts without RCCL and for rehearsing the rank > 0 paths with several ranks on one GPU.  Slow by construction
        (device -> host -> peers -> device per collective); `from_torch_dist` is the RCCL communicator."""
        import torch
        nv.require_one_hip_runtime("Comm.over_torch_dist")  # torch tensors over the library's device ranges
        rank, world = dist.get_rank(), dist.get_world_size()

        class _Raw:  # a device range as something torch.as_tensor understands
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        def all_gather(send, recv, nbytes, stream):
            dev = torch.device("cuda", int(device))
            s = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.stream(s):
                mine = torch.as_tensor(_Raw(send, nbytes), device=dev).cpu()  # waits for the stream's earlier work
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                torch.as_tensor(_Raw(recv, nbytes * world), device=dev).copy_(torch.cat(parts))
                s.synchronize()

        return cls.with_transport(device, rank, world, all_gather)

    @classmethod
    def in_process_group(cls, devices, timeout=600.0):
        """One communicator per entry of `devices` for ranks that are THREADS of this process (a host that drives its GPUs
        from one process, one thread and one engine context per rank; several ranks may share a device):
        helm_comm_create_in_process - the all-gather is device-to-device copies between the ranks' buffers inside the library,
        a barrier on either side.  Every rank must call the collectives from its own thread.  No RCCL, no torch.  A rank that
        fails breaks the group (`abort_group`), so the others get an error instead of waiting; `timeout` seconds bound every
        wait.  -> [Comm] in rank order."""
        world = len(devices)
        devs = (C.c_int * world)(*[int(d) for d in devices])
        outs = (nv.vp * world)()
        hip_check(hip.helm_comm_create_in_process(devs, world, float(timeout), outs))
        comms = []
        for r in range(world):
            c = cls.__new__(cls)
            c._h = nv.vp(outs[r])
            c.device = int(devices[r])
            comms.append(c)
        return comms

    def abort_group(self):
        """in_process_group: break the group's barrier so that no other rank thread waits for this one any more."""
        if getattr(self, "_h", None):
            hip.helm_comm_abort_group(self._h)

    def info(self):
        """What RCCL reports: rank, world size, device, library version."""
        r, w, d, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        hip_check(hip.helm_comm_info(self._h, C.byref(r), C.byref(w), C.byref(d), C.byref(v)))
        return {"rank": r.value, "world_size": w.value, "device": d.value, "rccl_version": v.value}

    def stats(self):
        n, b = C.c_int64(), C.c_int64()
        hip_check(hip.helm_comm_stats(self._h, C.byref(n), C.byref(b)))
        return {"collectives": n.value, "bytes_sent": b.value}

    def all_gather(self, send_ptr, recv_ptr, bytes_per_rank, stream_ptr):
        hip_check(hip.helm_comm_all_gather(self._h, nv.vp(send_ptr), nv.vp(recv_ptr), int(bytes_per_rank), nv.vp(stream_ptr)))

    def all_reduce(self, value, op="max"):
        v = C.c_double(float(value))
        hip_check(hip.helm_comm_all_reduce_f64(self._h, C.byref(v), {"sum": 0, "max": 1}[op]))
        return v.value

    def barrier(self):
        hip_check(hip.helm_comm_barrier(self._h))

    def destroy(self):
        if getattr(self, "_h", None):
            hip_check(hip.helm_comm_destroy(self._h))
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

"""Multi-GPU evaluation of a levelised netlist: one process per GPU, keys and the wire
table replicated, each level's gates split into `world` contiguous chunks and only the
chunk outputs (small LWE ciphertexts, (n+1) words per gate) all-gathered over RCCL.

The reference has no multi-GPU code; the level is its unit of
parallelism (src/circuit.rs:531) and is what is sharded here.

The per-level exchange is latency-bound (a 512-gate level of the boolean set moves
1.5 MB in total), so levels that one GPU can absorb in a single wave of workgroups
(<= `replicate_below` bootstraps: one per CU) are computed redundantly on every rank
instead of being sharded: that costs nothing in time and removes the collective.
"""
import numpy as np


class GpuLevelExecutor:
    """Level executor over the C ABI (helm_hip_program_*): the product path."""

    def __init__(self, program, wires):
        import torch
        from ._native import require_one_hip_runtime
        # torch streams, events and tensors meet the engine's kernels and buffers from here on
        require_one_hip_runtime("GpuLevelExecutor")
        self.torch = torch
        self.program, self.wires = program, wires
        self.n_levels = program.n_levels
        self.row_words = program.sk.params.n + 1
        self.device = torch.device("cuda", program.sk.device)

    def bind_stream(self):
        """Put the engine on torch's current stream of this device: the collective is issued there, and the
        shard kernels (before it) and the scatter (after it) are ordered with it only if they share it."""
        with self.torch.cuda.device(self.device):
            self.program.sk.set_stream(self.torch.cuda.current_stream(self.device).cuda_stream)

    def level_count(self, level):
        off = self.program.level_offsets
        return int(off[level + 1] - off[level])

    def level_pbs(self, level):
        return self.program.level_pbs(level)

    def chunk_rows(self, level, world):
        """Rows of one rank's slot in the level's all-gather (the largest chunk of the weight-balanced cut)."""
        return self.program.chunk_rows(level, world)

    def new_buffer(self, rows):
        return self.torch.empty((rows, self.row_words), dtype=self.torch.int32, device=self.device)

    def run_level(self, level):
        self.program.run(self.wires, level, level + 1)

    def shard_prepare(self, rank, world):
        """Plan and upload this rank's chunk of every launch once: the per-launch calls then only launch kernels."""
        self.program.shard_prepare(rank, world)

    def new_events(self):
        return self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)

    # ---- what the overlapped runner needs: a side stream, plain events, the engine put on a given stream -------------
    def new_stream(self):
        return self.torch.cuda.Stream(device=self.device)

    def new_event(self):
        return self.torch.cuda.Event()

    def current_stream(self):
        return self.torch.cuda.current_stream(self.device)

    def stream(self, s):
        return self.torch.cuda.stream(s)

    def engine_on(self, s):
        self.program.sk.set_stream(s.cuda_stream)

    def run_level_shard(self, level, rank, world, staging):
        self.program.run_level_shard(self.wires, level, rank, world, staging.data_ptr())

    def scatter_level(self, level, world, gathered):
        self.program.scatter_level(self.wires, level, world, gathered.data_ptr())

    def run_sharded(self, rank, world, stage, gather, capacity_rows, exchange, replicate_below):
        self.program.run_sharded(self.wires, rank, world, stage.data_ptr(), gather.data_ptr(), capacity_rows, exchange,
                                 replicate_below)

    def run_sharded_comm(self, comm, replicate_below, overlap=False):
        self.program.run_sharded_comm(self.wires, comm, replicate_below, overlap)

    def exchange_ms(self, reset=False):
        """Device time of the in-library all-gathers (helm_hip_timing.exchange_ms; needs timing_enable)."""
        t = self.program.sk.timing(reset=False)
        return t.exchange_ms


class ShardedRunner:
    """Drives one evaluation pass over all levels on `world` ranks.

    The collective is either the library's own (`comm`: helm_amd.comm.Comm, an RCCL communicator
    inside libhelm_hip.so - launch loop and ncclAllGather both native, no torch in the data path;
    what bench.py uses) or `dist`, torch.distributed (backend "nccl" = RCCL on GPUs; "gloo" in the
    CPU tests).  With world == 1 nothing is exchanged unless `force` (or a `comm`) asks for the
    sharded path anyway: every launch of more than `replicate_below` bootstraps then still goes
    stage -> all-gather -> scatter (the single-GPU test of the path over real RCCL)."""

    def __init__(self, executor, rank=0, world=1, dist=None, replicate_below=256, time_collective=False, depends_on=None,
                 ring=3, in_library=False, comm=None, force=False, overlap=False):
        """depends_on (optional, one entry per launch: the last EARLIER launch whose outputs this one reads, -1 for
        none; `launch_dependencies`) switches the overlapped schedule on: a sharded launch's all-gather and scatter go
        to a side stream and the next launches run meanwhile, each waiting only for the launch it depends on.
        With a `comm`, `overlap=True` asks the library for the same schedule natively (helm_hip_program_run_sharded_comm,
        overlap = 1: dependency table, second stream and events all inside libhelm_hip.so)."""
        self.ex, self.rank, self.world, self.dist, self.comm = executor, rank, world, dist, comm
        self.overlap = bool(overlap) and comm is not None
        self.replicate_below = replicate_below
        active = world > 1 or bool(force) or comm is not None
        self.active = active
        self.depends_on = None if depends_on is None or not active or comm is not None else [int(d) for d in depends_on]
        # in_library: the launch loop runs inside libhelm_hip.so (helm_hip_program_run_sharded) and calls back for the
        # all-gather only - the path a Rust host takes when it brings its own collective (Multi-GPU)
        self.in_library = bool(in_library) and active and comm is None and hasattr(executor, "run_sharded")
        self._ring_size = max(2, int(ring))
        self.time_collective = time_collective and active and hasattr(executor, "new_events")
        self._events = []
        self.sharded_levels = []
        self._staging, self._gathered = {}, {}
        if active and comm is not None:
            for l in range(executor.n_levels):
                if executor.level_pbs(l) > replicate_below:
                    self.sharded_levels.append(l)
            executor.shard_prepare(rank, world)
        elif active:
            if hasattr(executor, "bind_stream"):
                executor.bind_stream()  # not left to the caller: an unordered all-gather silently corrupts wires
            for l in range(executor.n_levels):
                if executor.level_pbs(l) > replicate_below:
                    self.sharded_levels.append(l)
            if hasattr(executor, "shard_prepare"):
                executor.shard_prepare(rank, world)
            rows = max([self._rows(l) for l in self.sharded_levels], default=0)
            if rows:
                self._stage = executor.new_buffer(rows)
                self._gather = executor.new_buffer(rows * world)
            if self.depends_on is not None and rows:
                assert len(self.depends_on) == executor.n_levels
                self._side = executor.new_stream()
                self._ring = [(executor.new_buffer(rows), executor.new_buffer(rows * world), executor.new_event())
                              for _ in range(self._ring_size)]
        self._sharded = set(self.sharded_levels)

    def _rows(self, level):
        """Rows one rank contributes to the level's all-gather: the executor's own cut when it has one (the engine cuts by
        bootstrap weight, helm_amd/csrc/shard_rule.h), contiguous chunks by gate count otherwise."""
        if hasattr(self.ex, "chunk_rows"):
            return int(self.ex.chunk_rows(level, self.world))
        return -(-self.ex.level_count(level) // self.world)

    def run(self):
        if self.comm is not None:
            if self.overlap:
                return self.ex.run_sharded_comm(self.comm, self.replicate_below, True)
            return self.ex.run_sharded_comm(self.comm, self.replicate_below)
        if self.in_library and self.sharded_levels:
            rows_cap = self._stage.shape[0]

            def exchange(_stage_ptr, _gather_ptr, rows):
                self.dist.all_gather_into_tensor(self._gather[:rows * self.world], self._stage[:rows])
                return 0
            return self.ex.run_sharded(self.rank, self.world, self._stage, self._gather, rows_cap, exchange, self.replicate_below)
        if self.depends_on is not None and self.sharded_levels:
            return self._run_overlapped()
        ex = self.ex
        for l in range(ex.n_levels):
            if l not in self._sharded:
                ex.run_level(l)
                continue
            rows = self._rows(l)
            stage = self._stage[:rows]
            gathered = self._gather[:rows * self.world]
            ex.run_level_shard(l, self.rank, self.world, stage)
            if self.time_collective:  # events on the stream the engine and the collective share
                a, b = ex.new_events()
                a.record()
                self.dist.all_gather_into_tensor(gathered, stage)
                b.record()
                self._events.append((a, b))
            else:
                self.dist.all_gather_into_tensor(gathered, stage)
            ex.scatter_level(l, self.world, gathered)

    def _run_overlapped(self):
        """Launch l's chunk is computed on the main stream; its all-gather and the scatter into the wire table follow on
        the side stream while the main stream goes on with launch l + 1.  A launch waits for the scatter of the launch it
        depends on (the side stream runs in order, so everything before that one is in the table as well); a staging
        pair of the ring is reused once the side stream is through with it.  Replicated launches run on the main stream
        as before.  Needs every wire written once per pass (checked by the caller: `launch_dependencies`)."""
        ex = self.ex
        main, side = ex.current_stream(), self._side
        import bisect
        done, done_at = [], []   # sharded launches in order, and the event recorded on the side stream after each scatter
        k = 0
        last = None
        for l in range(ex.n_levels):
            d = self.depends_on[l]
            if d >= 0 and done:
                w = bisect.bisect_right(done, d) - 1  # the latest sharded launch at or before d
                if w >= 0:
                    main.wait_event(done_at[w])
            if l not in self._sharded:
                ex.run_level(l)
                continue
            rows = self._rows(l)
            stage_full, gather_full, free = self._ring[k % self._ring_size]
            if k >= self._ring_size:
                main.wait_event(free)  # the side stream has finished with this pair
            k += 1
            stage, gathered = stage_full[:rows], gather_full[:rows * self.world]
            ex.run_level_shard(l, self.rank, self.world, stage)
            computed = ex.new_event()
            computed.record(main)
            side.wait_event(computed)
            with ex.stream(side):
                if self.time_collective:
                    a, b = ex.new_events()
                    a.record(side)
                    self.dist.all_gather_into_tensor(gathered, stage)
                    b.record(side)
                    self._events.append((a, b))
                else:
                    self.dist.all_gather_into_tensor(gathered, stage)
                ex.engine_on(side)
                ex.scatter_level(l, self.world, gathered)
                ex.engine_on(main)
                free.record(side)
                ev = ex.new_event()
                ev.record(side)
            done.append(l)
            done_at.append(ev)
            last = l
        if last is not None:
            main.wait_event(done_at[-1])

    def collective_ms(self, reset=False):
        """GPU time between the records around every all-gather since the last reset (time_collective=True);
        includes the wait for the slowest rank's chunk, which is what the exchange costs a launch."""
        if self.comm is not None:
            # the engine's own events around every ncclAllGather, accumulated since ITS last reset
            # (ServerKey.timing_enable / timing(reset=True)); `reset` is the engine's business here
            return self.ex.exchange_ms()
        if self._events:
            self._events[-1][1].synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._events)
        if reset:
            self._events = []
        return ms

    def exchanged_bytes_per_pass(self):
        ex = self.ex
        return sum(self._rows(l) * self.world * ex.row_words * 4 for l in self.sharded_levels)


def level_arrays(circuit, index):
    """level_map of a Circuit -> (opcode, in0, in1, in2, out, level_offsets) over wire rows
    given by `index` (name -> row)."""
    ops, i0, i1, i2, out, off = [], [], [], [], [], [0]
    lm = circuit.level_map()
    for lvl in sorted(lm):
        for gate in lm[lvl]:
            ins = [index[w] for w in gate.input_wires] + [-1, -1, -1]
            ops.append(int(gate.gate_type))
            i0.append(ins[0]); i1.append(ins[1]); i2.append(ins[2])
            out.append(index[gate.output_wire])
        off.append(len(ops))
    return (np.array(ops, np.int32), np.array(i0, np.int32), np.array(i1, np.int32), np.array(i2, np.int32),
            np.array(out, np.int32), np.array(off, np.int64))


def pack_levels(opcode, in0, in1, in2, out, level_offsets, quantum, quarter_cost=None):
    """Launch packing (helm_host_pack_levels, helm_amd/csrc/host/level_pack.cpp): the level schedule re-timed so
    that a launch holds a whole number of `quantum` bootstraps while that many gates are ready; dependency order
    kept, outputs bit-identical.  quarter_cost (ServerKey.launch_costs(): relative cost of a launch of at most 1/4,
    2/4, 3/4, 4/4 of a round) lets launches narrower than a round take the engine's most efficient width and leave
    the rest to the next launch.  -> (opcode, in0, in1, in2, out, launch_offsets, packed: bool)"""
    import ctypes as C
    from . import _host as H
    arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (opcode, in0, in1, in2, out)]
    off = np.ascontiguousarray(level_offsets, dtype=np.int64)
    total = len(arrs[0])
    order = np.zeros(total, dtype=np.int64)
    new_off = np.zeros(total + 1, dtype=np.int64)
    n = C.c_int64()
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    if quarter_cost is None:
        rc = H.host.helm_host_pack_levels(*[a.ctypes.data_as(i32p) for a in arrs], off.ctypes.data_as(i64p), len(off) - 1,
                                          int(quantum), order.ctypes.data_as(i64p), new_off.ctypes.data_as(i64p), C.byref(n))
    else:
        qc = (C.c_double * 4)(*[float(x) for x in quarter_cost])
        rc = H.host.helm_host_pack_levels_costed(*[a.ctypes.data_as(i32p) for a in arrs], off.ctypes.data_as(i64p), len(off) - 1,
                                                 int(quantum), qc, order.ctypes.data_as(i64p), new_off.ctypes.data_as(i64p),
                                                 C.byref(n))
    if rc < 0:
        raise H.Panic(H.host.helm_host_last_error().decode())
    return tuple(a[order] for a in arrs) + (new_off[:n.value + 1].copy(), rc == 0)


def gate_pbs(opcode):
    """Bootstraps of each gate: MUX 2, NOT / BUF / DFF / constants 0, every other gate 1 (helm_hip_program_level_pbs)."""
    op = np.asarray(opcode)
    w = np.ones(len(op), dtype=np.int64)
    w[op == 3] = 2                                   # HELM_GATE_MUX
    w[np.isin(op, (1, 6, 10, 11, 12))] = 0           # DFF, NOT, BUF, constants
    return w


def shard_bounds(opcode, world):
    """The engine's cut of one launch into `world` contiguous chunks (helm_amd/csrc/shard_rule.h, mirrored here for
    hosts and executors outside the library): by BOOTSTRAP weight - bounds[r] is the first gate at which the bootstraps
    before it reach r / world of the launch's total; a launch without bootstraps is cut by gate count.
    -> (bounds[0 .. world], rows of the largest chunk)"""
    w = gate_pbs(opcode)
    cnt, total = len(w), int(w.sum())
    if total == 0:
        chunk = -(-cnt // world)
        b = np.minimum(np.arange(world + 1, dtype=np.int64) * chunk, cnt)
    else:
        before = np.concatenate([[0], np.cumsum(w)])       # bootstraps of gates [0, g)
        # the first g with before[g] * world >= total * r
        b = np.searchsorted(before * world, total * np.arange(world + 1, dtype=np.int64), side="left").astype(np.int64)
        b[0], b[world] = 0, cnt
    return b, int(np.max(np.diff(b))) if world > 0 else 0


def split_launches(opcode, launch_offsets, max_pbs):
    """Cut every launch into sub-launches of at most `max_pbs` bootstraps (the gates of a launch are independent of
    each other, so any cut is valid): consecutive sub-launches of one launch can then overlap - the exchange of one with
    the bootstraps of the next.  -> new launch_offsets"""
    w = gate_pbs(opcode)
    off = [0]
    for l in range(len(launch_offsets) - 1):
        a, b = int(launch_offsets[l]), int(launch_offsets[l + 1])
        c = np.cumsum(w[a:b])
        parts = int(-(-int(c[-1]) // max_pbs)) if b > a else 1
        if parts > 1:  # cut where the running count passes each multiple of max_pbs (a MUX may put one part 1 over)
            cuts = a + np.searchsorted(c, np.arange(1, parts) * max_pbs, side="right")
            off.extend(int(x) for x in np.unique(cuts) if a < x < b)
        off.append(b)
    return np.array(off, dtype=np.int64)


def launch_dependencies(in0, in1, in2, out, launch_offsets, n_rows):
    """For every launch the last EARLIER launch that writes one of its input rows (-1: none).  Raises if a row is written
    twice in the pass (state-writing gates): the overlapped schedule needs every wire written once."""
    out = np.asarray(out)
    if len(np.unique(out)) != len(out):
        raise ValueError("a wire is written more than once per pass: the overlapped schedule does not apply")
    producer = np.full(int(n_rows), -1, dtype=np.int64)
    deps = []
    for l in range(len(launch_offsets) - 1):
        a, b = int(launch_offsets[l]), int(launch_offsets[l + 1])
        d = -1
        for arr in (in0, in1, in2):
            rows = np.asarray(arr[a:b])
            rows = rows[rows >= 0]
            if len(rows):
                d = max(d, int(producer[rows].max()))
        deps.append(d)
        producer[out[a:b]] = l
    return deps

"""`import helm_amd` BEFORE `import torch`, then torch handles cross into the library: the order that mapped two HIP
runtimes and aborted in round 5 (gpurun_out/r05/diag.log).  One fresh process:

  * exactly one libamdhip64 mapped after both imports (helm_amd/_native.py binds the process to the copy torch looks for);
  * the engine put on a torch stream (ServerKey.set_stream), the sharded pass through the library's RCCL communicator at
    world size 1 (every launch stage -> ncclAllGather -> scatter on that torch stream);
  * the sharded pass through torch.distributed's `nccl` backend: ShardedRunner binds torch's current stream, the staging
    buffers are torch tensors, the all-gather is torch's;
each must leave the wire table of helm_hip_program_run bit for bit (reference unit: the level of src/circuit.rs:531)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
import helm_amd                                   # FIRST
from helm_amd import _native as nv
before = nv.mapped_hip_runtimes()
import torch                                      # SECOND
import torch.distributed as dist
from helm_amd import comm as hc
from helm_amd.distributed import GpuLevelExecutor, ShardedRunner
res = {{"before_torch": before, "mapped": nv.mapped_hip_runtimes()}}
torch.cuda.set_device(0)
rng = np.random.default_rng(11)
ck = helm_amd.ClientKey.generate("toy_k2", seed=7)
sk = helm_amd.ServerKey(ck, device=0)
# a random levelised netlist: 6 levels of 37 gates of every bootstrapping type over 40 inputs, each gate a fresh row
n_in, width, levels = 40, 37, 6
ops_all = [0, 4, 5, 7, 8, 9, 3, 6]                # AND NAND NOR OR XNOR XOR MUX NOT
ops, i0, i1, i2, out, off = [], [], [], [], [], [0]
rows = n_in
for l in range(levels):
    for g in range(width):
        op = ops_all[int(rng.integers(len(ops_all)))]
        a, b, c = (int(x) for x in rng.integers(0, rows, size=3))
        ops.append(op); i0.append(a); i1.append(b if op != 6 else -1); i2.append(c if op == 3 else -1); out.append(rows + g)
    rows += width
    off.append(len(ops))
arrs = [np.array(x, np.int32) for x in (ops, i0, i1, i2, out)] + [np.array(off, np.int64)]
prog = helm_amd.Program(sk, *arrs)
enc = ck.encrypt(rng.integers(0, 2, size=n_in).astype(bool))
ref = sk.wires(rows)
ref.upload(np.arange(n_in, dtype=np.int32), enc)
prog.run(ref)
sk.sync()
want = ref.download()

def fresh():
    w = sk.wires(rows)
    w.upload(np.arange(n_in, dtype=np.int32), enc)
    return w

# (1) the engine on a torch stream, the library's own RCCL communicator (world size 1, every launch sharded)
s = torch.cuda.Stream(device=0)
sk.set_stream(s.cuda_stream)
c = hc.Comm.single(0)
w = fresh()
r = ShardedRunner(GpuLevelExecutor(prog, w), 0, 1, comm=c, replicate_below=0)
r.run(); r.run()
s.synchronize()
res["comm_on_torch_stream_same"] = bool(np.array_equal(w.download(), want))
res["comm_collectives"] = c.stats()["collectives"]
res["comm_info"] = c.info()
# (2) torch.distributed's nccl backend: torch's current stream bound by the runner, torch tensors as staging buffers
dist.init_process_group("nccl", init_method="file://" + sys.argv[1], rank=0, world_size=1)
w = fresh()
r = ShardedRunner(GpuLevelExecutor(prog, w), 0, 1, dist, replicate_below=0, force=True)
r.run(); r.run()
torch.cuda.synchronize()
res["torch_nccl_same"] = bool(np.array_equal(w.download(), want))
res["torch_nccl_sharded_levels"] = len(r.sharded_levels)
# (3) the overlapped torch schedule: a side torch stream, torch events, the engine moved between torch streams
from helm_amd.distributed import launch_dependencies
deps = launch_dependencies(arrs[1], arrs[2], arrs[3], arrs[4], arrs[5], rows)
w = fresh()
r = ShardedRunner(GpuLevelExecutor(prog, w), 0, 1, dist, replicate_below=0, force=True, depends_on=deps)
r.run(); r.run()
torch.cuda.synchronize()
res["torch_overlapped_same"] = bool(np.array_equal(w.download(), want))
dist.destroy_process_group()
res["mapped_at_end"] = nv.mapped_hip_runtimes()
c.destroy()
sk.close()
print("RESULT " + json.dumps(res))
'''


def test_helm_amd_imported_before_torch_shares_torchs_runtime_and_streams():
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("HELM_HIP_RUNTIME", None)
        r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), os.path.join(d, "store")], capture_output=True, text=True,
                           env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])   # (the round-5 failure was an abort: rc -6)
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][len("RESULT "):])
    assert len(res["before_torch"]) == 1 and res["mapped"] == res["before_torch"] == res["mapped_at_end"], res
    assert res["comm_on_torch_stream_same"] and res["comm_collectives"] >= 2 * 6
    assert res["comm_info"]["world_size"] == 1 and res["comm_info"]["rccl_version"] > 0
    assert res["torch_nccl_same"] and res["torch_nccl_sharded_levels"] == 6
    assert res["torch_overlapped_same"]

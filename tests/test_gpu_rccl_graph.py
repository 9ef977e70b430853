"""GPU: HIP-graph recording beside a live RCCL communicator.

bench.py records its step graph in a process that has joined a torch.distributed "nccl" (= RCCL) group, whose
watchdog threads talk to the HIP runtime in the background.  The ROCm runtime refuses some calls made by OTHER
threads while a stream is being recorded, so this is rehearsed here with a one-rank RCCL group on the single GPU
of the test box: barrier and all-reduce, record and replay steps, all-gather of the intervals."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import numpy as np, torch, torch.distributed as dist
from sxmc_amd import capi, workloads
from sxmc_amd.mcmc import MCMC
torch.cuda.set_device(0)
capi.call("sxmc_set_device", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = workloads.config3(0.003, nevents=2000)
eager = MCMC(w, seed=5, lut_output=False)
want = eager.walk(w.events, 120, 0.1, sync_interval=50)
m = MCMC(w, seed=5, stream=capi.new_stream(), lut_output=False)
dist.barrier()
t = torch.ones(1, device="cuda")
dist.all_reduce(t)                      # leaves work for the watchdog to poll
got = m.walk(w.events, 120, 0.1, sync_interval=50, graph_steps=8)
dist.barrier()
out = torch.empty(4, device="cuda")
dist.all_gather_into_tensor(out, torch.arange(4, dtype=torch.float32, device="cuda"))
torch.cuda.synchronize()
assert want[1] == got[1] and np.array_equal(want[0], got[0])
assert out.cpu().tolist() == [0.0, 1.0, 2.0, 3.0]
# bench.py at N > 1 also makes a communicator through the C ABI (librccl) inside this torch process, beside torch's
# own: the id travels over the process group (sxmc_amd.dist.RcclComm), the rank count is asked of the communicator
from sxmc_amd import dist as sd
comm = sd.RcclComm()
assert comm.query() == (0, 1, 0)
block = np.arange(60, dtype=np.float32)
assert np.array_equal(comm.allgather_f32(block), block.reshape(1, 60))
comm.close()
dist.destroy_process_group()
print("rccl-graph-ok")
"""


@pytest.mark.gpu
def test_graph_recording_beside_an_rccl_communicator():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-graph-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_c_abi_rccl_allgather_single_rank():
    """sxmc_comm_* (the multi-GPU exchange behind the C ABI, librccl): the one-process-per-GPU form -- unique id,
    init_rank, all-gather -- with a world of one on this box's GPU."""
    import ctypes as C

    import numpy as np

    from sxmc_amd import capi
    from sxmc_amd.capi import DeviceArray
    uid = C.create_string_buffer(128)
    capi.call("sxmc_comm_unique_id", uid, 128)
    comm = C.c_void_p(0)
    capi.call("sxmc_comm_init_rank", uid, 128, 1, 0, C.byref(comm))
    rank, n = C.c_int(-1), C.c_int(-1)
    capi.call("sxmc_comm_rank", comm, C.byref(rank), C.byref(n))
    assert (rank.value, n.value) == (0, 1)
    send = DeviceArray(np.arange(60, dtype=np.float32))
    recv = DeviceArray.zeros(60, np.float32)
    stream = capi.new_stream()
    capi.call("sxmc_comm_allgather_f32", comm, capi.ptr(send), capi.ptr(recv), 60, capi.ptr(stream))
    capi.call("sxmc_stream_synchronize", capi.ptr(stream))
    assert np.array_equal(recv.get(), np.arange(60, dtype=np.float32))
    capi.call("sxmc_comm_destroy", comm)

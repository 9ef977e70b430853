"""Multi-GPU layer: fake experiments shard embarrassingly over the GPUs of a node.

The reference runs its ensemble as a sequential host loop (src/sxmc.cpp:59-145); iterations are
independent, so experiment k goes to rank k mod G, every rank holds a full replica of the MC sample
tables, and the data path needs NO collective.  The only exchange is the gather of the
per-experiment intervals (interval.h:22-27: point_estimate, lower, upper, coverage per parameter)
at the end -- one small RCCL all_gather over xGMI (backend "nccl" is RCCL on ROCm; "gloo" on CPU
for the tests).  torch.distributed is plumbing here, as is torch itself.
"""
import os

import numpy as np

INTERVAL_FIELDS = 4     # point_estimate, lower, upper, coverage (interval.h:22-27)


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), \
        int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Join the job torch.distributed.run started (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the env).
    Returns (rank, local_rank, world).  A single-process run needs no process group."""
    rank, local_rank, world = env_world()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                import torch
                # SXMC_DIST_BACKEND=gloo rehearses a multi-rank run on a box with fewer GPUs than ranks
                backend = os.environ.get("SXMC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl":
                import torch
                kw["device_id"] = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def spawn_ranks(nranks, argv, env=None, poll_seconds=0.2, grace_seconds=10.0, out=None, err=None):
    """`python bench.py --gpus N` started BARE: the parent -- which has made no HIP call and has not asked torch
    about the GPU, and never does -- starts N fresh worker processes of `argv` (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, what torch.distributed.run would set), relays rank 0's
    stdout as its own and sends the other ranks' stdout to stderr.  Children, never a re-exec: a process that has
    touched the GPU must not be replaced.  The first rank to fail ends the job: the others are terminated (their own
    PIDs, SIGTERM then SIGKILL after `grace_seconds`) and its exit code is returned.  Returns 0 when every rank did."""
    import socket
    import subprocess
    import sys
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = out if out is not None else sys.stdout
    err = err if err is not None else sys.stderr
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL between processes needs on this pool
    base.update(WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                SXMC_LAUNCHED_BY="bench.py")
    procs = []
    for rank in range(nranks):
        e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), GROUP_RANK="0")
        procs.append(subprocess.Popen(list(argv), env=e, stdout=out if rank == 0 else err, stderr=err))
    code = 0
    live = set(range(nranks))
    while live and code == 0:
        time.sleep(poll_seconds)
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                code = rc if rc > 0 else 128 - rc
                print("bench launcher: rank %d exited with %d; ending the other ranks" % (r, rc), file=err, flush=True)
                break
    if live:
        for r in live:
            procs[r].terminate()
        deadline = time.time() + grace_seconds
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return code


class RcclComm:
    """One process per GPU: this rank's RCCL communicator made through the C ABI (sxmc_comm_unique_id on rank 0,
    the 128-byte id handed round over the torch.distributed process group that already exists, sxmc_comm_init_rank
    everywhere) -- the same entry points the C++ runner uses.  `query()` asks the communicator itself."""

    def __init__(self, init_timeout=None):
        """Collective over the existing process group: EVERY rank must call it, and every rank leaves it the same way.
        Rank 0 ALWAYS broadcasts -- the id, or None and the reason when it could not make one (so the others are never
        left waiting in a broadcast that will not come: ADVICE r3); a missing id raises the same error on every rank,
        before any rank enters ncclCommInitRank, which is itself a collective.  That call runs under a time limit
        (`init_timeout` seconds, SXMC_RCCL_INIT_TIMEOUT, default 120): a rank still inside it when the limit passes
        cannot be rescued -- its peers may have failed -- so the process says so and exits non-zero, which ends the job
        (spawn_ranks and torch.distributed.run both end the other ranks) instead of hanging it."""
        import ctypes as C
        import threading

        import torch.distributed as dist

        from . import capi
        rank, _, world = env_world()
        box = [None, None]
        if rank == 0:
            try:
                buf = C.create_string_buffer(128)
                capi.call("sxmc_comm_unique_id", buf, 128)
                box = [buf.raw, None]
            except Exception as exc:      # noqa: BLE001 (the other ranks must hear of it, whatever it is)
                box = [None, "%s: %s" % (type(exc).__name__, exc)]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise capi.SxmcError(capi.ERR_HIP, "rank 0 could not make an RCCL id: %s" % box[1])
        if init_timeout is None:
            init_timeout = float(os.environ.get("SXMC_RCCL_INIT_TIMEOUT", "120"))
        h = C.c_void_p(0)
        lib = capi.load()
        device = C.c_int(-1)
        # (the rank's card, asked of the runtime BEFORE the helper thread exists: a thread that could not be bound to it
        # would join the communicator on card 0 -- a duplicate device, reported by RCCL much later or never)
        rc = lib.sxmc_get_device(C.byref(device))
        if rc != capi.OK or device.value < 0:
            raise capi.SxmcError(rc if rc != capi.OK else capi.ERR_HIP,
                                 "RcclComm: the current device could not be asked for: %s" % lib.sxmc_last_error().decode())
        result = {}

        def join_communicator():          # (a new host thread starts on device 0: bind it to this rank's card first)
            try:
                rc_dev = lib.sxmc_set_device(device.value)
                if rc_dev != capi.OK:
                    result["rc"] = rc_dev
                    result["why"] = "sxmc_set_device(%d): %s" % (device.value, lib.sxmc_last_error().decode())
                    return
                result["rc"] = lib.sxmc_comm_init_rank(box[0], 128, world, rank, C.byref(h))
                if result["rc"] != capi.OK:
                    result["why"] = lib.sxmc_comm_last_error().decode()
            except BaseException as exc:      # noqa: BLE001 (re-raised on the caller's thread)
                result["exc"] = exc

        t = threading.Thread(target=join_communicator, daemon=True)
        t.start()
        t.join(init_timeout)
        if t.is_alive():
            import sys
            print("sxmc_amd.dist: rank %d is still inside ncclCommInitRank after %.0f s (a peer failed or never "
                  "arrived); ending the job" % (rank, init_timeout), file=sys.stderr, flush=True)
            os._exit(70)
        if "exc" in result:
            raise result["exc"]
        if result.get("rc") != capi.OK:
            raise capi.SxmcError(result.get("rc", capi.ERR_HIP), result.get("why") or "RcclComm: the helper thread "
                                 "returned nothing")
        self.h = h

    def query(self):
        import ctypes as C

        from . import capi
        r, n, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        rc = capi.load().sxmc_comm_query(self.h, C.byref(r), C.byref(n), C.byref(d))
        if rc != capi.OK:
            raise capi.SxmcError(rc, capi.load().sxmc_comm_last_error().decode())
        return r.value, n.value, d.value

    def allgather_f32(self, block):
        """block: float32 numpy array, same size on every rank -> [world, block.size] (through device buffers)."""
        from . import capi
        _, n, _ = self.query()
        block = np.ascontiguousarray(block, np.float32).ravel()
        send = capi.DeviceArray(block)
        recv = capi.DeviceArray.empty(n * block.size, np.float32)
        st = capi.new_stream()
        rc = capi.load().sxmc_comm_allgather_f32(self.h, capi.ptr(send), capi.ptr(recv), block.size, capi.ptr(st))
        if rc != capi.OK:
            raise capi.SxmcError(rc, capi.load().sxmc_comm_last_error().decode())
        capi.call("sxmc_stream_synchronize", capi.ptr(st))
        out = recv.get().reshape(n, block.size)
        capi.call("sxmc_stream_destroy", capi.ptr(st))
        return out

    def close(self):
        from . import capi
        if self.h:
            capi.load().sxmc_comm_destroy(self.h)
            self.h = None


def collective_record(device_index, device_info):
    """What the job's collectives really ran on, for the bench line: the backend torch.distributed uses, its world
    size, the rank count RCCL reports (from a communicator, not from the environment), an all-reduce of ones over
    the process group, and every rank's device (index, name, PCI bus id, pid, host) -- so a line tells eight ranks on
    eight cards from two ranks rehearsed over gloo on one.  Collective: every rank must call it.  None at world 1."""
    rank, local_rank, world = env_world()
    if world == 1:
        return None, None
    import socket

    import torch
    import torch.distributed as dist
    backend = dist.get_backend()
    t = torch.ones(1, dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t)
    mine = {"rank": rank, "local_rank": local_rank, "device_index": int(device_index), "name": device_info.get("name"),
            "pci_bus_id": device_info.get("pci_bus_id"), "pid": os.getpid(), "host": socket.gethostname()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    cards = {(d["host"], d["pci_bus_id"]) for d in everyone}
    comm, nranks, comm_device, note = None, None, None, None
    if backend == "nccl":
        # librccl through the C ABI, beside torch's own communicator.  Should it fail on some rank (it has never been
        # run on more than one card: no multi-GPU box was available to the builder), every rank learns of it and the
        # job goes on with torch.distributed's all_gather alone -- and says so in the line -- instead of stopping.
        # Every rank takes the same sequence of collectives whatever fails where: RcclComm's broadcast (rank 0 always
        # sends), ncclCommInitRank only if an id arrived (under a time limit: a rank stuck in it ends the job), then
        # the MIN all-reduce below.
        try:
            comm = RcclComm()
            _, nranks, comm_device = comm.query()
            mine_ok, why = 1.0, None
        except Exception as exc:      # noqa: BLE001 (whatever it is, the other ranks must hear of it)
            comm, mine_ok, why = None, 0.0, "%s: %s" % (type(exc).__name__, exc)
        ok = torch.tensor([mine_ok], dtype=torch.float64, device=_device_for_collectives())
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            if comm is not None:
                comm.close()
            comm, nranks, comm_device = None, None, None
            note = "the C-ABI RCCL communicator could not be made on every rank (%s); intervals gathered by " \
                   "torch.distributed's all_gather" % (why or "another rank failed")
    else:
        note = ("rehearsal: the ranks' collectives ran over %s (SXMC_DIST_BACKEND), not RCCL; %d rank(s) share a card"
                % (backend, world - len(cards)))
    rec = {"backend": backend, "library": "RCCL (torch.distributed 'nccl' on ROCm) + librccl through the C ABI "
           "(sxmc_comm_*)" if backend == "nccl" else backend,
           "world_size": dist.get_world_size(), "rccl_nranks": nranks, "rccl_device_of_rank0": comm_device,
           "allreduce_of_ones": float(t.item()), "distinct_cards": len(cards), "devices": everyone, "note": note,
           "launched_by": os.environ.get("SXMC_LAUNCHED_BY", "torch.distributed.run")}
    return rec, comm


def shutdown():
    _, _, world = env_world()
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def experiments_of_rank(nexperiments, rank, world):
    """Experiment k -> rank k mod world (round-robin keeps ranks within one experiment of each other)."""
    return list(range(rank, nexperiments, world))


def experiment_seed(base_seed, k):
    """Per-experiment seed: the reference draws all experiments from one sequential gRandom stream
    (sxmc.cpp:190-191), which cannot be sharded; a hash of (base, k) replaces it."""
    x = (int(base_seed) * 0x9E3779B97F4A7C15 + (int(k) + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    return x & 0x7FFFFFFFFFFFFFFF


def _device_for_collectives():
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier():
    _, _, world = env_world()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value):
    """MAX of a python float over the ranks (the bench contract's timing rule)."""
    _, _, world = env_world()
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    _, _, world = env_world()
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_intervals(local, nexperiments, nparameters, comm=None):
    """local: float32 [len(experiments_of_rank), nparameters, 4] in the rank's experiment order.
    Returns float32 [nexperiments, nparameters, 4] in experiment order on every rank (all_gather of
    equal-size padded blocks: ~61 KB at 256 experiments x 15 parameters, latency-bound).
    comm: an RcclComm -- the exchange then runs on librccl through the C ABI (sxmc_comm_allgather_f32, what the
    C++ runner calls) instead of torch.distributed's all_gather."""
    rank, _, world = env_world()
    local = np.ascontiguousarray(local, dtype=np.float32).reshape(-1, nparameters, INTERVAL_FIELDS)
    assert local.shape[0] == len(experiments_of_rank(nexperiments, rank, world))
    if world == 1:
        return local.copy()
    import torch
    import torch.distributed as dist
    per = (nexperiments + world - 1) // world
    pad = np.full((per, nparameters, INTERVAL_FIELDS), np.nan, dtype=np.float32)
    pad[: local.shape[0]] = local
    if comm is not None:
        blocks = comm.allgather_f32(pad).reshape((world,) + pad.shape)
    else:
        dev = _device_for_collectives()
        mine = torch.from_numpy(pad).to(dev)
        out = torch.empty((world,) + pad.shape, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out.view(-1), mine.view(-1)) if dist.get_backend() == "nccl" else \
            dist.all_gather(list(out.unbind(0)), mine)
        blocks = out.cpu().numpy()
    full = np.empty((nexperiments, nparameters, INTERVAL_FIELDS), dtype=np.float32)
    for r in range(world):
        ks = experiments_of_rank(nexperiments, r, world)
        full[ks] = blocks[r, : len(ks)]
    return full


def median(values):
    """utils.h:76-90 `median`: middle of the sorted list, mean of the two middle ones for even sizes."""
    v = sorted(float(x) for x in values)
    half = len(v) // 2
    return 1.0 * (v[half - 1] + v[half]) / 2 if len(v) % 2 == 0 else v[half]

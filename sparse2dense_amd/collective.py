"""Sum all-reduce of the small fp32 SyncBN vectors (82 per training step on this path).

Two routes.  (a) `torch.distributed.all_reduce` on the default process group: the DEFAULT whenever there is more than one rank -
SyncBN vectors and gradient buckets then share ONE communicator and are issued in host program order, which is the same on
every rank.  (b) The RCCL communicator of csrc/comm.hip, called on the stream the batch-norm kernels run on (no c10d stream
hand-off, ~3 us of host time per call instead of ~25): opt-in with `S2D_RCCL_DIRECT=1` at world > 1, because a second
communicator running beside c10d's has only ever been exercised with ONE rank here (no multi-GPU box inside a round; NCCL-family
libraries can deadlock when two communicators' collectives interleave differently across ranks).  With one rank
(`S2D_FORCE_DDP=1` measurement runs) it is on by default.  It is bootstrapped collectively by `init_direct()`; if that does not
succeed on EVERY rank within a time limit, all ranks fall back to route (a) together."""
import ctypes
import os
import threading

import torch
import torch.distributed as dist

_DIRECT = False
_DIRECT_PG = None      # the process group the direct communicator was bootstrapped over
_CANCEL = threading.Event()   # set when a bootstrap timed out: the helper thread then tears its communicator down


def sync_on():
    """One predicate for every self-synchronising batch norm of the path (FeatureBatchNorm1d, FastBatchNorm2d/3d): a
    process group is up and there is more than one rank (S2D_FORCE_DDP=1 exercises the route with one rank;
    S2D_DEBUG_NO_SYNCBN=1 is a measurement hook that splits gradient all-reduce cost from SyncBN cost)."""
    if not (dist.is_available() and dist.is_initialized()) or os.environ.get("S2D_DEBUG_NO_SYNCBN", "0") == "1":
        return False
    return dist.get_world_size() > 1 or os.environ.get("S2D_FORCE_DDP", "0") == "1"


def direct_enabled():
    return _DIRECT and dist.is_initialized() and _DIRECT_PG is dist.group.WORLD


_direct_call = None   # test hook: callable(tensor) standing in for s2d_comm_allreduce_sum_f32 - tests/test_dp_gloo.py drives the direct route's
#                       call sequence at world 2 through a mock communicator (a second gloo group) with it


def allreduce_sum_(t: torch.Tensor):
    """In-place sum of `t` over the ranks, ordered on the current stream."""
    if _direct_call is not None and direct_enabled() and t.is_contiguous():
        _direct_call(t)
        return t
    if direct_enabled() and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        from . import _lib
        lib = _lib.load()
        rc = lib.s2d_comm_allreduce_sum_f32(t.data_ptr(), t.numel(), torch._C._cuda_getCurrentRawStream(t.device.index))
        if rc:
            raise _lib.S2DError(f"s2d_comm_allreduce_sum_f32 failed ({rc}): {_lib.last_error()}")
        return t
    dist.all_reduce(t)
    return t


def init_direct(device_index, timeout_s=120.0):
    """Collective over the default process group (call it on every rank, after torch.cuda.set_device).  Returns True when
    the direct route is on.  Every step that can block runs in a helper thread under a time limit, and the outcome is
    agreed on with a torch.distributed all-reduce, so the ranks never end up on different routes."""
    global _DIRECT, _DIRECT_PG
    if direct_enabled():   # a second model wrapped in the same process group: the communicator is up already
        return True
    _DIRECT, _DIRECT_PG = False, None   # (a destroyed and re-created process group starts over)
    _CANCEL.clear()
    if not dist.is_initialized() or dist.get_backend() != "nccl":
        return False
    if os.environ.get("S2D_RCCL_DIRECT", "1" if dist.get_world_size() == 1 else "0") == "0":
        return False
    from . import _lib
    lib = _lib.load()
    world, rank = dist.get_world_size(), dist.get_rank()
    rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")   # the copy torch has mapped
    if os.path.exists(rccl):
        lib.s2d_comm_load_library(rccl.encode())
    ok = bool(lib.s2d_comm_available())
    ident = [None]
    if ok and rank == 0:
        buf = ctypes.create_string_buffer(128)
        ok = lib.s2d_comm_unique_id(buf) == 0
        ident[0] = bytes(buf.raw) if ok else None
    dist.broadcast_object_list(ident, src=0)
    ok = ok and ident[0] is not None
    result = {"ok": False, "why": "no unique id / RCCL not resolved"}

    def bootstrap():
        try:
            torch.cuda.set_device(device_index)
            if lib.s2d_comm_init(ident[0], world, rank) != 0:
                result["why"] = "comm_init: " + _lib.last_error()
                return
            if _CANCEL.is_set():   # the caller gave up while ncclCommInitRank was blocked: no probe, no stray communicator
                lib.s2d_comm_shutdown()
                return
            probe = torch.ones(8, dtype=torch.float32, device=torch.device("cuda", device_index))
            if lib.s2d_comm_allreduce_sum_f32(probe.data_ptr(), probe.numel(), torch._C._cuda_getCurrentRawStream(device_index)) != 0:
                result["why"] = "probe all-reduce: " + _lib.last_error()
                return
            torch.cuda.current_stream(device_index).synchronize()
            result["ok"] = bool((probe == float(world)).all().item())
            result["why"] = "probe sum mismatch"
        except Exception as e:   # any failure -> torch.distributed route
            result["ok"] = False
            result["why"] = repr(e)

    if ok:
        th = threading.Thread(target=bootstrap, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            _CANCEL.set()
            result["why"] = f"bootstrap did not finish within {timeout_s:.0f} s"
        ok = (not th.is_alive()) and result["ok"]
    if not ok:
        import sys
        print(f"[s2d] rank {rank}: direct RCCL route off ({result['why']}); using torch.distributed", file=sys.stderr, flush=True)
    flag = torch.tensor([1.0 if ok else 0.0], device=torch.device("cuda", device_index))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    _DIRECT = bool(flag.item() > 0.5)
    _DIRECT_PG = dist.group.WORLD if _DIRECT else None
    if not _DIRECT and ok:   # this rank's communicator came up but another rank's did not: drop it, all ranks use c10d
        lib.s2d_comm_shutdown()
    return _DIRECT

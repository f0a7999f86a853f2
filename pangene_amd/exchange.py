"""torch.distributed plumbing for the library's exchange hook (pg_set_exchange, include/pangene_amd.h).

One process per GPU; genomes are sharded across ranks; the library asks for a handful of small integer
all-reduces / all-gathers per round on buffers that live in its own memory: HBM for the
HIP backend (-> RCCL over xGMI through the "nccl" backend) or host memory for the oracle backend used
by the CPU tests (-> gloo).  This module only wraps those raw pointers as tensors and calls
torch.distributed; it never touches the data.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi

_DT = {0: (torch.int32, np.int32, 4), 1: (torch.int64, np.int64, 8)}


class _DevPtr:
    """Minimal __cuda_array_interface__ holder so torch can alias library-owned HBM without a copy."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _tensor(ptr: int, nbytes: int, is_device: int, device) -> torch.Tensor:
    if is_device:
        return torch.as_tensor(_DevPtr(ptr, nbytes), device=device)
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))


def install(lib: C.CDLL, device=None, group=None):
    """Registers the exchange callbacks on `lib`; returns an object that must be kept alive."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # HBM buffers over a host-only process group (gloo): staged through host memory.  Only meant for tests that run several
    # ranks of the HIP backend on ONE GPU, where RCCL refuses to put two ranks on a device.
    via_host = dist.get_backend(group) == "gloo"

    def allreduce(user, buf, count, dtype, op, is_device):
        try:
            tdt, _, sz = _DT[dtype]
            t = _tensor(buf, count * sz, is_device, device).view(tdt)
            rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
            if is_device and via_host:
                h = t.cpu()
                dist.all_reduce(h, op=rop, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=rop, group=group)
            if is_device:
                torch.cuda.current_stream(device).synchronize()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("[pangene_amd.exchange] allreduce failed:", e, flush=True)
            return -1

    def allgather(user, src, dst, nbytes, is_device):
        try:
            tin = _tensor(src, nbytes, is_device, device)
            tout = _tensor(dst, nbytes * world, is_device, device)
            if is_device and via_host:
                parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, tin.cpu(), group=group)
                tout.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(tout, tin, group=group)
            if is_device:
                torch.cuda.current_stream(device).synchronize()
            return 0
        except Exception as e:
            print("[pangene_amd.exchange] allgather failed:", e, flush=True)
            return -1

    x = capi.pg_exchange_t()
    x.rank, x.world, x.user = rank, world, None
    x.stream_ordered = 0  # torch's collectives run on torch's streams: the library waits for its own first
    keep = (capi.ALLREDUCE_CB(allreduce), capi.ALLGATHER_CB(allgather))
    x.allreduce, x.allgather = keep
    lib.pg_set_exchange(C.byref(x))
    return (x, keep)


def install_native(lib: C.CDLL, group=None) -> bool:
    """Sharded runs of the HIP backend: let the library talk to RCCL itself (pg_rccl_init, include/pangene_amd.h) --
    collectives are enqueued on the kernels' stream, no Python in the loop.  torch.distributed is only used to hand the
    128-byte bootstrap id from rank 0 to the others.  Returns False (with the reason on stderr) if RCCL cannot be set up;
    the caller then falls back to install()."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = C.create_string_buffer(128)
    ok = 1
    if rank == 0 and lib.pg_rccl_unique_id(uid) != 0:
        ok = 0
    box = [uid.raw if ok else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    if box[0] is None:
        if rank == 0:
            print("[pangene_amd.exchange] native RCCL exchange unavailable:", lib.pg_rccl_error().decode(), flush=True)
        return False
    uid = C.create_string_buffer(box[0], 128)
    rc = lib.pg_rccl_init(rank, world, uid)
    flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32)
    if world > 1:  # all ranks take the same path
        if dist.get_backend(group) == "nccl":
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        print("[pangene_amd.exchange] pg_rccl_init failed on some rank:", lib.pg_rccl_error().decode(), flush=True)
        lib.pg_rccl_finalize()
        return False
    return True


def uninstall(lib: C.CDLL):
    lib.pg_set_exchange(None)

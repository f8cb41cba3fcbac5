"""RCCL called directly (ctypes over librccl.so, the library PyTorch-ROCm ships): the communicator behind the frame-sharded /
CFG-parallel denoising step when its exchanges have to be part of a captured hipGraph.

Why not torch.distributed's process group for that: ProcessGroupNCCL runs every collective on its own stream under a watchdog
thread that polls the work's end event; an event recorded in a CAPTURING stream may not be queried, and the watchdog aborts
the process ("operation not permitted on an event last recorded in a capturing stream" -- observed on MI355X with torch
2.10 / RCCL 2.26 when `bench.py --parallel frames --graph` captured through the process group;
profiles/r03_graph_capture_pg_watchdog.txt).  A communicator of our own has no watchdog and enqueues exactly where we say:
on the caller's stream, or on this communicator's side stream fenced by events -- both capturable.

Bootstrap: the 128-byte ncclUniqueId of rank 0 travels through torch.distributed (any backend: `broadcast_object_list`), the
only use of the process group on this path.  One process per GPU, one communicator per (sub)group.

    comm = RcclComm.from_group(group)          # collective over the ranks of `group` (None = world)
    comm.all_reduce_(t)                        # in place, sum, on the current HIP stream
    comm.all_gather_into(out, inp)             # out [world * n] <- inp [n] of every rank; inp may be out's own slot
    comm.all_to_all_single(out, inp)           # equal splits
    comm.batch_p2p([("send", t, peer), ("recv", t, peer), ...])     # one ncclGroup; peers are ranks INSIDE the communicator
    ev = comm.side(lambda: comm.all_gather_into(out, inp))          # the same on the communicator's own stream, after what the current stream
    comm.join(ev)                                                   # has enqueued so far; join() makes the current stream wait for it
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch

_lib = None


class RcclError(RuntimeError):
    pass


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


_DTYPE = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7, torch.float64: 8, torch.bfloat16: 9}
_SUM, _MAX = 0, 2


def lib() -> C.CDLL:
    """librccl.so of the running PyTorch-ROCm (the same library torch.distributed's "nccl" backend uses)."""
    global _lib
    if _lib is None:
        path = os.environ.get("ME_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            raise RcclError(f"{path} not found (set ME_RCCL_LIB)")
        L = C.CDLL(path)
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RcclError(f"{what}: {lib().ncclGetErrorString(rc).decode(errors='replace')} (rc={rc})")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPE:
        raise ValueError(f"{name}: RCCL needs a contiguous device tensor of a supported dtype, got {t.dtype} {tuple(t.shape)} contiguous={t.is_contiguous()}")


class RcclComm:
    def __init__(self, rank: int, world: int, unique_id: bytes, device: Optional[torch.device] = None):
        if len(unique_id) != 128:
            raise ValueError("ncclUniqueId is 128 bytes")
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        uid = _UniqueId()
        C.memmove(C.byref(uid), unique_id, 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().ncclCommInitRank(C.byref(self._comm), world, uid, rank), "ncclCommInitRank")
        self._side = None

    @staticmethod
    def unique_id() -> bytes:
        uid = _UniqueId()
        _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        return bytes(uid.internal)

    @classmethod
    def from_group(cls, group=None) -> "RcclComm":
        """A communicator over the ranks of a torch.distributed group (None = the default group); the id is broadcast through that group."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        src = dist.get_process_group_ranks(group)[0] if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        return cls(rank, world, box[0])

    def destroy(self) -> None:
        if self._comm:
            lib().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()

    # ---- collectives on the current HIP stream ----------------------------------------------------------------------------
    def all_reduce_(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        _chk(t, "all_reduce_")
        _check(lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPE[t.dtype], _SUM if op == "sum" else _MAX, self._comm, _stream()), "ncclAllReduce")
        return t

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor) -> torch.Tensor:
        """out [world * inp.numel()] <- inp of every rank, rank-major.  inp may be the rank's own slot of out (in place)."""
        _chk(out, "all_gather_into.out")
        _chk(inp, "all_gather_into.inp")
        if out.numel() != self.world * inp.numel() or out.dtype != inp.dtype:
            raise ValueError("all_gather_into: out must hold world x inp")
        _check(lib().ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), _DTYPE[inp.dtype], self._comm, _stream()), "ncclAllGather")
        return out

    def batch_p2p(self, ops: Sequence[Tuple[str, torch.Tensor, int]]) -> None:
        """One ncclGroup of sends / receives; peers are ranks inside this communicator."""
        if not ops:
            return
        L, st = lib(), _stream()
        for kind, t, peer in ops:
            _chk(t, "batch_p2p")
            if kind not in ("send", "recv") or not (0 <= peer < self.world) or peer == self.rank:
                raise ValueError(f"batch_p2p: bad op ({kind}, peer {peer})")
        _check(L.ncclGroupStart(), "ncclGroupStart")
        try:
            for kind, t, peer in ops:
                fn = L.ncclSend if kind == "send" else L.ncclRecv
                _check(fn(t.data_ptr(), t.numel(), _DTYPE[t.dtype], peer, self._comm, st), "ncclSend" if kind == "send" else "ncclRecv")
        finally:
            _check(L.ncclGroupEnd(), "ncclGroupEnd")

    def all_to_all_single(self, out: torch.Tensor, inp: torch.Tensor) -> torch.Tensor:
        """Equal splits: chunk r of inp goes to rank r, chunk r of out comes from rank r (a group of world sends and receives; the rank's own
        chunk is a device copy)."""
        _chk(out, "all_to_all_single.out")
        _chk(inp, "all_to_all_single.inp")
        if out.numel() != inp.numel() or inp.numel() % self.world or out.dtype != inp.dtype:
            raise ValueError("all_to_all_single: equal-size tensors divisible by the world size")
        n = inp.numel() // self.world
        fi, fo = inp.reshape(-1), out.reshape(-1)
        fo[self.rank * n:(self.rank + 1) * n].copy_(fi[self.rank * n:(self.rank + 1) * n])
        ops = []
        for r in range(self.world):
            if r != self.rank:
                ops += [("send", fi[r * n:(r + 1) * n], r), ("recv", fo[r * n:(r + 1) * n], r)]
        self.batch_p2p(ops)
        return out

    # ---- the same on the communicator's own stream (overlap with what the caller enqueues next) ------------------------------
    def side(self, fn: Callable[[], object]) -> torch.cuda.Event:
        """Run `fn` (calls on this communicator) on the communicator's side stream, ordered after everything the current stream has enqueued;
        returns the event join() waits on.  Tensors `fn` touches must stay alive until the join."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            fn()
            ev = self._side.record_event()
        return ev

    @staticmethod
    def join(ev: Optional[torch.cuda.Event]) -> None:
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

"""Frame sharding of one clip over the GPUs of a node: each rank owns f/R consecutive frames of all
four batch rows, so the reconstruction -> editing K/V injection stays rank-local.  The cross-frame couplings of the
reference become these exchanges (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests):

  * spatial attn1 (keys of the previous frame only, attention_2d.py:732-740 and the edited variant): a ONE-frame halo
    of the layer's K|V rows from the previous rank (point-to-point; `PrevFrameHalo`), 1 / f_loc of the all-gather;
  * adapter sparse-causal attention (first / previous frame of an 8-frame chunk): at most TWO remote frames of the layer's
    K|V rows, point-to-point (`ChunkHalo`; the all-gather `FrameShard.start_kv` remains for `adapter="gather"`);
  * temporal attention (every pixel attends over all earlier frames; attention_2d.py:534-545): a frame<->pixel ALL-TO-ALL of the
    normed input rows -- each rank then holds all frames of N/R pixels, projects q|k|v there (a row-wise GEMM commutes with the
    row exchange), runs me_tattn on them (q_parts = kv_parts = R) and a second all-to-all returns the output rows to their
    frame owners: (R-1)/R * 2C columns per row cross the links instead of (R-1) * 2C for the all-gather (which stays as
    `temporal="gather"`, used when R does not divide the pixel count);
  * TemporalConv k=3: one-frame halos from both neighbours (point-to-point);
  * ResnetBlock2D / conv_norm_out GroupNorm (statistics span all frames): all-reduce of (sum, sum of squares).
ControlNet, cross-attention, feed-forward, spatial convolutions, per-frame GroupNorm, CFG and DDIM are rank-local.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

# data-path exchange accounting (bench.py reports it per step): kind -> [calls, payload bytes this rank sends or contributes]
STATS = {}
# How long the CALLER'S stream is held by every exchange (round 5: so that a first real scaling run can be read -- which exchange kinds the step waits for).
# TIMING on: every blocking exchange and every join of an asynchronous one is bracketed by two events on the current stream (CUDA) or two clock reads
# (CPU tensors); the interval covers waiting for the peers AND the transfer itself.  kind -> [event pairs] / accumulated seconds.  Off inside a graph
# capture (events cannot be timed there) and by default (the tests' counters do not need it).
TIMING = False
_WAIT_EVENTS = {}
_WAIT_SECONDS = {}
_last_kind = "?"


def _count(kind: str, t: torch.Tensor, frac: float = 1.0) -> None:
    global _last_kind
    _last_kind = kind
    c = STATS.setdefault(kind, [0, 0])
    c[0] += 1
    c[1] += int(t.numel() * t.element_size() * frac)


def _set_kind(kind: str) -> None:
    """Label the next exchange without counting it (a rank that only RECEIVES in an exchange has counted nothing)."""
    global _last_kind
    _last_kind = kind


class _timed:
    """with _timed(kind, cuda): ... -- the interval of the caller's stream (or of the host, for CPU tensors) spent inside one exchange."""

    def __init__(self, kind: str, cuda: bool):
        self.kind, self.cuda = kind, cuda and TIMING and not torch.cuda.is_current_stream_capturing()
        self.cpu = TIMING and not cuda

    def __enter__(self):
        if self.cuda:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        elif self.cpu:
            import time
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.cuda:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _WAIT_EVENTS.setdefault(self.kind, []).append((self.e0, e1))
        elif self.cpu:
            import time
            _WAIT_SECONDS[self.kind] = _WAIT_SECONDS.get(self.kind, 0.0) + time.perf_counter() - self.t0
        return False


class _Pending:
    """An asynchronous exchange in flight: the adapter's own handle plus the kind it was counted under (for the join's timing)."""

    def __init__(self, kind: str, inner, cuda: bool):
        self.kind, self.inner, self.cuda = kind, inner, cuda


def reset_stats() -> None:
    STATS.clear()
    _WAIT_EVENTS.clear()
    _WAIT_SECONDS.clear()


def stats_summary(steps: int = 1) -> dict:
    out = {k: {"calls_per_step": v[0] / steps, "mbytes_per_step": round(v[1] / steps / 1e6, 3)} for k, v in sorted(STATS.items())}
    out["total"] = {"calls_per_step": sum(v[0] for v in STATS.values()) / steps, "mbytes_per_step": round(sum(v[1] for v in STATS.values()) / steps / 1e6, 3)}
    if _WAIT_EVENTS or _WAIT_SECONDS:
        if _WAIT_EVENTS:
            torch.cuda.synchronize()
        tot = 0.0
        for k in set(_WAIT_EVENTS) | set(_WAIT_SECONDS):
            ms = sum(a.elapsed_time(b) for a, b in _WAIT_EVENTS.get(k, [])) + 1e3 * _WAIT_SECONDS.get(k, 0.0)
            out.setdefault(k, {})["stream_held_ms_per_step"] = round(ms / steps, 3)
            tot += ms
        out["total"]["stream_held_ms_per_step"] = round(tot / steps, 3)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Exchange backends.  Every data-path exchange of this module goes through one of these adapters (same interface):
#   TorchExchange -- torch.distributed's process group ("gloo" in the CPU tests, "nccl" = RCCL for eager multi-GPU runs): collectives run on the
#                    group's own stream under its watchdog; asynchronous forms return the Work objects.
#   HostStagedExchange -- verification only: several ranks on ONE GPU, exchanges staged through host memory and a CPU process group.
#   RcclExchange  -- RCCL called directly (motioneditor_amd/rccl.py): no watchdog, enqueued on the caller's stream or on the communicator's
#                    side stream behind an event -- the form a captured hipGraph can hold (MotionEditorPipeline.denoise_step_graphed).
# p2p peers are ranks INSIDE the group.
# ---------------------------------------------------------------------------------------------------------------------
class TorchExchange:
    kind = "torch"

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._ranks = dist.get_process_group_ranks(group) if group is not None else list(range(self.world))

    def all_reduce_(self, t: torch.Tensor, op: str = "sum") -> None:
        with _timed(_last_kind, t.is_cuda):
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=self.group)

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        with _timed(_last_kind, out.is_cuda):
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_gather_start(self, out: torch.Tensor, inp: torch.Tensor):
        return _Pending(_last_kind, dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True), out.is_cuda)   # in place: inp = this rank's slot of out

    def all_to_all(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        with _timed(_last_kind, recv.is_cuda):
            dist.all_to_all_single(recv, send, group=self.group)

    def p2p_start(self, ops_):
        if not ops_:
            return None
        return _Pending(_last_kind, dist.batch_isend_irecv([dist.P2POp(dist.isend if k == "send" else dist.irecv, t, self._ranks[peer], self.group) for k, t, peer in ops_]),
                        ops_[0][1].is_cuda)

    def finish(self, handle) -> None:
        if handle is None:
            return
        kind, cuda = "?", False
        if isinstance(handle, _Pending):
            kind, cuda, handle = handle.kind, handle.cuda, handle.inner
        with _timed("join " + kind, cuda):
            for r in (handle if isinstance(handle, (list, tuple)) else [handle]):
                r.wait()


class RcclExchange:
    kind = "rccl"

    def __init__(self, group=None):
        from . import rccl
        self.group = group
        self.comm = rccl.RcclComm.from_group(group)
        self.rank, self.world = self.comm.rank, self.comm.world

    def all_reduce_(self, t: torch.Tensor, op: str = "sum") -> None:
        with _timed(_last_kind, True):
            self.comm.all_reduce_(t, op)

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        with _timed(_last_kind, True):
            self.comm.all_gather_into(out, inp)

    def all_gather_start(self, out: torch.Tensor, inp: torch.Tensor):
        return _Pending(_last_kind, (self.comm.side(lambda: self.comm.all_gather_into(out, inp)), out, inp), True)     # tensors kept alive until finish()

    def all_to_all(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        with _timed(_last_kind, True):
            self.comm.all_to_all_single(recv, send)

    def p2p_start(self, ops_):
        if not ops_:
            return None
        return _Pending(_last_kind, (self.comm.side(lambda: self.comm.batch_p2p(ops_)), ops_), True)

    def finish(self, handle) -> None:
        if handle is not None:
            with _timed("join " + handle.kind, True):
                self.comm.join(handle.inner[0])


class HostStagedExchange(TorchExchange):
    """VERIFICATION adapter, not a data path: every exchange goes device -> host -> (a CPU process group, i.e. gloo) -> host -> device.  It lets several
    ranks share ONE GPU -- RCCL refuses two ranks on a device -- so that the real kernels meet the real, non-degenerate exchange pattern of the
    frame-sharded step (remote halos, the frame<->pixel all-to-all over R > 1 parts, statistics summed over ranks) on a one-GPU box
    (tests/test_frame_shard_gpu.py).  Blocking: `.cpu()` waits for the producing kernels, the copy back is ordered on the current stream; the
    asynchronous forms complete before they return."""
    kind = "staged"

    def all_reduce_(self, t: torch.Tensor, op: str = "sum") -> None:
        c = t.cpu()
        super().all_reduce_(c, op)
        t.copy_(c)

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        co = torch.empty(out.shape, dtype=out.dtype)
        super().all_gather_into(co, inp.cpu().contiguous())
        out.copy_(co)

    def all_gather_start(self, out: torch.Tensor, inp: torch.Tensor):
        self.all_gather_into(out, inp)
        return None

    def all_to_all(self, recv: torch.Tensor, send: torch.Tensor) -> None:
        cr = torch.empty(recv.shape, dtype=recv.dtype)
        super().all_to_all(cr, send.cpu().contiguous())
        recv.copy_(cr)

    def p2p_start(self, ops_):
        if not ops_:
            return None
        staged = [(k, t, (t.cpu().contiguous() if k == "send" else torch.empty(t.shape, dtype=t.dtype)), peer) for k, t, peer in ops_]
        super().finish(super().p2p_start([(k, c, peer) for k, _t, c, peer in staged]))
        for k, t, c, _peer in staged:
            if k == "recv":
                t.copy_(c)
        return None


_exchanges = {}


def exchange(group=None, kind: str = "torch"):
    """The exchange adapter of a process group (one per (group, kind); creating an RcclExchange is a collective over the group)."""
    if isinstance(group, (TorchExchange, RcclExchange)):
        return group
    key = (id(group) if group is not None else None, kind)
    x = _exchanges.get(key)
    if x is None:
        x = _exchanges[key] = {"rccl": RcclExchange, "staged": HostStagedExchange}.get(kind, TorchExchange)(group)
    return x


class FrameShard:
    def __init__(self, f_total: int, group=None, temporal: str = "a2a", adapter: str = "halo", comm: str = "torch", side_group=None):
        """side_group (round 6, `--shard-overlap`): a SECOND process group over the same ranks.  The content-aware adapter then runs on the step's side
        stream with its own FrameShard (`side_shard`: its chunk halos, TemporalConv halos and frame<->pixel all-to-alls travel on the second group's
        communicator), beside the UNet's down path and mid block -- as on one GPU -- instead of serialised behind them on the main stream.  Two
        communicators because collectives of ONE communicator execute in issue order: the main stream's GroupNorm all-reduce would otherwise queue
        behind an adapter exchange whose input the side stream has not produced yet.  Every rank issues the same host program, so each communicator sees
        one order on all ranks.  Creating the group is a collective over the default group (dist.new_group): the caller does it (bench.py, tests)."""
        if temporal not in ("a2a", "gather") or adapter not in ("halo", "gather"):
            raise ValueError("temporal must be 'a2a' or 'gather', adapter 'halo' or 'gather'")
        self.temporal, self.adapter = temporal, adapter
        self.group = group
        self.side_shard = FrameShard(f_total, side_group, temporal, adapter, comm) if side_group is not None else None
        self.x = exchange(group, comm)          # every exchange below goes through it
        self.rank, self.world = self.x.rank, self.x.world
        if f_total % self.world:
            raise ValueError(f"{f_total} frames do not split evenly over {self.world} ranks")
        self.f_total = f_total
        self.f_loc = f_total // self.world
        self.frame0 = self.rank * self.f_loc

    # ---- GroupNorm statistics -------------------------------------------------------------------------
    def allreduce_(self, t: torch.Tensor) -> None:
        _count("all_reduce(groupnorm stats)", t)
        self.x.all_reduce_(t)

    # ---- K|V rows of all frame shards, part-major -----------------------------------------------------
    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        _count("all_gather(K|V rows)", t)
        self.x.all_gather_into(out.reshape(self.world * t.shape[0], *t.shape[1:]), t)
        return out.reshape(self.world * t.shape[0], *t.shape[1:])

    def kv_buffer(self, rows: int, cols: int, B: int, npix: int, like: torch.Tensor):
        """(buffer for the completed K|V, the view of it this rank's projection GEMM writes into): the GEMM output lands in
        its slot of the gathered tensor, so no copy precedes the exchange."""
        ext = torch.empty((self.world, rows, cols), dtype=like.dtype, device=like.device)
        return ext, ext[self.rank]

    def start_kv(self, ext: torch.Tensor, B: int, npix: int, copy_rows=None):
        """Launch the exchange that completes `ext` WITHOUT waiting for it: RCCL runs it on its own stream behind the K|V
        projection, so the caller's next launches (the query projection) overlap the transfer; finish_kv() joins."""
        loc = ext[self.rank]
        _count("all_gather(K|V rows)", loc)
        work = self.x.all_gather_start(ext.reshape(-1, ext.shape[-1]), loc)   # in place: input = this rank's slot
        return (work, ext)

    def finish_kv(self, handle) -> torch.Tensor:
        work, ext = handle
        self.x.finish(work)
        return ext.reshape(self.world * ext.shape[1], ext.shape[2])

    def complete_kv(self, ext: torch.Tensor, B: int, npix: int, copy_rows=None) -> torch.Tensor:
        return self.finish_kv(self.start_kv(ext, B, npix, copy_rows))

    # ---- frame <-> pixel all-to-all for temporal attention -----------------------------------------------
    def pixel_sharded(self, npix: int) -> bool:
        return self.temporal == "a2a" and self.world > 1 and npix % self.world == 0

    def to_pixel_shards(self, x: torch.Tensor, BF: int, npix: int, copy_blocks) -> torch.Tensor:
        """x: this rank's rows (b, local frame, pixel) [BF*npix, W].  Returns [R*BF*Ns, W], part-major rows
        (source rank, b, local frame, pixel of THIS rank's slice), Ns = npix / R."""
        R, Ns = self.world, npix // self.world
        send = torch.empty((R * BF * Ns, x.shape[1]), dtype=x.dtype, device=x.device)
        copy_blocks(send, x, R, BF, Ns, ys0=BF * Ns, ys1=Ns, xs0=Ns, xs1=npix)      # (bf, j, pl) -> (j, bf, pl)
        recv = torch.empty_like(send)
        _count("all_to_all(temporal in)", send, (R - 1) / R)
        self.x.all_to_all(recv, send)
        return recv

    def to_frame_shards(self, o: torch.Tensor, BF: int, npix: int, copy_blocks) -> torch.Tensor:
        """Inverse of to_pixel_shards for the attention output [R*BF*Ns, C] -> rows (b, local frame, pixel) [BF*npix, C]."""
        R, Ns = self.world, npix // self.world
        recv = torch.empty_like(o)
        _count("all_to_all(temporal out)", o, (R - 1) / R)
        self.x.all_to_all(recv, o.contiguous())
        out = torch.empty((BF * npix, o.shape[1]), dtype=o.dtype, device=o.device)
        copy_blocks(out, recv, R, BF, Ns, ys0=Ns, ys1=npix, xs0=BF * Ns, xs1=Ns)       # (j, bf, pl) -> (bf, j, pl)
        return out

    def item(self, B: int, b: int, g: int) -> int:
        """kv item index of (batch row b, GLOBAL frame g) inside an all-gathered [world][B*f_loc items] tensor."""
        return (g // self.f_loc) * (B * self.f_loc) + b * self.f_loc + g % self.f_loc

    layout = "gather"   # part of the key-segment cache key (segments._gi)

    def gather_kv(self, kv: torch.Tensor, B: int, npix: int, copy_rows=None) -> torch.Tensor:
        return self.all_gather_rows(kv)

    def prev_frame_view(self) -> "PrevFrameHalo":
        """The view attn1 uses: only the previous rank's LAST frame is fetched."""
        return PrevFrameHalo(self)

    def chunk_view(self, chunk: int) -> "ChunkHalo":
        """The view the adapter's sparse-causal attention uses: the first frame of the chunk this rank's range starts in and
        the frame before the range -- at most two remote frames."""
        return ChunkHalo(self, chunk)

    # ---- one-frame halos for the temporal convolutions --------------------------------------------------
    def exchange_halos(self, x_ext: torch.Tensor, B: int, npix: int, copy_rows, defer: bool = False) -> tuple:
        """x_ext rows = [B*f_loc*npix local | B*npix halo of the previous rank | B*npix halo of the next rank].
        Sends this rank's first / last frame to its neighbours and receives theirs.  Returns (halo_prev_row,
        halo_next_row) with -1 where there is no neighbour."""
        rows = B * self.f_loc * npix
        hb = B * npix
        prev_blk, next_blk = x_ext[rows:rows + hb], x_ext[rows + hb:rows + 2 * hb]
        ops_ = []
        first = last = None
        if self.rank > 0:
            first = torch.empty_like(prev_blk)
            for b in range(B):
                copy_rows(first[b * npix:(b + 1) * npix], x_ext[(b * self.f_loc) * npix:(b * self.f_loc + 1) * npix])
            _count("p2p(TemporalConv halo)", first)
            ops_ += [("send", first, self.rank - 1), ("recv", prev_blk, self.rank - 1)]
        if self.rank < self.world - 1:
            last = torch.empty_like(next_blk)
            for b in range(B):
                copy_rows(last[b * npix:(b + 1) * npix], x_ext[(b * self.f_loc + self.f_loc - 1) * npix:(b * self.f_loc + self.f_loc) * npix])
            _count("p2p(TemporalConv halo)", last)
            ops_ += [("send", last, self.rank + 1), ("recv", next_blk, self.rank + 1)]
        _set_kind("p2p(TemporalConv halo)")
        reqs = self.x.p2p_start(ops_)
        hrows = (rows if self.rank > 0 else -1, rows + hb if self.rank < self.world - 1 else -1)
        if defer:      # the caller runs the interior frames (no remote data) while the halos travel, then joins: finish_halos(handle)
            return (reqs, hrows, first, last)
        self.x.finish(reqs)
        return hrows

    def finish_halos(self, handle) -> tuple:
        reqs, hrows, _first, _last = handle      # (the send buffers stay alive until the exchange is joined)
        self.x.finish(reqs)
        return hrows


class PrevFrameHalo:
    """K|V of this rank's frames preceded by a halo block with the previous rank's last frame of every batch row:
    rows = [B halo items | B * f_loc local items] (an item = npix rows).  Spatial attn1 -- plain [prev | cur] and the
    edited [src prev | src cur | own cur] -- reads nothing else ("1-frame halo send/recv to next rank")."""

    layout = "halo"

    def __init__(self, shard: FrameShard):
        self.s = shard
        self.world, self.rank, self.f_loc, self.f_total, self.frame0 = shard.world, shard.rank, shard.f_loc, shard.f_total, shard.frame0

    def item(self, B: int, b: int, g: int) -> int:
        if g == self.frame0 - 1 and self.rank > 0:
            return b
        if not (self.frame0 <= g < self.frame0 + self.f_loc):
            raise IndexError(f"frame {g} is neither local to rank {self.rank} nor its one-frame halo")
        return B + b * self.f_loc + (g - self.frame0)

    def kv_buffer(self, rows: int, cols: int, B: int, npix: int, like: torch.Tensor):
        """[B halo items | local items]: the projection GEMM writes the local part in place (no whole-tensor copy)."""
        ext = torch.empty((B * npix + rows, cols), dtype=like.dtype, device=like.device)
        return ext, ext[B * npix:]

    def start_kv(self, ext: torch.Tensor, B: int, npix: int, copy_rows):
        """Post the one-frame halo send / receive without waiting (the query projection overlaps it); finish_kv() joins."""
        s = self.s
        kv = ext[B * npix:]
        ops_ = []
        keep = None
        if self.rank < self.world - 1:
            last = keep = torch.empty((B * npix, kv.shape[1]), dtype=kv.dtype, device=kv.device)
            for b in range(B):
                copy_rows(last[b * npix:(b + 1) * npix], kv[(b * self.f_loc + self.f_loc - 1) * npix:(b * self.f_loc + self.f_loc) * npix])
            _count("p2p(attn1 K|V halo)", last)
            ops_.append(("send", last, self.rank + 1))
        if self.rank > 0:
            ops_.append(("recv", ext[:B * npix], self.rank - 1))
        else:
            copy_rows(ext[:B * npix], kv[:B * npix])      # never addressed (frame 0 has no predecessor); keep it finite
        _set_kind("p2p(attn1 K|V halo)")
        return (s.x.p2p_start(ops_), ext, keep)

    def finish_kv(self, handle) -> torch.Tensor:
        reqs, ext, _keep = handle
        self.s.x.finish(reqs)
        return ext

    def complete_kv(self, ext: torch.Tensor, B: int, npix: int, copy_rows) -> torch.Tensor:
        return self.finish_kv(self.start_kv(ext, B, npix, copy_rows))

    def gather_kv(self, kv: torch.Tensor, B: int, npix: int, copy_rows) -> torch.Tensor:
        ext, loc = self.kv_buffer(kv.shape[0], kv.shape[1], B, npix, kv)
        copy_rows(loc, kv)
        return self.complete_kv(ext, B, npix, copy_rows)


class ChunkHalo:
    """K|V of this rank's frames preceded by TWO halo blocks: rows = [B items: first frame of the chunk the rank's range starts in |
    B items: the frame before the range | B * f_loc local items].  The adapter's sparse-causal attention
    (controlnet_adapter.py:352-361: keys = [first frame of the 8-frame chunk | previous frame in the chunk]) reads nothing else,
    so at most two remote frames replace the all-gather of every rank's K|V.  The chunk's first frame may live several ranks
    back (f_loc < chunk): its owner sends it point-to-point to every rank whose range starts inside that chunk."""

    layout = "chunkhalo"

    def __init__(self, shard: FrameShard, chunk: int):
        self.s, self.chunk = shard, chunk
        self.world, self.rank, self.f_loc, self.f_total, self.frame0 = shard.world, shard.rank, shard.f_loc, shard.f_total, shard.frame0
        self.first, self.prev = self.needs(self.rank)

    def needs(self, r: int):
        """(global frame of the remote chunk-first frame or None, global frame of the remote previous frame or None) of rank r.
        A range that starts ON a chunk boundary needs neither; one that starts on the chunk's second frame needs the first only
        (its [first | previous] keys coincide, segments.first_prev_chunked)."""
        f0 = r * self.f_loc
        m = f0 % self.chunk
        return (f0 - m if m >= 1 else None, f0 - 1 if m >= 2 else None)

    def item(self, B: int, b: int, g: int) -> int:
        if self.frame0 <= g < self.frame0 + self.f_loc:
            return 2 * B + b * self.f_loc + (g - self.frame0)
        if g == self.first:
            return b
        if g == self.prev:
            return B + b
        raise IndexError(f"frame {g} is neither local to rank {self.rank} nor one of its two halo frames")

    def kv_buffer(self, rows: int, cols: int, B: int, npix: int, like: torch.Tensor):
        ext = torch.empty((2 * B * npix + rows, cols), dtype=like.dtype, device=like.device)
        return ext, ext[2 * B * npix:]

    def start_kv(self, ext: torch.Tensor, B: int, npix: int, copy_rows):
        s, hb, fl = self.s, B * npix, self.f_loc
        kv = ext[2 * hb:]

        def frame_rows(dst, fl_idx):   # the rows of local frame fl_idx of every batch row -> dst [B*npix, cols]
            for b in range(B):
                copy_rows(dst[b * npix:(b + 1) * npix], kv[(b * fl + fl_idx) * npix:(b * fl + fl_idx + 1) * npix])

        ops_, keep = [], []
        for r in range(self.world):          # what this rank owes the others (every rank derives the same schedule)
            if r == self.rank:
                continue
            first, prev = self.needs(r)
            mine = [g for g in (first, prev) if g is not None and g // fl == self.rank]
            if not mine:
                continue
            buf = torch.empty((len(mine) * hb, kv.shape[1]), dtype=kv.dtype, device=kv.device)
            for i, g in enumerate(mine):
                frame_rows(buf[i * hb:(i + 1) * hb], g - self.frame0)
            keep.append(buf)
            _count("p2p(adapter K|V halo)", buf)
            ops_.append(("send", buf, r))
        of = self.first // fl if self.first is not None else None
        op = self.prev // fl if self.prev is not None else None
        if of is not None and of == op:      # both frames from the same rank: one message [first | prev]
            ops_.append(("recv", ext[:2 * hb], of))
        else:
            if of is not None:
                ops_.append(("recv", ext[:hb], of))
            if op is not None:
                ops_.append(("recv", ext[hb:2 * hb], op))
        if of is None:
            copy_rows(ext[:hb], kv[:hb])          # never addressed; keep it finite
        if op is None:
            copy_rows(ext[hb:2 * hb], kv[:hb])
        _set_kind("p2p(adapter K|V halo)")
        return (s.x.p2p_start(ops_), ext, keep)

    def finish_kv(self, handle) -> torch.Tensor:
        reqs, ext, _keep = handle
        self.s.x.finish(reqs)
        return ext

    def complete_kv(self, ext: torch.Tensor, B: int, npix: int, copy_rows) -> torch.Tensor:
        return self.finish_kv(self.start_kv(ext, B, npix, copy_rows))

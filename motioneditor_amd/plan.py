"""Host side of the step-level C entry points (``me_plan_*`` / ``me_denoise_step``, csrc/plan.hip, include/motioned.h).

A denoising step is ~1100 kernel launches on two HIP streams.  ``StepPlan`` records ONE eager step -- every launch the library makes on
this thread, plus the cross-stream dependencies stated through the helpers below -- and afterwards re-issues the whole step from a single
C call: the per-op Python dispatch of the reference's loop body (pipeline_motion_editor.py:603-648) becomes ``me_denoise_step``.

What a recorded step may contain: ``libmotioned`` launches and the event record / wait pairs of this module -- nothing else.  A torch
kernel inside the step would run in the recording pass and silently be missing from every replay, so the few places of the launch graph
that used torch for data movement (``torch.cat`` of the latents, the fp16 cast of the text embeddings, one ``clone``) go through library
copies / casts on the GPU, and `torch_fallback` refuses while a plan records.

Memory: the recording pass allocates from a private ``torch.cuda.MemPool`` that the plan keeps; blocks the pass frees go back to THAT pool,
which nobody allocates from again, so every address a recorded launch holds stays the plan's.  Within the pass blocks are reused as in any
eager step -- same-stream reuse is ordered by the stream, in the replay as in the recording -- and a tensor handed to another stream
(`share`) is kept alive for the life of the plan: the caching allocator would otherwise re-issue its block once the other stream's use had
COMPLETED in the recording pass (a host-side observation), which a replay with different timing could not rely on.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import List, Optional

import torch

from . import capi

ACTIVE: Optional["StepPlan"] = None   # the plan this thread is recording into


def _h(stream: torch.cuda.Stream) -> int:
    return stream.cuda_stream


def record_event(stream: torch.cuda.Stream):
    """stream.record_event(), stated to the recording plan as well."""
    ev = stream.record_event()
    if ACTIVE is not None:
        i = C.c_int32(-1)
        capi.check(capi.lib().me_plan_event_record(_h(stream), C.byref(i)), "me_plan_event_record")
        ACTIVE._event_ids[id(ev)] = i.value
        ACTIVE._keep.append(ev)      # id() stays unique while the event lives
    return ev


def wait_event(stream: torch.cuda.Stream, ev) -> None:
    """stream.wait_event(ev), stated to the recording plan as well (the event must have been recorded through `record_event` inside the same plan)."""
    stream.wait_event(ev)
    if ACTIVE is not None:
        i = ACTIVE._event_ids.get(id(ev))
        if i is None:
            raise RuntimeError("plan: waiting on an event that was not recorded inside the recording step")
        capi.check(capi.lib().me_plan_event_wait(_h(stream), i), "me_plan_event_wait")


def wait_stream(waiter: torch.cuda.Stream, signaller: torch.cuda.Stream) -> None:
    """waiter.wait_stream(signaller)."""
    wait_event(waiter, record_event(signaller))


def share(t: torch.Tensor, stream: torch.cuda.Stream) -> None:
    """t.record_stream(stream); a recording plan additionally keeps `t` allocated for its whole life (module docstring)."""
    t.record_stream(stream)
    if ACTIVE is not None:
        ACTIVE._keep.append(t)


def torch_fallback(what: str) -> None:
    """Called where the launch graph is about to use a torch kernel for data movement: fine in an eager step, fatal for a step being recorded."""
    if ACTIVE is not None:
        raise RuntimeError(f"plan: {what} would run a torch kernel inside a recorded step (it would be missing from every replay)")


class StepPlan:
    """One recorded denoising step.  Usage (pipelines.MotionEditorPipeline.denoise_step_planned):

        pl = StepPlan()
        with pl.recording():
            out = step(lat, emb)            # eager, with ops.STEP_PARAMS = params
        pl.bind(lat, emb, params, out)
        new = pl.step(latents, emb_now, t, guidance, ca, cb)
    """

    def __init__(self) -> None:
        self._handle = C.c_void_p()
        self._pool = None
        self._keep: List[object] = []
        self._event_ids = {}
        self._bound = None
        self.main_stream = None

    @contextlib.contextmanager
    def recording(self):
        global ACTIVE
        if ACTIVE is not None:
            raise RuntimeError("plan: a step is already being recorded")
        L = capi.lib()
        self.main_stream = torch.cuda.current_stream()
        self._pool = torch.cuda.MemPool()
        capi.check(L.me_plan_begin(C.byref(self._handle), _h(self.main_stream)), "me_plan_begin")
        ACTIVE = self
        ok = False
        try:
            with torch.cuda.use_mem_pool(self._pool):
                yield self
            ok = True
        finally:
            ACTIVE = None
            rc = L.me_plan_end(self._handle)
            if ok:
                capi.check(rc, "me_plan_end")
            else:   # the step raised: drop the half-recorded plan, let the step's exception propagate
                L.me_plan_destroy(self._handle)
                self._handle = C.c_void_p()

    def bind(self, lat_in: torch.Tensor, text: Optional[torch.Tensor], params: torch.Tensor, lat_out: torch.Tensor) -> None:
        if lat_in.dtype != torch.float32 or lat_out.dtype != torch.float32 or not lat_in.is_contiguous() or not lat_out.is_contiguous() or lat_in.shape != lat_out.shape:
            raise ValueError("plan.bind: latents in / out must be contiguous fp32 tensors of one shape")
        if params.dtype != torch.float32 or params.numel() != 4 or not params.is_cuda:
            raise ValueError("plan.bind: params must be a CUDA fp32 [4] tensor {t, guidance, ca, cb}")
        if text is not None and not text.is_contiguous():
            raise ValueError("plan.bind: text embeddings must be contiguous")
        nb_text = 0 if text is None else text.numel() * text.element_size()
        capi.check(capi.lib().me_plan_bind(self._handle, lat_in.data_ptr(), lat_in.numel() * 4, None if text is None else text.data_ptr(), nb_text,
                                           params.data_ptr(), lat_out.data_ptr()), "me_plan_bind")
        self._bound = (lat_in, text, params, lat_out)

    def step(self, latents: torch.Tensor, text: Optional[torch.Tensor], t: float, guidance: float, ca: float, cb: float) -> torch.Tensor:
        """One denoising step on torch's current stream; returns the updated latents (a fresh tensor)."""
        if self._bound is None:
            raise RuntimeError("plan.step: bind() first")
        lat_in, btext, _, lat_out = self._bound
        if latents.dtype != torch.float32 or not latents.is_contiguous() or latents.shape != lat_in.shape or latents.device != lat_in.device:
            raise ValueError("plan.step: latents must be contiguous fp32 of the recorded shape, on the recorded device")
        if text is not None:
            if btext is None or text.dtype != btext.dtype or text.shape != btext.shape or not text.is_contiguous() or text.device != btext.device:
                raise ValueError("plan.step: text embeddings must match the recorded ones in dtype, shape and device (contiguous)")
        out = torch.empty_like(lat_in)
        cur = torch.cuda.current_stream()
        capi.check(capi.lib().me_denoise_step(self._handle, latents.data_ptr(), out.data_ptr(), None if text is None else text.data_ptr(),
                                              float(t), float(guidance), float(ca), float(cb), _h(cur)), "me_denoise_step")
        return out

    def stats(self) -> dict:
        st = capi.PlanStats()
        capi.check(capi.lib().me_plan_info(self._handle, C.byref(st)), "me_plan_info")
        return {k: getattr(st, k) for k, _ in capi.PlanStats._fields_}

    def nodes(self):
        """(kind, stream index, event id, grid, block) of every node, in order (tests)."""
        n = self.stats()
        info = capi.PlanNodeInfo()
        out = []
        for i in range(n["launches"] + n["event_records"] + n["event_waits"]):
            capi.check(capi.lib().me_plan_node(self._handle, i, C.byref(info), None, 0), "me_plan_node")
            out.append((info.kind, info.stream, info.event, tuple(info.grid), tuple(info.block)))
        return out

    def close(self) -> None:
        if self._handle:
            capi.lib().me_plan_destroy(self._handle)
            self._handle = C.c_void_p()
        self._bound = None
        self._keep.clear()
        self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

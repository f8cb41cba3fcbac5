"""Key-segment tables for the fused attention kernel (``me_attn``): for every query item (one frame
of one batch row) the kv items it attends and how.  Tables are tiny int32 device tensors, built once
per (pattern, shape) and cached -- the editors' gating is a pure function of (step, layer), so the
same tables are replayed every step.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .capi import SEG_DUAL_BIN, SEG_DUAL_CUR, SEG_DUAL_PREV, SEG_PLAIN

_cache: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
GENERAL_DUAL: Dict[int, bool] = {}  # seg_mode.data_ptr() -> table uses the general (mask-reading) dual modes
BINARY_DUAL: Dict[int, bool] = {}   # seg_mode.data_ptr() -> table has DUAL_BIN segments (me_attn then needs the vsum scratch)
SEG_COUNT: Dict[int, int] = {}  # seg_item.data_ptr() -> number of valid (item, segment) entries = key segments actually multiplied
KEY_UNITS: Dict[int, int] = {}  # seg_item.data_ptr() -> sum over items of (1 per plain, 2 per dual segment); FLOP accounting only


ITEM_ORDER: Dict[int, torch.Tensor] = {}    # seg_item.data_ptr() -> int32 permutation of the query items (me_attn_args.item_order), for tables that want one


def _mk(key, rows_item, rows_mode, device, ref_units=None, order=None):
    """ref_units: reference-semantics key units of the table when it differs from what the rows spell out (collapsed duplicates).
    order: the processing order of the query items inside a head's run of the attention kernel's heads-slowest block order (scheduling only)."""
    hit = _cache.get((key, str(device)))
    if hit is None:
        hit = (torch.tensor(rows_item, dtype=torch.int32, device=device), torch.tensor(rows_mode, dtype=torch.int32, device=device))
        _cache[(key, str(device))] = hit
        if order is not None:
            assert sorted(order) == list(range(len(rows_item)))
            ITEM_ORDER[hit[0].data_ptr()] = torch.tensor(order, dtype=torch.int32, device=device)
        GENERAL_DUAL[hit[1].data_ptr()] = any(m in (SEG_DUAL_CUR, SEG_DUAL_PREV) for rm in rows_mode for m in rm)
        BINARY_DUAL[hit[1].data_ptr()] = any(m == SEG_DUAL_BIN for rm in rows_mode for m in rm)
        SEG_COUNT[hit[0].data_ptr()] = sum(1 for ri in rows_item for i_ in ri if i_ >= 0)
        KEY_UNITS[hit[0].data_ptr()] = ref_units if ref_units is not None else \
            sum((2 if m != SEG_PLAIN else 1) for ri, rm in zip(rows_item, rows_mode) for i_, m in zip(ri, rm) if i_ >= 0)
    return hit


def self_items(n_items: int, device):
    """every item attends itself (ControlNet attn1, adapter attn_pose)."""
    return _mk(("self", n_items), [[i] for i in range(n_items)], [[SEG_PLAIN]] * n_items, device)


def cross_text(B: int, f: int, device):
    """item (b, fr) attends text row b -- K/V projected once per b, not per frame (attention_2d.py:343)."""
    return _mk(("cross", B, f), [[i // f] for i in range(B * f)], [[SEG_PLAIN]] * (B * f), device)


def cross_interleaved(n_items: int, n_text: int, device, row_offset: int = 0):
    """ControlNet prompt quirk: embeds.repeat(f,1,1) on "(b f)" rows -> row r reads text r % 2
    (pipeline_motion_editor.py:615,621).  row_offset = index of this tensor's first row in the full batch."""
    return _mk(("crossil", n_items, n_text, row_offset), [[(i + row_offset) % n_text] for i in range(n_items)], [[SEG_PLAIN]] * n_items, device)


def _gi(shard, B, f):
    """(b, GLOBAL frame) -> kv item index: plain (b*f + g) unsharded; when sharded, whatever layout the shard view gives the
    gathered K|V (parallel.FrameShard: part-major all-gather; parallel.PrevFrameHalo: [halo | local])."""
    if shard is None:
        return (lambda b, g: b * f + g), 0, ()
    return (lambda b, g: shard.item(B, b, g)), shard.frame0, (shard.world, shard.rank, getattr(shard, "layout", "gather"))


def prev_cur(B: int, f: int, device, shard=None):
    """MotionFrameAttention: keys = [frame max(i-1,0) | frame i] (attention_2d.py:732-740).  f = local frames."""
    gi, f0, key = _gi(shard, B, f)
    # global frame 0 attends [frame 0 | frame 0]: a softmax over every key twice equals the softmax over every key once (each
    # weight doubles in numerator and denominator), so that item gets ONE segment -- same result, half its work
    rows = [([gi(b, f0 + i), -1] if f0 + i == 0 else [gi(b, f0 + i - 1), gi(b, f0 + i)]) for b in range(B) for i in range(f)]
    return _mk(("prevcur", B, f) + key, rows, [[SEG_PLAIN, SEG_PLAIN]] * (B * f), device, ref_units=2 * B * f)


def first_prev_chunked(B: int, f: int, chunk: int, device, shard=None):
    """Adapter sparse-causal attention on independent chunks of `chunk` frames:
    keys = [first frame of chunk | previous frame in chunk] (controlnet_adapter.py:352-361,414,472)."""
    gi, f0, key = _gi(shard, B, f)
    rows = []
    for b in range(B):
        for i in range(f):
            g = f0 + i
            c0 = g - g % chunk
            # the first two frames of a chunk attend [first | first]: duplicated keys, one segment gives the same softmax
            rows.append([gi(b, c0), -1] if g % chunk <= 1 else [gi(b, c0), gi(b, c0 + g % chunk - 1)])
    return _mk(("firstprev", B, f, chunk) + key, rows, [[SEG_PLAIN, SEG_PLAIN]] * (B * f), device, ref_units=2 * B * f)


def edited_spatial(f: int, device, binary_mask: bool = False, B: int = 4, shard=None):
    """FullySelfAttentionControlMask on batch 4 = [u.rec, u.edit, c.rec, c.edit] (fully_control.py:425-447;
    B = 2 is one (rec, edit) pair, i.e. one classifier-free-guidance half on a CFG-parallel rank):
    recon rows keep [prev | cur]; edit rows attend [src prev (fg/bg dual, mask frame max(head-1,0)) |
    src cur (dual, mask frame head) | own cur]; the edit branch's prev-frame K/V are dropped
    (k[:, 3N:], fully_control.py:383).  With a binary mask (the reference's man.mask PNGs are 0/255) the
    fg/bg pair of every source key weighs exp(s) + exp(0) whichever way the bit points, so the kernel's
    DUAL_BIN mode needs no mask read."""
    gi, f0, key = _gi(shard, B, f)
    rows, modes = [], []
    for b in range(B):
        for i in range(f):
            g = f0 + i
            if b % 2 == 0:
                rows.append([gi(b, max(g - 1, 0)), gi(b, g), -1])
                modes.append([SEG_PLAIN, SEG_PLAIN, SEG_PLAIN])
            else:
                rows.append([gi(b - 1, max(g - 1, 0)), gi(b - 1, g), gi(b, g)])
                modes.append([SEG_DUAL_BIN, SEG_DUAL_BIN, SEG_PLAIN] if binary_mask else [SEG_DUAL_PREV, SEG_DUAL_CUR, SEG_PLAIN])
    # processing order: (recon frame i, edit frame i, recon frame i + 1, ...) per (recon, edit) pair -- an edit item reads its source's K | V right after
    # the reconstruction item did, while they are still in the XCD's L2 (ascending order puts f items between the two)
    order = [b * f + i + r * f for b in range(0, B - 1, 2) for i in range(f) for r in (0, 1)] + ([(B - 1) * f + i for i in range(f)] if B % 2 else [])
    return _mk(("edited", f, binary_mask, B) + key, rows, modes, device, order=order)


def has_dual(seg_mode: torch.Tensor) -> bool:
    """True when the table holds any dual (masked / binary) segment.  Tables built here answer from the cache; a foreign table is inspected."""
    gd, bd = GENERAL_DUAL.get(seg_mode.data_ptr()), BINARY_DUAL.get(seg_mode.data_ptr())
    if gd is None or bd is None:
        return bool((seg_mode != SEG_PLAIN).any().item())
    return gd or bd


_inverse: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}


def inverse(seg_item: torch.Tensor, n_kv_items: int):
    """CSR inverse of a segment table for the key-centric attention backward (me_attn_bwd): (inv_ptr int32 [n_kv_items + 1], inv_item int32)
    with inv_item[inv_ptr[k] : inv_ptr[k + 1]] = the query items that list kv item k, once per listing, in ascending order."""
    key = (seg_item.data_ptr(), tuple(seg_item.shape), n_kv_items, str(seg_item.device))
    hit = _inverse.get(key)
    if hit is None:
        lists = [[] for _ in range(n_kv_items)]
        for it, row in enumerate(seg_item.tolist()):
            for kit in row:
                if kit < 0:
                    break
                lists[kit].append(it)
        ptr = [0]
        for l_ in lists:
            ptr.append(ptr[-1] + len(l_))
        flat = [i for l_ in lists for i in l_] or [0]
        hit = (torch.tensor(ptr, dtype=torch.int32, device=seg_item.device), torch.tensor(flat, dtype=torch.int32, device=seg_item.device), seg_item)   # seg_item kept alive: the key is its address
        _inverse[key] = hit
    return hit[0], hit[1]

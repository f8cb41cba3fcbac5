"""Exact-math stand-in for xformers.ops.memory_efficient_attention on (B*H, M, dh) tensors."""
import torch


def memory_efficient_attention(q, k, v, attn_bias=None, p=0.0, scale=None, op=None):
    s = (q.shape[-1] ** -0.5) if scale is None else scale
    scores = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype), q, k.transpose(-1, -2), beta=0, alpha=s)
    if attn_bias is not None:
        scores = scores + attn_bias
    return torch.bmm(scores.softmax(dim=-1), v)

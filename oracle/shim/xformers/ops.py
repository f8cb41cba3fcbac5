"""Exact-math stand-in for xformers.ops.memory_efficient_attention on (B*H, M, dh) tensors.

Test infrastructure (container-only, see oracle/shim/README.md).  The (batch x head) axis is processed in chunks so that the materialised score
matrix stays under ~1 GB: every (batch, head) row is an independent softmax(q k^T) v, so chunking that axis changes no arithmetic -- it only lets
the reference UNet run at production token counts (8 frames x 64x64 latents: 128 x 4096 x 8192 fp32 scores = 17 GB in one piece) inside
oracle/make_golden.py --only-single."""
import torch

_MAX_SCORE_BYTES = 1 << 30


def memory_efficient_attention(q, k, v, attn_bias=None, p=0.0, scale=None, op=None):
    s = (q.shape[-1] ** -0.5) if scale is None else scale
    per = q.shape[1] * k.shape[1] * q.element_size()
    step = max(1, min(q.shape[0], _MAX_SCORE_BYTES // max(per, 1)))
    outs = []
    for i in range(0, q.shape[0], step):
        qi, ki, vi = q[i:i + step], k[i:i + step], v[i:i + step]
        scores = torch.baddbmm(torch.empty(qi.shape[0], qi.shape[1], ki.shape[1], dtype=q.dtype), qi, ki.transpose(-1, -2), beta=0, alpha=s)
        if attn_bias is not None:
            scores = scores + (attn_bias[i:i + step] if attn_bias.dim() == 3 and attn_bias.shape[0] == q.shape[0] else attn_bias)
        outs.append(torch.bmm(scores.softmax(dim=-1), vi))
    return outs[0] if len(outs) == 1 else torch.cat(outs)

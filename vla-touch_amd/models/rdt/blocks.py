"""Init-time helpers of the RDT model — mirror of the position-embedding functions of the reference's
VLA/models/rdt/blocks.py:209-306 (`get_1d_sincos_pos_embed_from_grid`, `get_nd_sincos_pos_embed_from_grid`,
`get_multimodal_cond_pos_embed`).  They only run when a model is constructed without a checkpoint (the
learned tables in a checkpoint replace them), in numpy float64 like the reference.

The compute classes of that file (TimestepEmbedder, CrossAttention, RDTBlock, FinalLayer) have no Python
counterpart here: their arithmetic is the HIP RDT driver (csrc/vt_rdt.hip) behind vt_rdt_forward / vt_rdt_sample.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """positions (M,) -> (M, embed_dim) = [sin(p w_k) | cos(p w_k)], w_k = 10000^(-2k/embed_dim)."""
    assert embed_dim % 2 == 0
    half = embed_dim // 2
    freq = 1.0 / (10000.0 ** (np.arange(half, dtype=np.float64) / half))
    p = np.asarray(pos, dtype=np.float64).reshape(-1)
    ang = p[:, None] * freq[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_nd_sincos_pos_embed_from_grid(embed_dim, grid_sizes):
    """Grid of sizes (g0, .., gK-1) -> (g0, .., gK-1, embed_dim): every axis with size > 1 gets an equal, even share of
    the channels (axes of size <= 1 contribute nothing)."""
    grid_sizes = tuple(grid_sizes)
    live = [i for i, g in enumerate(grid_sizes) if g > 1]
    out = np.zeros(grid_sizes + (embed_dim,))
    share = embed_dim // len(live)
    share -= share % 2
    for slot, axis in enumerate(live):
        e = get_1d_sincos_pos_embed_from_grid(share, np.arange(grid_sizes[axis]))
        shape = [1] * len(grid_sizes) + [share]
        shape[axis] = -1
        out[..., slot * share:(slot + 1) * share] += e.reshape(shape)
    return out


def get_multimodal_cond_pos_embed(embed_dim, mm_cond_lens: OrderedDict, embed_modality=True):
    """Concatenated position table for an ordered set of modalities.  With `embed_modality` the first half of the
    channels carries a per-modality sin-cos code and the second half the within-modality position; otherwise all
    channels carry the position.  A negative length (or negative grid entry of an "image" tuple) means "tokens exist but
    get no positional code along that axis"."""
    names = list(mm_cond_lens.keys())
    mod = np.zeros((len(names), embed_dim))
    if embed_modality:
        mod[:, :embed_dim // 2] = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, np.arange(len(names)))
        pdim = embed_dim // 2
    else:
        pdim = embed_dim
    parts = []
    for idx, name in enumerate(names):
        ln = mm_cond_lens[name]
        if name == "image" and isinstance(ln, (tuple, list)):
            full = tuple(abs(x) for x in ln)
            grid = tuple(x if x > 0 else 1 for x in ln)
            tab = np.zeros(full + (embed_dim,))
            tab[..., -pdim:] += get_nd_sincos_pos_embed_from_grid(pdim, grid)
            tab = tab.reshape(-1, embed_dim)
        else:
            tab = np.zeros((abs(ln), embed_dim))
            tab[:, -pdim:] += get_1d_sincos_pos_embed_from_grid(pdim, np.arange(ln if ln > 0 else 1))
        parts.append(tab + mod[idx])
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, embed_dim))

"""Multi-GPU plumbing: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The refinement path shards by episode: every rank refines its own batch with replicated frozen weights, so the only
communication is the ONE-TIME broadcast of the packed weight blobs from rank 0 (few, large messages; ~0.3 GB for
DINOv2-B + the two U-Nets) and an optional gather of the tiny results.  There is no collective in the step loop.
Works with the gloo backend on CPU tensors too (used by the world_size-2 CPU tests).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous split of `n_items` episodes over `world` ranks (first n % world ranks get one extra)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_tensors(tensors: Iterable[torch.Tensor], src: int = 0, bucket_bytes: int = 256 << 20) -> int:
    """In-place broadcast of a list of tensors from `src`; small tensors are coalesced into flat buckets so the
    transfer is a few large messages (xGMI links are per-peer, large messages amortise the ring latency)."""
    import torch.distributed as dist
    total = 0
    by_key = {}
    host_staged = dist.get_backend() == "gloo"     # gloo moves device tensors through host memory: do it explicitly, once per bucket

    def bcast(t: torch.Tensor):
        if host_staged and t.is_cuda:
            h = t.cpu()
            dist.broadcast(h, src)
            t.copy_(h)
        else:
            dist.broadcast(t, src)
    for t in tensors:
        if t is None:
            continue
        by_key.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in by_key.items():
        bucket: List[torch.Tensor] = []
        size = 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            if len(bucket) == 1 and bucket[0].is_contiguous():
                bcast(bucket[0])
            else:
                flat = torch.cat([b.reshape(-1) for b in bucket])
                bcast(flat)
                off = 0
                for b in bucket:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
            bucket, size = [], 0

        for t in ts:
            nb = t.numel() * t.element_size()
            total += nb
            if size + nb > bucket_bytes:
                flush()
            bucket.append(t)
            size += nb
        flush()
    return total


def controller_weight_tensors(ctrl) -> List[torch.Tensor]:
    """Every packed device tensor the controller's engines read (DINOv2, observation MLP, sampler U-Nets)."""
    out: List[torch.Tensor] = []
    enc = ctrl.image_encoder.engine
    out += [w for w in enc._weights if w is not None]
    dev = torch.device(ctrl.device)
    mlp = ctrl.state_encoder.engine(dev)
    out += list(mlp.W) + list(mlp.b)
    net = ctrl.diffusion_model
    nets = ("v_net", "s_net") if net.sde_type == "vs" else ("b_net", "s_net")
    eng = net._sampler_engine(nets, dev)
    out += [w for w in eng._weights if w is not None]
    return out


def broadcast_controller_weights(ctrl, src: int = 0) -> int:
    """Rank `src`'s packed weights overwrite everyone else's (in place: engine handles keep their pointers)."""
    ts = controller_weight_tensors(ctrl)
    # cached position-embedding tables are derived from weights: broadcast them too once they exist
    ts += list(ctrl.image_encoder.engine._pos_cache.values())
    n = broadcast_tensors(ts, src)
    # derived copies follow the received weights: the fused sampler path's pre-split convolution weights are re-packed locally
    # (device-side, no extra traffic on the links)
    net = ctrl.diffusion_model
    nets = ("v_net", "s_net") if net.sde_type == "vs" else ("b_net", "s_net")
    net._sampler_engine(nets, torch.device(ctrl.device)).repack()
    rp = getattr(ctrl.image_encoder.engine, "repack", None)
    if rp is not None:
        rp()                                    # the DINOv2 engine's fragment-packed fc1 copy
    for k, v in ctrl.stats.items():
        broadcast_tensors([v], src)
    return n


def gather_results(local: torch.Tensor, dst: int = 0):
    """Optional: collect the tiny [B_local, T, 10] outputs on `dst` for verification (not in the step loop)."""
    import torch.distributed as dist
    world = dist.get_world_size()
    outs = [torch.empty_like(local) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(local, outs, dst=dst)
    return outs

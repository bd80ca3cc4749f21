"""Worker of tests/test_gpu_multiproc.py: one of two processes sharing cuda:0 (the GPU box has ONE MI355X), torch.distributed
over gloo.  Rank 1 starts from DIFFERENT weights; after the one-time broadcast both ranks must produce bit-identical refined
chunks for rank 0's shard, and the step loop must not touch a collective (SURVEY §8e / DESIGN §7)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vla-touch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    torch.set_grad_enabled(False)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)

    from tests import cases
    from tests.test_gpu_rdt import make_runner
    from vlatouch.dist import broadcast_controller_weights, broadcast_tensors, gather_results, shard_range
    from residual_controller.bridge_controller import DiffusionController

    # ---- every rank builds its engines; rank 1 then corrupts its packed weights (a rank that never received the broadcast would
    #      refine with these)
    ctrl = cases.build_controller(DiffusionController, precision="bf16", device=dev)
    B_total, T = 4, 16
    inp = {k: v.to(dev) for k, v in cases.predict_inputs(B_total, T, 224).items()}
    z = inp.pop("z")
    lo, hi = shard_range(B_total, 0, world)                   # rank 0's shard, refined by BOTH ranks for the comparison
    sl = slice(lo, hi)
    args = lambda: (inp["state"][sl], inp["vla"][sl], inp["cam1"][sl], inp["cam2"][sl], inp["forces"][sl])
    before = ctrl.predict(*args(), noise=z[:, sl].contiguous())        # builds every engine (weights packed on device)
    r = make_runner(cases.RDT_TINY, torch.bfloat16)
    ri = {k: v.to(dev) for k, v in cases.rdt_inputs(cases.RDT_TINY, 2, 12, dtype=torch.bfloat16).items()}
    rdt_args = (ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"])
    chunk_before = r.predict_action(*rdt_args, x_init=ri["x_init"])
    if rank == 1:
        from vlatouch.dist import controller_weight_tensors
        for w in controller_weight_tensors(ctrl) + list(r.engine()._weights):
            w.mul_(0.5)
        ctrl.image_encoder.engine.repack()     # derived copies (the DINOv2 engine's fragment-packed fc1) follow the corrupted weights: the broadcast must rebuild them
        torch.cuda.synchronize()
        wrong = ctrl.predict(*args(), noise=z[:, sl].contiguous())
        assert not torch.equal(wrong, before), "corrupting the weights must change the result"

    # ---- the one-time weight broadcast (bench.py's start-up path)
    n1 = broadcast_controller_weights(ctrl, src=0)
    n2 = broadcast_tensors(r.engine()._weights, src=0)
    r.engine().repack()
    assert n1 > 0 and n2 > 0

    # ---- the step loop: no collective may be called
    names = ["broadcast", "all_reduce", "all_gather", "gather", "scatter", "reduce", "barrier", "all_to_all", "send", "recv", "reduce_scatter"]
    saved = {n: getattr(dist, n) for n in names}

    def boom(*a, **k):
        raise AssertionError("collective called inside the step loop")
    for n in names:
        setattr(dist, n, boom)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        zz = z[:, sl].contiguous()
        holder = {}
        with torch.cuda.stream(s):
            holder["out"] = ctrl.predict(*args(), noise=zz)
            s.synchronize()
            with torch.cuda.graph(g, stream=s):
                holder["out"] = ctrl.predict(*args(), noise=zz)
            g.replay()
            s.synchronize()
        after = holder["out"].clone()
        chunk_after = r.predict_action(*rdt_args, x_init=ri["x_init"])
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(dist, n, saved[n])

    outs = gather_results(after.cpu(), dst=0)
    chunks = gather_results(chunk_after.float().cpu(), dst=0)
    if rank == 0:
        assert torch.equal(after, before) and torch.equal(chunk_after, chunk_before), "rank 0's own results must not change"
        assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
        assert torch.equal(chunks[0], chunks[1]), float((chunks[0] - chunks[1]).abs().max())
        print(f"MP_OK broadcast {n1 + n2} bytes; ranks bit-equal on pi_I {tuple(outs[0].shape)} and RDT {tuple(chunks[0].shape)}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Worker of tests/test_gpu_multiproc.py::test_rccl_path_runs_on_one_gpu: ONE process, torch.distributed over the "nccl" backend (= RCCL on ROCm)
with world_size 1 on cuda:0 — the production start-up path of bench.py (communicator bound to the device, bucketed weight broadcast, repack of
the derived copies, barrier, max-reduce of the timing) executed for real before the driver's 8-GPU run does it.  RCCL refuses two ranks on
one device, so world_size 1 is what a one-GPU box can run; the two-rank logic is covered over gloo by tests/_mp_gpu_worker.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vla-touch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist


def main():
    port = sys.argv[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)            # bench.py's call, verbatim
    assert dist.get_backend() == "nccl"

    from tests import cases
    from tests.test_gpu_rdt import make_runner
    from vlatouch.dist import broadcast_controller_weights, broadcast_tensors, controller_weight_tensors, gather_results
    from residual_controller.bridge_controller import DiffusionController

    ctrl = cases.build_controller(DiffusionController, precision="bf16", device=dev)
    inp = {k: v.to(dev) for k, v in cases.predict_inputs(3, 16, 224).items()}
    z = inp.pop("z")
    args = (inp["state"], inp["vla"], inp["cam1"], inp["cam2"], inp["forces"])
    before = ctrl.predict(*args, noise=z)
    r = make_runner(cases.RDT_TINY, torch.bfloat16)
    ri = {k: v.to(dev) for k, v in cases.rdt_inputs(cases.RDT_TINY, 2, 12, dtype=torch.bfloat16).items()}
    rdt_args = (ri["lang_tokens"], ri["lang_mask"], ri["img_tokens"], ri["state_tokens"], ri["action_mask"], ri["freq"])
    chunk_before = r.predict_action(*rdt_args, x_init=ri["x_init"])
    sums_before = [float(w.double().abs().sum()) for w in controller_weight_tensors(ctrl) + list(r.engine()._weights)]

    # ---- the one-time weight broadcast over RCCL (device buffers handed to ncclBroadcast, flat buckets for the small tensors)
    n1 = broadcast_controller_weights(ctrl, src=0)
    n2 = broadcast_tensors(r.engine()._weights, src=0)
    r.engine().repack()
    torch.cuda.synchronize()
    assert n1 > 0 and n2 > 0
    sums_after = [float(w.double().abs().sum()) for w in controller_weight_tensors(ctrl) + list(r.engine()._weights)]
    assert sums_before == sums_after, "a broadcast from rank 0 to itself must leave every weight as it was"

    # ---- the collectives bench.py's timing bracket uses
    dist.barrier()
    tt = torch.tensor([1.25, 0.5], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert tt.tolist() == [1.25, 0.5]

    after = ctrl.predict(*args, noise=z)
    chunk_after = r.predict_action(*rdt_args, x_init=ri["x_init"])
    torch.cuda.synchronize()
    assert torch.equal(after, before), float((after - before).abs().max())
    assert torch.equal(chunk_after, chunk_before)
    outs = gather_results(after, dst=0)                                                # device tensors through RCCL's gather
    assert len(outs) == 1 and torch.equal(outs[0], after)
    print(f"NCCL_OK backend {dist.get_backend()} broadcast {n1 + n2} bytes; results bit-equal before / after")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Multi-GPU readiness on a ONE-GPU box: two processes on cuda:0, gloo backend, real engines.
See tests/_mp_gpu_worker.py for what each rank does and asserts."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_broadcast_then_bit_equal_and_no_step_collectives():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, "-m", "tests._mp_gpu_worker", str(r), "2", port], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
    assert "MP_OK" in outs[0], outs[0][-2000:]


def test_bench_self_launches_two_ranks_and_prints_one_json_line():
    """`python bench.py --gpus 2` (what the driver's SCALE run calls) must start its own ranks under torch.distributed.run and print ONE
    rank-0 JSON line with n_gpus = 2 and the one-time weight broadcast.  The test box has one MI355X: VLATOUCH_BENCH_SHARE_GPU=1 puts both
    ranks on cuda:0 over gloo (the production path is one GPU per rank over RCCL); everything else is the bench's own multi-rank code."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VLATOUCH_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["config"]["global_batch"] == 8 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["config"]["weight_broadcast"]["bytes"] > 2e9          # RDT-1B + DINOv2-B + U-Nets
    assert "cpu_baseline" not in r                                                     # N = 1 only
    assert r["roofline"]["frac"] > 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_rccl_path_runs_on_one_gpu():
    """The "nccl" (= RCCL) backend itself, world_size 1 on cuda:0: init_process_group(device_id=...), broadcast_controller_weights +
    broadcast_tensors(engine weights) + repack, barrier, all_reduce(MAX), gather — librccl is loaded and its kernels run once on this box
    before the driver's multi-GPU bench is the first to do so (see tests/_nccl_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "tests._nccl_worker", _free_port()], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "NCCL_OK backend nccl" in p.stdout, p.stdout[-2000:]


def test_bench_under_torch_distributed_run_with_one_gpu_uses_rccl():
    """The driver's launch line with N = 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ... bench.py --gpus 1`.
    WORLD_SIZE is set, so bench.py builds the RCCL process group, broadcasts the frozen weights (to itself), brackets the timed region with
    dist.barrier() and max-reduces the elapsed time — every line of the multi-rank code path except having a second rank."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VLATOUCH_BENCH_SHARE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", _free_port(),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    wb = r["config"]["weight_broadcast"]
    assert r["n_gpus"] == 1 and wb["backend"] == "nccl" and wb["bytes"] > 2e9 and r["value"] > 0
    assert r["latency_mode"]["chunks_per_s"] > 0 and r["config"]["batches_in_flight"] == 3

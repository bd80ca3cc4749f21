#!/usr/bin/env python
"""bench.py — refined action chunks / s on MI355X for the VLA-Touch action-refinement path.

    python bench.py [--gpus N --steps K --warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one `DiffusionController.predict` over a batch of `--batch` (32) episodes per GPU: 2 camera batches of
synthetic 224x224 frames -> DINOv2 CLS x2 -> state/force MLP -> 10-step velocity-score SDE over v_net/s_net ->
denormalised refined chunks [B, T, 10].  Inputs are resident in HBM before the timed region; the sampler's Gaussian
noise is drawn on device inside the step (as the reference does).  Episodes are independent: ranks take disjoint
batches (weak scaling), frozen weights are broadcast once from rank 0 over RCCL, no collective in the step loop.

Prints ONE JSON line (rank 0) with the BASELINE.json metric plus `roofline` (dominant kernel class = the 256-square ping-pong MFMA GEMM tile,
gemm_pt_kernel / gemm_pp256d_kernel, timed live with HIP events on its launch stream; `roofline_other` carries the same
figures for the 128-column GEMM kernel) and `cpu_baseline` (the oracle on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# the CPU-baseline leg sweeps torch's intra-op thread count: OpenMP workers must SLEEP between parallel regions (the default spin-wait makes large teams
# stall each other); read by the OpenMP runtime when torch loads it, so set before `import torch`.  No effect on the GPU path.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("KMP_BLOCKTIME", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "vla-touch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="episodes per GPU per step")
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--dino", default="base", choices=["small", "base"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="full", choices=["full", "pi_refine", "dino_mlp", "rdt", "siglip", "robot", "lstm", "marker", "train", "train_lstm"],
                    help="full: BASELINE configs[3] = one RDT-1B chunk (5-step DPM-Solver++) + DINOv2 x2 + MLP + interpolant sampler per "
                         "refined chunk; pi_refine: the same without the RDT chunk generator; dino_mlp: configs[1]; rdt: configs[2]")
    ap.add_argument("--rdt-steps", type=int, default=5, help="RDT denoising steps (upstream RDT-1B config: 5)")
    ap.add_argument("--rms-mode", default="meansq", choices=["meansq", "var"],
                    help="timm RmsNorm arithmetic of RDT: meansq = timm >= 1.0.9; var = timm <= 1.0.8 incl. the timm==1.0.3 upstream RDT-1B pins (what its "
                         "released checkpoints need): no score bound exists there, so the cached cross-attention runs its ONLINE softmax")
    ap.add_argument("--rdt-compute", default="auto", choices=["auto", "f16", "bf16"],
                    help="16-bit activation / MFMA operand type of the bf16 RDT-1B: auto (the product default) = IEEE fp16 under the engine's range guard, falling "
                         "back to bf16 when a value leaves the fp16 range (the record's `range_guard` says what happened); f16 / bf16 pin the type (f16: the bf16 "
                         "weights convert exactly, same width and MFMA rate, |chunk - fp32 oracle| 8x smaller; bf16 = the reference's own execution dtype)")
    ap.add_argument("--lang-len", type=int, default=32)
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a captured hipGraph")
    ap.add_argument("--streams", type=int, default=0,
                    help="batches in flight per GPU: consecutive steps (independent batches of --batch episodes) are enqueued round-robin on this "
                         "many HIP streams, each with its own workspaces and hipGraph, so the launch-bound phases of one batch (pi_I U-Nets, the "
                         "small per-denoise-step RDT launches) run beside the MFMA-bound phases of another.  1 = one batch at a time (latency "
                         "mode).  0 = the default of the workload (full / rdt / pi_refine / lstm: 3, robot: 2, else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", dest="no_overlap", action="store_false", default=True,
                    help="run the observation encoding on a second HIP stream beside the RDT chunk generation (measured: <1 %% gain)")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--cpu-threads", default="32,64,128", help="thread counts of the CPU-baseline sweep (comma separated, each capped by the affinity mask)")
    ap.add_argument("--no-var-chain", action="store_true", help="skip the cpu_baseline leg's chained-parity check with timm==1.0.3's variance RmsNorm (one more oracle episode, ~10 s)")
    ap.add_argument("--cpu-batch", type=int, default=8, help="episodes in the CPU-baseline sample of the pi_I leg")
    ap.add_argument("--force-dim", type=int, default=3,
                    help="width of the tactile vector m_t fed to the observation MLP (reference default 3 = the marker tracker's force estimate, "
                         "bridge_controller.py:25; BASELINE.json's synthetic workload names a 64-d tactile vector: --force-dim 64)")
    ap.add_argument("--alt-compute-steps", type=int, default=12,
                    help="steps of the extra timed pass with the OTHER 16-bit activation type of RDT-1B (reported as `alt_rdt_compute`: a second runner on the "
                         "same weights, its own graphs; workloads with an RDT chunk only; 0 = skip)")
    ap.add_argument("--latency-steps", type=int, default=6,
                    help="steps of the extra ONE-batch-at-a-time pass run after the timed region (reported as `latency_mode`; 0 = skip)")
    return ap.parse_args()


def synth_inputs(B, T, res, seed, device, force_dim=3):
    """frames 0.2+0.8*U(0,1) (batch mean ~0.6 -> the reference's normalise branch), state/force N(0,1),
    vla U(0,1), unit stats."""
    from vlatouch import synth
    g = synth.inputs_rng(seed)
    mk = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return dict(
        cam1=mk((0.2 + 0.8 * g.random((B, 3, res, res), dtype=np.float32))),
        cam2=mk((0.2 + 0.8 * g.random((B, 3, res, res), dtype=np.float32))),
        state=mk(g.standard_normal((B, 10), dtype=np.float32)),
        forces=mk(g.standard_normal((B, force_dim), dtype=np.float32)),
        vla=mk(g.uniform(0, 1, (B, T, 10)).astype(np.float32)),
    )


def main():
    args = parse()
    if args.workload in ("train", "train_lstm"):     # the controller training step (one JSON line, tools/train_bench.py; 1 GPU)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import train_bench
        B = args.batch if args.batch != 32 else 128                   # the reference's training default (bridge_train.py:698)
        fn = train_bench.bench_si if args.workload == "train" else train_bench.bench_lstm
        print(json.dumps(fn(B, args.steps)))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` launches its own ranks: re-exec under torch.distributed.run, one process per GPU, rendezvous on
        # 127.0.0.1 (the container hostname may not resolve).  Rank 0 of the relaunched job prints the ONE JSON line.
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (python bench.py --gpus N does it itself)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path in the product)"
    # VLATOUCH_BENCH_SHARE_GPU=1: every rank on cuda:0 over gloo — the one-GPU test box's stand-in for N GPUs over RCCL (tests/test_gpu_multiproc.py)
    share = os.environ.get("VLATOUCH_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_grad_enabled(False)
    dist = None
    bcast_bytes, bcast_s = 0, 0.0
    # a process group whenever the launcher set one up (torch.distributed.run exports WORLD_SIZE — also for `--gpus 1`, so that the RCCL path
    # — communicator, weight broadcast, barrier, max-reduce of the timing — runs on one GPU exactly as it does on eight); plain
    # `python bench.py` (N = 1) stays free of torch.distributed
    use_dist = "WORLD_SIZE" in os.environ
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from vlatouch import _lib as L
    from vlatouch import synth
    from residual_controller.bridge_controller import DiffusionController

    # ---- frozen weights: rank 0 generates the deterministic synthetic set, the others receive it over RCCL
    t0 = time.time()
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):       # the mirrored reference classes print while they build ("init SI without model args"): stdout carries ONE JSON line
        ctrl = synth.build_controller(DiffusionController, precision=args.precision, device=dev, size=args.dino, stats=synth.unit_stats(), force_dim=args.force_dim)
    if use_dist:
        from vlatouch.dist import broadcast_controller_weights
        torch.cuda.synchronize(dev)
        tb = time.time()
        bcast_bytes += broadcast_controller_weights(ctrl, src=0)
        torch.cuda.synchronize(dev)
        bcast_s += time.time() - tb
    setup_s = time.time() - t0
    B, T = args.batch, args.horizon
    # batches in flight by default: 3 for the RDT / pi_I workloads (round 4, measured on one box: full 410 -> 422 chunks/s, rdt 459 -> 464, pi_refine 4 277 -> 4 536,
    # lstm 6 952 -> 7 342; 4 in flight: 404), 2 for `robot` (137 vs 134 with 3), 1 for the rest
    n_streams = args.streams if args.streams > 0 else (3 if args.workload in ("full", "rdt", "pi_refine", "lstm") else (2 if args.workload == "robot" else 1))
    # every batch in flight reads its OWN input set (frames, state, tactile vector; below: its own condition tokens): independent batches of a
    # real pipeline share nothing but the weights, and the metric must not profit from cache reuse between slots
    inps = [synth_inputs(B, T, args.res, 1234 + rank + 1000 * si, dev, args.force_dim) for si in range(n_streams)]
    inp = inps[0]
    noise_bufs = [torch.empty(10, B, T, 10, dtype=torch.float32, device=dev) for _ in range(n_streams)]
    out_holders = [{} for _ in range(n_streams)]
    vla_bufs = [torch.empty(B, T, 10, dtype=torch.float32, device=dev) for _ in range(n_streams)]
    xinit_bufs = [torch.empty(B, 64, 128, dtype=torch.float32, device=dev) for _ in range(n_streams)]
    from vlatouch import ops as _ops
    rng = _ops.DeviceRng(20250928 + rank, dev)

    # ---- RDT-1B chunk generator (config NOT in the reference: upstream RDT-1B values, SURVEY §8a-8 [assumed-upstream])
    rdt = rin = None
    RDT1B = dict(hidden=2048, depth=28, heads=32, horizon=64, action_dim=128, lang_token_dim=4096, img_token_dim=1152,
                 state_token_dim=128, max_lang_cond_len=1024, img_cond_len=4374)
    if args.workload in ("full", "rdt", "robot"):
        from models.rdt_runner import RDTRunner
        rdt_dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
        cfg = {"rdt": {"hidden_size": 2048, "depth": 28, "num_heads": 32, "rms_norm": args.rms_mode}, "lang_adaptor": "mlp2x_gelu", "img_adaptor": "mlp2x_gelu",
               "state_adaptor": "mlp3x_gelu",
               "noise_scheduler": {"num_train_timesteps": 1000, "num_inference_timesteps": args.rdt_steps, "beta_schedule": "squaredcos_cap_v2",
                                   "prediction_type": "sample", "clip_sample": False}}
        rdt = RDTRunner(action_dim=128, pred_horizon=64, config=cfg, lang_token_dim=4096, img_token_dim=1152, state_token_dim=128,
                        max_lang_cond_len=1024, img_cond_len=4374, dtype=rdt_dtype, device=dev, init_weights=False, compute_dtype=args.rdt_compute)
        rdt.load_state_dict(synth.fill_state_dict_device(synth.rdt_runner_shapes(**RDT1B), dev, rdt_dtype, seed=7), assign=True)
        eng = rdt.engine()
        if use_dist:
            from vlatouch.dist import broadcast_tensors
            torch.cuda.synchronize(dev)
            tb = time.time()
            bcast_bytes += broadcast_tensors(eng._weights, src=0)
            torch.cuda.synchronize(dev)
            bcast_s += time.time() - tb
            eng.repack()                                                 # the fragment-packed copies follow the received weights
        g = torch.Generator(device=dev).manual_seed(4321 + rank)
        # condition tokens are handed over in the engine's activation type (what an upstream encoder on this device writes: no cast inside the step)
        in_dtype = rdt.compute_dtype
        rn = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32).to(in_dtype)
        amask = torch.zeros(B, 1, 128, device=dev, dtype=in_dtype)
        amask[:, :, :10] = 1.0                                       # the 10 EEF dims of the unified action vector
        mk_rin = lambda: dict(lang=rn(B, args.lang_len, 4096), mask=torch.ones(B, args.lang_len, dtype=torch.bool, device=dev), img=rn(B, 4374, 1152),
                              state=rn(B, 1, 128), amask=amask, freq=torch.full((B,), 10.0, device=dev))
        rins = [mk_rin() for _ in range(n_streams)]               # 322 MB of image tokens per slot
        rin = rins[0]
    lstm = lstm_in = None
    dino_c = synth.DINOV2_CONFIGS[args.dino]
    dino_sd = synth.torch_state_dict(synth.dinov2_shapes(dino_c["hidden"], dino_c["layers"]), prefix=f"dinov2-{args.dino}.")
    if args.workload == "lstm":         # the alternative residual head: obs encoding + T sequential LSTM ticks per chunk
        from residual_controller.lstm_step_controller import TactileLSTMController
        lstm = TactileLSTMController(device=dev, precision=args.precision, image_model_path=f"facebook/dinov2-{args.dino}",
                                     image_state_dict=dino_sd)
        for name, shp in synth.lstm_controller_shapes(dino_c["hidden"]).items():
            getattr(lstm, name).load_state_dict(synth.torch_state_dict(shp, prefix=f"lstm_ctrl.{name}."))
        lstm.to(dev)
        lstm.stats = {k: v.to(dev) for k, v in synth.unit_stats().items()}
        lstm_ins = [dict(forces=torch.randn(B, T, 3, device=dev)) for _ in range(n_streams)]
        lstm_in = lstm_ins[0]
    mk = mk_frames = None
    if args.workload == "marker":       # GelSight frames -> marker displacements + force estimate, a 256-frame stream per step
        from residual_controller.tactile.marker.marker_tracker import EnhancedMarkerTracker
        rngm = np.random.default_rng(77)
        base_frames = [synth.synth_gel_frame(rngm, shift=(1.5 * np.sin(i), 1.0 * np.cos(i)), bulge=0.5 * i) for i in range(8)]
        mk_frames = torch.from_numpy(np.stack([base_frames[i % 8] for i in range(8 * B)])).to(dev)       # [8B, 240, 320, 3] uint8
        mk = EnhancedMarkerTracker(7, 9, device=dev)
        mk.calibrate(base_frames[0])
        args.no_graph = True                                                # the stream API returns host arrays (counts are read back)
    sig = sig_px = None
    if args.workload in ("siglip", "robot"):       # the RDT image tower on the 6 frames of every chunk (so400m, 384x384, 729 tokens each)
        from vlatouch.engine import SiglipEngine
        c = synth.SIGLIP_CONFIGS["so400m"]
        wdt = torch.float32
        ssd = synth.fill_state_dict_device(synth.siglip_shapes(**c), dev, wdt, seed=9)
        sig = SiglipEngine({k: v.cpu() for k, v in ssd.items()}, heads=c["heads"], precision="fp16" if args.precision == "bf16" else "fp32", device=dev)
        del ssd
        sig_pxs = [(2.0 * torch.rand(6 * B, 3, 384, 384, device=dev) - 1.0) for _ in range(n_streams)]
        sig_px = sig_pxs[0]
        if args.workload == "robot":
            tok_bufs = [torch.empty(6 * B, 729, 1152, dtype=rdt.compute_dtype, device=dev) for _ in range(n_streams)]
    setup_s = time.time() - t0

    ctx = {"rdt": rdt, "rins": rins if rdt is not None else None}       # which RDT runner the step drives (the alt-compute pass swaps it)

    def step(slot=0):
        out_holder, noise_buf = out_holders[slot], noise_bufs[slot]
        inp = inps[slot]
        rdt = ctx["rdt"]
        rin = ctx["rins"][slot] if rdt is not None else None
        lstm_in = lstm_ins[slot] if lstm is not None else None
        sig_px = sig_pxs[slot] if sig is not None else None
        if args.workload == "lstm":
            obs = lstm.encode_observation(inp["state"], inp["cam1"], inp["cam2"])
            out_holder["out"] = lstm.predict_sequence(obs, inp["vla"], lstm_in["forces"])
            return
        if args.workload == "siglip":
            out_holder["out"] = sig.forward(sig_px)
            return
        if args.workload == "robot":
            # the robot step (franka_model_eef.py:283-313 + frank_inference_eef.py:495-533): 6 camera frames per chunk through the SigLIP tower
            # -> image tokens (cast to the RDT dtype) -> RDT chunk -> first T ticks x 10 EEF dims -> DINOv2 x2 + MLP + interpolant SDE
            tok = sig.forward(sig_px)                                            # [6B, 729, 1152] fp32
            img_tok = _ops.cast(tok, rdt.compute_dtype, out=tok_bufs[slot]).view(B, 6 * tok.shape[1], tok.shape[2])
            x0 = rng.normal_(xinit_bufs[slot], round_bf16=args.precision == "bf16")
            chunk = rdt.predict_action(rin["lang"], rin["mask"], img_tok, rin["state"], rin["amask"], rin["freq"], x_init=x0, return_fp32=True)
            out_holder["chunk"] = chunk
            vla_r = _ops.slice_cast(chunk, T, 10, out=vla_bufs[slot])
            rng.normal_(noise_buf)
            out_holder["out"] = ctrl.predict(inp["state"], vla_r, inp["cam1"], inp["cam2"], inp["forces"], noise=noise_buf)
            return
        if args.workload == "marker":
            out_holder["out"] = mk.track_frames(mk_frames)
            return
        if args.workload == "dino_mlp":
            out_holder["out"] = ctrl.encode_observation(inp["state"], inp["cam1"], inp["cam2"], inp["forces"])
            return
        vla = inp["vla"]
        obs = None
        if rdt is not None:
            if args.workload == "full" and not args.no_overlap:
                # the observation encoding (DINOv2 x2 + MLP) does not depend on the RDT chunk: run it on a second HIP stream so
                # it fills the CUs the small per-step RDT GEMMs leave idle (fork/join is captured in the hipGraph)
                main = torch.cuda.current_stream(dev)
                side_stream = side_streams[slot]                     # one side stream per batch in flight (a shared one serialised the slots)
                side_stream.wait_stream(main)
                with torch.cuda.stream(side_stream):
                    obs = ctrl.encode_observation(inp["state"], inp["cam1"], inp["cam2"], inp["forces"])
            # a_t = RDT chunk [B, 64, 128] -> the 10 EEF dims of the first T ticks feed the controller (frank_inference_eef.py:495-517)
            x0 = rng.normal_(xinit_bufs[slot], round_bf16=args.precision == "bf16")      # the N(0,1) start (rdt_runner.py:136), vt_randn
            chunk = rdt.predict_action(rin["lang"], rin["mask"], rin["img"], rin["state"], rin["amask"], rin["freq"], x_init=x0, return_fp32=True)
            if obs is not None:
                torch.cuda.current_stream(dev).wait_stream(side_stream)
            out_holder["chunk"] = chunk
            if args.workload == "rdt":
                return
            vla = _ops.slice_cast(chunk, T, 10, out=vla_bufs[slot])       # chunk[:, :T, :10] as fp32, one kernel
        rng.normal_(noise_buf)       # the reference's torch.randn_like draws (bridge_model.py:372), on device (vt_randn: Philox)
        out_holder["out"] = ctrl.predict(inp["state"], vla, inp["cam1"], inp["cam2"], inp["forces"], noise=noise_buf, obs_cond=obs)

    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    stream = streams[0]
    side_streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    graphs = [None] * n_streams
    for si, st in enumerate(streams):
        with torch.cuda.stream(st):
            for _ in range(max(1, args.warmup) if si == 0 else 1):       # warm-up also sizes every workspace (no allocation inside the graph)
                step(si)
            st.synchronize()
            if not args.no_graph:
                try:
                    g_ = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_, stream=st):
                        step(si)
                    g_.replay()
                    st.synchronize()
                    graphs[si] = g_
                except Exception as e:          # pragma: no cover
                    if rank == 0:
                        print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
    graph = graphs[0] if all(g_ is not None for g_ in graphs) else None

    def run(i):
        si = i % n_streams
        with torch.cuda.stream(streams[si]):
            if graph is not None:
                graphs[si].replay()
            else:
                step(si)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        # a few untimed replays in the timed pattern (the first replays of a graph pay one-time costs)
        for i in range(n_streams * 2):
            run(i)
        # ---- timed region: exactly K steps, barrier + synchronize on both sides.  With S streams, step i runs on stream i % S: up to
        #      S batches are in flight; a step's latency is measured from its own enqueue point on its stream to its completion.
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        barrier()
        t_start = time.perf_counter()
        for i in range(args.steps):
            st = streams[i % n_streams]
            ev0[i].record(st)
            run(i)
            ev1[i].record(st)
        barrier()
        elapsed = time.perf_counter() - t_start
        lat = sorted(ev0[i].elapsed_time(ev1[i]) for i in range(args.steps))
        p50 = lat[len(lat) // 2]
        out_holder = out_holders[0]

        # ---- latency mode inside the same run: ONE batch at a time (slot 0's graph back to back on its stream) — what a caller that waits for
        #      each refined batch sees; same barrier + synchronize bracket, same max over ranks
        lat_elapsed = None
        if args.latency_steps > 0 and n_streams > 1:
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.latency_steps):
                run(0)
            barrier()
            lat_elapsed = time.perf_counter() - t1

        # ---- roofline leg: one eager step with HIP events around every launch of the dominant GEMM kernel
        lib = L.lib()
        prof = {}
        for mode in (2, 3, 4, 6):    # 2 = 256-square GEMM tile, 3 = 160 / 128-column GEMM tiles, 4 = cached cross-attention, 6 = fused U-Net convolution (one eager step each)
            lib.vt_prof_enable(mode)
            step()
            stream.synchronize()
            ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_long()
            L.check(lib.vt_prof_collect(C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "vt_prof_collect")
            lib.vt_prof_enable(0)
            prof[mode] = (ms.value, fl.value, by.value, n.value)

    # ---- the same timed pattern with the other 16-bit activation type of RDT-1B (f16 <-> bf16): a second runner on the SAME weights, its own
    #      workspaces and graphs; reported beside the headline so that the record carries what the choice of the default costs / buys
    # the 16-bit type RDT-1B actually ran in ("auto" starts in fp16 and falls back to bf16 only if the range guard fires) + the guards' sticky words, read once here
    rdt_used = "f16" if (rdt is not None and rdt.compute_dtype == torch.float16) else "bf16"
    range_guard = None
    if rdt is not None and args.precision == "bf16":
        bits = rdt.overflowed()
        range_guard = {"rdt_compute_requested": args.rdt_compute, "rdt_compute_used": rdt_used, "rdt_flag": bits, "rdt_flag_names": L.range_names(bits),
                       "note": "sticky device word OR-ed by the kernels of every step of this run (warm-up, graph replays, timed region): 0 = no value left the compute type's range"}
    alt_elapsed = None
    if rdt is not None and args.precision == "bf16" and args.alt_compute_steps > 0 and args.workload in ("full", "rdt"):
        try:                                            # a side measurement: it must never take the headline line down with it
            from models.rdt_runner import RDTRunner as _RR
            alt_mode = "bf16" if rdt_used == "f16" else "f16"
            rdt_alt = _RR(action_dim=128, pred_horizon=64, config=cfg, lang_token_dim=4096, img_token_dim=1152, state_token_dim=128, max_lang_cond_len=1024,
                          img_cond_len=4374, dtype=rdt_dtype, device=dev, init_weights=False, compute_dtype=alt_mode)
            rdt_alt.load_state_dict(rdt.state_dict(), assign=True)
            alt_in = rdt_alt.compute_dtype
            rins_alt = [{k: (v.to(alt_in) if v.is_floating_point() and k != "freq" else v) for k, v in r_.items()} for r_ in rins]
            ctx["rdt"], ctx["rins"] = rdt_alt, rins_alt
            graphs_alt = []
            for si, st in enumerate(streams):
                with torch.cuda.stream(st):
                    step(si)
                    st.synchronize()
                    g_ = None
                    if graph is not None:
                        g_ = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_, stream=st):
                            step(si)
                        g_.replay()
                        st.synchronize()
                    graphs_alt.append(g_)

            def run_alt(i):
                si = i % n_streams
                with torch.cuda.stream(streams[si]):
                    if graphs_alt[si] is not None:
                        graphs_alt[si].replay()
                    else:
                        step(si)
            for i in range(n_streams * 2):
                run_alt(i)
            barrier()
            t1 = time.perf_counter()
            for i in range(args.alt_compute_steps):
                run_alt(i)
            barrier()
            alt_elapsed = time.perf_counter() - t1
            ctx["rdt"], ctx["rins"] = rdt, rins
            del graphs_alt, rins_alt, rdt_alt
            torch.cuda.empty_cache()
        except Exception as e:      # pragma: no cover
            ctx["rdt"], ctx["rins"] = rdt, rins
            alt_elapsed = None
            if rank == 0:
                print(f"[bench] alt-compute pass failed ({type(e).__name__}: {e}); the headline is unaffected", file=sys.stderr)

    if dist is not None:
        tt = torch.tensor([elapsed, lat_elapsed or 0.0, alt_elapsed or 0.0], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
        if lat_elapsed is not None:
            lat_elapsed = float(tt[1].item())
        if alt_elapsed is not None:
            alt_elapsed = float(tt[2].item())
    total_chunks = B * world * args.steps * (8 if args.workload == "marker" else 1)
    value = total_chunks / elapsed

    WL = {
        "full": ("refined action chunks/sec (RDT-1B chunk + DINOv2 x2 + pi_I)",
                 "full = BASELINE configs[3]: per refined chunk one RDT-1B action chunk (D2048 L28 H32, 64x128, %d-step DPM-Solver++, %d lang + "
                 "4374 img condition tokens, cross-attn K/V cached per chunk) -> 10 EEF dims x first %d ticks -> 2x DINOv2-%s CLS @%d + "
                 "state/force MLP + 10-step interpolant SDE (v_net+s_net)" % (args.rdt_steps, args.lang_len, T, args.dino, args.res)),
        "pi_refine": ("refined action chunks/sec (pi_I only, no RDT chunk generation)",
                      "pi_refine: 2x DINOv2-%s CLS @%d + state/force MLP + 10-step interpolant SDE (v_net+s_net), T=%d; BASELINE configs[3] "
                      "WITHOUT the RDT-1B chunk generator" % (args.dino, args.res, T)),
        "dino_mlp": ("encoded observations/sec", "dino_mlp = BASELINE configs[1]: 2x DINOv2-%s @%d + state/force MLP" % (args.dino, args.res)),
        "lstm": ("refined action chunks/sec (LSTM residual head, no RDT chunk generation)", "lstm (SURVEY 8a-7): 2x DINOv2-%s CLS @%d + obs MLP + "
                 "%d sequential ticks of force MLP -> 2-layer LSTM -> residual head per chunk" % (args.dino, args.res, T)),
        "robot": ("refined action chunks/sec (SigLIP image tower + RDT-1B chunk + DINOv2 x2 + pi_I)",
                  "robot = the reference's whole robot step (franka_model_eef.py:283-313, frank_inference_eef.py:495-533): 6 x 384x384 frames per chunk -> "
                  "SigLIP-so400m tower -> 4374 image tokens -> RDT-1B chunk (%d-step DPM-Solver++, %d lang tokens) -> first %d ticks x 10 EEF dims -> 2x "
                  "DINOv2-%s @%d + MLP + 10-step interpolant SDE; batch %d chunks" % (args.rdt_steps, args.lang_len, T, args.dino, args.res, B)),
        "siglip": ("chunks' worth of image tokens/sec (6 frames per chunk)", "siglip (SURVEY 8f-1): SigLIP-so400m-patch14-384 tower, 6 x 384x384 frames per "
                   "chunk -> 6 x 729 x 1152 image tokens, batch %d chunks (%d images per step)" % (B, 6 * B)),
        "marker": ("GelSight frames/sec (marker displacements + force estimate m_t)", "marker (SURVEY 8f-3): %d frames of 240x320x3 per step: blur -> adaptive "
                   "threshold -> open -> components -> contour centroids -> nearest-baseline displacement -> force; value counts FRAMES" % (8 * B)),
        "rdt": ("RDT-1B action chunks/sec", "rdt = BASELINE configs[2] shape: RDT-1B, %d-step DPM-Solver++, cached T5-sized (4096-d) language "
                "tokens, batch %d" % (args.rdt_steps, B)),
    }[args.workload]
    res = {
        "metric": WL[0],
        "value": round(value, 2), "unit": "frames/s" if args.workload == "marker" else "chunks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000 * elapsed / args.steps, 4), "p50_step_latency_ms": round(p50, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16 model, 16-bit activations in IEEE fp16 (RDT-1B --rdt-compute f16, DINOv2 tower; same width and MFMA rate as bf16)" if rdt_used == "f16"
                  else "bf16 (DINOv2 tower: IEEE fp16)") if args.precision == "bf16" else "fp32",
        "dtypes": ({"rdt": ("bf16 weights converted exactly to IEEE fp16, fp16 activations + f16 MFMA" if rdt_used == "f16" else "bf16 storage + bf16 MFMA")
                           + ", fp32 residual stream, accumulation and solver state", "dinov2": "IEEE fp16 storage + f16 MFMA, fp32 residual stream",
                    "siglip": "IEEE fp16 storage + f16 MFMA", "obs_mlp": "fp32", "unet_sampler": "fp32 storage, split-bf16 (3 bf16 MFMAs per product)",
                    "lstm_head": "fp32 storage, split-bf16"} if args.precision == "bf16" else {"all": "fp32 storage + fp32 MFMA"}),
        "data": "synthetic",
        "config": {
            "workload": WL[1],
            "batch_per_gpu": B, "global_batch": B * world, "horizon": T, "parallelism": f"dp{world} (episodes sharded, no step collectives)",
            "hipgraph": graph is not None, "batches_in_flight": n_streams, "inputs": "one synthetic input set per batch in flight (no sharing between slots)",
            "tactile_vector_dim": args.force_dim,
            "ms_per_step_semantics": "wall time of the timed region / steps; with batches_in_flight > 1 consecutive steps (independent batches) "
                                     "overlap on separate HIP streams, so p50_step_latency_ms (enqueue -> completion of one batch) exceeds ms_per_step", "rdt_mode": (("IEEE fp16" if rdt_used == "f16" else "bf16") + " storage + MFMA (bf16 model), fp32 residual stream") if args.precision == "bf16" else "fp32",
            "dino_mode": "IEEE fp16 storage + f16 MFMA, fp32 residual stream" if args.precision == "bf16" else "fp32", "unet_mode": "split-bf16 (3 bf16 MFMAs/k-step, fp32 storage)" if args.precision == "bf16" else "fp32 MFMA",
            "weights": "random-init synthetic of the named architectures (no checkpoints offline; RDT-1B hyper-parameters are upstream's, "
                       "not in the reference)", "setup_s": round(setup_s, 1),
        },
    }
    # one batch at a time: chunks/s and ms per step with nothing else in flight (`--streams 1`); with one stream the timed region IS that mode
    if lat_elapsed is not None:
        res["latency_mode"] = {"chunks_per_s": round(B * world * args.latency_steps * (8 if args.workload == "marker" else 1) / lat_elapsed, 2),
                               "ms_per_step": round(1000 * lat_elapsed / args.latency_steps, 4), "steps": args.latency_steps, "batches_in_flight": 1,
                               "note": "same graphs, slot 0 replayed back to back after the timed region; `value` above is the throughput mode"}
    elif n_streams == 1:
        res["latency_mode"] = {"chunks_per_s": res["value"], "ms_per_step": res["ms_per_step"], "steps": args.steps, "batches_in_flight": 1,
                               "note": "the timed region itself (one batch in flight)"}
    if range_guard is not None:
        res["range_guard"] = range_guard
    if alt_elapsed is not None:
        res["alt_rdt_compute"] = {"rdt_compute": "bf16" if rdt_used == "f16" else "f16", "chunks_per_s": round(B * world * args.alt_compute_steps / alt_elapsed, 2),
                                  "ms_per_step": round(1000 * alt_elapsed / args.alt_compute_steps, 4), "steps": args.alt_compute_steps, "batches_in_flight": n_streams,
                                  "note": "the same step with the other 16-bit activation type of RDT-1B (bf16 = the reference's execution dtype: |chunk - fp32 oracle| "
                                          "~8e-3..1.1e-2; f16 = the default: ~1e-3), same weights, same box, timed right after the headline"}
    # end-to-end matrix-pipe fraction: SURVEY 8(d)'s algorithmic GFLOP per chunk (2 MAC, cached condition K/V, no recompute credit) x chunks / wall time
    gf_chunk = {"full": (1030.0 + 168.0 * args.rdt_steps + 57.0) + 99.3, "rdt": 1030.0 + 168.0 * args.rdt_steps + 57.0, "pi_refine": 99.3, "dino_mlp": 92.6}.get(args.workload)
    if gf_chunk is not None and args.dino == "base" and args.horizon == 16:
        tf = gf_chunk * total_chunks / elapsed / 1e3
        res["end_to_end_mfma"] = {"algorithmic_gflop_per_chunk": round(gf_chunk, 1), "achieved_tflops": round(tf, 1), "peak_tflops": PEAK_BF16_TFLOPS * world,
                                  "frac": round(tf / (PEAK_BF16_TFLOPS * world), 4), "note": "whole step incl. HBM- and launch-bound phases against the dense bf16 MFMA peak"}
    if dist is not None:
        res["config"]["weight_broadcast"] = {"bytes": int(bcast_bytes), "seconds": round(bcast_s, 3), "backend": dist.get_backend(),
                                             "note": "one-time, before the timed region; no collective inside the step loop"}
    # HBM-side traffic per launch comes from separate rocprofv3 --pmc passes (tools/pmc_summary.py); the committed summary of
    # the last such run is attached when present (null otherwise: counters cannot be read from inside this process)
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        pass
    # the counters are a BUILDER-side measurement (separate rocprofv3 --pmc passes, tools/profile_round.sh) read from a committed file, not taken in this run:
    # stamped with the fingerprint of the kernel sources they were taken on and whether that is the tree this bench runs from
    traffic_source = {"file": "profiles/pmc_traffic.json", "measured_in_this_run": False, "csrc_sha16_of_counters": pmc.get("csrc_sha16"),
                      "csrc_sha16_of_this_tree": L.csrc_sha16(), "same_kernel_sources": pmc.get("csrc_sha16") == L.csrc_sha16()} if pmc else None

    def roof(mode, name, pmc_key):
        ms_, fl_, by_, n_ = prof[mode]
        if n_ <= 0 or ms_ <= 0:
            return None
        tf = fl_ / (ms_ * 1e-3) / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_BF16_TFLOPS, 4), "launches_per_step": n_, "avg_launch_us": round(1000 * ms_ / n_, 2),
                "algorithmic_gflop_per_step": round(fl_ / 1e9, 1), "algorithmic_gbytes_per_step": round(by_ / 1e9, 3),
                "share_of_step_time": round(ms_ / (1000 * elapsed / args.steps), 3),
                "traffic": (round(pmc[pmc_key]["per_launch_bytes"] / 1e9, 4) if pmc_key in pmc else None),
                "traffic_unit": "GB per launch (PMC 2*FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json)",
                "traffic_source": traffic_source,
                "algorithmic_gbytes_per_launch": round(by_ / 1e9 / n_, 4)}
    # on-box measured peaks (report fractions of the nominal AND of a measured peak): a large square bf16 GEMM on the
    # same kernel family and a device-to-device copy (read + write bytes)
    onbox = None
    if rank == 0:
        from vlatouch import ops as _ops
        with torch.cuda.stream(stream):
            a8 = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
            w8 = (torch.randn(8192, 8192, device=dev) * 0.01).to(torch.bfloat16)
            o8 = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
            src = torch.empty(1 << 28, device=dev, dtype=torch.float32)            # 1 GiB
            dst = torch.empty_like(src)
            for _ in range(2):
                _ops.gemm(a8, w8, out=o8, out_dtype=torch.bfloat16)
                dst.copy_(src)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(stream)
            for _ in range(5):
                _ops.gemm(a8, w8, out=o8, out_dtype=torch.bfloat16)
            e[1].record(stream)
            for _ in range(5):
                dst.copy_(src)
            e[2].record(stream)
            stream.synchronize()
            onbox = {"bf16_gemm_8192_tflops": round(5 * 2 * 8192 ** 3 / (e[0].elapsed_time(e[1]) * 1e-3) / 1e12, 1),
                     "d2d_copy_GBs": round(5 * 2 * src.numel() * 4 / (e[1].elapsed_time(e[2]) * 1e-3) / 1e9, 1)}
            del a8, w8, o8, src, dst

    r_pp = roof(2, "gemm_pt_kernel + gemm_pp256d_kernel (256x256x64 ping-pong tile, 16-bit MFMA; gemm_pt = persistent, operand stream continuous across tiles, "
                   "epilogue on registers inside the k-loop: fused condition K|V projections of RDT, image / language adaptors, DINOv2 qkv / fc1 / out-projection / fc2; "
                   "gemm_pp256d = the same tile, one launch block per tile, for the epilogue kinds the persistent kernel does not have)", "gemm_pp256")
    r_gl = roof(3, "gemm_pw_kernel (160x128x64 tile, frozen fragment-packed weights streamed global -> VGPR, activations through an LDS-DMA ring; with the few "
                   "gemm_glds_kernel launches: the per-denoise-step Linears of RDT, M = batch x 67 rows, incl. the fused qkv projection; the RMSNorm between a "
                   "residual Linear and the next Linear rides on their epilogues: x * gain + sums of squares out of the first, a row scale in the second)", "gemm_pw")
    r_at = None
    if prof.get(4, (0, 0, 0, 0))[3] > 0 and prof[4][0] > 0:
        ms_, fl_, by_, n_ = prof[4]
        gbs = by_ / (ms_ * 1e-3) / 1e9
        r_at = {"kernel": "attn_kvt_ring_kernel (RDT cross-attention against the cached condition K / Vt tile stream, re-read every denoise step)",
                "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                "launches_per_step": n_, "avg_launch_us": round(1000 * ms_ / n_, 2), "algorithmic_gbytes_per_step": round(by_ / 1e9, 3),
                "algorithmic_gbytes_per_launch": round(by_ / 1e9 / n_, 4),
                "traffic": (round(pmc["attn_kvt"]["per_launch_bytes"] / 1e9, 4) if "attn_kvt" in pmc else None),
                "traffic_unit": "GB per launch averaged over image- and language-layer calls (PMC 2*FETCH_SIZE + WRITE_SIZE)",
                "share_of_step_time": round(ms_ / (1000 * elapsed / args.steps), 3)}
        if rdt is not None:      # which softmax ran: the fixed-maximum form needs a load-time score bound per block (mean-square RmsNorm only) <= 40
            sb = [b for b in rdt.engine().score_bounds]
            fixed_on = os.environ.get("VLATOUCH_ATTN_FIXEDMAX", "1") != "0"
            n_fixed = sum(1 for b in sb if 0.0 < b <= 40.0) if fixed_on else 0
            r_at["softmax"] = "fixed" if n_fixed == len(sb) else ("online" if n_fixed == 0 else f"fixed in {n_fixed} of {len(sb)} blocks")
            r_at["score_bound_max"] = round(max(sb), 2) if sb else None
            r_at["rms_mode"] = rdt.rms_mode
        r_at["tile_dma_cache_policy"] = "default" if os.environ.get("VLATOUCH_KVT_NT", "1") == "0" else "nt"
    r_uc = None
    if prof.get(6, (0, 0, 0, 0))[3] > 0 and prof[6][0] > 0:
        ms_, fl_, by_, n_ = prof[6]
        r_uc = {"kernel": "uconv_kernel (fused conditional 1-D U-Net convolution of the pi_I sampler: GroupNorm + Mish + FiLM + residual of the input resolved in "
                          "the prologue, split-bf16 MFMA with weights streamed global -> VGPR; a dependent chain of 300 launches per refined batch)",
                "bound": "mfma", "achieved": round(fl_ / (ms_ * 1e-3) / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl_ / (ms_ * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "launches_per_step": n_, "avg_launch_us": round(1000 * ms_ / n_, 2),
                "issued_mfma_gflop_per_step": round(fl_ / 1e9, 1), "algorithmic_gflop_per_step": round(fl_ / 3e9, 1),
                "flop_note": "achieved / frac count the bf16 MFMA flops ISSUED (split-bf16: 3 products per fp32 product); the algorithmic figure is a third",
                "algorithmic_gbytes_per_step": round(by_ / 1e9, 3),
                "achieved_GBs": round(by_ / (ms_ * 1e-3) / 1e9, 1), "share_of_step_time": round(ms_ / (1000 * elapsed / args.steps), 3),
                "note": "latency-bound by construction (M = batch x T_l <= 512 rows per launch): neither roof is near; tools/uconv_phases.py has the "
                        "in-kernel phase times", "traffic": None}
    if onbox is not None:
        res["onbox_peaks"] = onbox
        for r in (r_pp, r_gl):
            if r is not None:
                # NOT an independent peak: this library's own 256-square tile on an 8192^3 product on this box (what the clocks of THIS box give a
                # long MFMA-bound kernel); the independent figure is `peak` (2.5 PF dense; 2 495 TF measured by an MFMA-only loop, MI355X_MICROARCH.md)
                r["frac_of_onbox_gemm_8192"] = round(r["achieved"] / onbox["bf16_gemm_8192_tflops"], 4)
    if r_pp is not None:
        res["roofline"] = r_pp
        res["roofline_other"] = [r for r in (r_gl, r_at, r_uc) if r is not None]
    elif r_gl is not None:
        res["roofline"] = r_gl
        res["roofline_other"] = [r for r in (r_at, r_uc) if r is not None]

    # ---- CPU baseline: the oracle (fp32 torch ops on the host cores), rank 0, N=1 only, on a BOUNDED sample:
    #      pi_I leg on CB episodes; RDT leg on whole episodes (the reference's own schedule: K/V re-projected every step).
    #      chunks/s = 1 / (t_pi / CB + t_rdt).  Round 6: the thread count is SWEPT (--cpu-threads 32,64,128, passive OpenMP waits — set at the top of this
    #      file, before torch loads its OpenMP runtime) on the pi_I leg and on a one-denoise-step probe of the RDT leg; the timed legs run at the best count of
    #      each, `cores` = the count of the dominant (RDT) leg, the rest of the sweep is in `cores_sweep`.
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("pi_refine", "full"):
        from oracle import controller as oc
        avail = len(os.sched_getaffinity(0))
        cand = sorted({max(1, min(avail, int(c))) for c in str(args.cpu_threads).split(",") if c.strip()})
        CB = min(B, args.cpu_batch)
        cpu = {k: v[:CB].cpu() for k, v in inp.items()}
        z = torch.randn(10, CB, T, 10)
        sds = (dino_sd, synth.torch_state_dict(synth.state_encoder_shapes(2 * dino_c["hidden"] + 10 + args.force_dim), prefix="state_encoder."),
               synth.torch_state_dict(synth.si_net_shapes(10, 256), prefix="si.", salt="ema"), synth.unit_stats())
        heads = 12 if args.dino == "base" else 6
        f = lambda: oc.predict(sds[0], heads, sds[1], sds[2], sds[3], cpu["state"], cpu["vla"], cpu["cam1"], cpu["cam2"], cpu["forces"], z)
        torch.set_num_threads(cand[0])
        ref = f()                                                          # untimed first call (one-time costs); also the parity check of the benchmarked config
        sweep = {}
        for c in cand:
            torch.set_num_threads(c)
            ts = []
            for _ in range(max(1, args.cpu_iters)):
                t1 = time.perf_counter(); f(); ts.append(time.perf_counter() - t1)
            sweep[c] = {"pi_s_per_chunk": round(float(np.median(ts)) / CB, 4)}
        c_pi = min(cand, key=lambda c: sweep[c]["pi_s_per_chunk"])
        t_pi = sweep[c_pi]["pi_s_per_chunk"]
        cores = c_pi
        got = ctrl.predict(inp["state"][:CB], inp["vla"][:CB], inp["cam1"][:CB], inp["cam2"][:CB], inp["forces"][:CB], noise=z.to(dev)).cpu()
        sample = f"pi_I: {max(1, args.cpu_iters)} oracle predict() call(s) on {CB} episodes at {c_pi} threads"
        t_rdt = 0.0
        parity = {}
        if args.workload == "full":
            from oracle import rdt as orr
            sd_cpu = {k: v.float().cpu() for k, v in rdt.state_dict().items()}
            ep = lambda b: {k: (v[b:b + 1].float().cpu() if v.dtype != torch.bool else v[b:b + 1].cpu()) for k, v in rin.items()}
            # the SAME start noise for the GPU batch and the oracle's episodes: the timed oracle run doubles as the parity check of
            # the benchmarked configuration (B = 32 rows through the large-batch kernels; oracle = fp32 math on the bf16-rounded weights)
            gx = torch.Generator(device=dev).manual_seed(99)
            x_init = torch.randn(B, 64, 128, generator=gx, device=dev, dtype=torch.float32).to(rdt_dtype)
            zc = torch.randn(10, B, T, 10)

            def gpu_chain_of(runner):
                with torch.cuda.stream(stream):
                    # the chunk as the step hands it over (fp32 buffer of the solver), then the CHAIN of the step on the same inputs: first T ticks x 10 EEF dims -> predict
                    ch = runner.predict_action(rin["lang"], rin["mask"], rin["img"], rin["state"], rin["amask"], rin["freq"], x_init=x_init, return_fp32=True)
                    an = ctrl.predict(inp["state"], _ops.slice_cast(ch, T, 10), inp["cam1"], inp["cam2"], inp["forces"], noise=zc.to(dev))
                    stream.synchronize()
                return ch.float().cpu(), an.cpu()

            def oracle_episode(b, rms, steps=args.rdt_steps):
                c1 = ep(b)
                t1 = time.perf_counter()
                rc = orr.predict_action(sd_cpu, c1["lang"], c1["mask"], c1["img"], c1["state"], c1["amask"], c1["freq"], x_init[b:b + 1].float().cpu(),
                                        heads=32, horizon=64, num_inference_steps=steps, rms_mode=rms)
                return rc, time.perf_counter() - t1

            def chain_diffs(b, rc, gpu_chunk, gpu_chain):
                one = slice(b, b + 1)
                cpu1 = {k: v[one].cpu() for k, v in inp.items()}
                ref_chain = oc.predict(sds[0], heads, sds[1], sds[2], sds[3], cpu1["state"], rc[:1, :T, :10], cpu1["cam1"], cpu1["cam2"], cpu1["forces"], zc[:, one])
                return float((gpu_chunk[b] - rc[0]).abs().max()), float((gpu_chain[b] - ref_chain[0]).abs().max()), float(rc.abs().max())

            gpu_chunk, gpu_chain = gpu_chain_of(rdt)
            # one-denoise-step probe of the RDT leg at every thread count (a fifth of an episode's work each), then whole episodes at the best count
            for c in cand:
                torch.set_num_threads(c)
                sweep[c]["rdt_one_step_probe_s"] = round(oracle_episode(0, rdt.rms_mode, steps=1)[1], 3)
            cores = min(cand, key=lambda c: sweep[c]["rdt_one_step_probe_s"])
            torch.set_num_threads(cores)
            rows = sorted({0, B - 1})
            t_eps = []
            for b in rows:
                rc, dt_ = oracle_episode(b, rdt.rms_mode)
                t_eps.append(dt_)
                e_chunk, e_chain, sc = chain_diffs(b, rc, gpu_chunk, gpu_chain)
                parity[f"row{b}"] = {"chunk": e_chunk, "a_hat_chain": e_chain, "chunk_scale": sc}
            t_rdt = float(np.mean(t_eps))
            rdt_diff, chain_diff, rdt_scale = parity["row0"]["chunk"], max(v["a_hat_chain"] for v in parity.values()), parity["row0"]["chunk_scale"]
            sample += f"; RDT-1B: {len(rows)} oracle predict_action calls on 1 episode each (rows {rows}, {args.rdt_steps} steps, fp32) at {cores} threads"
            # the chain in the arithmetic released RDT-1B checkpoints need (timm==1.0.3 variance RmsNorm, models/rdt/blocks.py:22): a second runner on the same weights, episode 0
            if not args.no_var_chain and rdt.rms_mode != "var":
                try:
                    from models.rdt_runner import RDTRunner as _RV
                    cfg_v = dict(cfg, rdt=dict(cfg["rdt"], rms_norm="var"))
                    rdt_v = _RV(action_dim=128, pred_horizon=64, config=cfg_v, lang_token_dim=4096, img_token_dim=1152, state_token_dim=128, max_lang_cond_len=1024,
                                img_cond_len=4374, dtype=rdt_dtype, device=dev, init_weights=False, compute_dtype=args.rdt_compute)
                    rdt_v.load_state_dict(rdt.state_dict(), assign=True)
                    ch_v, an_v = gpu_chain_of(rdt_v)
                    rc_v, _ = oracle_episode(0, "var")
                    e_chunk, e_chain, sc = chain_diffs(0, rc_v, ch_v, an_v)
                    parity["row0_var_rmsnorm"] = {"chunk": e_chunk, "a_hat_chain": e_chain, "chunk_scale": sc, "range_flag": rdt_v.overflowed()}
                    del rdt_v, ch_v, an_v
                    torch.cuda.empty_cache()
                except Exception as e:          # pragma: no cover  (a side check: it must not take the line down)
                    parity["row0_var_rmsnorm"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:
            cpu_model = "unknown"
        try:        # physical cores = distinct (socket, core id) pairs
            phys, cur = set(), {}
            for l in open("/proc/cpuinfo"):
                if ":" in l:
                    k_, v_ = (x.strip() for x in l.split(":", 1))
                    cur[k_] = v_
                elif cur:
                    phys.add((cur.get("physical id"), cur.get("core id"))); cur = {}
            n_phys = len(phys) or None
        except Exception:
            n_phys = None
        res["cpu_baseline"] = {"value": round(1.0 / (t_pi + t_rdt), 3), "unit": "chunks/s", "cores": cores, "cpu_model": cpu_model,
                               "host_logical_cpus": os.cpu_count(), "host_physical_cores": n_phys,
                               "cores_sweep": {str(c): v for c, v in sweep.items()},
                               "cores_note": f"torch intra-op threads swept over {cand} (OMP_WAIT_POLICY={os.environ.get('OMP_WAIT_POLICY')}, KMP_BLOCKTIME={os.environ.get('KMP_BLOCKTIME')}) of "
                                             f"{n_phys} physical cores / {os.cpu_count()} logical CPUs; each leg timed at its best count (pi_I {c_pi}, RDT {cores}); `cores` = the RDT leg's (97 % of the time); "
                                             "a bounded sample, not the same amount of work as a GPU step",
                               "kind": "port", "sample": sample,
                               "pi_s_per_chunk": round(t_pi, 4), "rdt_s_per_chunk": round(t_rdt, 3),
                               "max_abs_diff_vs_gpu_pi": float((got - ref).abs().max())}
        if args.workload == "full":
            res["cpu_baseline"].update({"max_abs_diff_vs_gpu_chain": chain_diff,
                                        "chain_parity_note": "a_hat, worst of episodes %s: GPU RDT-1B bf16-model chunk (B=%d) -> slice -> predict vs oracle chunk -> oracle predict, "
                                                             "same start noise and SDE noise; north-star tolerance 1e-2 (flat); per-episode figures and the timm==1.0.3 variance-RmsNorm "
                                                             "chain in `parity`" % (sorted({0, B - 1}), B),
                                        "parity": parity,
                                        "max_abs_diff_vs_gpu_rdt": rdt_diff, "rdt_output_scale": rdt_scale,
                                        "rdt_parity_note": "GPU batch row 0 (bf16 model, %s activations, B=%d, fp32 hand-over) vs oracle fp32 on the same bf16-rounded weights / "
                                                           "inputs / start noise" % (rdt_used, B)})
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

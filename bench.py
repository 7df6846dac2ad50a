#!/usr/bin/env python3
"""Benchmark of the DDPM hot path on MI355X (contract: python bench.py --gpus N --steps K --warmup W).

A step = one training step of the denoiser on one synthetic batch already resident in HBM:
    randint t -> q_sample -> UNet forward -> ELBO -> backward -> (gradient all-reduce when N > 1) -> Adam step
on the workload of BASELINE.json configs[1] (32x32x1 images, batch 256 per GPU, stages [32,64,128,256], T=120,
bf16 MFMA compute with fp32 accumulation / fp32 master weights, dropout 0.1 active).  value = trained images/s of
the whole job.  Also reported on the same line: sampled images/s of the full T-step ancestral chain, a roofline
entry for the dominant kernel and a bounded CPU baseline (the oracle timed on the host cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dm_amd_bootstrap  # noqa: E402,F401
from dm_amd.losses.ELBO import ELBOLoss  # noqa: E402
from dm_amd.neural_nets.denoise_models import ddpm as dm_ddpm, unets as dm_unets  # noqa: E402
from dm_amd.optimizer import FusedAdam  # noqa: E402
from dm_amd.runtime import graphs  # noqa: E402

YAML = dict(stages_channels=[32, 64, 128, 256], expand_factor=2, drop_p=0.1, T=120, beta=0.25, t_embed_dim=32, activation="gelu",
            base_groups_norm=4, initializer="he_normal", dtype=None)


def build_model(dev, channels, attention_stages=(), activation=None):
    torch.manual_seed(42 + int(os.environ.get("RANK", 0)))
    extra = {"attention_stages": list(attention_stages)} if attention_stages else {}
    m = dm_ddpm.DDPM(dm_unets.BaseUNet, **{**YAML, "activation": activation or YAML["activation"]}, is_conditional=False, img_channels=channels, reconstruction_method="mse",
                     latent_dim=512, device=dev, **extra)
    return m.to(dev)


def cpu_baseline(channels, size, batch, steps):
    """The oracle (CPU restatement of the reference) timed on the host cores: same train step, bounded sample."""
    from oracle import ref_cpu as O
    torch.manual_seed(42)
    cfg = O.UNetCfg(img_channels=channels, drop_p=0.1)
    p = {k: v.clone().requires_grad_(True) for k, v in O.hash_params(cfg).items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-3)
    s = O.make_schedule(120, 0.25)
    x0 = torch.rand(batch, channels, size, size) * 2 - 1
    gen = torch.Generator().manual_seed(1)

    def step():
        t = torch.randint(1, 121, (batch,))
        masks = O.draw_drop_masks(p, cfg, x0, t, gen)
        x_t = O.q_sample(s, x0, t, torch.randn_like(x0))
        pred = O.unet_forward(p, cfg, x_t, t, drop=O.DropMasks(masks))
        loss = O.elbo_loss(s, t, pred, x0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    # sampling: a few timed steps of the ancestral chain (extrapolated to T = 120), then one full chain
    with torch.no_grad():
        frozen = {k: v.detach() for k, v in p.items()}
        x = torch.randn(batch, channels, size, size)
        ssteps = 3
        O.sample_chain(frozen, cfg, s, x, 120, [torch.randn_like(x)], steps=1)
        t1 = time.perf_counter()
        O.sample_chain(frozen, cfg, s, x, 120, [torch.randn_like(x) for _ in range(ssteps)], steps=ssteps)
        ds = (time.perf_counter() - t1) / ssteps
        t1 = time.perf_counter()
        O.sample_chain(frozen, cfg, s, x, 120, [torch.randn_like(x) for _ in range(120)])
        dfull = time.perf_counter() - t1
    return {"value": batch * steps / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} train steps of batch {batch} at {size}x{size}x{channels} (oracle/ref_cpu.py, fp32, dropout on, Adam; "
                      "the reference trainer's per-iteration checkpoint, EMA and sleeps excluded)",
            "sampling": {"value": batch / dfull, "unit": "images/s", "sample": f"one full 120-step chain of batch {batch}",
                         "extrapolated_from_3_steps": batch / (120 * ds)}}


def dropin_eager(dev, channels, size, batch, micro_steps=30):
    """The reference trainer's own loop shape through the drop-in modules, no graph driver (base_trainer.py:196-235 there):
    randint -> q -> model -> criterion -> loss.backward(retain_graph=True) -> optimizer.step() + zero_grad on every 3rd
    micro-batch (gradient accumulation 3, no 1/3 scaling)."""
    m = build_model(dev, channels)
    m.train()
    crit = ELBOLoss("mse")
    opt = FusedAdam(m.parameters(), lr=1e-3)
    x0 = torch.rand(batch, channels, size, size, device=dev) * 2 - 1

    def micro(i):
        t = torch.randint(1, m.T + 1, (batch,), device=dev)
        x_t = m.q(x0, t)
        loss = crit(m, t, m(x_t, t), x0, is_ddp=False)["Loss"]
        loss.backward(retain_graph=True)
        if i % 3 == 0:
            opt.step()
            opt.zero_grad(set_to_none=True)

    # untimed: the first optimizer step moves the parameters into FusedAdam's flat buffer, and the module captures its graphs
    # on the second pass after that
    for i in range(1, 10):
        micro(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, micro_steps + 1):
        micro(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / micro_steps
    return {"value": batch / dt, "unit": "images/s", "ms_per_micro_step": 1e3 * dt, "gradient_accumulation_steps": 3,
            "what": "the reference trainer's loop (randint, q, model, criterion, backward(retain_graph), step every 3rd) on the module "
                    "surface without a graph driver: the module replays its own captured forward / backward graphs from the second "
                    "pass of a shape on"}


def timed(step_fn, steps, warmup, barrier, dist, dev):
    for _ in range(warmup):
        step_fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        import torch.distributed as td
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = tt.item()
    return dt


def extra_configs(dev, world, dist, barrier, reducer_factory, use_graph):
    """The other GPU configs BASELINE.json names, same code, shorter runs: Flowers102-shaped 64x64x3 training (batch 128 per
    GPU, data parallel when N > 1), 128x128x3 sampling with self-attention at the 16x16 stage (128 images per GPU) and latent
    diffusion on 256x256x3 images (autoencoder + the same denoiser on the latents)."""
    out = {}
    # -- 64x64x3 training
    m = build_model(dev, 3)
    m.train()
    opt = FusedAdam(m.parameters(), lr=1e-3)
    b = 128
    x0 = torch.rand(b, 3, 64, 64, device=dev) * 2 - 1
    red = reducer_factory(m) if dist else None
    step = graphs.PipelinedTrainStep(m, ELBOLoss("mse"), opt, x0, red, use_graph=use_graph)
    dt = timed(step.replay, 40, 15, barrier, dist, dev)
    out["train_64x64x3"] = {"value": b * world * 40 / dt, "unit": "images/s", "ms_per_step": 25 * dt, "batch_per_gpu": b, "n_gpus": world,
                            "workload": "DDPM train step, 64x64x3 (Flowers102-shaped), batch 128/GPU, drop_p=0.1, Adam"}
    m.eval()
    smp = graphs.GraphedSampler(m, 256, (64, 64), use_graph=use_graph)
    dt = timed(smp.sample, 1, 1, barrier, dist, dev)
    out["sample_64x64x3"] = {"value": 256 * world / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / 120, "batch_per_gpu": 256, "n_gpus": world,
                             "workload": "DDPM full T=120 ancestral sampling, 64x64x3, 256 images/GPU"}
    del step, smp, m, opt
    torch.cuda.empty_cache()
    # -- 128x128x3 sampling with attention at the 16x16 stage
    m = build_model(dev, 3, [2])
    m.eval()
    nb = max(32, 1024 // world)          # the configuration's job is 1024 images, sharded over the GPUs
    smp = graphs.GraphedSampler(m, nb, (128, 128), use_graph=use_graph)
    smp.begin()
    for _ in range(3):                   # warm-up: a few steps (a full chain of 1024 images takes 6 s)
        smp.step()
    dt = timed(smp.sample, 1, 0, barrier, dist, dev)
    out["sample_128x128x3_attention"] = {"value": nb * world / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / 120, "batch_per_gpu": nb,
                                         "n_gpus": world, "workload": "DDPM full T=120 ancestral sampling, 128x128x3, self-attention at the "
                                                                      f"16x16 stage (C=128, 256 tokens, 8 heads), 1024 images over {world} GPU(s)"}
    del smp, m
    torch.cuda.empty_cache()
    # -- latent diffusion on 256x256x3 images: frozen Encoder (1/4 resolution, 32 latent channels), the same UNet denoiser on the
    #    64x64x32 latents, Decoder after sampling.  A training step = encode the image batch + one denoiser step on the latents.
    from dm_amd.neural_nets.autoencoders import Decoder, Encoder
    torch.manual_seed(42 + int(os.environ.get("RANK", 0)))
    enc = Encoder(3, (32, 32), 32, 2, drop_p=0.0, activation="gelu").to(dev)
    dec = Decoder(3, 32, (32,), 2, drop_p=0.0, activation="gelu").to(dev)
    m = dm_ddpm.LatentDDPM(dm_unets.BaseUNet, enc, dec, **YAML, is_conditional=False, img_channels=32, reconstruction_method="mse",
                           latent_dim=512, device=dev).to(dev)
    m.train()
    b = 32
    images = torch.rand(b, 3, 256, 256, device=dev) * 2 - 1
    lat = torch.tanh(m.encode(images)).contiguous()
    opt = FusedAdam([q for q in m.parameters() if q.requires_grad], lr=1e-3)
    red = reducer_factory(m) if dist else None
    step = graphs.PipelinedTrainStep(m, ELBOLoss("mse"), opt, lat, red, use_graph=use_graph)

    def ldm_step():
        lat.copy_(torch.tanh(m.encode(images)))
        step.replay()

    dt = timed(ldm_step, 20, 10, barrier, dist, dev)
    out["train_ldm_256x256x3"] = {"value": b * world * 20 / dt, "unit": "images/s", "ms_per_step": 50 * dt, "batch_per_gpu": b, "n_gpus": world,
                                  "workload": "latent diffusion train step on 256x256x3 images: frozen Encoder (eager, 32 latent channels at "
                                              "64x64) + denoiser step on the latents, batch 32/GPU"}
    m.eval()
    smp = graphs.GraphedSampler(m, 32, (64, 64), use_graph=use_graph)
    dt = timed(lambda: m.decode(smp.sample()), 1, 1, barrier, dist, dev)
    out["sample_ldm_256x256x3"] = {"value": 32 * world / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / 120, "batch_per_gpu": 32, "n_gpus": world,
                                   "workload": "latent diffusion sampling of 256x256x3 images: T=120 chain on 64x64x32 latents + Decoder, "
                                               "32 images/GPU"}
    return out


def _count_gpus_sysfs():
    """GPUs of this node as the kernel driver lists them (topology nodes with compute units), or None where that listing
    does not exist.  Reading sysfs opens no device: the launcher parent stays a process that has never touched a GPU."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for f in nodes:
        try:
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            return None
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis is not None and vis.strip() != "":
        n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
    return n


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh processes of this script, one per GPU, with the
    rendezvous environment torch.distributed reads (the reference spawns one process per GPU the same way,
    train_diff.py:114-126 there).  Runs before this process has touched the GPU; the children inherit stdout, so rank 0's
    JSON line is the job's.  Returns the first non-zero exit status."""
    import socket
    import subprocess
    if n < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    have = _count_gpus_sysfs()                # (never through the HIP runtime: the parent must not open the GPU before it spawns)
    if "--dry-launch" not in argv and have is not None and n > have:
        sys.exit(f"bench.py: --gpus {n} but this node has {have} GPU(s)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live and not rc:
        time.sleep(0.05)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                rc = rc or p.returncode
    for p in live:                           # a failed rank leaves the others waiting in a collective: end exactly those children
        p.kill()
        p.wait()
    if torch.cuda.is_initialized():          # the launcher's contract: it starts ranks, it never touches a GPU itself
        print("bench.py: the launcher process initialised the GPU runtime", file=sys.stderr)
        return rc or 3
    if "--dry-launch" in argv:
        print(json.dumps({"launcher": True, "gpu_runtime_initialised": torch.cuda.is_initialized(), "gpus_seen_in_sysfs": have}), flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--sample-batch", type=int, default=1024, help="images per GPU per sampling job (BASELINE configs[3] samples 1024 images)")
    ap.add_argument("--sample-streams", type=int, default=2, help="chains of sample-batch/streams images advanced on this many streams")
    ap.add_argument("--attention-stages", default="", help="comma-separated stages that carry a self-attention ConditionalModule (other configs)")
    ap.add_argument("--activation", default="gelu", help="the network's activation (the reference's training config uses gelu; its block default is swish)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--torch-adam", action="store_true", help="use torch.optim.Adam(capturable) instead of the fused flat Adam")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one all-reduce after backward instead of the overlapped two-phase exchange")
    ap.add_argument("--phases", type=int, default=4, help="backward phases of the pipelined step: 2, 4, 5, or larger = one per stage")
    ap.add_argument("--single-graph", action="store_true", help="N = 1: the whole step as one HIP graph on one stream (through autograd)")
    ap.add_argument("--bucket-mb", type=int, default=64, help="largest single all-reduce message")
    ap.add_argument("--wire", choices=("fp32", "bf16"), default="fp32",
                    help="number format of the gradients on the links (bf16: half the bytes, one more rounding; see runtime/dp.py)")
    ap.add_argument("--no-sample", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs (64x64x3 training, 128x128x3 sampling), the "
                                                            "gradient-accumulation-3 step and the eager drop-in loop")
    ap.add_argument("--detail", default="", help="write a per-launch-shape timing table of the roofline pass to this file")
    ap.add_argument("--dry-launch", action="store_true", help="launcher check: every rank prints its rendezvous environment and exits (no GPU work)")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.dry_launch):
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", 1)) != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} disagrees with WORLD_SIZE={os.environ.get('WORLD_SIZE', 1)} "
                 "(one rank per GPU: launch with --nproc-per-node equal to --gpus, or let bench.py start the ranks itself)")
    if a.dry_launch:
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}), flush=True)
        return
    # libraries (the collective library's version banner, for one) write to stdout: keep the real stdout for the one JSON
    # line and send everything else to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # DDPM_BENCH_REHEARSE_DP=1 takes the N > 1 code path (process group, overlapped exchange with real RCCL calls) with a
    # single rank, to rehearse it on a one-GPU box; the reported line is then not a scaling result
    rehearse = world == 1 and os.environ.get("DDPM_BENCH_REHEARSE_DP") == "1"
    dist = world > 1 or rehearse
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist:
        import torch.distributed as td
        if rehearse:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            td.init_process_group("nccl", device_id=dev)

    m = build_model(dev, a.channels, [int(v) for v in a.attention_stages.split(",") if v.strip()], a.activation)
    m.train()
    crit = ELBOLoss("mse")
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=0) if not a.torch_adam else torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    x0 = (torch.rand(a.batch, a.channels, a.size, a.size, device=dev) * 2 - 1)

    if dist:
        # data parallel: full replica per GPU; the decoder half of the flat gradient buffer is all-reduced (RCCL, AVG) while
        # the encoder half of the backward pass still runs, the encoder half right after it
        from dm_amd.runtime.dp import GradReducer
        reducer = GradReducer(list(m.parameters()), 2 if rehearse else world, bucket_bytes=a.bucket_mb << 20, wire=a.wire)
        reducer.broadcast_parameters()
        if a.no_overlap:
            step = graphs.GraphedTrainStep(m, crit, opt, x0, after_backward=reducer.allreduce, use_graph=not a.no_graph)
        else:
            step = graphs.OverlappedTrainStep(m, crit, opt, x0, reducer, phases=a.phases, use_graph=not a.no_graph)
    elif a.single_graph or a.torch_adam:
        step = graphs.GraphedTrainStep(m, crit, opt, x0, use_graph=not a.no_graph)
    else:
        step = graphs.PipelinedTrainStep(m, crit, opt, x0, None, phases=a.phases, use_graph=not a.no_graph)

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step.replay()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step.replay()
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = tt.item()
    loss = step.replay().item()
    train_ips = a.batch * world * a.steps / dt

    shaped = {(32, 1): " (FashionMNIST-shaped)", (64, 3): " (Flowers102-shaped)"}.get((a.size, a.channels), "")
    out = {
        "metric": "train imgs/sec (q_sample + UNet fwd + ELBO + bwd + Adam step), sampled imgs/sec reported alongside",
        "value": train_ips, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"DDPM train step, {a.size}x{a.size}x{a.channels}{shaped}, batch {a.batch}/GPU, "
                               f"stages [32,64,128,256], T=120, drop_p=0.1, Adam (lr 1e-3), gradient_accumulation_steps=1"
                               + (f", self-attention at stages {a.attention_stages}" if a.attention_stages else "")
                               + (f", activation {a.activation}" if a.activation != "gelu" else ""),
                   "global_batch": a.batch * world, "parallelism": f"dp{world}", "hip_graph": not a.no_graph,
                   "grad_exchange": ("none" if not dist else "one all-reduce after backward" if a.no_overlap else
                                     "flat ranges of one gradient buffer: decoder half after the 2nd of 4 backward phases, the encoder's deep stages + latent after the 3rd, the small rest at the end; all but the last travel during the backward pass"),
                   "grad_wire": a.wire if dist else None},
        "final_loss": loss,
    }

    if not a.no_sample:
        m.eval()
        nb = a.sample_batch
        if a.sample_streams > 1:
            sampler = graphs.ConcurrentSampler(m, nb // a.sample_streams, (a.size, a.size), a.sample_streams, use_graph=not a.no_graph,
                                               first_image=rank * nb, seed=20240229)
        else:
            sampler = graphs.GraphedSampler(m, nb, (a.size, a.size), use_graph=not a.no_graph, first_image=rank * nb, seed=20240229)
        if dist:
            # the job's images: every rank runs its own chains, one gather to rank 0 at the end of each job
            from dm_amd.runtime.dp import sample_sharded
            run_chain = lambda: sample_sharded(lambda n, first: sampler.sample(), nb * world)      # noqa: E731
        else:
            run_chain = sampler.sample
        run_chain()
        barrier()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            run_chain()
        barrier()
        ds = (time.perf_counter() - t0) / reps
        if dist:
            tt = torch.tensor([ds], device=dev, dtype=torch.float64)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            ds = tt.item()
        out["sampling"] = {"value": nb * world / ds, "unit": "images/s", "chain_steps": 120, "batch_per_gpu": nb, "streams": a.sample_streams,
                           "ms_per_chain": 1e3 * ds, "ms_per_step": 1e3 * ds / 120}
        m.train()

    if rank == 0 and not a.no_roofline:
        from dm_amd.runtime import roofline
        rows = [] if a.detail else None
        out["roofline"] = roofline.measure(step.local_body, traffic_file=os.path.join(ROOT, "profiles", "hbm_traffic.json"), detail=rows)
        pk = os.path.join(ROOT, "profiles", "r01_peaks.json")      # what this kind of device delivered to a plain read / MFMA loop (tools/peaks)
        if os.path.exists(pk):
            with open(pk) as f:
                m_ = json.load(f)
            r_ = out["roofline"]
            r_["peak_measured"] = m_["hbm_read_GBs"] if r_["bound"] == "hbm" else m_["mfma_bf16_dense_TFLOPs"]
            r_["frac_of_measured"] = r_["achieved"] / r_["peak_measured"]
        if rows:
            with open(a.detail, "w") as f:
                for k, cnt, us, gbs, tf in sorted(rows, key=lambda r: -r[1] * r[2]):
                    f.write(f"{k:60s} x{cnt:3d} {us:8.1f} us {gbs:8.1f} GB/s {tf:7.1f} TF/s  {cnt * us:8.1f} us/step\n")
    if not a.no_extra:
        del step
        torch.cuda.empty_cache()
        if not dist:
            # the reference trainer's default: gradients of 3 micro-batches summed, optimizer step on the 3rd (Q8)
            m3 = build_model(dev, a.channels)
            m3.train()
            s3 = graphs.PipelinedTrainStep(m3, crit, FusedAdam(m3.parameters(), lr=1e-3), x0, None, phases=a.phases, use_graph=not a.no_graph,
                                           gradient_accumulation_steps=3)
            dt3 = timed(s3.replay, 30, 6, barrier, dist, dev)
            out["gradient_accumulation_3"] = {"value": a.batch * 30 / dt3, "unit": "images/s", "ms_per_micro_step": 1e3 * dt3 / 30,
                                              "what": "same workload, optimizer step on every 3rd micro-batch (the reference trainer's default)"}
            del s3, m3
            torch.cuda.empty_cache()
            out["dropin_eager"] = dropin_eager(dev, a.channels, a.size, a.batch)
        if dist:
            from dm_amd.runtime.dp import GradReducer
            def rf(mm):
                r_ = GradReducer(list(mm.parameters()), 2 if rehearse else world, bucket_bytes=a.bucket_mb << 20, wire=a.wire)
                r_.broadcast_parameters()
                return r_
        else:
            rf = None
        out["extra_configs"] = extra_configs(dev, world, dist, barrier, rf, not a.no_graph)
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(a.channels, a.size, 64, 10)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist:
        barrier()                      # rank 0 may still be timing kernels: nobody tears the group down before it is done
        td.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: VQ-IMG 256x256 images/s (recon + VQ, fwd + bwd + Adam) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one per-GPU batch of synthetic images:
``rec, q = VQBASE(x); (|x-rec|.mean() + q).backward(); Adam.step()`` on the model block of the
reference's conf/img_config.yaml (BASELINE.json configs[1]: VQ-IMG 256x256, codebook 8192, per-GPU
batch 32, bf16 activations, fp32 accumulate).  Inputs are resident in HBM before the timed region.
N > 1: one process per GPU, the reference's own data-parallel scheme (train.py:24,32: NCCL==RCCL process
group + DistributedDataParallel bucketed gradient all-reduce overlapped with backward; SyncBatchNorm's
statistics all-gather), weak scaling (per-GPU batch fixed).

Rank 0 prints ONE JSON line; besides the driver's contract it carries
  "roofline":     the dominant kernel (3x3 128->128 conv at 256^2, fwd/dgrad implicit GEMM) timed live with
                  HIP events on the launch stream inside the timed steps, against the bf16 MFMA peak;
  "cpu_baseline": the CPU oracle (oracle/vq_oracle.py, a port of the reference's arithmetic) timed on this
                  host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "make-a-scene_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

IMG_CFG = dict(ddconfig=dict(z_channels=256, in_channels=3, out_channels=3, channels=[128, 128, 128, 256, 512, 512],
                             num_res_blocks=2, resolution=512, attn_resolutions=[32], dropout=0.0),
               n_embed=8192, embed_dim=256, init_steps=3000, reservoir_size=12500)   # reference conf/img_config.yaml:19-34
PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
FWD_BWD_GFLOP_PER_IMG = 1337.53   # SURVEY.md section 8(d), counted on the reference with torch flop_counter


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=15,
                    help="untimed steps; the first ~1 s of load on a cold MI355X runs 5-8 %% slower (clock ramp), so the default covers it")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE config: 32)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dp", default="mas", choices=["mas", "ddp"],
                    help="N>1 gradient averaging: mas_hip.dp.GradReducer (default) or torch DistributedDataParallel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-batch", type=int, default=1)
    return ap.parse_args()


def _cpu_baseline_worker(batch, threads):
    """child process: prints the seconds of one timed CPU-oracle fwd+bwd step"""
    from oracle import vq_oracle as O
    torch.set_num_threads(threads)
    sd = O.synth_state_dict(IMG_CFG["ddconfig"], IMG_CFG["n_embed"], IMG_CFG["embed_dim"], seed=0)
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    x = O.synth_image_batch(batch, 3, 256, seed=0)

    def step():
        for v in sd.values():
            v.grad = None
        dec, qq, _, _ = O.vqbase_forward(sd, x, IMG_CFG["ddconfig"], training=True)
        O.recon_vq_loss(x, dec, qq).backward()

    step()                                      # warm-up (allocator, oneDNN primitive cache)
    t0 = time.perf_counter()
    step()
    print("CPU_BASELINE_SECONDS", time.perf_counter() - t0, flush=True)


def cpu_baseline(batch, budget_s=90):
    """Times the CPU oracle (fp32 port of the reference's arithmetic) on a BOUNDED sample: `batch` images, 1 warm-up +
    1 timed fwd+bwd step, in a child process that is killed after `budget_s` seconds (the host of a GPU box can have
    hundreds of slow hardware threads: round 1 measured 277 s/step with all 256).  A reported baseline, not the target."""
    import subprocess
    threads = min(os.cpu_count() or 1, 32)
    sample = f"oracle/vq_oracle.py fp32 fwd+bwd, B={batch} x 256x256, 1 warm-up + 1 timed step, {threads} threads, torch CPU {torch.__version__}"
    out = {"value": None, "unit": "images/s", "cores": threads, "kind": "port", "sample": sample}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(batch), str(threads)],
                           capture_output=True, text=True, timeout=budget_s)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE_SECONDS"):
                out["value"] = round(batch / float(line.split()[1]), 4)
        if out["value"] is None:
            out["sample"] += " -- worker failed: " + r.stderr[-200:]
    except subprocess.TimeoutExpired:
        out["sample"] += f" -- exceeded the {budget_s}s budget"
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no CPU fallback for the product path)", file=sys.stderr)
        sys.exit(2)
    # MAS_BENCH_SHARE_GPU=1 + MAS_BENCH_BACKEND=gloo: functional check of the N>1 code path (reducer, SyncBatchNorm
    # exchange, max-over-ranks timing) with several ranks on ONE GPU -- RCCL refuses two ranks per device, gloo does not
    if os.environ.get("MAS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("MAS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ddp = world > 1 or os.environ.get("MAS_BENCH_FORCE_DDP") == "1"     # the env knob exercises the N>1 code path on one GPU
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=dev)
        else:
            dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank)

    from mas_hip import ops
    from models import VQBASE
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    ops.set_compute_dtype(cdt)

    torch.manual_seed(0)                                   # identical replicas (DDP would broadcast anyway)
    model = VQBASE(**IMG_CFG)
    with torch.no_grad():                                  # post-k-means-like codebook scale
        model.quantize.embedding.weight.normal_(0.0, 1.0)
    model = model.to(dev).train()
    model.quantize.q_counter = model.quantize.q_re_end      # steady state: VQ lookup on the path, no warm-up bypass
    net, reducer = model, None
    if ddp and args.dp == "ddp":                            # the reference's wrapper (train.py:31-34)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], gradient_as_bucket_view=True)
    elif ddp:                                               # same semantics, 3 flat buckets instead of 345 per-parameter copies
        from mas_hip.dp import GradReducer
        reducer = GradReducer(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=5e-6, betas=(0.5, 0.9), fused=True)   # conf/img_config.yaml:36-41

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)          # distinct data per rank
    x = torch.rand(args.batch, 3, 256, 256, generator=g).to(dev)

    # ---- live timing of the dominant kernel (HIP events on the launch stream) -------------
    dom = {"events": [], "on": False}

    def hook(kind, shape, launch):
        # shape = (n,h,w,cin,ho,wo,cout,ks,stride)
        if dom["on"] and kind == "conv_fwd" and shape[3] == 128 and shape[6] == 128 and shape[7] == 3 and shape[8] == 1 \
                and shape[4] == 256 and shape[5] == 256 and shape[1] == 256:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            dom["events"].append((e0, e1))
        else:
            launch()

    ops.set_launch_hook(hook)

    def step():
        rec, q = net(x)
        loss = (x - rec).abs().mean() + q
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    dom["on"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dom["on"] = False
    ops.set_launch_hook(None)
    if ddp:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss)
    spread = None
    if ddp:                                                 # replicas must still hold identical weights after the timed steps
        with torch.no_grad():
            cs = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = float((hi - lo).item())

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        out = {
            "metric": "VQ-IMG 256x256 images/sec/node (recon+VQ fwd+bwd)", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "VQ-IMG 256x256, codebook 8192x256, conf/img_config.yaml model block (95.2 M params), "
                                   "fwd+bwd of L1+q_loss + Adam step", "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" + ((" (DistributedDataParallel" if args.dp == "ddp" else " (mas_hip.dp.GradReducer: 128 MiB flat buckets,")
                                                       + " RCCL all-reduce overlapped with backward + SyncBatchNorm)" if ddp else "")},
            "final_loss": round(final_loss, 5),
            "replica_weight_checksum_spread": spread,
            "model_tflops_per_gpu": round(value / world * FWD_BWD_GFLOP_PER_IMG / 1e3, 1),
        }
        if dom["events"]:
            ms = [a.elapsed_time(b) for a, b in dom["events"]]
            avg_ms = sum(ms) / len(ms)
            px = args.batch * 256 * 256
            flops = 2.0 * 9 * 128 * 128 * px                 # algorithmic FLOPs of one launch
            esz = 2 if args.dtype == "bf16" else 4
            bytes_ = 2.0 * px * 128 * esz                    # one read of the input, one write of the output
            ach = flops / (avg_ms * 1e-3) / 1e12
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
            out["roofline"] = {"kernel": "conv_fwd_kernel<3x3,s1,128->128> @256x256 (fwd + dgrad launches)", "bound": "mfma",
                               "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "traffic": _pmc_traffic(), "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(ms),
                               "algorithmic_gflop_per_launch": round(flops / 1e9, 1),
                               "algorithmic_hbm_gbs": round(bytes_ / (avg_ms * 1e-3) / 1e9, 1),
                               "hbm_frac": round(bytes_ / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        out["encoder_stack"] = _encoder_stack(model, x, args.batch, args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_batch)
        print(json.dumps(out), flush=True)
    if ddp:
        dist.destroy_process_group()


def _encoder_stack(model, x, batch, dtype):
    """The north_star's own target line: the fused conv+GN+SiLU encoder stack (forward) against the
    HBM and MFMA rooflines, from SURVEY.md section 8(d)'s per-image figures (152.34 Melem of
    algorithmic traffic, 152.19 GFLOP).  Timed after the bench region, outside `value`."""
    esz = 2 if dtype == "bf16" else 4
    with torch.no_grad():
        for _ in range(2):
            model.encoder(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            model.encoder(x)
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = 152.34e6 * esz * batch / (ms * 1e-3) / 1e9
    tfs = 152.19e9 * batch / (ms * 1e-3) / 1e12
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else 157.3
    return {"what": "Encoder.forward, all 23 layers, batch %d" % batch, "fwd_ms": round(ms, 3),
            "algorithmic_hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
            "tflops": round(tfs, 1), "mfma_frac": round(tfs / peak, 4), "binding_bound": "mfma",
            "north_star_target_hbm_frac": 0.40}


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary, if any."""
    p = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    try:
        with open(p) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:
        return None


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()

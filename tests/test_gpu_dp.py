"""Row (e) on the real HIP path: two data-parallel ranks (gloo -- RCCL refuses two ranks on one device -- both on cuda:0),
each running the tiny VQBASE on its half of the batch with torch's SyncBatchNorm exchange and mas_hip.dp.GradReducer,
must reproduce the single-process full-batch loss and gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = dict(ddconfig=dict(z_channels=32, in_channels=3, out_channels=3, channels=[32, 32, 64, 64],
                          num_res_blocks=1, resolution=32, attn_resolutions=[8], dropout=0.0),
            n_embed=64, embed_dim=32, init_steps=3000, reservoir_size=12500)


def _model():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "make-a-scene_amd"))
    from models import VQBASE
    from mas_hip import ops
    from oracle.vq_oracle import synth_state_dict, synth_image_batch
    ops.set_compute_dtype(torch.float32)
    m = VQBASE(**TINY)
    m.load_state_dict(synth_state_dict(TINY["ddconfig"], TINY["n_embed"], TINY["embed_dim"], seed=0), strict=True)
    m = m.to("cuda:0").train()
    m.quantize.q_counter = m.quantize.q_re_end
    return m, synth_image_batch(4, 3, 32, seed=0).to("cuda:0")


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, x = _model()
    from mas_hip.dp import GradReducer
    red = GradReducer(m.parameters(), bucket_bytes=256 << 10)
    xs = x[rank * 2:(rank + 1) * 2]
    rec, q = m(xs)
    loss = (xs - rec).abs().mean() + q
    loss.backward()
    red.finish()
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    if rank == 0:
        np.savez(out, loss=float(lsum) / world, nbuckets=len(red.buckets),
                 **{k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_full_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp_gpu.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["nbuckets"]) >= 2
    m, x = _model()
    rec, q = m(x)                                    # no process group here: SyncBatchNorm == BatchNorm over the full batch
    loss = (x - rec).abs().mean() + q
    loss.backward()
    assert abs(float(loss) - float(got["loss"])) < 1e-4 * abs(float(loss))
    checked = 0
    for k, p in m.named_parameters():
        if p.grad is not None:
            ref = p.grad.detach().cpu().numpy()
            assert np.abs(got[k] - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-6, k
            checked += 1
    assert checked > 40


def _ddp_worker(rank, world, port, out):
    """the reference's own wrapper (train.py:32: DistributedDataParallel(model, device_ids=[device]))"""
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, x = _model()
    if rank == 1:                                     # DDP must broadcast rank 0's parameters at construction
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0])
        xs = x[rank * 2:(rank + 1) * 2]
        res = {}
        for step in range(2):                         # second step: bucket views rebuilt, gradients accumulate into DDP's buckets
            net.zero_grad(set_to_none=(step == 0))
            rec, q = net(xs)
            loss = (xs - rec).abs().mean() + q
            loss.backward()
            torch.cuda.synchronize()
        lsum = loss.detach().clone()
        dist.all_reduce(lsum)
    stride_warn = [str(w.message) for w in caught if "strides" in str(w.message).lower()]
    if rank == 0:
        np.savez(out, loss=float(lsum) / world, n_stride_warnings=len(stride_warn),
                 **{k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_wrapper_two_ranks_equal_full_batch(tmp_path):
    """`DistributedDataParallel` over our modules: rank-sharded fwd+bwd (torch SyncBatchNorm exchange, DDP's bucketed
    gradient average) == single-process full batch, and no gradient reaches DDP with strides that differ from its bucket
    view (the 1x1-conv weight gradients used to: VERDICT r1 weak #3)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "ddp_gpu.npz")
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["n_stride_warnings"]) == 0
    m, x = _model()
    rec, q = m(x)
    loss = (x - rec).abs().mean() + q
    loss.backward()
    bad = [k for k, p in m.named_parameters() if p.grad is not None and p.grad.stride() != p.stride()]
    assert not bad, bad                                # what DDP compares against its bucket views
    assert abs(float(loss) - float(got["loss"])) < 1e-4 * abs(float(loss))
    checked = 0
    for k, p in m.named_parameters():
        if p.grad is not None:
            ref = p.grad.detach().cpu().numpy()
            assert np.abs(got[k] - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-6, k
            checked += 1
    assert checked > 40


def test_bench_py_multi_rank_branch_end_to_end():
    """bench.py's N > 1 branch (process group, GradReducer, SyncBatchNorm exchange, barrier + max-over-ranks timing, replica
    checksum) run end to end by pytest: two ranks share the one GPU over gloo (RCCL refuses two ranks per device; the 8-GPU RCCL
    run is the driver's).  Small batch, 1 + 1 steps."""
    import json
    import socket
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MAS_BENCH_SHARE_GPU="1", MAS_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                                       "--batch", "2", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-1500:] + outs[1][1][-1500:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 4 and j["scaling"] == "weak" and j["value"] > 0
    assert j["replica_weight_checksum_spread"] == 0.0                  # both replicas applied the same averaged gradients
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]   # only rank 0 prints


def test_bench_py_eight_ranks_through_the_drivers_launcher():
    """the command line the driver uses for N = 8 -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...`` -- as a dry run: eight ranks share the one GPU over gloo
    (MAS_BENCH_SHARE_GPU / MAS_BENCH_BACKEND), so that the first real 8-GPU RCCL run is not also the first 8-rank run of the
    argument / rendezvous / reducer-bucket / SyncBatchNorm / max-over-ranks plumbing.  Checked: the line's shape (n_gpus 8, global batch
    8 x per-GPU batch, weak scaling), identical replicas after the steps (checksum spread 0), one JSON line (rank 0 only)."""
    import json
    import socket
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MAS_BENCH_SHARE_GPU="1", MAS_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 16 and j["config"]["per_gpu_batch"] == 2 and j["scaling"] == "weak"
    assert j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0 and abs(j["value"] - 16 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-2 * j["value"]
    assert j["replica_weight_checksum_spread"] == 0.0
    assert "dp8" in j["config"]["parallelism"]


def test_bench_py_multi_rank_gradient_accumulation_no_sync():
    """the accumulation branch of the N > 1 path -- GradReducer.no_sync() on all but the last micro-batch, as the
    reference accumulates (conf/img_config.yaml:13) -- end to end through `bench.py --workload e2e` (frozen VQ encode -> transformer,
    micro-batches of 1 accumulated twice per step), two gloo ranks on the one GPU.  Both replicas must end with identical weights
    (checksum spread 0): a reduction fired per micro-batch, or skipped on the last one, would break that or the averaging."""
    import json
    import socket
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MAS_BENCH_SHARE_GPU="1", MAS_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--workload", "e2e", "--gpus", "2", "--steps", "1",
                                       "--warmup", "1", "--batch", "2", "--micro-batch", "1"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-1500:] + outs[1][1][-1500:]
    j = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 4 and j["value"] > 0
    assert "accumulation" in j["config"]["workload"]
    assert j["replica_weight_checksum_spread"] == 0.0

"""bench.py --gpus N launches its own N ranks: on a box with >= 2 GPUs over RCCL
("nccl"), otherwise the one-device self-test (all ranks on cuda:0, rendezvous over gloo).  Also the
sharded module's gather over the real backend when two devices are there."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks():
    env = dict(os.environ, RROI_BENCH_E2E="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["RROI_BENCH_ONE_DEVICE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["warmup"] == 5 and r["scaling"] == "weak"
    assert r["config"]["rois_total"] == 1024 and r["config"]["rois_per_gpu"] == 512
    assert r["value"] > 0 and abs(r["value"] - 1024 / (r["ms_per_step"] * 1e-3)) < 1e-3 * r["value"]
    ex = r["extra"]
    assert "with_gather_ms" in ex and "with_allreduce_grad_ms" in ex
    # the line verifies itself: every rank reports what it ran on and its own time
    assert len(ex["ranks"]) == 2 and [q["rank"] for q in ex["ranks"]] == [0, 1]
    for q in ex["ranks"]:
        assert q["arch"].startswith("gfx") and q["uuid"] and q["ms_per_step"] > 0
    assert max(q["ms_per_step"] for q in ex["ranks"]) <= r["ms_per_step"] * 1.001
    assert ex["with_allreduce_grad_ms"] is not None and ex["with_allreduce_grad_ms"] > 0, ex.get("with_allreduce_grad")
    if torch.cuda.device_count() >= 2:
        assert ex["with_gather_ms"] > r["ms_per_step"]
        assert len({q["uuid"] for q in ex["ranks"]}) == 2


def test_bench_refuses_more_gpus_than_visible_without_the_self_test_switch():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RROI_BENCH_ONE_DEVICE")}
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "visible" in (p.stderr + p.stdout)


def _nccl_worker(rank, world, port, q):
    import numpy as np
    import torch.distributed as dist
    sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
    import workloads as Wk
    from rroi_align.sharded import ShardedRRoiAlign
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        f, r = Wk.bench_inputs(R=64, C=32, seed=4)
        feats = torch.from_numpy(f).cuda().requires_grad_(True)
        crops = ShardedRRoiAlign(8, 64, 0.25, gather=True)(feats, torch.from_numpy(r).cuda())
        crops.sum().backward()
        q.put((rank, crops.detach().cpu().numpy(), feats.grad.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_sharded_gather_over_rccl(oracle):
    import numpy as np
    import torch.multiprocessing as mp
    import workloads as Wk
    from test_sharded import _free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    f, r = Wk.bench_inputs(R=64, C=32, seed=4)
    want = oracle.forward_c(f, r, 8, 64, 0.25, threads=8)
    for rank, crops, g in res:
        assert np.array_equal(crops, want)            # rows back in global order, bit-identical
        lo, hi = (0, 32) if rank == 0 else (32, 64)
        gw = oracle.backward_c(np.full_like(want[lo:hi], 2.0), r[lo:hi], f.shape, 0.25)  # both ranks' losses see every crop
        assert np.abs(g - gw).max() <= 1e-4 * max(1.0, np.abs(gw).max())

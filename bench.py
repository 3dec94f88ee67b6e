#!/usr/bin/env python3
"""bench.py -- RoIRotate forward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one forward pass of the hot path (prologue relayout + gather kernel)
over one batch of synthetic input already resident in HBM:  BASELINE.json
configs[1]  -- features 1x256x160x160 fp32 (NCHW, the reference contract), 512
random rotated ROIs, pooled 8x64, spatial_scale 0.25.  With N > 1 the ROIs of a
512*N set are row-sharded 512 per rank with no data-path collective (configs[3],
weak scaling); the time is the max over ranks.  N > 1 runs either under
torch.distributed.run (one rank per GPU, RANK/LOCAL_RANK/WORLD_SIZE in the
environment) or, when those are absent, spawns its N ranks itself.

Prints ONE JSON line on rank 0.
  value            ROIs/s of the whole job: K steps after W warm-up steps, wall clock between
                   barrier + synchronize on both sides, max over ranks.
  roofline         the dominant kernel (rroi_fwd_split_kernel): algorithmic bytes of one launch /
                   its average duration INSIDE the step (`kernel_ms.avg` = the whole call minus the
                   prologue, each between two HIP events on the launch stream over fixed loops that
                   do not depend on --steps/--warmup and run BEFORE the timed steps, on a chip
                   brought up to speed).  `kernel_ms.alone` / `frac_alone`: the same kernel launched
                   back to back by itself (300 warm-up + 500 timed).  `whole_call_frac`: the same
                   bytes over `ms_per_step` (both launches); `whole_call_ms_spread`: median / p10 /
                   p90 over 200 individually bracketed calls.
  cpu_baseline     the oracle (a C port of the reference's per-element semantics; the reference has
                   no runnable CPU path) timed on this host's cores -- a reported baseline, not the
                   thing measured.
  extra            `sensitivity`: the call with every bin active and with axis-aligned ROIs,
                   next to a control run of the default draw in the same loop; `two_calls_in_flight`: the same call
                   with consecutive calls alternating between two streams (round 6; not `value`);
                   N > 1: the step followed by the RCCL all_gather of the crops into one
                   preallocated (512*N, 256, 8, 64) buffer (`with_gather_ms`) and the 25 MiB all_reduce
                   of the feature gradient (`with_allreduce_grad_ms`), reported beside the kernel-only
                   step as SURVEY 8(e) asks; `ranks`: what every rank ran on (device UUID, architecture)
                   and its own ms_per_step -- two ranks on one device fail the run (outside the
                   one-device self-test).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import tempfile
import sys
import time

import numpy as np
import torch

from bench_legs import beyond_cache, mixed_layout, train_regime, train_step, two_calls_in_flight

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fots.pytorch_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_WARM, KERNEL_TIMED = 300, 500  # the roofline loop, fixed
TOUCHED_PIXELS_RANK0 = 25278  # distinct map pixels the 512 seeded ROIs of configs[1] read (oracle.touched_pixels)

CFG = dict(R=512, C=256, H=160, W=160, img=640, PH=8, PW=64, scale=0.25)


def make_inputs(R, C=256, H=160, W=160, img=640, seed=0):
    """features ~ N(0,1); cx,cy ~ U[0,img), h ~ U[16,64), w = h*U[4,8),
    angle ~ U[-90,90) degrees, batch index 0."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((1, C, H, W), dtype=np.float32)
    cx = rng.uniform(0, img, R)
    cy = rng.uniform(0, img, R)
    h = rng.uniform(16, 64, R)
    w = h * rng.uniform(4, 8, R)
    ang = rng.uniform(-90, 90, R)
    rois = np.stack([np.zeros(R), cx, cy, h, w, ang], 1).astype(np.float32)
    return feats, rois


def cpu_baseline(feats, rois):
    """Oracle timed on the host (rank 0, N = 1 only).  Imports oracle/ -- allowed here only."""
    sys.path.insert(0, ROOT)
    from oracle import rroi_align_oracle as O
    c = CFG
    from fots_e2e.hostcpus import effective_cpus
    # OpenMP's default is the machine's hardware threads; the cgroup quota is what this process gets
    cores = max(1, min(O.max_threads(), effective_cpus()))
    # bounded sample: 128 of the 512 ROIs on 1 thread (~1 s) and the full workload 6x on all cores
    t0 = time.perf_counter()
    O.forward_c(feats, rois[:128], c["PH"], c["PW"], c["scale"], threads=1)
    t1 = (time.perf_counter() - t0) * 4.0  # 128 of 512 ROIs
    best = float("inf")
    for _ in range(6):
        t0 = time.perf_counter()
        O.forward_c(feats, rois, c["PH"], c["PW"], c["scale"], threads=cores)
        best = min(best, time.perf_counter() - t0)
    # the reference's own structure (one "thread" per output element, geometry recomputed for
    # every channel, kernel.cu:28-162), single thread, 4 of the 512 ROIs
    t0 = time.perf_counter()
    O.forward_literal_c(feats, rois[:4], c["PH"], c["PW"], c["scale"])
    t_lit = (time.perf_counter() - t0) / 4.0
    touched = O.touched_pixels(rois, 1, c["H"], c["W"], c["PH"], c["PW"], c["scale"])
    # configs[2], BASELINE.md section 2 last row: the backward of the same shapes (oracle, hoisted, a thread owns
    # a set of channels; double accumulation), single thread and all cores, full workload
    gout = np.random.default_rng(1).standard_normal((len(rois), c["C"], c["PH"], c["PW"]), dtype=np.float32)
    t0 = time.perf_counter()
    O.backward_c(gout, rois, feats.shape, c["scale"], threads=1)
    tb1 = time.perf_counter() - t0
    tbn = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        O.backward_c(gout, rois, feats.shape, c["scale"], threads=cores)
        tbn = min(tbn, time.perf_counter() - t0)
    del gout
    backward = {"ms_per_call": round(tbn * 1e3, 2), "cores": cores, "ms_per_call_single_thread": round(tb1 * 1e3, 2),
                "unit": "ms", "kind": "port",
                "sample": "full workload (grad_out 512 x 256 x 8 x 64 -> grad_in 1 x 256 x 160 x 160), oracle "
                          "rroi_oracle_backward_mt (kernel.cu:193-278 semantics, geometry once per bin, channels dealt to "
                          "the threads, double accumulation), best of 3 on all cores / one run on one thread"}
    return {
        "value": round(len(rois) / best, 1), "unit": "ROIs/s", "cores": cores, "kind": "port",
        "sample": "full workload (512 ROIs x 256 ch x 8x64), oracle/rroi_align_oracle.c hoisted "
                  "forward, OpenMP over ROIs, best of 6; single-thread (128-ROI sample x4): "
                  "%.1f ROIs/s; literal per-element form of the reference kernel, single thread "
                  "(4-ROI sample): %.1f ROIs/s" % (len(rois) / t1, 1.0 / t_lit),
        "ms_per_step": round(best * 1e3, 2),
        "backward": backward,
    }, touched


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawned(local_rank, world, port, args):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    run(args)


def live_traffic(mode, timeout_s=150):
    """HBM-side bytes per launch of every rroi_* kernel of tools/traffic_probe.py (`mode`: "forward" = the configs[1]
    call, "backward" = the configs[2] call), measured NOW under `rocprofv3 --kernel-trace --pmc <one counter>` for
    FETCH_SIZE and WRITE_SIZE in separate passes (never with another trace domain), as MI355X_MICROARCH.md's HBM section
    prescribes: both counters are in KiB, FETCH_SIZE is doubled on gfx950.  Returns {short kernel name: {"FETCH_SIZE",
    "WRITE_SIZE" (KiB per launch), "us" (average duration in the counter passes), "launches"}} or None when rocprofv3 is
    absent, when this process is itself being profiled, when RROI_BENCH_TRAFFIC=0, or when anything about a pass fails
    -- the caller then falls back to the round's committed passes under profiles/."""
    if os.environ.get("RROI_BENCH_TRAFFIC", "1") == "0" or shutil.which("rocprofv3") is None:
        return None
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    import csv
    import glob

    def short(name):
        return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]

    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rroi_traffic_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--",
                            sys.executable, os.path.join(ROOT, "tools", "traffic_probe.py"), mode],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            per_dispatch, span = {}, {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as fh:
                    for row in csv.DictReader(fh):
                        k = short(row["Kernel_Name"])
                        if k.startswith("rroi_") and row["Counter_Name"] == counter:
                            key = (k, path, row["Dispatch_Id"])
                            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
                            span[key] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
            if not per_dispatch:
                return None
            for k in {key[0] for key in per_dispatch}:
                vals = [v for key, v in per_dispatch.items() if key[0] == k]
                us = [v for key, v in span.items() if key[0] == k]
                e = res.setdefault(k, {})
                e[counter] = sum(vals) / len(vals)
                e["launches"] = len(vals)
                e["us"] = round(min(e.get("us", 1e30), sum(us) / len(us)), 1)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if any("FETCH_SIZE" not in e or "WRITE_SIZE" not in e for e in res.values()):
        return None
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults sized so that the run ends in well under a minute; the driver passes its own K / W
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" in os.environ:          # launched by torch.distributed.run: one rank per GPU
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            ap.error("--gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]))
        run(args)
    elif args.gpus == 1:
        run(args)
    else:                                    # self-launch: N ranks of this script, one per GPU
        one_device = os.environ.get("RROI_BENCH_ONE_DEVICE") == "1"
        if torch.cuda.device_count() < args.gpus and not one_device:
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible (RROI_BENCH_ONE_DEVICE=1 runs the "
                     "multi-rank path on one device as a self-test)" % (args.gpus, torch.cuda.device_count()))
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(args.gpus, _free_port(), args), nprocs=args.gpus, join=True)


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # RROI_BENCH_ONE_DEVICE=1 (self-test only): all ranks share cuda:0 and rendezvous over gloo, so
    # the multi-rank code path can be exercised on a one-GPU box.  Never set by the driver.
    one_device = os.environ.get("RROI_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # the ranks of a node share its CPU quota: keep every rank's intra-op pool inside its share
    from fots_e2e.hostcpus import effective_cpus
    torch.set_num_threads(max(1, min(torch.get_num_threads(), effective_cpus() // world)))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    from rroi_align._ext import rroi_align as ext  # fails loudly if the HIP library is missing
    from rroi_align.sharded import gather_crops, shard_bounds

    c = CFG
    feats_np, rois_all = make_inputs(c["R"] * world, c["C"], c["H"], c["W"], c["img"])
    lo, hi = shard_bounds(len(rois_all), world, rank)
    rois_np = rois_all[lo:hi]
    feats = torch.from_numpy(feats_np).to(dev)
    rois = torch.from_numpy(rois_np).to(dev)
    R = rois.shape[0]

    # pre-allocated output and workspace; inputs resident in HBM before timing
    out = torch.empty((R, c["C"], c["PH"], c["PW"]), dtype=torch.float32, device=dev)
    nbytes = ext._lib.rroi_align_forward_workspace_bytes(1, c["C"], c["H"], c["W"], R, ext.LAYOUT_NCHW)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def launch(stages):
        st = ext._lib.rroi_align_forward_stages_hip(
            feats.data_ptr(), ext.LAYOUT_NCHW, c["scale"], 1, R, c["H"], c["W"], c["C"], c["PH"],
            c["PW"], rois.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, ext.PATH_TILED, stages,
            stream)
        if st != 1:
            raise RuntimeError(f"rroi_align_forward_stages_hip -> {st}")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def event_loop(fn, warm, timed):
        """average duration of `timed` back-to-back calls of fn between two events on the launch stream"""
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(timed):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / timed

    # ---- roofline of the dominant kernel: its own fixed loop ------------------------------------------
    # The chip comes up to speed on plain elementwise traffic first (~60 ms of read + write of non-zero data), so
    # that a profiler's per-kernel average over EVERY launch of this process (profiles/*_bench_kernel_stats.csv) is
    # not an average over the ramp.  The ramp is not the shader clock alone: after idle time, or after a stretch
    # of zero fills, this kernel needs a few hundred launches to settle from ~48-50 to ~44.5 us (rocprofv3 kernel
    # trace, launch by launch) -- which is also why the zero-fill calibration below runs LAST.
    warm_src = torch.randn(out.numel() // 4, dtype=torch.float32, device=dev)
    warm_dst = torch.empty_like(warm_src)
    for _ in range(int(os.environ.get("RROI_BENCH_WARM", "1200"))):
        torch.mul(warm_src, 1.0001, out=warm_dst)
    del warm_src, warm_dst
    launch(ext.STAGE_ALL)
    gather_ms = event_loop(lambda: launch(ext.STAGE_GATHER), KERNEL_WARM, KERNEL_TIMED)
    prologue_ms = event_loop(lambda: launch(ext.STAGE_PROLOGUE), 100, 200)
    # the whole call by the same clock (two HIP events around back-to-back calls): a second, independent
    # reading of the step next to the wall-clock one below, and the state the timed region starts from
    step_events_ms = event_loop(lambda: launch(ext.STAGE_ALL), 100, 300)

    # The first short burst after a long back-to-back run pays for the runtime's housekeeping of the thousands
    # of completed launches (measured in a probe of the same shape: first 20-step segment 60-62 us per step, every later one
    # 53.7-54.3): two untimed bursts with a synchronisation each settle it before W and K below.
    for _ in range(2):
        for _ in range(32):
            launch(ext.STAGE_ALL)
        torch.cuda.synchronize()

    # ---- the metric: W untimed warm-up steps, then exactly K timed steps --------------------------
    for _ in range(args.warmup):
        launch(ext.STAGE_ALL)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch(ext.STAGE_ALL)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1, on the side: the step followed by the all_gather of the crops -------------------
    with_gather_ms = gather_err = None
    if world > 1:
        try:
            full = torch.empty((R * world, c["C"], c["PH"], c["PW"]), dtype=torch.float32, device=dev)

            def step_and_gather():
                launch(ext.STAGE_ALL)
                gather_crops(out, R * world, out=full)
            with torch.no_grad():
                for _ in range(3):
                    step_and_gather()
                barrier()
                t0 = time.perf_counter()
                n_g = max(1, min(args.steps, 20))
                for _ in range(n_g):
                    step_and_gather()
                barrier()
                tg = torch.tensor([(time.perf_counter() - t0) / n_g * 1e3], dtype=torch.float64,
                                  device="cpu" if one_device else dev)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            with_gather_ms = float(tg.item())
            del full
        except Exception as e:  # a backend that cannot gather device tensors (self-test over gloo)
            gather_err = repr(e)[:200]

    # ---- N > 1: the feature gradient's all_reduce (25 MiB) and what every rank ran on -----
    with_allreduce_ms = allreduce_err = ranks_info = None
    if world > 1:
        try:
            grad = torch.zeros((1, c["C"], c["H"], c["W"]), dtype=torch.float32, device="cpu" if one_device else dev)
            for _ in range(3):
                dist.all_reduce(grad)
            barrier()
            t0 = time.perf_counter()
            n_a = max(1, min(args.steps, 20))
            for _ in range(n_a):
                launch(ext.STAGE_ALL)
                dist.all_reduce(grad)
            barrier()
            ta = torch.tensor([(time.perf_counter() - t0) / n_a * 1e3], dtype=torch.float64,
                              device="cpu" if one_device else dev)
            dist.all_reduce(ta, op=dist.ReduceOp.MAX)
            with_allreduce_ms = float(ta.item())
            del grad
        except Exception as e:
            allreduce_err = repr(e)[:200]
        prop = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local, "device": torch.cuda.current_device(),
                "uuid": str(getattr(prop, "uuid", "")), "arch": getattr(prop, "gcnArchName", ""),
                "name": prop.name, "ms_per_step": round(elapsed_local / args.steps * 1e3, 5)}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        uuids = [r["uuid"] for r in ranks_info]
        if not one_device and len(set(uuids)) != world:
            raise SystemExit("bench.py: %d ranks but the devices are %s -- two ranks share a GPU" % (world, uuids))

    # spread of the call (median / p10 / p90 over individually bracketed calls; an event between
    # two launches costs ~1-2 us of its own, so these sit above the back-to-back average) and the two
    # sensitivity points -- every bin active (w / h = 8) and axis-aligned ROIs (angle = 0)
    step_stats = sensitivity = None
    # RROI_BENCH_SENSITIVITY=0: tools/profile_round.sh's kernel-stats pass, so that the profiler's per-kernel
    # average is over launches of the BASELINE workload only (no sensitivity points, no channels-last call)
    if world == 1 and os.environ.get("RROI_BENCH_SENSITIVITY", "1") == "1":
        sensitivity = {}
        keep = rois.clone()
        for name, edit in (("default_draw", lambda r: None), ("all_active", lambda r: r[:, 4].copy_(r[:, 3] * 8.0)),
                           ("axis_aligned", lambda r: r[:, 5].zero_())):
            rois.copy_(keep)
            edit(rois)
            # a change of the ROI set is followed by a transient of a few hundred launches (seen up to 70 us per
            # call decaying to 57; tools/sensitivity_probe.py): warm up in chunks until two agree within 1 %
            prev = event_loop(lambda: launch(ext.STAGE_ALL), 50, 100)
            for _ in range(20):
                cur = event_loop(lambda: launch(ext.STAGE_ALL), 0, 100)
                settled = abs(cur - prev) <= 0.01 * prev
                prev = cur
                if settled:
                    break
            ms = event_loop(lambda: launch(ext.STAGE_ALL), 0, 200)
            sensitivity[name] = {"ms_per_call": round(ms, 5), "ROIs/s": round(R / (ms * 1e-3), 1)}
        sensitivity["what"] = ("the same call, same loop: the default draw (control), w = 8 h for every ROI (no masked "
                               "bins), angle = 0 (rows of a crop are rows of the map); not part of `value`")
        rois.copy_(keep)
        del keep

    # on the side (not part of `value`): the same call when the producer hands over channels-last
    # features -- consumed in place, no relayout
    nhwc_ms = None
    if world == 1 and os.environ.get("RROI_BENCH_SENSITIVITY", "1") == "1":
        feats_cl = feats.contiguous(memory_format=torch.channels_last)  # storage (B, H, W, C)
        nb_cl = ext._lib.rroi_align_forward_workspace_bytes(1, c["C"], c["H"], c["W"], R, ext.LAYOUT_NHWC)
        ws_cl = torch.empty(max(nb_cl, 1), dtype=torch.uint8, device=dev)

        def fwd_cl():
            st = ext._lib.rroi_align_forward_hip(feats_cl.data_ptr(), ext.LAYOUT_NHWC, c["scale"], 1, R, c["H"],
                                                 c["W"], c["C"], c["PH"], c["PW"], rois.data_ptr(),
                                                 out.data_ptr(), ws_cl.data_ptr(), nb_cl, ext.PATH_TILED, stream)
            if st != 1:
                raise RuntimeError(f"rroi_align_forward_hip(NHWC) -> {st}")
        nhwc_ms = event_loop(fwd_cl, 100, 300)
        launch(ext.STAGE_ALL)  # leave the NCHW result in `out`
        torch.cuda.synchronize()
        del feats_cl, ws_cl

    # configs[2] on the side (not part of `value`): backward w.r.t. the features, same shapes
    bwd_ms = bwd_cl_ms = None
    if world == 1:
        gout = torch.randn_like(out)
        nb_b = ext._lib.rroi_align_backward_workspace_bytes(1, c["C"], c["H"], c["W"], R, c["PH"], c["PW"])
        ws_b = torch.empty(nb_b, dtype=torch.uint8, device=dev)
        gin = torch.empty((1, c["C"], c["H"], c["W"]), dtype=torch.float32, device=dev)

        def bwd():
            st = ext._lib.rroi_align_backward_hip(gout.data_ptr(), c["scale"], 1, R, c["H"], c["W"], c["C"],
                                                  c["PH"], c["PW"], rois.data_ptr(), gin.data_ptr(),
                                                  ws_b.data_ptr(), nb_b, ext.PATH_TILED, stream)
            if st != 1:
                raise RuntimeError(f"rroi_align_backward_hip -> {st}")
        bwd_ms = event_loop(bwd, 10, 50)
        gout_cl = gout.contiguous(memory_format=torch.channels_last)  # storage (R, PH, PW, C)

        def bwd_cl():
            st = ext._lib.rroi_align_backward_layout_hip(gout_cl.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NHWC, c["scale"], 1, R,
                                                         c["H"], c["W"], c["C"], c["PH"], c["PW"], rois.data_ptr(),
                                                         gin.data_ptr(), ws_b.data_ptr(), nb_b, ext.PATH_TILED, stream)
            if st != 1:
                raise RuntimeError(f"rroi_align_backward_layout_hip -> {st}")
        bwd_cl_ms = event_loop(bwd_cl, 10, 50)
        del gout, gout_cl, ws_b, gin
    # ... and in the hand-off the reference's pipeline can reach (round 6): NCHW features / gradient, channels-last crops / top_diff
    mixed = None
    if world == 1 and os.environ.get("RROI_BENCH_SENSITIVITY", "1") == "1":
        try:
            mixed = mixed_layout(ext, dev, event_loop, c, feats, rois)
            launch(ext.STAGE_ALL)
        except Exception as e:
            mixed = {"error": repr(e)[:300]}

    # what ONE call costs when it is not the 1,000th of a back-to-back run -- after >= 2 ms of an
    # idle GPU and after unrelated kernels (an elementwise pass over 64 MB and a 2048^3 matrix product) have had the caches:
    # one HIP event before and one after the single call (the pair costs ~2-3 us of its own: it brackets an EMPTY stream at
    # the `empty_pair_ms` figure)
    isolated = None
    if world == 1 and os.environ.get("RROI_BENCH_SENSITIVITY", "1") == "1":
        other_a = torch.randn(16 << 20, dtype=torch.float32, device=dev)
        other_m = torch.randn(2048, 2048, dtype=torch.float32, device=dev)

        def one_call(idle, unrelated, empty=False):
            if unrelated:
                torch.mul(other_a, 1.0001, out=other_a)
                torch.mm(other_m, other_m)
            torch.cuda.synchronize()
            if idle:
                time.sleep(0.002)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if not empty:
                launch(ext.STAGE_ALL)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)
        def stats(v):
            v = sorted(v)
            return {"median": round(v[len(v) // 2], 5), "min": round(v[0], 5), "max": round(v[-1], 5), "n": len(v)}
        isolated = {"after_idle_ms": stats([one_call(True, False) for _ in range(15)]),
                    "after_unrelated_kernels_ms": stats([one_call(False, True) for _ in range(15)]),
                    "after_idle_and_unrelated_kernels_ms": stats([one_call(True, True) for _ in range(15)]),
                    "empty_pair_ms": stats([one_call(False, False, empty=True) for _ in range(15)]),
                    "what": "ONE forward call (prologue + gather, configs[1]) between two HIP events: after 2 ms of idle GPU; "
                            "right after unrelated kernels (an elementwise pass over 64 MB, a 2048^3 product); after both. "
                            "empty_pair_ms = the two events with nothing between them. `value` is the steady state of a "
                            "back-to-back run; this is what a call costs in the middle of other work"}
        del other_a, other_m

    # two calls in flight on two streams (round 6): what the write path does when it is never left idle
    inflight = None
    if world == 1 and os.environ.get("RROI_BENCH_SENSITIVITY", "1") == "1":
        try:
            inflight = two_calls_in_flight(ext, dev, c, feats, rois, R * c["C"] * c["PH"] * c["PW"] * 4 + R * 24
                                           + TOUCHED_PIXELS_RANK0 * c["C"] * 4)
        except Exception as e:
            inflight = {"error": repr(e)[:300]}

    train = None
    if world == 1 and os.environ.get("RROI_BENCH_TRAIN", "1") == "1":
        try:
            train = train_regime(ext, dev, event_loop)
        except Exception as e:
            train = {"error": repr(e)[:300]}

    tstep = None
    if world == 1 and os.environ.get("RROI_BENCH_TRAIN", "1") == "1":
        try:
            tstep = train_step(ext, dev, event_loop)
        except Exception as e:
            tstep = {"error": repr(e)[:300]}

    big = None
    if world == 1 and os.environ.get("RROI_BENCH_BIG", "1") == "1":
        try:
            big = beyond_cache(ext, dev, event_loop)
        except Exception as e:
            big = {"error": repr(e)[:300]}

    # spread of the call, measured LAST among the kernel timings (200 interleaved event records leave the
    # runtime ~3.5 us per call slower for what follows in the process -- measured, cause not pursued)
    if world == 1:
        n_s = 200
        for _ in range(20):
            launch(ext.STAGE_ALL)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_s + 1)]
        evs[0].record()
        for i in range(n_s):
            launch(ext.STAGE_ALL)
            evs[i + 1].record()
        torch.cuda.synchronize()
        d = np.sort(np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n_s)]))
        del evs
        step_stats = {"n": n_s, "median": round(float(np.median(d)), 5), "p10": round(float(d[n_s // 10]), 5),
                      "p90": round(float(d[(9 * n_s) // 10]), 5),
                      "how": "one HIP event after every call, 200 calls, differences of consecutive events"}

    # calibration of the bound on this box: what plain writes of the same 256 MiB buffer reach.  Zeros are a special
    # case on this chip (8 TB/s; the data-dependent part of the power budget is idle); the gather writes feature
    # data, so its ceiling is a write of NON-ZERO data: a constant (fill_(1.0)) and values that differ from store to store
    # (the library's write probe: one 16-byte plain store per thread, torch's elementwise launch shape).  Run last: a stretch of zero fills slows what follows.
    flat = out.view(-1)
    fill_one_ms = event_loop(lambda: out.fill_(1.0), 10, 40)
    def write_probe():
        if ext._lib.rroi_align_write_probe_hip(out.data_ptr(), flat.numel(), stream) != 1:
            raise RuntimeError("rroi_align_write_probe_hip")
    probe_ms = event_loop(write_probe, 10, 40)
    fill_ms = event_loop(lambda: out.fill_(0.0), 10, 40)
    launch(ext.STAGE_ALL)  # leave the real result in `out`
    torch.cuda.synchronize()

    # configs[4] on the side: the end-to-end inference pipeline (backbone + RoIRotate + CRNN head)
    e2e = None
    if world == 1 and os.environ.get("RROI_BENCH_E2E", "1") == "1":
        try:
            sys.path.insert(0, ROOT)
            from bench_e2e import measure as e2e_measure
            e2e = e2e_measure(dev)
        except Exception as e:
            e2e = {"error": repr(e)[:300]}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = R * world * args.steps / elapsed

    cpu, touched = (None, None)
    if world == 1 and not args.no_cpu_baseline:
        cpu, touched = cpu_baseline(feats_np, rois_np)
    # algorithmic bytes of one gather launch: output + rois + unique feature taps
    bytes_out = R * c["C"] * c["PH"] * c["PW"] * 4
    if touched is None:  # no oracle pass in this run: the count the oracle gives for rank 0's 512 seeded ROIs
        touched = TOUCHED_PIXELS_RANK0 if world == 1 else c["H"] * c["W"]  # N > 1: another draw, the upper bound
    elif world == 1 and touched != TOUCHED_PIXELS_RANK0:
        print("bench.py: touched pixels %d != the recorded %d (the oracle's count is used)" % (touched, TOUCHED_PIXELS_RANK0),
              file=sys.stderr)
    bytes_feat = touched * c["C"] * 4
    b_alg = bytes_out + R * 24 + bytes_feat
    # The dominant kernel's duration INSIDE the step (the timed region's launches): the step minus the prologue,
    # both between HIP events.  Launched alone, back to back, the same kernel is ~1 us faster (its map slices
    # are still in the L2s from the previous launch); the profiler's per-kernel average is over both kinds.
    # (never below the kernel launched by itself: the subtraction understates the gather when the prologue's own loop is
    # slowed by something the step is not -- a profiler's per-launch interception)
    gather_in_step_ms = max(step_events_ms - prologue_ms, gather_ms, 1e-6)
    achieved = b_alg / (gather_in_step_ms * 1e-3) / 1e9
    fill_gbs = out.numel() * 4 / (fill_ms * 1e-3) / 1e9
    fill_one_gbs = out.numel() * 4 / (fill_one_ms * 1e-3) / 1e9
    probe_gbs = out.numel() * 4 / (probe_ms * 1e-3) / 1e9
    nonzero_gbs = max(fill_one_gbs, probe_gbs)

    bwd_prof = None
    bpath = os.path.join(ROOT, "profiles", "bwd_traffic.json")  # rocprofv3 kernel trace + PMC passes over the backward
    if os.path.exists(bpath):
        try:
            bwd_prof = json.load(open(bpath))
            bwd_prof["traffic_measured"] = False   # the round's committed passes, not this run
        except Exception:
            bwd_prof = None

    traffic = None
    traffic_source = None
    traffic_measured = False     # True only when the counter passes ran inside THIS bench run
    how_live = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                "tools/traffic_probe.py %s; FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 B), WRITE_SIZE as is")
    if world == 1:
        live = live_traffic("forward")   # two short counter passes over the configs[1] call
        g = None if live is None else next((v for k, v in live.items() if k.startswith("rroi_fwd_split_kernel")), None)
        if g is not None:
            traffic = int(g["FETCH_SIZE"] * 2048 + g["WRITE_SIZE"] * 1024)
            traffic_measured = True
            traffic_source = how_live % ("(the configs[1] call, %d launches per pass: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB)"
                                         % (g["launches"], g["FETCH_SIZE"], g["WRITE_SIZE"]))
        live_b = live_traffic("backward")   # ... and over the configs[2] backward call
        if live_b is not None:
            per = {k: {"read_bytes": int(v["FETCH_SIZE"] * 2048), "written_bytes": int(v["WRITE_SIZE"] * 1024)}
                   for k, v in live_b.items()}
            bwd_prof = {"traffic_measured": True, "source": how_live % "backward (the configs[2] call)",
                        "what": "rroi_align_backward_hip, PATH_TILED, BASELINE configs[2]",
                        "kernel_us": {k: v["us"] for k, v in live_b.items()},
                        "kernel_us_how": "average duration of each launch in the counter passes (profiled clocks run ~3 % "
                                         "below unprofiled ones)",
                        "traffic_per_kernel": per,
                        "traffic_bytes_per_call": sum(v["read_bytes"] + v["written_bytes"] for v in per.values())}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # the round's committed counter passes: the fallback
    if traffic is None and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("gather_kernel_bytes_per_launch")
            traffic_source = ("profiles/traffic.json: FETCH_SIZE (doubled) + WRITE_SIZE of separate rocprofv3 --pmc passes over "
                              "this kernel, cold caches -- collected once per round, not in this run (no rocprofv3 here, or "
                              "RROI_BENCH_TRAFFIC=0, or the pass failed)")
        except Exception:
            traffic = None

    line = {
        "metric": "RoIRotate forward ROIs/sec", "value": round(value, 1), "unit": "ROIs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[1]" if world == 1 else "configs[3] (%d ROIs sharded over %d GPUs)" % (R * world, world))
                               + ": features 1x256x160x160 fp32 NCHW%s, %d rotated ROIs/GPU, "
                               "pooled 8x64, spatial_scale 0.25, forward" % (" replicated" if world > 1 else "", R),
                   "rois_per_gpu": R, "rois_total": R * world, "channels": c["C"],
                   "pooled": [c["PH"], c["PW"]], "path": "prologue(relayout+affine) + tiled gather (gatherer + storer waves)",
                   "parallelism": "roi-shard x%d, no data-path collective" % world},
        "roofline": {"bound": "hbm", "kernel": "rroi_fwd_split_kernel", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "whole_call_frac": round(b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     # the same bytes over the per-call wall time with TWO calls in flight on two streams (extra.two_calls_in_flight;
                     # `value` and whole_call_frac stay the sequential call)
                     "whole_call_frac_two_in_flight": None if not inflight or "whole_call_frac" not in inflight else inflight["whole_call_frac"],
                     "traffic": traffic,
                     "traffic_measured": traffic_measured,
                     "traffic_source": traffic_source,
                     "algorithmic_bytes": b_alg,
                     "kernel_ms": {"avg": round(gather_in_step_ms, 5), "alone": round(gather_ms, 5),
                                   "how": "avg: inside the step = max(whole_call_ms_events - prologue_ms_avg, alone) (300 steps "
                                          "and 200 prologues between two HIP events each); alone: %d back-to-back "
                                          "launches of the kernel by itself between two HIP events after %d warm-up "
                                          "launches (map slices still in the L2s; includes the ~1 us launch-to-"
                                          "launch gap the rocprofv3 kernel trace does not)" % (KERNEL_TIMED, KERNEL_WARM)},
                     "frac_alone": round(b_alg / (gather_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "prologue_ms_avg": round(prologue_ms, 5),
                     "whole_call_ms_events": round(step_events_ms, 5),
                     "whole_call_ms_spread": step_stats,
                     "calibrated": {"what": "torch fill_(0.0) of the 256 MiB output buffer on this GPU (40 back-to-back): "
                                            "ZEROS, which this chip writes faster than any other data -- kept for "
                                            "continuity with rounds 1-3, not the gather's ceiling",
                                    "GB/s": round(fill_gbs, 1), "frac_of_it": round(achieved / fill_gbs, 4)},
                     "calibrated_nonzero": {
                         "what": "plain writes of NON-ZERO data into the same 256 MiB buffer, 40 back-to-back each: "
                                 "fill_(1.0) (a constant) and rroi_align_write_probe_hip (values that differ from store to "
                                 "store, one 16-byte plain store per thread = torch's elementwise launch shape); the better "
                                 "of the two is the ceiling the gather's output stream is held against "
                                 "(profiles/r04_write_ceiling.txt: the same by store policy, tools/kbench wceil)",
                         "fill_one_GB/s": round(fill_one_gbs, 1), "nonconstant_GB/s": round(probe_gbs, 1),
                         "GB/s": round(nonzero_gbs, 1),
                         "frac_of_it": round(achieved / nonzero_gbs, 4),
                         "output_stream_frac_of_it": round(bytes_out / (gather_in_step_ms * 1e-3) / 1e9 / nonzero_gbs, 4)}},
        "cpu_baseline": cpu,
        "extra": {
            "sensitivity": sensitivity,
            "isolated_call_ms": isolated,
            "two_calls_in_flight": inflight,
            "ranks": ranks_info,
            "with_gather_ms": None if with_gather_ms is None else round(with_gather_ms, 4),
            "with_allreduce_grad_ms": None if with_allreduce_ms is None else round(with_allreduce_ms, 4),
            "with_allreduce_grad": None if world == 1 else (
                allreduce_err or "step + all_reduce of the (1, 256, 160, 160) fp32 feature gradient (25 MiB), max over ranks"),
            "with_gather": None if world == 1 else (
                gather_err or "step + all_gather_into_tensor of the crops into one preallocated "
                              "(%d, 256, 8, 64) buffer (%.2f GiB), max over ranks" % (R * world, R * world * 524288 / 2 ** 30)),
            "channels_last_call": None if nhwc_ms is None else {
                "what": "the same forward call with channels-last feature storage, consumed in place (no relayout); "
                        "not part of `value` (BASELINE's contract is NCHW)",
                "ms_per_call": round(nhwc_ms, 5), "ROIs/s": round(R / (nhwc_ms * 1e-3), 1),
                "frac_of_peak_whole_call": round(b_alg / (nhwc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "backward": None if bwd_ms is None else {
                "what": "configs[2]: grad w.r.t. the features, same shapes, rroi_align_backward_hip (gather path), "
                        "50 back-to-back calls; not part of `value`",
                "ms_per_call": round(bwd_ms, 5),
                # grad_out read + grad_in write (+ rois) over the whole call, against the same peak
                "algorithmic_bytes": int(R * c["C"] * c["PH"] * c["PW"] * 4 + c["C"] * c["H"] * c["W"] * 4 + R * 24),
                "frac_of_peak_whole_call": round((R * c["C"] * c["PH"] * c["PW"] * 4 + c["C"] * c["H"] * c["W"] * 4 + R * 24)
                                                 / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "ms_per_call_channels_last": round(bwd_cl_ms, 5),  # top_diff and the feature gradient both channels_last
                "mixed": mixed,   # NCHW features / feature gradient, channels-last crops / top_diff (+ the forward of that pair)
                # per-kernel durations and fabric traffic of the same call: rocprofv3 kernel trace + separate --pmc passes
                # (tools/profile_bwd.sh -> profiles/bwd_traffic.json), collected once per round, not in this run
                "roofline": None if bwd_prof is None else dict(
                    bwd_prof, bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s",
                    achieved=round((R * c["C"] * c["PH"] * c["PW"] * 4 + c["C"] * c["H"] * c["W"] * 4 + R * 24)
                                   / (bwd_ms * 1e-3) / 1e9, 1),
                    wasted_traffic_ratio=None if not bwd_prof.get("traffic_bytes_per_call") else round(
                        bwd_prof["traffic_bytes_per_call"]
                        / (R * c["C"] * c["PH"] * c["PW"] * 4 + c["C"] * c["H"] * c["W"] * 4 + R * 24), 3))},
            "train_regime": train,
            "train_step": tstep,
            "beyond_cache": big,
            "e2e": e2e,
        },
    }
    print(json.dumps(line))
    sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

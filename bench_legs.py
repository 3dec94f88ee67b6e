"""bench_legs.py -- the side legs of bench.py's line (`extra.train_regime`, `extra.beyond_cache`, `extra.mixed_layout`,
`extra.train_step`, `extra.two_calls_in_flight`): benchmark code, not part of the product package; none of it enters `value`.

Round 6 moved the first two out of bench.py and added the other two: the hand-off layouts the FOTS pipeline can actually
reach (VERDICT r05 item 1 -- the reference's backbone emits NCHW features, tools/models.py:387-457, and its head consumes
the crops directly, src/ocr_process.py:266-267, :295) and the training caller's step (train.py:79-119 ->
src/ocr_process.py:259-301).
"""
import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def settled(event_loop, fn, timed):
    """event_loop after ~20 ms of the same call: every shape starts behind host-side set-up (uploads, allocations) and
    the first few hundred calls after such a pause run ~2.5 us slower than the ones that follow (tools/merge_ab.py and
    tools/groups_ab.py time every form twice for that reason: 35.4 then 32.8 us for the same launch).  Median of three
    loops: at R = 32 a call is 8-9 us of GPU time and the host is barely ahead of it, so one host-side hiccup inside a
    2 ms loop shows (seen once: 14.2 us for a call that is 7.8 in every other run)."""
    per_call_ms = max(event_loop(fn, 5, 30), 1e-3)
    return sorted([event_loop(fn, int(20.0 / per_call_ms) + 1, timed), event_loop(fn, 0, timed), event_loop(fn, 0, timed)])[1]


def train_regime(ext, dev, event_loop):
    """the reference's OWN training call (src/ocr_process.py:259-267: pooled_height 11,
    pooled_width = ceil(11 * max w / h) -- any integer), on its 64-channel 1/4 map: forward and backward per shape
    with algorithmic bytes and the fraction of the 8 TB/s peak.  Not part of `value`."""
    B, C, H, W, scale = 2, 64, 120, 160, 0.25
    stream = torch.cuda.current_stream().cuda_stream
    rows = {}

    for PW in (83, 100, 96):          # 11 x 96 is the aligned shape the SHIFT kernels are held against
        for R in (32, 512):
            rng = np.random.default_rng(1000 + R + PW)
            feats = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).to(dev)
            h = rng.uniform(16, 64, R)
            rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                              h * rng.uniform(2, PW / 11.0, R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).to(dev)
            out = torch.empty((R, C, 11, PW), dtype=torch.float32, device=dev)
            gout = torch.randn_like(out)
            gin = torch.empty_like(feats)
            nf = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, ext.LAYOUT_NCHW)
            nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, 11, PW)
            ws = torch.empty(max(nf, nb, 1), dtype=torch.uint8, device=dev)

            def fwd():
                st = ext._lib.rroi_align_forward_hip(feats.data_ptr(), ext.LAYOUT_NCHW, scale, B, R, H, W, C, 11, PW,
                                                     rois.data_ptr(), out.data_ptr(), ws.data_ptr(), nf, ext.PATH_AUTO, stream)
                if st != 1:
                    raise RuntimeError(f"train_regime forward -> {st}")

            def bwd():
                st = ext._lib.rroi_align_backward_hip(gout.data_ptr(), scale, B, R, H, W, C, 11, PW, rois.data_ptr(),
                                                      gin.data_ptr(), ws.data_ptr(), nb, ext.PATH_AUTO, stream)
                if st != 1:
                    raise RuntimeError(f"train_regime backward -> {st}")
            f_ms, b_ms = settled(event_loop, fwd, 200), settled(event_loop, bwd, 100)
            # the same pair with channels-last tensors at both ends (what the callers' modules hand over when the backbone
            # runs channels_last): features consumed in place, crops / gradients channels-last
            feats_cl = feats.contiguous(memory_format=torch.channels_last)
            out_cl = torch.empty((R, C, 11, PW), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
            gout_cl = gout.contiguous(memory_format=torch.channels_last)
            nf_cl = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, ext.LAYOUT_NHWC)
            ws_cl = torch.empty(max(nf_cl, nb, 1), dtype=torch.uint8, device=dev)

            def fwd_cl():
                st = ext._lib.rroi_align_forward_layout_hip(feats_cl.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NHWC, scale, B, R, H, W, C,
                                                            11, PW, rois.data_ptr(), out_cl.data_ptr(), ws_cl.data_ptr(), nf_cl,
                                                            ext.PATH_AUTO, stream)
                if st != 1:
                    raise RuntimeError(f"train_regime forward (channels-last) -> {st}")

            def bwd_cl():
                st = ext._lib.rroi_align_backward_layout_hip(gout_cl.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NHWC, scale, B, R, H, W, C,
                                                             11, PW, rois.data_ptr(), gin.data_ptr(), ws_cl.data_ptr(), nb,
                                                             ext.PATH_AUTO, stream)
                if st != 1:
                    raise RuntimeError(f"train_regime backward (channels-last) -> {st}")
            fcl_ms, bcl_ms = settled(event_loop, fwd_cl, 200), settled(event_loop, bwd_cl, 100)
            # MIXED (round 6): NCHW features and NCHW feature gradient -- what the reference's backbone emits
            # and expects -- with channels-last crops / top_diff: no pixel-major copy of top_diff in the backward

            def fwd_mx():
                st = ext._lib.rroi_align_forward_layout_hip(feats.data_ptr(), ext.LAYOUT_NCHW, ext.LAYOUT_NHWC, scale, B, R, H, W, C,
                                                            11, PW, rois.data_ptr(), out_cl.data_ptr(), ws.data_ptr(), nf,
                                                            ext.PATH_AUTO, stream)
                if st != 1:
                    raise RuntimeError(f"train_regime forward (mixed) -> {st}")

            def bwd_mx():
                st = ext._lib.rroi_align_backward_layout_hip(gout_cl.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NCHW, scale, B, R, H, W, C,
                                                             11, PW, rois.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb,
                                                             ext.PATH_AUTO, stream)
                if st != 1:
                    raise RuntimeError(f"train_regime backward (mixed) -> {st}")
            fmx_ms, bmx_ms = settled(event_loop, fwd_mx, 200), settled(event_loop, bwd_mx, 100)
            del feats_cl, out_cl, gout_cl, ws_cl
            # algorithmic bytes: crops + rois + the map once (an upper bound of the touched pixels; at R = 32 most of the
            # map is not touched, so the forward's fraction is an overestimate there -- the call still relays it out)
            crops, fmap = R * C * 11 * PW * 4, B * C * H * W * 4
            fb, bb = crops + R * 24 + fmap, crops + R * 24 + fmap
            rows["11x%d_R%d" % (PW, R)] = {
                "forward_us": round(f_ms * 1e3, 2), "backward_us": round(b_ms * 1e3, 2),
                "forward_algorithmic_bytes": fb, "backward_algorithmic_bytes": bb,
                "forward_frac_of_peak": round(fb / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "backward_frac_of_peak": round(bb / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "forward_us_channels_last": round(fcl_ms * 1e3, 2), "backward_us_channels_last": round(bcl_ms * 1e3, 2),
                "backward_frac_of_peak_channels_last": round(bb / (bcl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "forward_us_mixed": round(fmx_ms * 1e3, 2), "backward_us_mixed": round(bmx_ms * 1e3, 2),
                "backward_frac_of_peak_mixed": round(bb / (bmx_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del feats, rois, out, gout, gin, ws
    for PW in (83, 100):
        for R in (32, 512):
            rows["11x%d_R%d" % (PW, R)]["forward_vs_aligned_11x96"] = round(
                rows["11x%d_R%d" % (PW, R)]["forward_us"] / rows["11x96_R%d" % R]["forward_us"] / (PW / 96.0), 3)
    rows["what"] = ("the reference's training call (src/ocr_process.py:259-267): %d images of %d x %d x %d, pooled 11 x PW, "
                    "R ROIs over the images, PATH_AUTO, 200 / 100 back-to-back calls between HIP events after ~20 ms of the same call, median of three such loops; bytes = crops + "
                    "rois + the whole map once; forward_vs_aligned_11x96 = time per output byte against the 11 x 96 shape "
                    "(rows of whole 64-byte sectors) at the same R; *_channels_last = the same calls with channels-last features, crops "
                    "and gradients (no relayout on either side); *_mixed = NCHW features and NCHW feature gradient (the reference's backbone) with "
                    "channels-last crops / top_diff (channels_last_out=True): the layout pair the existing pipeline can reach -- whether it PAYS "
                    "there is extra.train_step" % (B, C, H, W))
    return rows


def beyond_cache(ext, dev, event_loop):
    """crops LARGER than the 256 MB memory-side cache, forward only -- rows of whole 64-byte sectors
    (configs[1]'s pooled size with twice the ROIs; the 64-channel training shape at 11 x 96) and rows that are not (the
    line-aligned windows of round 5: C = 256, 11 x 100 -- the shape the verdict names -- and C = 64, 11 x 83).  Not part of
    `value`."""
    stream = torch.cuda.current_stream().cuda_stream
    rows = {}
    for (tag, B, C, H, W, R, PH, PW) in (("C256_8x64_R1024", 1, 256, 160, 160, 1024, 8, 64), ("C256_11x100_R600", 1, 256, 160, 160, 600, 11, 100),
                                         ("C64_11x96_R2048", 2, 64, 120, 160, 2048, 11, 96), ("C64_11x83_R2048", 2, 64, 120, 160, 2048, 11, 83)):
        rng = np.random.default_rng(1000 + R + PW)
        feats = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).to(dev)
        h = rng.uniform(16, 64, R)
        rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                          h * rng.uniform(2, PW / float(PH), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).to(dev)
        out = torch.empty((R, C, PH, PW), dtype=torch.float32, device=dev)
        nf = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, ext.LAYOUT_NCHW)
        ws = torch.empty(max(nf, 1), dtype=torch.uint8, device=dev)

        def fwd():
            st = ext._lib.rroi_align_forward_hip(feats.data_ptr(), ext.LAYOUT_NCHW, 0.25, B, R, H, W, C, PH, PW, rois.data_ptr(),
                                                 out.data_ptr(), ws.data_ptr(), nf, ext.PATH_AUTO, stream)
            if st != 1:
                raise RuntimeError(f"beyond_cache forward -> {st}")
        ms = sorted(event_loop(fwd, 30 if i == 0 else 0, 60) for i in range(3))[1]
        crops = R * C * PH * PW * 4
        rows[tag] = {"forward_us": round(ms * 1e3, 1), "crops_MB": round(crops / 1e6, 1),
                     "crops_TBps": round(crops / (ms * 1e-3) / 1e12, 2), "rows_are_whole_sectors": PH * PW % 16 == 0}
        del feats, rois, out, ws
    rows["what"] = ("forward calls whose crops exceed the 256 MB memory-side cache (PATH_AUTO, NCHW, median of three loops of 60 "
                    "back-to-back calls): bytes of crops / time.  Rows that are not whole sectors take the line-aligned windows "
                    "(32 own bins of 64 gathered) there; round 4: 2.2-3.0 TB/s")
    return rows




def mixed_layout(ext, dev, event_loop, cfg, feats, rois):
    """configs[2]'s shapes in the MIXED hand-off: NCHW features -> channels-last crops (two launches: prologue +
    the channels-last gather), channels-last top_diff -> NCHW feature gradient (no pixel-major copy of top_diff: the gather reads
    the caller's tensor in place).  Next to the NCHW pair of the same run.  Not part of `value`."""
    c = cfg
    stream = torch.cuda.current_stream().cuda_stream
    R = rois.shape[0]
    out_cl = torch.empty((R, c["C"], c["PH"], c["PW"]), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
    gout_cl = torch.randn((R, c["C"], c["PH"], c["PW"]), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
    gin = torch.empty((1, c["C"], c["H"], c["W"]), dtype=torch.float32, device=dev)
    nf = ext._lib.rroi_align_forward_workspace_bytes(1, c["C"], c["H"], c["W"], R, ext.LAYOUT_NCHW)
    nb = ext._lib.rroi_align_backward_workspace_bytes(1, c["C"], c["H"], c["W"], R, c["PH"], c["PW"])
    ws = torch.empty(max(nf, nb, 1), dtype=torch.uint8, device=dev)

    def fwd():
        st = ext._lib.rroi_align_forward_layout_hip(feats.data_ptr(), ext.LAYOUT_NCHW, ext.LAYOUT_NHWC, c["scale"], 1, R, c["H"], c["W"],
                                                    c["C"], c["PH"], c["PW"], rois.data_ptr(), out_cl.data_ptr(), ws.data_ptr(), nf,
                                                    ext.PATH_TILED, stream)
        if st != 1:
            raise RuntimeError(f"mixed forward -> {st}")

    def bwd():
        st = ext._lib.rroi_align_backward_layout_hip(gout_cl.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NCHW, c["scale"], 1, R, c["H"],
                                                     c["W"], c["C"], c["PH"], c["PW"], rois.data_ptr(), gin.data_ptr(), ws.data_ptr(),
                                                     nb, ext.PATH_TILED, stream)
        if st != 1:
            raise RuntimeError(f"mixed backward -> {st}")
    f_ms, b_ms = event_loop(fwd, 100, 300), event_loop(bwd, 10, 50)
    return {"forward_ms_per_call": round(f_ms, 5), "backward_ms_per_call": round(b_ms, 5), "forward_plus_backward_ms": round(f_ms + b_ms, 5),
            "what": "configs[2] with NCHW features / NCHW feature gradient and channels-last crops / top_diff "
                    "(rroi_align_forward_layout_hip(NCHW, NHWC), rroi_align_backward_layout_hip(NHWC, NCHW)): the backward skips the "
                    "pixel-major copy of top_diff.  Same loops as extra.backward.ms_per_call (NCHW at both ends)"}


def train_step(ext, dev, event_loop):
    """The training caller's recognition branch as the reference runs it (train.py:79-119 -> src/ocr_process.py:259-301): op forward
    on the 64-channel 1/4 map of a batch of two 512 x 512 images (train.py: batch_size 2, input_size 512), 32 ROIs (:253-255),
    pooled 11 x 96 -> forward_ocr -> CTC loss -> backward through the head and the op.  Timed with NCHW crops (the reference's
    contract) and with channels-last crops (`channels_last_out=True`, the mixed hand-off); `head_*` = the head alone on crops that
    are already there.  The head's weights stay in the reference's NCHW format.  Not part of `value`."""
    import torch.nn.functional as F
    from fots_e2e.model import FOTSNet
    from fots_e2e.weights import deterministic_init
    from rroi_align.modules.rroi_align import _RRoiAlign
    B, C, H, W, R, PH, PW, nclass = 2, 64, 128, 128, 32, 11, 96, 87
    rng = np.random.default_rng(77)
    feats = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).to(dev)
    h = rng.uniform(16, 64, R)
    rois = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                      h * rng.uniform(2, PW / float(PH), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).to(dev)
    lens = [int(v) for v in rng.integers(3, 10, R)]
    targets = torch.from_numpy(rng.integers(1, nclass, sum(lens)).astype(np.int64)).to(dev)
    net = deterministic_init(FOTSNet(nclass)).to(dev).train()
    params = list(net.parameters())

    def ctc(preds):      # src/ocr_process.py:296-301
        return F.ctc_loss(preds.permute(2, 0, 1), targets, [PW] * R, lens, blank=0, reduction="sum") / R
    rows, fns = {}, {}
    for name, cl in (("crops_nchw", False), ("crops_channels_last", True)):
        op = _RRoiAlign(PH, PW, 0.25, channels_last_out=cl)
        crops0 = op(feats, rois).detach()
        leaf = feats.clone().requires_grad_(True)

        def head(crops0=crops0):
            x = crops0.clone().requires_grad_(True)
            ctc(net.forward_ocr(x)).backward()
            for p in params:
                p.grad = None

        def step(op=op, leaf=leaf):
            leaf.grad = None
            ctc(net.forward_ocr(op(leaf, rois))).backward()
            for p in params:
                p.grad = None
        fns[name] = (head, step)
        event_loop(step, 10, 10)
    # A / B / A / B ...: the head's own run-to-run spread (tens of us) is larger than what the op's layout changes
    samples = {name: ([], []) for name in fns}
    for _ in range(5):
        for name, (head, step) in fns.items():
            samples[name][0].append(event_loop(head, 3, 25) * 1e3)
            samples[name][1].append(event_loop(step, 3, 25) * 1e3)
    for name, (h_, s_) in samples.items():
        h_, s_ = sorted(h_), sorted(s_)
        rows[name] = {"head_forward_backward_us": {"median": round(h_[2], 1), "min": round(h_[0], 1), "max": round(h_[-1], 1)},
                      "step_us": {"median": round(s_[2], 1), "min": round(s_[0], 1), "max": round(s_[-1], 1)}}
    rows["what"] = ("recognition branch of one training step: _RRoiAlign forward (2 x 64 x 128 x 128 map, 32 ROIs, 11 x 96) -> "
                    "FOTSNet.forward_ocr -> CTC -> backward through head and op; head_* = the head alone on crops that are already "
                    "there (fp32, MIOpen, NCHW weights); five interleaved loops of 25 per figure.  The op's two calls are ~30 us of a "
                    "~4.7 ms step (0.6 %).  Measured on every box so far: channels-last crops make the head (NCHW weights) 70-100 us SLOWER per "
                    "step -- three times the op's whole cost -- so the callers' modules keep the reference's contract for NCHW features "
                    "(DESIGN.md 5.5)")
    return rows


def two_calls_in_flight(ext, dev, c, feats, rois, algorithmic_bytes, calls=600, warm=300):
    """Round 6: configs[1] with TWO calls in flight -- consecutive calls alternate between two HIP streams, each with its own
    crops and its own workspace (every call does all of its work and writes all of its output; the two crops tensors are
    compared bit for bit), so that one call's prologue and start-up run beside the other's store stream.  Wall clock per call
    over `calls` calls after `warm`, synchronised on both sides; the one-stream figure by the same loop beside it.  Not
    `value`: the headline stays the SEQUENTIAL call, which is what a caller with one feature map at a time gets."""
    import time
    R = rois.shape[0]
    nb = ext._lib.rroi_align_forward_workspace_bytes(1, c["C"], c["H"], c["W"], R, ext.LAYOUT_NCHW)

    def run(nstreams):
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        outs = [torch.empty((R, c["C"], c["PH"], c["PW"]), dtype=torch.float32, device=dev) for _ in range(nstreams)]
        wss = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
        torch.cuda.synchronize()

        def call(i):
            s = i % nstreams
            st = ext._lib.rroi_align_forward_stages_hip(feats.data_ptr(), ext.LAYOUT_NCHW, c["scale"], 1, R, c["H"], c["W"], c["C"],
                                                        c["PH"], c["PW"], rois.data_ptr(), outs[s].data_ptr(), wss[s].data_ptr(), nb,
                                                        ext.PATH_TILED, ext.STAGE_ALL, streams[s].cuda_stream)
            if st != 1:
                raise RuntimeError(f"two_calls_in_flight -> {st}")
        for i in range(warm):
            call(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(calls):
            call(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / calls * 1e3
        return ms, all(torch.equal(outs[0], o) for o in outs[1:])
    one_ms, _ = run(1)
    two_ms, same = run(2)
    return {"ms_per_call": round(two_ms, 5), "ROIs/s": round(R / (two_ms * 1e-3), 1),
            "whole_call_frac": round(algorithmic_bytes / (two_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "one_stream_ms_per_call": round(one_ms, 5),
            "one_stream_whole_call_frac": round(algorithmic_bytes / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "crops_identical": bool(same),
            "what": "configs[1], consecutive calls alternating between two HIP streams (own crops and workspace each): wall clock per "
                    "call over %d calls after %d, synchronised on both sides; one_stream_* = the same loop on one stream.  The "
                    "sequential call leaves the write path idle while a prologue and a gather's first taps run; a second call in "
                    "flight fills that.  Not `value`" % (calls, warm)}

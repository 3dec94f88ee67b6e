"""fots_e2e -- the callers' side of RoIRotate, end to end (BASELINE configs[4]).

The reference's inference driver (`test.py:44-127`) is image -> shared backbone + detection heads
(`tools/models.py:387-457`) -> boxes (`nms.get_boxes`) -> per detected word: host-built ROI, one
RoIRotate launch with R = 1, the recognition head (`models.py:334-379`), arg max and a Python CTC
decode (`tools/ocr_utils.py:131-199`).  This package is that pipeline on PyTorch-ROCm around the HIP op in
its MI355X shape (`pipeline.batched`, `pipeline.infer_image`) -- ROIs built on the device, ONE RoIRotate
launch per image, the head once per pooled-width bucket, one batched greedy-CTC launch; `pipeline.infer_batch` (round 6)
takes SEVERAL images of one size through one pass of the network and one RoIRotate launch (the op's batch index), and
`pipeline.infer_stream` keeps two such batches in flight.  The reference's
per-word structure is kept with the checkers (`oracle/e2e_loop_oracle.py`: the tests' comparison, the
benchmark's baseline leg), not here.

The network (`model.FOTSNet`) is stock torch.nn running on MIOpen / rocBLAS: it is the producer
and the consumer of the op's tensors, not part of the hot path, and is restated here only because
no reference source travels to the GPU box.  Its `state_dict` keys equal the reference's
(`ModelResNetSep2`), so the reference's checkpoints load unchanged; it is pinned against the
reference's own module by `tests/golden/make_e2e_golden.py`.
"""

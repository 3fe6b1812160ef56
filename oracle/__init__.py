"""oracle/ - CPU restatement of the reference's caption hot path.  TEST INFRASTRUCTURE.

This package is the *checker*, never the product:
  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
    `bench.py` may import it;
  * the shipped package (`transform-and-tell_amd/`, import alias `tell_amd`)
    never imports it and fails loudly if its HIP library is missing.

Arithmetic is plain PyTorch fp32 on CPU (the path is floating point).  Every
function cites the reference file:line (relative to /root/reference) whose
algorithm it restates.  It is written from the algorithm, not from the text of
the reference: direct K-tap gathers instead of band matrices, one table-driven
decoder instead of per-model copies, static-shape adaptive softmax.

Pinning:
  * decoder-side modules, decoders, criterion, model wrappers and greedy
    generation: pinned against fixtures in `tests/golden/*.npz`, produced by
    `tests/golden/make_golden.py` which imports and runs the real reference in
    the authoring container;  `make_positions` additionally against the
    reference's own known-answer test (tell/modules/token_embedders/tests/
    test_positional.py:8-44).
  * ResNet-152 (torchvision 0.6.1 Bottleneck v1.5, absent here), RoBERTa-large
    (fairseq @2f7e3f3323 via torch.hub, absent here) and BertAdam
    (pytorch_pretrained_bert, absent here): restated from their published
    architecture / update rule - **parity unpinned** for those three (RoBERTa is
    cross-checked structurally against `transformers.RobertaModel` in the
    authoring container, see tests/test_oracle_encoders.py).
"""

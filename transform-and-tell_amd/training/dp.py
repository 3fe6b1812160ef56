"""Data-parallel pieces that are independent of the device (exercised on CPU with gloo in
tests/test_dp_gloo.py, run over RCCL/xGMI on the MI355X node)."""
import torch


def loss_weight(n_local, dist, world):
    """Weight that makes mean-of-rank-losses equal the loss of ONE process on the concatenated
    batch: the reference normalises by the local non-pad token count
    (transformer_faces_objects.py:88), so rank r must count n_r * world / sum_r n_r.
    n_local: float tensor [1] on the compute device (no host sync)."""
    n_global = n_local.clone()
    dist.all_reduce(n_global)
    return n_local * world / n_global


def all_reduce_flat(flat_grad, dist, bucket_elems):
    """Sum-all-reduce of the whole flat fp32 gradient buffer in few, large buckets (the xGMI ring is
    per-link bound; every rank always contributes every bucket, zeros included)."""
    handles = [dist.all_reduce(flat_grad[s:s + bucket_elems], async_op=True)
               for s in range(0, flat_grad.numel(), bucket_elems)]
    for h in handles:
        h.wait()


def all_reduce_flat_bf16(flat_grad, wire, dist, bucket_elems, cast):
    """The same reduction with bf16 on the wire: the xGMI ring is per-link bound (2 GPUs share ONE link: 802 MB of
    fp32 gradients would take ~10 ms of a 14.5 ms step), so the gradients are rounded to bf16 for the exchange -
    the precision every activation and working weight of the bf16 path already has - and widened back into the
    fp32 buffer the optimizer reads.  `wire`: persistent bf16 buffer of flat_grad's size; `cast(src, dst)`: the
    device cast (tell_cast).  Pre-scaling is not needed: sums of <= 8 gradient values stay far inside bf16 range."""
    cast(flat_grad, wire)
    handles = [dist.all_reduce(wire[s:s + bucket_elems], async_op=True)
               for s in range(0, wire.numel(), bucket_elems)]
    for h in handles:
        h.wait()
    cast(wire, flat_grad)

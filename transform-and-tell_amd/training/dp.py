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

"""Episode sharding over the GPUs of one node (not in the reference, which is single-process: fumi/main.py:145-146).

Episodes of a meta-batch are independent given the meta-parameters (fumi/models/fumi.py:182,187:
L = 1/B sum_b L_b), so rank r of R owns the contiguous episode block [r*B/R, (r+1)*B/R) of the SAME sampled
meta-batch; each rank writes grad_scale=1/B times the sum of its local per-episode gradients into one flat fp32
buffer [grads | sum loss | sum acc] and a single all-reduce(sum) (RCCL over xGMI; gloo in the CPU tests) makes the
buffer identical everywhere, after which every rank applies the identical optimizer step."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(B):
    """Episode range [lo, hi) of this rank; B must divide evenly (equal shards keep mean-of-means exact)."""
    rank, size = world()
    if size == 1:
        return 0, B
    if B % size:
        raise ValueError(f"meta-batch of {B} episodes does not shard evenly over {size} ranks")
    per = B // size
    return rank * per, (rank + 1) * per


def all_reduce_sum_(flat):
    if world()[1] > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_gather_rows(t):
    """Concatenate equal row-blocks of every rank along dim 0."""
    _, size = world()
    if size == 1:
        return t
    out = [torch.empty_like(t) for _ in range(size)]
    dist.all_gather(out, t.contiguous())
    return torch.cat(out, 0)

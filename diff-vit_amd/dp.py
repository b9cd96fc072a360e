"""Data-parallel execution of the quantized forward over the GPUs of one node.

The path shards naturally: images are independent and every scale is frozen at calibration, so there is no
cross-sample statistic.  One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in
the CPU tests), the frozen plan is replicated, the global batch is split contiguously, and the only exchange step is one
all-gather of the logits per batch: fp32 [B_local, classes] (1 MB per GPU at 256 x 1000) or, with
``codes=True``, the int8 logit codes (4x smaller; logits are codes * act_out scale).

The collective itself costs ~10 us per step (tools/gather_cost.py: 2.636 -> 2.643 ms with a one-rank RCCL group).  What does cost is
INITIALISING RCCL with the runtime's default of four hardware queues: its streams take queues, two of the three slice streams of the
forward then share one, and the step takes 3.6 ms on every rank.  ``import diff_vit_amd`` therefore defaults ``GPU_MAX_HW_QUEUES`` to 8
(it must be in the environment before the process makes its first HIP call).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """contiguous split of n items; the first n % world ranks get one more."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class DataParallelForward:
    """``forward_fn(local_images) -> local_logits`` on every rank, then all-gather.  ``forward_fn`` is the product's
    quantized forward on GPU ranks (``lambda x: model(x, bits)[0]``)."""

    def __init__(self, forward_fn, num_classes, group=None, always_gather=False):
        """``always_gather``: run the collective even in a one-rank group (exercises the RCCL path on a single GPU)."""
        self.fn = forward_fn
        self.num_classes = num_classes
        self.group = group
        self.always_gather = bool(always_gather) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def __call__(self, global_images):
        """every rank passes the same global batch (or only its own slice via ``local``)."""
        lo, hi = shard_bounds(global_images.shape[0], self.world, self.rank)
        return self.local(global_images[lo:hi], global_images.shape[0])

    def _all_gather(self, gathered, shard):
        """one all-gather on the tensors' own device (RCCL over xGMI for GPU tensors under backend "nccl").  The gloo backend of a
        ROCm build has no device collectives: a GPU shard is then staged through the host (rehearsal runs of N ranks on one GPU)."""
        if shard.is_cuda and dist.get_backend(self.group) == 'gloo':
            host = torch.empty(gathered.shape, dtype=gathered.dtype)
            dist.all_gather_into_tensor(host, shard.cpu(), group=self.group)
            gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(gathered, shard, group=self.group)
        return gathered

    def local(self, local_images, global_n):
        out = self.fn(local_images)
        if self.world == 1 and not self.always_gather:
            return out
        sizes = [shard_bounds(global_n, self.world, r) for r in range(self.world)]
        counts = [b - a for a, b in sizes]
        if len(set(counts)) == 1:
            gathered = torch.empty(global_n, self.num_classes, dtype=out.dtype, device=out.device)
            return self._all_gather(gathered, out.contiguous())
        # ragged global batch: collectives need equal shapes, so pad every shard to the largest one
        cmax = max(counts)
        padded = torch.zeros(cmax, self.num_classes, dtype=out.dtype, device=out.device)
        padded[:out.shape[0]] = out
        gathered = self._all_gather(torch.empty(self.world * cmax, self.num_classes, dtype=out.dtype, device=out.device), padded)
        return torch.cat([gathered[r * cmax: r * cmax + counts[r]] for r in range(self.world)], 0)

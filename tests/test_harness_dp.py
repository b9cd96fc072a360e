"""CPU: harness (row H) and the data-parallel runner (world_size-2 gloo)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def test_accuracy_and_meter():
    import diff_vit_amd as dva
    out = torch.tensor([[0.1, 0.9, 0.0, 0.0, 0.0, 0.0], [0.5, 0.1, 0.2, 0.3, 0.05, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    p1, p5 = dva.harness.accuracy(out, torch.tensor([1, 5, 5]), topk=(1, 5))
    assert abs(p1.item() - 200.0 / 3) < 1e-4 and abs(p5.item() - 200.0 / 3) < 1e-4
    m = dva.harness.AverageMeter()
    m.update(2.0, 2); m.update(5.0, 1)
    assert m.val == 5.0 and abs(m.avg - 3.0) < 1e-9 and m.count == 3
    args = dva.harness.build_parser().parse_args([])
    assert args.model == 'deit_tiny' and args.calib_batchsize == 50 and args.val_batchsize == 50 and args.seed == 0
    assert dva.harness.str2model('deit_small') is dva.deit_small_patch16_224
    with pytest.raises(KeyError):
        dva.harness.str2model('resnet50')


def test_shard_bounds():
    from diff_vit_amd.dp import shard_bounds
    for n, w in ((2048, 8), (10, 4), (3, 8), (0, 2)):
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(x[1] - x[0] for x in b) - min(x[1] - x[0] for x in b) <= 1


def _dp_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import diff_vit_amd as dva
    torch.manual_seed(0)
    w = torch.randn(7, 3 * 8 * 8)
    calls = []
    fn = lambda x: x.reshape(x.shape[0], -1) @ w.t()           # stand-in forward (the GPU engine is not available on CPU)
    x = dva.synth.images(5, n, 8)
    runner = dva.dp.DataParallelForward(fn, 7, always_gather=(world == 1))
    if world == 1:                                             # the one-rank group still goes through the collective (bench.py --force-dist)
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    out = runner(x)
    if world == 1:
        assert calls == [1]
    if rank == 0:
        q.put((out.numpy(), fn(x).numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n', [8, 7])
def test_data_parallel_allgather_gloo(n):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, ref = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(got, ref)          # even (all_gather_into_tensor) and ragged (all_gather) global batches


def test_one_rank_group_still_gathers():
    """bench.py --force-dist: a one-rank process group runs the all-gather (the RCCL branch on a single GPU; gloo here)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_dp_worker, args=(0, 1, 29500 + (os.getpid() + 31) % 2000, 6, q))
    p.start()
    got, ref = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0 and np.array_equal(got, ref)

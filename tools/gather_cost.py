"""GPU (one rank): where the time of the data-parallel step's all-gather goes.  Runs under a one-rank NCCL (= RCCL) process group:
host time of the collective call, step time with / without it, variants (preallocated output, async_op, side stream).
usage: python tools/gather_cost.py"""
import os, socket, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch.distributed as dist
import diff_vit_amd as dva
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)
arch = dva.synth.ARCHS['deit_small']
m = dva.deit_small_patch16_224(cfg=dva.Config(True, True, 'minmax'))
m.load_state_dict(dva.synth.vit_state_dict(arch, 3), strict=False)
m = m.cuda().eval()
dva.harness.calibrate_model(m, dva.synth.images(3, 2, 224).cuda(), where='host')
plan = m.freeze('cuda')
bits, B = [8] * 50, 256
x = dva.synth.images(1000, B, 224).cuda()
out = torch.empty(B, 1000, device='cuda')
gat = torch.empty(B, 1000, device='cuda')
side = torch.cuda.Stream()


def run(name, step, n=60):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        h = step()
        host += h or 0.0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('%-46s %.3f ms / step   host time inside the collective call %.3f ms' % (name, dt * 1e3, host / n * 1e3), flush=True)


def fwd():
    plan.forward_streams(x, bits, out, 3)


def s_plain():
    fwd()


def s_sync_alloc():
    fwd()
    t = time.perf_counter()
    g = torch.empty(B, 1000, device='cuda')
    dist.all_gather_into_tensor(g, out)
    return time.perf_counter() - t


def s_prealloc():
    fwd()
    t = time.perf_counter()
    dist.all_gather_into_tensor(gat, out)
    return time.perf_counter() - t


pending = [None]


def s_async():
    fwd()
    t = time.perf_counter()
    if pending[0] is not None:
        pending[0].wait()
    pending[0] = dist.all_gather_into_tensor(gat, out, async_op=True)
    return time.perf_counter() - t


def s_copy_only():      # what a one-rank gather amounts to: a device copy on the same stream
    fwd()
    gat.copy_(out)


run('forward only', s_plain)
run('forward + all_gather (new output tensor)', s_sync_alloc)
run('forward + all_gather (preallocated output)', s_prealloc)
run('forward + all_gather async_op, waited a step later', s_async)
run('forward + device copy of the logits', s_copy_only)
run('forward only (again)', s_plain)
dist.destroy_process_group()

"""GPU: the PCIe-inclusive rate of the headline configuration - images start in (pinned) HOST memory, as the reference's DataLoader hands them
over (test_quant.py:418-447: `data.cuda()` per batch).  `value` of bench.py has the inputs resident in HBM; this is the other number.
Double-buffered: batch i + 1 is copied on a copy stream while batch i runs.  python tools/pcie_rate.py [steps]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 256
m = dva.deit_small_patch16_224(pretrained=False, cfg=dva.Config()).cuda().eval()
m.load_state_dict(dva.synth.vit_state_dict(dva.synth.ARCHS['deit_small'], 5), strict=False)
dva.harness.calibrate_model(m, dva.synth.images(5, 2, 224).cuda())
plan = m.freeze()
bc = [8] * 50
base = dva.synth.images(5, 64, 224, offset=100)
host = [base.repeat(4, 1, 1, 1).contiguous().pin_memory() for _ in range(2)]
dev = [torch.empty_like(host[0], device='cuda') for _ in range(2)]
out = torch.empty(B, 1000, device='cuda')
nbytes = host[0].numel() * 4
cur = torch.cuda.current_stream()
dev[0].copy_(host[0])
for _ in range(5):                                   # the side streams are chosen here, before the copy stream exists
    plan.forward_streams(dev[0], bc, out, 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    plan.forward_streams(dev[0], bc, out, 3)
torch.cuda.synchronize()
resident = B * steps / (time.perf_counter() - t0)
cp = torch.cuda.Stream()
with torch.cuda.stream(cp):
    for _ in range(3):
        dev[1].copy_(host[1], non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(cp):
    for _ in range(10):
        dev[1].copy_(host[1], non_blocking=True)
torch.cuda.synchronize()
h2d = 10 * nbytes / (time.perf_counter() - t0) / 1e9
copied = [torch.cuda.Event() for _ in range(2)]
used = [torch.cuda.Event() for _ in range(2)]
def loop(n):
    for i in range(2):
        used[i].record(cur)
    with torch.cuda.stream(cp):
        dev[0].copy_(host[0], non_blocking=True)
        copied[0].record(cp)
    for i in range(n):
        a, b = i & 1, (i + 1) & 1
        with torch.cuda.stream(cp):                  # the next batch, once the forward that read that buffer is done
            cp.wait_event(used[b])
            dev[b].copy_(host[b], non_blocking=True)
            copied[b].record(cp)
        cur.wait_event(copied[a])
        plan.forward_streams(dev[a], bc, out, 3)
        used[a].record(cur)
loop(5)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(steps)
torch.cuda.synchronize()
piped = B * steps / (time.perf_counter() - t0)
# the naive form: copy, then forward, on one stream
t0 = time.perf_counter()
for i in range(steps):
    dev[0].copy_(host[i & 1], non_blocking=True)
    plan.forward_streams(dev[0], bc, out, 3)
torch.cuda.synchronize()
serial = B * steps / (time.perf_counter() - t0)
# the product's path: harness.DevicePrefetcher (copies on engine.copy_stream - the pool's third, probed side stream - and a sliced forward that
# keeps to two side streams + the caller's) around the module forward
class Batches:
    def __init__(self, n):
        self.n = n
    def __len__(self):
        return self.n
    def __iter__(self):
        for i in range(self.n):
            yield host[i & 1], torch.zeros(B, dtype=torch.long)
def loop_p(n):
    for x, _ in dva.harness.DevicePrefetcher(Batches(n), 'cuda:0'):
        m(x, bc)
loop_p(5)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop_p(steps)
torch.cuda.synchronize()
piped_b = B * steps / (time.perf_counter() - t0)
print(json.dumps({'batch': B, 'input_bytes_per_batch': nbytes, 'h2d_GBps_pinned': round(h2d, 1), 'img_s_inputs_resident': round(resident),
                  'img_s_pcie_double_buffer_on_a_fifth_stream': round(piped), 'img_s_pcie_prefetcher': round(piped_b), 'compute_side_streams_after': dva.engine.compute_side_streams(0), 'img_s_pcie_copy_then_forward': round(serial),
                  'img_s_bound_by_copy': round(h2d * 1e9 / (nbytes / B)), 'side_streams': dva.engine.SIDE_STREAM_REPORT.get(0)}), flush=True)

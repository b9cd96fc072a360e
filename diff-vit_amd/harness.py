"""Evaluation harness of the hot path: the counterpart of the reference driver's ``validate / accuracy /
AverageMeter / seed / str2model`` and of its calibration sequence (test_quant.py:56-86,214-249,418-501).

Differences the build owns: a synthetic-data mode (no ImageFolder / torchvision / network),
``--mixed`` off by default (the reference's Hessian + Pareto search never completes as shipped, test_quant.py:186),
and images/sec reporting.  Everything else -- flag names, model-name map, the calibration call order, unpacking three
returns from ``model(data, bit_config, plot)``, top-k by ``output.topk(maxk, 1, True, True)`` and the print format --
follows the reference.
"""
import argparse
import contextlib
import io
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from . import synth
from .config import Config
from . import vit


def build_parser():
    """same flags and defaults as test_quant.py:18-53 (plus --synthetic/--n-val for the offline mode)."""
    p = argparse.ArgumentParser(description='P2-ViT PoT-PTQ on MI355X')
    p.add_argument('--model', default='deit_tiny')
    p.add_argument('--data', default='/home/ubuntu/imagenet')
    p.add_argument('--quant', default=False, action='store_true')
    p.add_argument('--ptf', default=True)
    p.add_argument('--lis', default=True)
    p.add_argument('--quant-method', default='minmax', choices=['minmax', 'ema', 'omse', 'percentile'])
    p.add_argument('--mixed', default=False, action='store_true')
    p.add_argument('--calib-batchsize', default=50, type=int, help='batchsize of calibration set')
    p.add_argument('--mode', default=1, type=int, help='calibration data: 1 Gaussian noise (offline default), 0/2 need real data')
    p.add_argument('--calib-iter', default=6, type=int)
    p.add_argument('--val-batchsize', default=50, type=int, help='batchsize of validation set')
    p.add_argument('--num-workers', default=16, type=int)
    p.add_argument('--device', default='cuda', type=str, help='device')
    p.add_argument('--print-freq', default=100, type=int, help='print frequency')
    p.add_argument('--seed', default=0, type=int, help='seed')
    p.add_argument('--synthetic', default=True, action='store_true', help='synthetic ImageNet-shaped data and random-init weights')
    p.add_argument('--real-data', default=False, action='store_true',
                   help='evaluate on the ImageFolder tree under --data (val/, and train/ for --mode 0 calibration) through the PIL port of '
                        'build_transform (data.py; its equality with torchvision is unpinned)')
    p.add_argument('--pretrained', default=False, action='store_true', help='like the reference (test_quant.py:95): the checkpoint of the '
                   'torch-hub cache, <TORCH_HOME>/hub/checkpoints/<file> (checkpoint.PRETRAINED_FILES); never downloaded here')
    p.add_argument('--checkpoint', default='', help='local .pth / .npz checkpoint (checkpoint.load_checkpoint); default: seeded synthetic weights')
    p.add_argument('--n-val', default=500, type=int, help='number of synthetic validation images')
    p.add_argument('--bits', default=8, type=int, choices=[4, 8], help='uniform bit_config for the validation run')
    p.add_argument('--calib-on', default='host', choices=['host', 'model'], help="where the calibration pass runs: 'host' reproduces the reference's exponents exactly")
    p.add_argument('--search-pop', default=25, type=int, help='--mixed: population size (test_quant.py:340)')
    p.add_argument('--search-iter', default=8, type=int, help='--mixed: evolutionary iterations (test_quant.py:343)')
    p.add_argument('--search-max-configs', default=50, type=int, help='--mixed: Pareto candidates kept (test_quant.py:281)')
    p.add_argument('--search-slack', default=1.1, type=float, help='--mixed: size constraint = slack x the all-4-bit model (test_quant.py:262)')
    return p


def str2model(name):
    """test_quant.py:56-68"""
    from . import swin
    d = {'deit_tiny': vit.deit_tiny_patch16_224, 'deit_small': vit.deit_small_patch16_224,
         'deit_base': vit.deit_base_patch16_224, 'vit_base': vit.vit_base_patch16_224,
         'vit_large': vit.vit_large_patch16_224, 'swin_tiny': swin.swin_tiny_patch4_window7_224,
         'swin_small': swin.swin_small_patch4_window7_224, 'swin_base': swin.swin_base_patch4_window7_224}
    print('Model: %s' % name)
    return d[name]


def _is_swin(model):
    from .swin import SwinTransformer
    return isinstance(model, SwinTransformer)


def _forward(model, data, bit_config=None):
    """``model(data, bit_config, plot)`` -> (output, FLOPs, distance).  The reference's Swin returns the logits alone and takes
    no bit_config (swin_quant.py:813-817): a uniform bit_config selects its weight width, FLOPs/distance stay empty."""
    if _is_swin(model):
        bits = int(bit_config[0]) if bit_config else 8
        return model(data, bits=bits), [], []
    return model(data, bit_config, False)


def seed(seed=0):
    """test_quant.py:71-86"""
    os.environ['PYTHONHASHSEED'] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


class AverageMeter(object):
    """test_quant.py:469-485"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """precision@k (test_quant.py:488-501)"""
    maxk = max(topk)
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0).mul_(100.0 / batch_size) for k in topk]


def calibrate_model(model, calibrate_data, where='host'):
    """the reference's calibration sequence (test_quant.py:235-249): one forward with calibrate + last_calibrate
    open, then close and switch to quant.

    ``where='host'`` (default): the one-off float pass and the observers' power-of-two searches run on the host CPU, whatever device
    the model lives on (it is moved there and back).  The searches pick the minimum of four nearly equal MSE scores per channel
    (minmax.py:198-240, ptf.py:96-134); only the host path reproduces the reference's choices exponent for exponent (DeiT-S: 495 of
    495 calibrated tensors, tests/test_module_surface.py and the GPU test of the same name), and the batched search makes it a
    couple of seconds instead of the reference's minute.  ``where='model'`` runs everything on the model's device (GPU: faster
    again, scores in fp64, but the float pass rounds differently there and a few hundred of 262 760 exponents move by one)."""
    if where not in ('host', 'model'):
        raise ValueError("where must be 'host' or 'model'")
    dev = next(model.parameters()).device
    on_host = where == 'host' and dev.type != 'cpu'
    if on_host:
        model.cpu()
        calibrate_data = calibrate_data.cpu()
    # the pass is hundreds of small tensor operations: on a many-core host torch's default thread count makes it an order of magnitude
    # slower (EPYC 9575F, DeiT-S: 14.4 s at 128 threads, 2.2 s at 32, 0.9 s at 16 - tools/calib_threads.py; the calibrated state is identical)
    n_threads = torch.get_num_threads()
    cpu_pass = where == 'host' or dev.type == 'cpu'
    if cpu_pass and n_threads > 16:
        torch.set_num_threads(16)
    try:
        model.model_open_calibrate()
        with torch.no_grad():
            model.model_open_last_calibrate()
            output, FLOPs, global_distance = _forward(model, calibrate_data)
        model.model_close_calibrate()
        model.model_quant()
    finally:
        if torch.get_num_threads() != n_threads:
            torch.set_num_threads(n_threads)
    if on_host:
        model.to(dev)
    return output, FLOPs, global_distance


class SyntheticLoader:
    """ImageNet-shaped batches from the deterministic generator; labels = argmax of a float teacher pass are supplied
    by the caller (there is no dataset offline), default labels are a fixed pseudo-random class per image."""

    def __init__(self, n, batch_size, img_size=224, num_classes=1000, seed=0, device='cpu', targets=None):
        self.n, self.bs, self.img, self.seed, self.device = n, batch_size, img_size, seed, device
        self.targets = targets if targets is not None else torch.from_numpy(
            (synth._stream(seed, 'labels', n) % np.uint64(num_classes)).astype(np.int64))

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(0, self.n, self.bs):
            k = min(self.bs, self.n - i)
            yield synth.images(self.seed, k, self.img, offset=i).to(self.device), self.targets[i:i + k].to(self.device)


class DevicePrefetcher:
    """``for data, target in DevicePrefetcher(loader, device)``: the batches of ``loader`` on ``device``, batch i + 1 copied while batch i
    runs.  The reference's loop does ``data.cuda()`` and then the forward, one after the other (test_quant.py:425-431); at 256 x 3 x 224^2 fp32
    the copy (154 MB, 2.8 ms at 55 GB/s) is longer than the DeiT-S forward (2.5 ms).  The copies run on ``engine.copy_stream`` - a stream
    probed to have a dispatch pipe of its own, which the sliced forward then leaves alone: 82 k img/s from pinned host memory against 45 k
    for copy-then-forward and 50 k for a double buffer on an arbitrary fifth stream (profiles/r04_pcie.txt).  Pinned batches
    (``DataLoader(pin_memory=True)``, as the reference's loaders are) copy asynchronously; pageable ones work, without the overlap."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        dev = self.device
        if dev.type != 'cuda':
            for data, target in self.loader:
                yield data.to(dev), target.to(dev)
            return
        from . import engine as E
        st = E.copy_stream(dev)
        try:
            ahead = None
            for data, target in self.loader:
                cur = torch.cuda.current_stream(dev)
                st.wait_stream(cur)                       # (the caching allocator may hand the copy a block the current stream just released)
                with torch.cuda.stream(st):
                    d, t = data.to(dev, non_blocking=True), target.to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(st)
                if ahead is not None:
                    yield self._hand_over(ahead, dev)
                ahead = (d, t, ev)
            if ahead is not None:
                yield self._hand_over(ahead, dev)
        finally:
            E.release_copy_stream(dev)                    # (also when the consumer stops early): all side streams serve the forward again

    @staticmethod
    def _hand_over(item, dev):
        d, t, ev = item
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(ev)
        d.record_stream(cur)
        t.record_stream(cur)
        return d, t


def validate(args, val_loader, model, criterion, device, bit_config=None):
    """test_quant.py:418-466; additionally returns images/sec of the forward calls."""
    batch_time, losses, top1, top5 = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    model.eval()
    val_start_time = end = time.time()
    n_img, fwd = 0, 0.0
    for i, (data, target) in enumerate(DevicePrefetcher(val_loader, device)):
        t0 = time.time()
        with torch.no_grad():
            output, FLOPs, distance = _forward(model, data, bit_config)
        if data.is_cuda:
            torch.cuda.synchronize()
        fwd += time.time() - t0
        n_img += data.size(0)
        loss = criterion(output, target)
        prec1, prec5 = accuracy(output.data, target, topk=(1, 5))
        losses.update(loss.data.item(), data.size(0))
        top1.update(prec1.data.item(), data.size(0))
        top5.update(prec5.data.item(), data.size(0))
        batch_time.update(time.time() - end)
        end = time.time()
        if i % args.print_freq == 0:
            print('Test: [{0}/{1}]\t'
                  'Time {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                  'Loss {loss.val:.4f} ({loss.avg:.4f})\t'
                  'Prec@1 {top1.val:.3f} ({top1.avg:.3f})\t'
                  'Prec@5 {top5.val:.3f} ({top5.avg:.3f})'.format(i, len(val_loader), batch_time=batch_time, loss=losses,
                                                                 top1=top1, top5=top5))
    val_end_time = time.time()
    print(' * Prec@1 {top1.avg:.3f} Prec@5 {top5.avg:.3f} Time {time:.3f}'.format(top1=top1, top5=top5,
                                                                              time=val_end_time - val_start_time))
    print(' * forward throughput %.1f images/sec' % (n_img / max(fwd, 1e-9)))
    return losses.avg, top1.avg, top5.avg


def main(argv=None):
    args = build_parser().parse_args(argv)
    seed(args.seed)
    device = torch.device(args.device)
    cfg = Config(args.ptf, args.lis, args.quant_method)
    model = str2model(args.model)(pretrained=args.pretrained, cfg=cfg)
    arch = model.arch
    if args.pretrained:
        pass
    elif _is_swin(model):
        model.load_state_dict(synth.swin_state_dict(model.state_dict(), args.seed))
    else:
        model.load_state_dict(synth.vit_state_dict(arch, args.seed), strict=False)
    if args.checkpoint:
        from .checkpoint import load_checkpoint
        load_checkpoint(model, args.checkpoint)
    model = model.to(device).eval()
    train_loader = None
    if args.real_data:
        # test_quant.py:118-144: ImageFolder val / train trees with the model family's mean / std / crop
        from .data import build_loaders
        loader, train_loader = build_loaders(args.data, args.model, args.val_batchsize, args.calib_batchsize, 0)
    else:
        # labels: the float model's own top-1 ("agreement with fp32"), the metric BASELINE.json names besides images/sec
        loader = SyntheticLoader(args.n_val, args.val_batchsize, arch['img_size'], arch['num_classes'], args.seed, device)
        with torch.no_grad():
            tgt = torch.cat([_forward(model, d)[0].argmax(1).cpu() for d, _ in loader])
        loader = SyntheticLoader(args.n_val, args.val_batchsize, arch['img_size'], arch['num_classes'], args.seed, device, tgt)
    criterion = nn.CrossEntropyLoss().to(device)
    bit_config = None
    if args.quant:
        if args.mode == 0 and train_loader is not None:
            print('Calibrating with real data...')                       # test_quant.py:214-233, mode 0: the first training batch
            calib_data = next(iter(train_loader))[0].to(device)
        else:
            print('Calibrating with Gaussian noise...')
            calib_data = synth.images(args.seed + 1, args.calib_batchsize, arch['img_size']).to(device)
        _, FLOPs, global_distance = calibrate_model(model, calib_data, where=args.calib_on)
        if args.mixed and not _is_swin(model):
            # test_quant.py:253-408 on the fast path: every candidate bit_config is one validate() over the HIP engine (the frozen
            # plan holds both weight widths per layer, so switching configurations costs nothing)
            from .search import mixed_precision_search
            quiet = argparse.Namespace(**{**vars(args), 'print_freq': 10 ** 9})

            def score(bc):
                with contextlib.redirect_stdout(io.StringIO()):
                    return validate(quiet, loader, model, criterion, device, bc)[1]
            ranked, pop = mixed_precision_search(score, FLOPs, global_distance, seed=args.seed, pop_size=args.search_pop,
                                                 evo_iter=args.search_iter, max_configs=args.search_max_configs, slack=args.search_slack)
            print('best mixed-precision configuration: Prec@1 %.3f' % pop[0][1])
            print(pop[0][0])
            return validate(args, loader, model, criterion, device, pop[0][0]) + (pop[0][0],)
        bit_config = [args.bits] * ((4 * arch['depth'] + 2) if 'depth' in arch else 1)
        print(bit_config)
    return validate(args, loader, model, criterion, device, bit_config)


if __name__ == '__main__':
    main()

"""GPU: largest |fast GELU - fp64 GELU| per input range (p2v_gelu_err_sweep), the bound behind the arithmetic GELU epilogue's margin test."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine; L = E.lib()
err = torch.zeros(1, dtype=torch.float32, device='cuda')
for lo_, hi_ in ((2.0**-20, 0.5), (0.5, 2.0), (2.0, 4.0), (4.0, 6.0), (6.0, 32.0)):
    for sign in (0, 0x80000000):
        err.zero_()
        lo, hi = np.float32(lo_).view(np.uint32), np.float32(hi_).view(np.uint32)
        E.check(L.p2v_gelu_err_sweep(int(lo) | sign, int(hi - lo), E.ptr(err), E.stream_ptr()))
        torch.cuda.synchronize()
        print('range %s[%g,%g): max |fast - exact| = %.3e' % ('-' if sign else '+', lo_, hi_, float(err.item())))
y = dva.synth.normal(9, 'gelu', (1 << 24,), 2.0).cuda()
for e in (3, 4, 5, 6):
    codes = torch.empty(y.numel(), dtype=torch.int8, device='cuda'); flags = torch.zeros(1, dtype=torch.int64, device='cuda')
    E.check(L.p2v_gelu_quant_f32(E.ptr(y), y.numel(), 2.0 ** e, E.ptr(codes), E.ptr(flags), 0, E.stream_ptr()))
    torch.cuda.synchronize()
    print('inv_s=2^%d: slow-path lanes %d of %d = %.2e' % (e, int(flags.item()), y.numel(), flags.item() / y.numel()))

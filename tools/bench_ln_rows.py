"""GPU: isolated launch time of the stand-alone LayerNorm for the wide models' shapes, per rows-per-wave setting (p2v_set_tuning ln_rows)
and with / without constants folded ahead (p2v_ln_prefold).   usage: python tools/bench_ln_rows.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine; L = E.lib(); S = dva.synth
for C_, rows in ((512, 25088), (768, 50432), (1024, 6272), (384, 50432)):
    codes = torch.clamp(torch.round(S.normal(3, 'x%d' % C_, (rows, C_), 35.0)), -128, 127).to(torch.int8).cuda()
    in_scale = 0.0123 * 2.0 ** torch.floor(S.uniform(3, 'm%d' % C_, (C_,), 0, 3.99))
    s1 = in_scale.min()
    dev = [t.contiguous().cuda() for t in (torch.round(in_scale / s1), S.uniform(3, 'g%d' % C_, (C_,), 0.5, 1.5), S.normal(3, 'b%d' % C_, (C_,), 0.3),
                                           torch.full((C_,), 16.0), torch.ones(C_))]
    ln = E.Ln(float(s1), *[E.ptr(t) for t in dev])
    nb = L.p2v_ln_prefold_bytes(C_); buf = torch.empty(nb // 4, device='cuda')
    out = torch.empty(rows, C_, dtype=torch.int8, device='cuda')
    for pre in (0, 1):
        if pre:
            E.check(L.p2v_ln_prefold(C.byref(ln), C_, E.ptr(buf), nb))
        for r in (2, 4, 8, 16, 32):
            L.p2v_set_tuning(b'ln_rows', r)
            for _ in range(5):
                L.p2v_int_layernorm(E.ptr(codes), C_, rows, C_, C.byref(ln), E.ptr(out), C_, E.stream_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                L.p2v_int_layernorm(E.ptr(codes), C_, rows, C_, C.byref(ln), E.ptr(out), C_, E.stream_ptr())
            e1.record(); torch.cuda.synchronize()
            print('C %4d rows %6d prefolded %d rows/wave %2d: %.2f us' % (C_, rows, pre, r, e0.elapsed_time(e1) * 1e3 / 50))
    L.p2v_set_tuning(b'ln_rows', 4)

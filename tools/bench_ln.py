"""time p2v_int_layernorm alone at the DeiT-S batch-256 shape (rows 50432, C 384); env P2V_LN_GENERIC=1 selects the generic chain."""
import sys, ctypes as C, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_vit_amd as dva
E, S = dva.engine, dva.synth
rows, C_ = 256 * 197, int(sys.argv[1]) if len(sys.argv) > 1 else 384
codes = torch.clamp(torch.round(S.normal(3, 'lx', (rows, C_)) * 35.0), -128, 127)
in_scale = 0.0123 * 2.0 ** torch.floor(S.uniform(3, 'lm', (C_,), 0, 3.99))
gamma = S.uniform(3, 'lg', (C_,), 0.2, 1.5); beta = S.normal(3, 'lb', (C_,), 0.3)
cs = 2.0 ** torch.floor(S.uniform(3, 'lc', (C_,), -2, 2.99))
s_a = 2.0 ** -4
out_scale = s_a * cs
s1 = in_scale.min()
dev = [t.contiguous().cuda() for t in (codes.to(torch.int8), torch.round(in_scale / s1), gamma, beta, 1.0 / out_scale, out_scale / cs / s_a)]
lnp = E.Ln(float(s1), *[E.ptr(t) for t in dev[1:]])
out = torch.zeros(rows, C_, dtype=torch.int8, device='cuda')
L = E.lib()
def run(n):
    for _ in range(n):
        E.check(L.p2v_int_layernorm(E.ptr(dev[0]), C_, rows, C_, C.byref(lnp), E.ptr(out), C_, E.stream_ptr()))
run(5); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(200); b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 1000 / 200
print('layernorm rows %d C %d: %.2f us/launch, %.0f GB/s (in+out)' % (rows, C_, us, 2 * rows * C_ / us / 1e3), 'checksum', int(out.int().sum()))

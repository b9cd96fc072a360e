"""GPU micro-benchmark + A/B parity of the fused LayerNorm+GEMM launch (p2v_ln_gemm_i8): the 4-wave kernel of round 2
("ln_gemm_version" 1) against the pipelined kernel with 4 waves (2) and 8 waves (3), same inputs, outputs compared bit for bit.
usage: python tools/bench_ln_gemm.py [images=86] [C=384] [iters=30]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine
L = E.lib()
images = int(sys.argv[1]) if len(sys.argv) > 1 else 86
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 384
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
M = images * 197
g = torch.Generator().manual_seed(1)
x = torch.randint(-128, 128, (M, Cc), dtype=torch.int8, generator=g).cuda()
vec = [torch.ones(Cc), torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1, torch.full((Cc,), 16.0), torch.ones(Cc)]
vec = [t.cuda() for t in vec]
ln = E.Ln(0.02, *[E.ptr(t) for t in vec])
for name, kind, N in (('qkv', E.EPI_REQUANT, 3 * Cc), ('fc1', E.EPI_GELU, 4 * Cc)):
    n_pad, k_pad = (N + 127) // 128 * 128, (Cc + 63) // 64 * 64
    w = torch.zeros(n_pad, k_pad, dtype=torch.int8)
    w[:N, :Cc] = torch.randint(-128, 128, (N, Cc), dtype=torch.int8, generator=g)
    cs = torch.full((n_pad,), 2.0 ** -12).cuda(); b = (torch.randn(n_pad, generator=g)).cuda()
    wf = E.fragment_order(w).cuda(); wd = w.cuda()
    lin = E.Linear(E.ptr(wd), E.ptr(cs), E.ptr(b), E.ptr(wf)); epi = E.Epilogue(); epi.inv_s_out = 2.0 ** 4
    if kind == E.EPI_GELU:
        epi.gelu = E.gelu_table(2.0 ** 4, 'cuda')
    outs = {}
    for ver in (1, 2, 3):
        E.check(L.p2v_set_tuning(b'ln_gemm_version', ver))
        out = torch.zeros(M, N, dtype=torch.int8, device='cuda')
        run = lambda: E.check(L.p2v_ln_gemm_i8(kind, E.ptr(x), Cc, M, Cc, C.byref(ln), N, C.byref(lin), C.byref(epi), E.ptr(out), N, None, E.stream_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        outs[ver] = out.cpu()
        print('%s  images %d  C %d  N %d  version %d: %.2f us / launch  (%.0f TOP/s)' % (name, images, Cc, N, ver, us, 2.0 * M * Cc * N / us / 1e6))
    for v in (2, 3):
        same = torch.equal(outs[1], outs[v])
        print('%s  v1 == v%d bit for bit: %s%s' % (name, v, same, '' if same else '  MISMATCHES %d of %d' % (int((outs[1] != outs[v]).sum()), outs[1].numel())))
E.check(L.p2v_set_tuning(b'ln_gemm_version', 2))

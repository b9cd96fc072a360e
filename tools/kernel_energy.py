"""GPU: joules per launch of the hot kernels.  The default step runs at the board's power limit (tools/smi_during_bench.sh: ~1365 W,
shader clock 2.15 GHz instead of 2.4), so what a kernel costs the step is its ENERGY, not its isolated time.  Each kernel kind is
launched back to back for a few seconds at DeiT-S shapes (all images of the batch in one launch) while rocm-smi is sampled:
energy per launch = average socket power x time per launch (idle socket power is printed for reference).
usage: python tools/kernel_energy.py [images=256] [seconds=4]"""
import ctypes as C, os, subprocess, sys, threading, time, re
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine
L = E.lib()
images = int(sys.argv[1]) if len(sys.argv) > 1 else 256
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
T, D, H, HID = 197, 384, 6, 1536
M = images * T
g = torch.Generator().manual_seed(1)
ri = lambda *s: torch.randint(-128, 128, s, dtype=torch.int8, generator=g).cuda()


class Smi(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True); self.stop = False; self.p = []; self.f = []
    def run(self):
        while not self.stop:
            o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
            p = re.search(r'Power \(W\): ([0-9.]+)', o); f = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', o)
            if p and f:
                self.p.append(float(p.group(1))); self.f.append(float(f.group(1)))
            time.sleep(0.2)


def measure(name, fn, work):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    smi = Smi(); smi.start()
    while time.perf_counter() - t0 < secs:
        for _ in range(200):
            fn()
        torch.cuda.synchronize(); n += 200
    dt = time.perf_counter() - t0
    smi.stop = True; smi.join()
    ps, fs = smi.p[1:], smi.f[1:]          # the first sample straddles the ramp
    pw, fq = (sum(ps) / len(ps), sum(fs) / len(fs)) if ps else (float('nan'), float('nan'))
    us = dt / n * 1e6
    print('%-12s %8.1f us/launch  %7.0f W  %5.0f MHz  %7.2f mJ/launch  (%d samples)  %s' % (name, us, pw, fq, pw * us * 1e-3, len(ps), work), flush=True)
    return pw * us * 1e-6


time.sleep(2)
o = subprocess.run(['rocm-smi', '--showpower'], capture_output=True, text=True).stdout
print('idle:', re.search(r'Power \(W\): ([0-9.]+)', o).group(0))
tot = 0.0
# fused LayerNorm + GEMM (version from the environment / default)
vec = [torch.ones(D), torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1, torch.full((D,), 16.0), torch.ones(D)]
vec = [t.cuda() for t in vec]
ln = E.Ln(0.02, *[E.ptr(t) for t in vec])
x = ri(M, D)
for name, kind, N in (('ln_gemm_qkv', E.EPI_REQUANT, 3 * D), ('ln_gemm_fc1', E.EPI_GELU, HID)):
    w = torch.randint(-128, 128, (N, D), dtype=torch.int8, generator=g)
    cs = torch.full((N,), 2.0 ** -12).cuda(); b = torch.randn(N, generator=g).cuda()
    wf = E.fragment_order(w).cuda(); wd = w.cuda()
    lin = E.Linear(E.ptr(wd), E.ptr(cs), E.ptr(b), E.ptr(wf)); epi = E.Epilogue(); epi.inv_s_out = 2.0 ** 4
    if kind == E.EPI_GELU:
        epi.gelu = E.gelu_table(2.0 ** 4, 'cuda')
    out = torch.zeros(M, N, dtype=torch.int8, device='cuda')
    fn = lambda: E.check(L.p2v_ln_gemm_i8(kind, E.ptr(x), D, M, D, C.byref(ln), N, C.byref(lin), C.byref(epi), E.ptr(out), N, None, E.stream_ptr()))
    tot += 12 * measure(name, fn, '%.1f GOP' % (2e-9 * M * D * N))
# attention
qkv = (torch.randn(M, 3 * D, generator=g) * 30).round().clamp(-128, 127).to(torch.int8).cuda()
s_q1, s_at, s_a2 = 2.0 ** -4, 2.0 ** -4, 2.0 ** -3
x0, bb, cc = dva.plan.lis_consts(torch.tensor([s_at]))
at = E.Attn(s_q1 * s_q1, 0.125, 1.0 / s_at, s_q1 / s_a2, x0, bb, cc)
ao = torch.zeros(M, D, dtype=torch.int8, device='cuda')
tot += 12 * measure('attention', lambda: E.check(L.p2v_lis_attention(E.ptr(qkv), images, T, H, 64, C.byref(at), E.ptr(ao), None, E.stream_ptr())), '%.1f GOP' % (4e-9 * images * H * T * T * 64))
# RESID GEMMs
res = ri(M, D)
for name, K in (('gemm_proj', D), ('gemm_fc2', HID)):
    a = ri(M, K)
    w = torch.randint(-128, 128, (D, K), dtype=torch.int8, generator=g).cuda()
    cs = torch.full((D,), 2.0 ** -13).cuda(); b = torch.randn(D, generator=g).cuda()
    sm, sr, sn = [(torch.rand(D, generator=g) * 0.02 + 0.01).cuda() for _ in range(3)]
    lin = E.Linear(E.ptr(w), E.ptr(cs), E.ptr(b), None)
    epi = E.Epilogue(); epi.s_mid, epi.s_res, epi.s_next, epi.residual = E.ptr(sm), E.ptr(sr), E.ptr(sn), E.ptr(res)
    out = torch.zeros(M, D, dtype=torch.int8, device='cuda')
    fn = lambda: E.check(L.p2v_gemm_i8(E.EPI_RESID, E.ptr(a), K, M, K, D, C.byref(lin), C.byref(epi), E.ptr(out), D, None, E.stream_ptr()))
    tot += 12 * measure(name, fn, '%.1f GOP' % (2e-9 * M * K * D))
print('12 blocks: %.2f J per %d images = %.2f mJ / image' % (tot, images, tot / images * 1e3))

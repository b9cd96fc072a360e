"""GPU diagnostic (diag build: make -C diff-vit_amd/csrc diag): per-workgroup phase stamps of one register-staged tiled GEMM launch
(k_gemm_i8_w4, fc1 shape).  The stamps exist only in libp2vit_hip_diag.so; the product library carries none."""
import ctypes as C, os, sys
os.environ['P2V_GEMM_STAGES'] = '0'          # the stamped kernel is the register-staged one
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine
E.LIB_PATH = os.path.join(ROOT, 'diff-vit_amd', 'csrc', 'libp2vit_hip_diag.so')
L = E.lib()
L.p2v_debug_set_gemm_stamps.argtypes = [C.c_void_p]; L.p2v_debug_set_gemm_stamps.restype = None
M, K, N = 50432, 384, int(sys.argv[1]) if len(sys.argv) > 1 else 1536
kind = int(sys.argv[2]) if len(sys.argv) > 2 else E.EPI_GELU
x = torch.randint(-128, 128, (M, K), dtype=torch.int8, device='cuda')
w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device='cuda')
cs = torch.full((N,), 2.0 ** -12, device='cuda'); b = torch.randn(N, device='cuda')
out = torch.empty(M, N, dtype=torch.int8, device='cuda')
lin = E.Linear(E.ptr(w), E.ptr(cs), E.ptr(b)); epi = E.Epilogue(); epi.inv_s_out = 2.0 ** 4
nblk = (M // 128) * (N // 128)
st = torch.zeros(nblk * 6, dtype=torch.int64, device='cuda')
for it in range(3):
    L.p2v_debug_set_gemm_stamps(C.c_void_p(st.data_ptr()) if it == 2 else None)
    E.check(L.p2v_gemm_i8(kind, E.ptr(x), K, M, K, N, C.byref(lin), C.byref(epi), E.ptr(out), N, None, E.stream_ptr()))
    torch.cuda.synchronize()
L.p2v_debug_set_gemm_stamps(None)
s = st.cpu().numpy().reshape(nblk, 6).astype(np.int64)
t0 = s[:, 0].min()
start, k0, k1, e1 = (s[:, i] - t0 for i in range(4))
print('blocks', nblk, 'kernel span (ticks)', (s[:, 3] - t0).max())
print('per-block mean ticks: prologue %.0f  k-loop %.0f  epilogue %.0f  total %.0f' % ((k0 - start).mean(), (k1 - k0).mean(), (e1 - k1).mean(), (e1 - start).mean()))
print('percentiles total', np.percentile(e1 - start, [5, 50, 95]))
print('percentiles prologue', np.percentile(k0 - start, [5, 50, 95]), 'kloop', np.percentile(k1 - k0, [5, 50, 95]), 'epi', np.percentile(e1 - k1, [5, 50, 95]))
hw = s[:, 5]; xcc = s[:, 4] & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 7)
# concurrency on one CU over time
one = np.where(cu == cu[0])[0]
print('blocks on the CU of block 0:', len(one))
ev = sorted([(start[i], 1) for i in one] + [(e1[i], -1) for i in one])
cur = 0; mx = 0
for t, d in ev:
    cur += d; mx = max(mx, cur)
print('max concurrent blocks on that CU', mx)
order = one[np.argsort(start[one])][:12]
for i in order:
    print('  blk %5d start %7d  prolog %6d kloop %6d epi %6d' % (i, start[i], k0[i] - start[i], k1[i] - k0[i], e1[i] - k1[i]))
print('starts per first 20k ticks histogram', np.histogram(start, bins=10)[0])

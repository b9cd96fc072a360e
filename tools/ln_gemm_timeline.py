"""GPU diagnostic (diag build: make -C diff-vit_amd/csrc diag): per-workgroup phase stamps of one fused LayerNorm+GEMM launch.
usage: python tools/ln_gemm_timeline.py [N=1536] [kind=1 (GELU) | 0 (REQUANT)] [images=83] [version=2]
(version 1 = the 4-wave kernel: stamps 4+2j / 5+2j after the k-loop / epilogue of column tile j; version 2 / 3 = the pipelined
kernel with 4 / 8 waves, wave 0 of group 0: stamp 4 after the first k-loop, 5+it after the it-th tile body)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine
E.LIB_PATH = os.path.join(ROOT, 'diff-vit_amd', 'csrc', 'libp2vit_hip_diag.so')       # the stamping build, never the product library
L = E.lib()
L.p2v_debug_set_gemm_stamps.argtypes = [C.c_void_p]; L.p2v_debug_set_gemm_stamps.restype = None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
kind = int(sys.argv[2]) if len(sys.argv) > 2 else E.EPI_GELU
M, Cc = (int(sys.argv[3]) if len(sys.argv) > 3 else 83) * 197, 384
ver = int(sys.argv[4]) if len(sys.argv) > 4 else 2
E.check(L.p2v_set_tuning(b'ln_gemm_version', ver))
x = torch.randint(-128, 128, (M, Cc), dtype=torch.int8, device='cuda')
w = torch.randint(-128, 128, (N, Cc), dtype=torch.int8, device='cuda')
cs = torch.full((N,), 2.0 ** -12, device='cuda'); b = torch.randn(N, device='cuda')
vec = [torch.ones(Cc, device='cuda'), torch.rand(Cc, device='cuda') + 0.5, torch.randn(Cc, device='cuda') * 0.1, torch.full((Cc,), 16.0, device='cuda'),
       torch.ones(Cc, device='cuda')]
ln = E.Ln(0.02, *[E.ptr(t) for t in vec])
out = torch.empty(M, N, dtype=torch.int8, device='cuda')
wf = E.fragment_order(w.cpu()).cuda()
lin = E.Linear(E.ptr(w), E.ptr(cs), E.ptr(b), E.ptr(wf)); epi = E.Epilogue(); epi.inv_s_out = 2.0 ** 4
if kind == E.EPI_GELU:
    epi.gelu = E.gelu_table(2.0 ** 4, 'cuda')
nblk = (M + 63) // 64
st = torch.zeros(nblk * 64, dtype=torch.int64, device='cuda')
for it in range(3):
    L.p2v_debug_set_gemm_stamps(C.c_void_p(st.data_ptr()) if it == 2 else None)
    E.check(L.p2v_ln_gemm_i8(kind, E.ptr(x), Cc, M, Cc, C.byref(ln), N, C.byref(lin), C.byref(epi), E.ptr(out), N, None, E.stream_ptr()))
    torch.cuda.synchronize()
L.p2v_debug_set_gemm_stamps(None)
s = st.cpu().numpy().reshape(nblk, 64).astype(np.int64)
tn = (N + 127) // 128
t0 = s[:, 0].min()
ph = lambda a_, b_: np.percentile(s[:, b_] - s[:, a_], [5, 50, 95]).astype(int)
if ver == 1:
    end = s[:, 5 + 2 * (tn - 1)]
    print('version 1  blocks', nblk, 'column tiles', tn, 'kernel span (ticks)', (end - t0).max())
    print('issue W + constants', ph(0, 1), ' LayerNorm', ph(1, 2), ' barrier', ph(2, 3))
    kl = np.stack([s[:, 4 + 2 * j] - (s[:, 3] if j == 0 else s[:, 3 + 2 * j]) for j in range(tn)], 1)
    ep = np.stack([s[:, 5 + 2 * j] - s[:, 4 + 2 * j] for j in range(tn)], 1)
    print('k-loop per column tile (median over blocks):', np.median(kl, 0).astype(int))
    print('epilogue per column tile (median):          ', np.median(ep, 0).astype(int))
else:
    ng = tn if ver == 2 else (tn + 1) // 2      # tiles of wave group 0 (version 2: one group; 3: two groups)
    end = s[:, 5 + ng - 1]
    print('version', ver, ' blocks', nblk, 'column tiles', tn, '(group 0:', ng, ') kernel span (ticks)', (end - t0).max())
    print('issue W + constants', ph(0, 1), ' LayerNorm', ph(1, 2), ' barrier', ph(2, 3), ' first k-loop', ph(3, 4))
    tb = np.stack([s[:, 5 + i] - s[:, 4 + i] for i in range(ng)], 1)
    print('tile bodies of group 0 (median over blocks):', np.median(tb, 0).astype(int))
print('total per block', np.percentile(end - s[:, 0], [5, 50, 95]).astype(int), ' start spread', np.percentile(s[:, 0] - t0, [50, 95, 100]).astype(int))

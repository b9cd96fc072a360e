"""GPU diagnostic (diag build): per-workgroup phase stamps of one tiled-GEMM launch (k_gemm_dma).
usage: python tools/gemm_timeline.py [M=50432] [K=768] [N=3072] [kind=1 GELU|0 REQUANT|2 RESID] [tile=0|128|256]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine
E.LIB_PATH = os.path.join(ROOT, 'diff-vit_amd', 'csrc', 'libp2vit_hip_diag.so')
L = E.lib()
L.p2v_debug_set_gemm_stamps.argtypes = [C.c_void_p]; L.p2v_debug_set_gemm_stamps.restype = None
M, K, N, kind, tile = [int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 50432), (2, 768), (3, 3072), (4, 1), (5, 0))]
E.check(L.p2v_set_tuning(b'gemm_tile', tile))
x = torch.randint(-128, 128, (M, K), dtype=torch.int8, device='cuda')
w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device='cuda')
cs = torch.full((N,), 2.0 ** -12, device='cuda'); b = torch.randn(N, device='cuda')
lin = E.Linear(E.ptr(w), E.ptr(cs), E.ptr(b), None, 0); epi = E.Epilogue(); epi.inv_s_out = 2.0 ** 4
keep = []
if kind == E.EPI_GELU:
    epi.gelu = E.gelu_table(2.0 ** 4, 'cuda')
if kind == E.EPI_RESID:
    keep = [(torch.rand(N, device='cuda') * 0.02 + 0.01) for _ in range(3)] + [torch.randint(-128, 128, (M, N), dtype=torch.int8, device='cuda')]
    epi.s_mid, epi.s_res, epi.s_next, epi.residual = [E.ptr(t) for t in keep]
out = torch.empty(M, N, dtype=torch.int8, device='cuda')
nblk = ((M + 127) // 128) * ((N + 127) // 128)
st = torch.zeros(nblk * 8, dtype=torch.int64, device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    L.p2v_debug_set_gemm_stamps(C.c_void_p(st.data_ptr()) if it == 3 else None)
    if it == 2: ev[0].record()
    E.check(L.p2v_gemm_i8(kind, E.ptr(x), K, M, K, N, C.byref(lin), C.byref(epi), E.ptr(out), N, None, E.stream_ptr()))
    if it == 2: ev[1].record()
    torch.cuda.synchronize()
L.p2v_debug_set_gemm_stamps(None)
s = st.cpu().numpy().reshape(nblk, 8).astype(np.int64)
s = s[s[:, 4] > 0]
ph = lambda a_, b_: np.percentile(s[:, b_] - s[:, a_], [5, 50, 95]).astype(int)
print('M %d K %d N %d kind %d tile %d: %.1f us / launch (%.0f TOP/s), %d workgroups stamped' % (M, K, N, kind, tile, ev[0].elapsed_time(ev[1]) * 1e3,
      2.0 * M * K * N / ev[0].elapsed_time(ev[1]) / 1e9, len(s)))
print('prologue', ph(0, 1), ' k-loop', ph(1, 2), '(per k-tile %d)' % (np.median(s[:, 2] - s[:, 1]) / (K // 64)), ' barrier', ph(2, 3), ' epilogue', ph(3, 4), ' total', ph(0, 4))
print('kernel span %d ticks' % (s[:, 4].max() - s[:, 0].min()))

import ctypes as C, os, sys, numpy as np, torch
os.environ['P2V_GEMM_RESIDENT'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine; L = E.lib()
L.p2v_debug_set_gemm_stamps.argtypes = [C.c_void_p]; L.p2v_debug_set_gemm_stamps.restype = None
M, K, N = 50432, 384, int(sys.argv[1]) if len(sys.argv) > 1 else 1536
kind = int(sys.argv[2]) if len(sys.argv) > 2 else E.EPI_GELU
x = torch.randint(-128, 128, (M, K), dtype=torch.int8, device='cuda'); w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device='cuda')
cs = torch.full((N,), 2.0 ** -12, device='cuda'); b = torch.randn(N, device='cuda'); out = torch.empty(M, N, dtype=torch.int8, device='cuda')
lin = E.Linear(E.ptr(w), E.ptr(cs), E.ptr(b)); epi = E.Epilogue(); epi.inv_s_out = 2.0 ** 4
st = torch.zeros(8192 + 16 * 8 * 32, dtype=torch.int64, device='cuda')
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    L.p2v_debug_set_gemm_stamps(C.c_void_p(st.data_ptr()) if it == 2 else None)
    ev0.record()
    E.check(L.p2v_gemm_i8(kind, E.ptr(x), K, M, K, N, C.byref(lin), C.byref(epi), E.ptr(out), N, None, E.stream_ptr()))
    ev1.record(); torch.cuda.synchronize()
    print('launch %d: %.1f us' % (it, ev0.elapsed_time(ev1) * 1e3))
L.p2v_debug_set_gemm_stamps(None)
allst = st.cpu().numpy().astype(np.int64)
s = allst[:8192].reshape(256, 8, 4)
pw = allst[8192:].reshape(16, 8, 32)
ok = s[:, :, 0] > 0
main = (s[:, :, 1] - s[:, :, 0])[ok]; epi_ = (s[:, :, 2] - s[:, :, 1])[ok]; bar = (s[:, :, 3] - s[:, :, 2])[ok]
nxt = (s[:, 1:, 0] - s[:, :-1, 3])[ok[:, 1:] & ok[:, :-1]]
print('iterations stamped', ok.sum())
for nm, v in (('mainloop (incl. wait for barrier 1 entry)', main), ('epilogue', epi_), ('barrier2 wait (DMA + stragglers)', bar), ('stores+loop overhead', nxt)):
    print('%-45s mean %7.0f  p5 %7.0f p50 %7.0f p95 %7.0f' % (nm, v.mean(), *np.percentile(v, [5, 50, 95])))
print('block 0 iterations:'); print(s[0] - s[0, 0, 0])

for blk in (0, 5):
    for it in (1, 2):
        b1 = s[blk, it, 1]
        print('block %d it %d: per-wave epilogue end (cycles after barrier 1):' % (blk, it), (pw[blk, it, :16] - b1).tolist())
        print('              per-wave vmcnt(0) satisfied:', (pw[blk, it, 16:] - b1).tolist())

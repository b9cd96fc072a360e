"""GPU diagnostic (diag build: make -C diff-vit_amd/csrc diag): phase stamps (wave 0 of every workgroup) of one k_lis_attention launch.
usage: python tools/attn_timeline.py [images=85]"""
import ctypes as C, math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import diff_vit_amd as dva
E = dva.engine
E.LIB_PATH = os.environ.get('P2V_DIAG_LIB', os.path.join(ROOT, 'diff-vit_amd', 'csrc', 'libp2vit_hip_diag.so'))   # the stamping build, never the product library
L = E.lib()
L.p2v_debug_set_gemm_stamps.argtypes = [C.c_void_p]; L.p2v_debug_set_gemm_stamps.restype = None
B, N, H, hd = int(sys.argv[1]) if len(sys.argv) > 1 else 85, 197, 6, 64
D = H * hd
qkv = torch.clamp(torch.round(torch.randn(B, N, 3 * D) * 30), -128, 127).to(torch.int8).cuda()
s_q1, s_at, s_a2 = 2.0 ** -4, 2.0 ** -4, 2.0 ** -3
x0 = math.floor(-0.6931 / s_at); bb = math.floor((0.96963238 / 0.35815147) / s_at); cc = math.floor((1 / 0.35815147) / s_at ** 2)
at = E.Attn(s_q1 * s_q1, float(np.float32(hd ** -0.5)), 1.0 / s_at, s_q1 / s_a2, x0, bb, cc)
out = torch.zeros(B * N, D, dtype=torch.int8, device='cuda')
nblk = B * H
st = torch.zeros(nblk * 16, dtype=torch.int64, device='cuda')
for it in range(3):
    L.p2v_debug_set_gemm_stamps(C.c_void_p(st.data_ptr()) if it == 2 else None)
    E.check(L.p2v_lis_attention(E.ptr(qkv), B, N, H, hd, C.byref(at), E.ptr(out), None, E.stream_ptr()))
    torch.cuda.synchronize()
L.p2v_debug_set_gemm_stamps(None)
s = st.cpu().numpy().reshape(nblk, 16).astype(np.int64)
ph = lambda a_, b_: np.percentile(s[:, b_] - s[:, a_], [5, 50, 95]).astype(int)
print('workgroups', nblk, ' kernel span', int((s[:, 11] - s[:, 0].min()).max()))
print('exp table', ph(0, 1), ' K/V staging', ph(1, 2), ' first Q + barrier', ph(2, 3))
for blk, o in (('query block 0', 4), ('query block 8', 8)):
    print(blk, ': scores (Q wait + 14 MFMA)', ph(o - 1, o), ' codes, max, exp sum', ph(o, o + 1), ' quotients, 2^-k, P.V', ph(o + 1, o + 2), ' requant + store', ph(o + 2, o + 3))
print('wave 0 total', ph(0, 11), ' start spread', np.percentile(s[:, 0] - s[:, 0].min(), [50, 95, 100]).astype(int))

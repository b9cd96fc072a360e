"""ctypes binding of the C-ABI engine ``csrc/libp2vit_hip.so`` (declared in ``include/p2vit.h``).

The library is the product: there is no eager/CPU fallback.  Anything that needs the quantized forward
calls :func:`lib`, which raises ``RuntimeError`` when the shared object has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C diff-vit_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libp2vit_hip.so')

P2V_ABI_VERSION = 5
EPI_REQUANT, EPI_GELU, EPI_RESID, EPI_EMBED, EPI_HEAD = 0, 1, 2, 3, 4
E_ARG, E_BITS, E_SHAPE, E_UNSUPPORTED, E_WORKSPACE, E_LAUNCH, E_STATE = -1, -2, -3, -4, -5, -6, -7

KERNEL_KINDS = ('patchify', 'gemm_embed', 'fill_cls', 'layernorm', 'gemm_qkv', 'attention', 'gemm_proj', 'gemm_fc1',
                'gemm_fc2', 'gemm_head', 'ln_gemm_qkv', 'ln_gemm_fc1', 'event_gap')

_f, _i, _p, _ll = C.c_float, C.c_int32, C.c_void_p, C.c_longlong


class ModelDesc(C.Structure):
    _fields_ = [('abi_version', _i), ('img_size', _i), ('patch_size', _i), ('in_chans', _i), ('embed_dim', _i),
                ('depth', _i), ('num_heads', _i), ('mlp_hidden', _i), ('num_classes', _i)]


class Linear(C.Structure):
    _fields_ = [('w_codes', _p), ('colscale', _p), ('bias', _p), ('w_frag', _p), ('packed4', _i)]


class LnPre(C.Structure):
    """``p2v_ln_pre``: LayerNorm constants folded ahead of the launches (filled by ``p2v_ln_prefold``)."""
    _fields_ = [('gm', _p), ('bt', _p), ('gmin', _f), ('gmax', _f), ('bmax', _f), ('pot', _i), ('pm_one', _i)]


class Ln(C.Structure):
    _fields_ = [('s1', _f), ('mask', _p), ('gamma', _p), ('beta', _p), ('inv_out', _p), ('post_mul', _p), ('out_scale', _p), ('pre', LnPre)]


class Attn(C.Structure):
    _fields_ = [('s_qkv_sq', _f), ('qk_scale', _f), ('inv_s_attn', _f), ('av_mul', _f), ('x0_int', _i), ('b_int', _i),
                ('c_int', _i)]


class WinAttn(C.Structure):
    _fields_ = [('s_q1', _f), ('qk_scale', _f), ('s_attn', _f), ('s_table', _f), ('s_q2', _f), ('s_q3', _f), ('x0_int', _i),
                ('b_int', _i), ('c_int', _i), ('table_codes', _p), ('win_index', _p), ('region', _p), ('ws', _i), ('n_windows', _i), ('qkv_stride', _i), ('out_stride', _i)]


class GeluTab(C.Structure):
    """exact GELU -> requant threshold table (``p2v_gelu_tab``); ``table`` None = arithmetic evaluation."""
    _fields_ = [('table', _p), ('k', _f), ('off', _f), ('cells', _i)]


class Epilogue(C.Structure):
    _fields_ = [('inv_s_out', _f), ('s_out', _f), ('s_mid', _p), ('s_res', _p), ('s_next', _p), ('residual', _p),
                ('inv_s_pe', _f), ('pe_to_embed', _f), ('s_embed', _f), ('pos_deq', _p), ('patches', _i), ('gelu', GeluTab),
                ('tap_out', _p), ('resid_tab', _p)]


class Block(C.Structure):
    _fields_ = [('ln1', Ln * 2), ('inv_s_qkv', _f * 2), ('attn', Attn), ('proj_epi', Epilogue), ('ln2', (Ln * 2) * 2),
                ('inv_s_fc1', _f), ('gelu_fc1', GeluTab), ('fc2_epi', Epilogue)]


OP_PATCHIFY, OP_GEMM, OP_LAYERNORM, OP_WINATTN, OP_MERGE, OP_AVGPOOL, OP_LN_GEMM = range(7)


class Op(C.Structure):
    """one record of a replayable launch sequence (``p2v_op`` in include/p2vit.h)."""
    _fields_ = [('kind', _i), ('epi', _i), ('inp', _p), ('out', _p), ('M', _i), ('K', _i), ('N', _i), ('lda', _i), ('ldo', _i),
                ('i0', _i), ('i1', _i), ('i2', _i), ('i3', _i), ('i4', _i), ('i5', _i), ('f0', _f), ('f1', _f),
                ('lin', Linear), ('ep', Epilogue), ('ln', Ln), ('wa', WinAttn)]


class P2VError(RuntimeError):
    pass


_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """Load (once) and return the engine library; loud failure when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('HIP engine %s is not built: run `make -C %s` (or __graft_entry__.build()). '
                           'The quantized forward has no CPU fallback.' % (LIB_PATH, os.path.dirname(LIB_PATH)))
    L = C.CDLL(LIB_PATH)
    L.p2v_last_error.restype = C.c_char_p
    L.p2v_abi_version.restype = _i
    L.p2v_plan_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(_p)]
    L.p2v_plan_destroy.argtypes = [_p]
    L.p2v_plan_destroy.restype = None
    L.p2v_plan_set_linear.argtypes = [_p, _i, _i, C.POINTER(Linear)]
    L.p2v_plan_set_embed.argtypes = [_p, _f, C.POINTER(Epilogue), _p]
    L.p2v_plan_set_block.argtypes = [_p, _i, C.POINTER(Block)]
    L.p2v_plan_set_head.argtypes = [_p, C.POINTER(Ln), _f, _f]
    L.p2v_plan_block_prefolded.argtypes = [_p, _i]
    L.p2v_plan_resid_prefolded.argtypes = [_p, _i]
    L.p2v_ln_prefold_bytes.argtypes = [_i]
    L.p2v_ln_prefold_bytes.restype = C.c_size_t
    L.p2v_ln_prefold.argtypes = [C.POINTER(Ln), _i, _p, C.c_size_t]
    L.p2v_max_tokens.argtypes = [_i]
    L.p2v_resident_tokens.argtypes = [_i]
    L.p2v_resid_prefold_bytes.argtypes = [_i]
    L.p2v_resid_prefold_bytes.restype = C.c_size_t
    L.p2v_resid_prefold.argtypes = [C.POINTER(Linear), C.POINTER(Epilogue), _i, _p, C.c_size_t, C.POINTER(_i), _p]
    L.p2v_workspace_bytes.argtypes = [_p, _i]
    L.p2v_workspace_bytes.restype = C.c_size_t
    L.p2v_workspace_view.argtypes = [_p, _i, C.c_char_p]
    L.p2v_workspace_view.restype = _ll
    L.p2v_forward.argtypes = [_p, _p, _i, C.POINTER(C.c_int8), _i, _p, _p, C.c_size_t, _i, _p]
    L.p2v_forward_taps.argtypes = [_p, _p, _i, C.POINTER(C.c_int8), _i, _p, _p, C.c_size_t, C.POINTER(_p), C.POINTER(_p), _p]
    L.p2v_forward_profile.argtypes = [_p, _p, _i, C.POINTER(C.c_int8), _i, _p, _p, C.c_size_t, _p, C.POINTER(C.c_float),
                                      C.POINTER(C.c_int32), _i]
    L.p2v_forward_profile_begin.argtypes = [_p, _p, _i, C.POINTER(C.c_int8), _i, _p, _p, C.c_size_t, _p, C.POINTER(_p)]
    L.p2v_forward_profile_end.argtypes = [_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), _i]
    L.p2v_quantize_patchify.argtypes = [_p, _i, _i, _i, _i, _i, _f, _p, _i, _p]
    L.p2v_gemm_i8.argtypes = [_i, _p, _i, _i, _i, _i, C.POINTER(Linear), C.POINTER(Epilogue), _p, _i, _p, _p]
    L.p2v_int_layernorm.argtypes = [_p, _ll, _i, _i, C.POINTER(Ln), _p, _ll, _p]
    L.p2v_lis_attention.argtypes = [_p, _i, _i, _i, _i, C.POINTER(Attn), _p, _p, _p]
    L.p2v_ln_gemm_i8.argtypes = [_i, _p, _ll, _i, _i, C.POINTER(Ln), _i, C.POINTER(Linear), C.POINTER(Epilogue), _p, _i, _p, _p]
    L.p2v_ln_gemm_fusable.argtypes = [_i, _i, _i, _i]
    L.p2v_set_tuning.argtypes = [C.c_char_p, _i]
    L.p2v_ln_gemm_fusable.restype = _i
    L.p2v_run_ops.argtypes = [C.POINTER(Op), _i, _p]
    L.p2v_run_ops_profile.argtypes = [C.POINTER(Op), _i, _p, C.POINTER(C.c_float)]
    L.p2v_patch_merge_gather.argtypes = [_p, _i, _i, _i, _i, _p, _p]
    L.p2v_avgpool_quant.argtypes = [_p, _i, _i, _i, _f, _f, _p, _p]
    L.p2v_window_attention.argtypes = [_p, _i, _i, _i, _i, C.POINTER(WinAttn), _p, _p, _p]
    L.p2v_fake_quant_f32.argtypes = [_p, _ll, _p, _i, _ll, _i, _i, _p, _p, _p]
    L.p2v_gelu_quant_f32.argtypes = [_p, _ll, _f, _p, _p, _i, _p]
    L.p2v_gelu_err_sweep.argtypes = [C.c_uint32, C.c_uint32, _p, _p]
    L.p2v_stream_probe.argtypes = [_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.p2v_gelu_table_plan.argtypes = [_f, C.POINTER(GeluTab)]
    L.p2v_gelu_table_scratch_bytes.argtypes = [_i]
    L.p2v_gelu_table_scratch_bytes.restype = C.c_size_t
    L.p2v_gelu_table_build.argtypes = [_f, C.POINTER(GeluTab), _p, C.c_size_t, _p]
    L.p2v_gelu_table_check.argtypes = [_f, C.POINTER(GeluTab), _p, _p]
    if L.p2v_abi_version() != P2V_ABI_VERSION:
        raise RuntimeError('libp2vit_hip.so ABI %d != binding %d: rebuild' % (L.p2v_abi_version(), P2V_ABI_VERSION))
    _lib = L
    return L


def check(rc):
    """Map C status codes onto the exceptions the reference raises at the same places."""
    if rc == 0:
        return
    msg = lib().p2v_last_error().decode()
    if rc == E_BITS:
        raise ValueError(msg)                      # bit_pool.index(bit) -> ValueError (vit_fquant.py:282)
    if rc == E_SHAPE:
        raise AssertionError(msg)                  # PatchEmbed size assert (layers_quant.py:437-439)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)             # quantizer/base.py:28-30
    raise P2VError('p2vit error %d: %s' % (rc, msg))


def stream_ptr(device=None):
    """the caller's current HIP stream ON ``device`` (default: the current device)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_GELU_TABLES = {}     # (device index, 1/scale) -> (GeluTab, table tensor): built once per process and device


def gelu_table(inv_s, device):
    """threshold table of GELU -> QAct for the power-of-two multiplier ``inv_s`` on ``device`` (p2v_gelu_table_build: an
    exhaustive fp32 sweep on the GPU, ~0.1 s, cached), or an empty descriptor when the scale has no table."""
    import torch
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), float(inv_s))
    if key not in _GELU_TABLES:
        L = lib()
        t = GeluTab()
        if L.p2v_gelu_table_plan(float(inv_s), C.byref(t)) != 0:
            _GELU_TABLES[key] = (GeluTab(), None)
        else:
            with torch.cuda.device(device):
                tab = torch.empty(t.cells * 2, dtype=torch.int32, device=device)
                scratch = torch.empty(L.p2v_gelu_table_scratch_bytes(t.cells), dtype=torch.uint8, device=device)
                t.table = ptr(tab)
                rc = L.p2v_gelu_table_build(float(inv_s), C.byref(t), ptr(scratch), scratch.numel(), stream_ptr(device))
            if rc == E_UNSUPPORTED:
                _GELU_TABLES[key] = (GeluTab(), None)
            else:
                check(rc)
                _GELU_TABLES[key] = (t, tab)
    return _GELU_TABLES[key][0]


def fragment_order(wp):
    """int8 weight codes [n_pad (x128)][k_pad (x64)] -> the MFMA-fragment order of ``p2v_linear.w_frag``:
    [column tile][wave][k-step of 32][lane = 32*h + r][16 bytes] with W[128*tile + 32*wave + r][32*kstep + 16*h + b]."""
    n_pad, k_pad = wp.shape
    assert n_pad % 128 == 0 and k_pad % 64 == 0
    return wp.reshape(n_pad // 128, 4, 32, k_pad // 32, 2, 16).permute(0, 1, 3, 4, 2, 5).contiguous()


def fragment_order_packed4(wp):
    """4-bit codes in [-8, 7], [n_pad][k_pad] -> ``p2v_linear.w_frag`` for ``packed4`` layers: the fragment order of
    :func:`fragment_order` with the 16 codes of a lane in 8 bytes (byte j of dword 0 = code[j] | code[4+j] << 4, of dword 1 =
    code[8+j] | code[12+j] << 4: the chunk format of :func:`pack_int4_tiles`); uint8 [.., 64 lanes, 8]."""
    import torch
    assert int(wp.min()) >= -8 and int(wp.max()) <= 7
    f = (fragment_order(wp).to(torch.int16) & 15).to(torch.uint8)              # [tile, wave, kstep, h, r, 16]
    lo = f[..., [0, 1, 2, 3, 8, 9, 10, 11]]
    hi = f[..., [4, 5, 6, 7, 12, 13, 14, 15]]
    return (lo | (hi << 4)).contiguous()


def pack_int4_tiles(wp):
    """int8 codes in [-8, 7], [n_pad (x128)][k_pad (x64)] -> the packed layout of ``p2v_linear.packed4`` (include/p2vit.h): uint8
    [n_pad/128][k_pad/64][128][32]: two codes per byte, the LDS image of every weight tile (chunk swizzle included)."""
    import torch
    n_pad, k_pad = wp.shape
    assert n_pad % 128 == 0 and k_pad % 64 == 0 and int(wp.min()) >= -8 and int(wp.max()) <= 7
    w = (wp.to(torch.int16) & 15).to(torch.uint8).reshape(n_pad // 128, 128, k_pad // 64, 4, 16)       # [tile, row, ktile, chunk, 16 codes]
    lo = w[..., [0, 1, 2, 3, 8, 9, 10, 11]]
    hi = w[..., [4, 5, 6, 7, 12, 13, 14, 15]]
    b = lo | (hi << 4)                                                                           # [tile, row, ktile, chunk, 8 bytes]
    rows = torch.arange(128)
    src = torch.arange(4).reshape(1, 4) ^ ((rows >> 3) & 3).reshape(128, 1)                      # position c' holds chunk c' ^ f(row)
    b = torch.gather(b, 3, src.reshape(1, 128, 1, 4, 1).expand(b.shape[0], 128, b.shape[2], 4, 8))
    return b.permute(0, 2, 1, 3, 4).reshape(n_pad // 128, k_pad // 64, 128, 32).contiguous()


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


# ---- side HIP streams of the batch slicing: ONE probed set per device for the whole process ------------------------------------------
# A HIP stream gets a hardware queue when it first carries work, in order (GPU_MAX_HW_QUEUES of them, then round again), and the queues are
# served by FOUR dispatch pipes: queue i and queue i + 4 share one.  Two streams with work in them at the same time on one pipe do not just
# serialise - every kernel waits for a queue switch, and the sliced forward falls BELOW the one-stream rate (profiles/r04_stream_pool.txt:
# DeiT-S 101 k img/s -> 57 - 68 k as soon as the process had used one to three other streams before the plan took its own; 82 k on one stream).
# So (a) plans do not own streams, they borrow this one set, and (b) the set is PROBED: candidates are kept only if a train of timed
# kernels (p2v_stream_probe: 64 workgroups that wait 20 us) on them runs beside the same train on the caller's stream and on the streams already kept.
import threading as _threading
_SIDE_STREAMS = {}
_SIDE_LOCK = _threading.Lock()   # the pool is chosen once per device, whichever host thread asks first
SIDE_STREAM_REPORT = {}          # per device index: what the probe saw (bench.py prints it)
MAX_SIDE_STREAMS = 3


def _pair_ms(a, b, kernels=10, usec=20, wgs=64, lds=0):
    """wall time (ms) of a train of timed kernels on stream ``a`` and the same train on ``b``, enqueued together; best of three tries (host wall clock)"""
    import time
    import torch
    L = lib()
    best = 1e9
    for _ in range(3):
        a.synchronize(); b.synchronize()
        t0 = time.perf_counter()
        check(L.p2v_stream_probe(C.c_void_p(a.cuda_stream), kernels, usec, wgs, lds))
        check(L.p2v_stream_probe(C.c_void_p(b.cuda_stream), kernels, usec, wgs, lds))
        a.synchronize(); b.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def side_streams(device, n):
    """the first ``n`` shared side streams of ``device`` (useful: up to MAX_SIDE_STREAMS); chosen once, on first use, among up to twelve
    candidates, by the probe above against the stream that is current at that moment (the caller's) and against each other."""
    import torch
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _SIDE_STREAMS.get(idx)
    if pool is None and torch.cuda.is_current_stream_capturing():
        # a first call inside a graph capture cannot synchronise, i.e. cannot probe: plain streams for this capture, the pool stays unchosen
        return [torch.cuda.Stream(device=idx) for _ in range(n)]
    if pool is None:
        with _SIDE_LOCK:
            pool = _SIDE_STREAMS.get(idx)
            if pool is None:
                pool = _SIDE_STREAMS[idx] = _choose_side_streams(idx)
    if n > MAX_SIDE_STREAMS:
        import warnings
        warnings.warn('%d side streams: with the caller\'s stream more than four streams carry work, and the step collapses to about half its '
                      'speed (profiles/r04_slices.txt); %d is the useful maximum' % (n, MAX_SIDE_STREAMS), RuntimeWarning, stacklevel=3)
        with _SIDE_LOCK:
            while len(pool) < n:
                pool.append(torch.cuda.Stream(device=idx))
    return pool[:n]


def _choose_side_streams(idx):
    """probe up to twelve candidate streams against the caller's stream and against each other (see above); ~10 ms, once"""
    import torch
    with torch.cuda.device(idx):
        cur = torch.cuda.current_stream(idx)
        kernels, usec, wgs = 10, 20, 64
        alone = kernels * usec * 1e-3                       # one train: 0.2 ms; measured: 0.26 - 0.27 ms for two trains side by side,
        limit = 2.0 * alone                                 # 0.55 - 0.60 ms when the two queues share a pipe or the streams a queue
        pool, tried, rejected = [], 0, []
        while len(pool) < MAX_SIDE_STREAMS and tried < 12:
            st = torch.cuda.Stream(device=idx)
            tried += 1
            worst = max(_pair_ms(st, o, kernels, usec, wgs) for o in [cur] + pool)
            if worst <= limit:
                pool.append(st)
            else:
                rejected.append(round(worst, 3))
        probed = len(pool)
        while len(pool) < MAX_SIDE_STREAMS:                 # nothing better found: plain streams (and say so)
            pool.append(torch.cuda.Stream(device=idx))
        SIDE_STREAM_REPORT[idx] = {'candidates': tried, 'kept_by_probe': probed, 'rejected_pair_ms': rejected, 'limit_ms': round(limit, 3)}
        if probed < MAX_SIDE_STREAMS:
            import warnings
            warnings.warn('diff_vit_amd: only %d of %d side streams run beside the caller\'s stream on device %d (other streams of the '
                          'process hold the hardware queues): the sliced forward may run below the one-stream rate; pass n_streams=1'
                          % (probed, MAX_SIDE_STREAMS, idx), RuntimeWarning, stacklevel=3)
    return pool


_COPY_STREAM = {}          # device index -> number of input pipelines currently copying on the pool's third side stream


def _dev_index(device):
    import torch
    dev = torch.device(device)
    return dev.index if dev.index is not None else torch.cuda.current_device()


def copy_stream(device):
    """the stream for the host -> device copies of an input pipeline that prefetches the next batch while this one runs
    (``harness.DevicePrefetcher``): the pool's THIRD side stream, i.e. one that was probed to have a dispatch pipe of its own.  Until the
    matching ``release_copy_stream`` the sliced forward of the device keeps to two side streams + the caller's (three slices in flight): a
    fifth busy stream has to share a pipe with a slice, and the double-buffered DeiT-S loop then runs at 50 k img/s instead of 84 k (copy
    bound at 53 - 57 GB/s: 88 - 95 k; profiles/r04_pcie.txt)."""
    idx = _dev_index(device)
    st = side_streams(idx, MAX_SIDE_STREAMS)[MAX_SIDE_STREAMS - 1]
    _COPY_STREAM[idx] = _COPY_STREAM.get(idx, 0) + 1
    return st


def release_copy_stream(device):
    """the input pipeline is done: the sliced forward may use all side streams again"""
    idx = _dev_index(device)
    if _COPY_STREAM.get(idx, 0) > 0:
        _COPY_STREAM[idx] -= 1


def compute_side_streams(device):
    """side streams the sliced forward may use on ``device``: MAX_SIDE_STREAMS, one less while a ``copy_stream`` is handed out"""
    return MAX_SIDE_STREAMS - (1 if _COPY_STREAM.get(_dev_index(device), 0) > 0 else 0)


# ---- host threads that enqueue the side streams' slices (FrozenPlan.forward_streams) ----------------------------------------------------
THREADED_ENQUEUE = os.environ.get('P2V_THREADED_ENQUEUE', '1') != '0'
_ENQUEUE_POOL = None


def enqueue_pool():
    """MAX_SIDE_STREAMS daemon worker threads, created on first use"""
    global _ENQUEUE_POOL
    if _ENQUEUE_POOL is None:
        with _SIDE_LOCK:
            if _ENQUEUE_POOL is None:
                from concurrent.futures import ThreadPoolExecutor
                _ENQUEUE_POOL = ThreadPoolExecutor(MAX_SIDE_STREAMS, thread_name_prefix='p2v-enqueue')
    return _ENQUEUE_POOL

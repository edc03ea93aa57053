"""ctypes binding of libgpv_hip.so (C ABI in include/gpv_hip.h).

This is the ONLY compute backend of the package: there is no CPU / eager fallback.  If the
shared library is missing or a kernel returns an error, a RuntimeError is raised.
Tensors are passed as raw device pointers (``tensor.data_ptr()``) + the current HIP stream.
"""
import ctypes as C
import os
import threading

import torch

BF16, F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
KMAJOR, TRANS = 0, 1

_LIB = None
# GPV_TUNING_LIB=1 (tools/ only): the -DGPV_TUNING build of the same sources -- environment knobs of the launch heuristics, timing ablations,
# experimental kernels (`make -C gpv-1_amd/csrc tuning`).  The production library reads no environment.
TUNING = os.environ.get('GPV_TUNING_LIB', '0') == '1'
_LIB_PATH = os.environ.get('GPV_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc',
                                                          'libgpv_hip_tuning.so' if TUNING else 'libgpv_hip.so')


class GemmArgs(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p),
                ('M', C.c_int), ('N', C.c_int), ('K', C.c_int), ('batch', C.c_int),
                ('lda', C.c_int64), ('ldb', C.c_int64), ('ldc', C.c_int64),
                ('sA', C.c_int64), ('sB', C.c_int64), ('sC', C.c_int64),
                ('layoutA', C.c_int), ('layoutB', C.c_int), ('dtype_in', C.c_int), ('dtype_out', C.c_int),
                ('alpha', C.c_float), ('rowscale', C.c_void_p), ('bias', C.c_void_p),
                ('res', C.c_void_p), ('ldr', C.c_int64), ('sR', C.c_int64),
                ('relu_mask', C.c_void_p), ('ldm', C.c_int64),
                ('act', C.c_int), ('drop_p', C.c_float), ('seed', C.c_uint64),
                ('accumulate', C.c_int), ('split_k', C.c_int), ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64),
                ('a_rowsum', C.c_void_p), ('flags', C.c_int)]


class ConvArgs(C.Structure):
    _fields_ = [('mode', C.c_int), ('x', C.c_void_p), ('w', C.c_void_p), ('y', C.c_void_p),
                ('B', C.c_int), ('IH', C.c_int), ('IW', C.c_int), ('Cs', C.c_int), ('Cin', C.c_int),
                ('OH', C.c_int), ('OW', C.c_int), ('Cout', C.c_int),
                ('KH', C.c_int), ('KW', C.c_int), ('SH', C.c_int), ('SW', C.c_int), ('PH', C.c_int), ('PW', C.c_int),
                ('dtype_in', C.c_int), ('dtype_out', C.c_int),
                ('rowscale', C.c_void_p), ('bias', C.c_void_p), ('res', C.c_void_p), ('relu_mask', C.c_void_p),
                ('act', C.c_int), ('split_k', C.c_int), ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64),
                ('y_mask_bits', C.c_void_p), ('relu_mask_bits', C.c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [('q', C.c_void_p), ('k', C.c_void_p), ('v', C.c_void_p), ('o', C.c_void_p),
                ('q_bs', C.c_int64), ('q_rs', C.c_int64), ('k_bs', C.c_int64), ('k_rs', C.c_int64),
                ('v_bs', C.c_int64), ('v_rs', C.c_int64), ('o_bs', C.c_int64), ('o_rs', C.c_int64),
                ('B', C.c_int), ('H', C.c_int), ('Sq', C.c_int), ('Sk', C.c_int), ('dh', C.c_int),
                ('scale', C.c_float), ('kpm', C.c_void_p), ('causal', C.c_int),
                ('drop_p', C.c_float), ('seed', C.c_uint64), ('lse', C.c_void_p), ('dtype', C.c_int),
                ('dout', C.c_void_p), ('do_bs', C.c_int64), ('do_rs', C.c_int64),
                ('dq', C.c_void_p), ('dk', C.c_void_p), ('dv', C.c_void_p)]


class TTProblem(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('a_rowsum', C.c_void_p),
                ('M', C.c_int), ('N', C.c_int), ('K', C.c_int), ('lda', C.c_int), ('ldb', C.c_int), ('ldc', C.c_int)]


class FoldProblem(C.Structure):
    _fields_ = [('partials', C.c_void_p), ('out0', C.c_void_p), ('out1', C.c_void_p), ('nblk', C.c_int), ('cols', C.c_int)]


class ConvWgradProblem(C.Structure):
    _fields_ = [('x', C.c_void_p), ('dy', C.c_void_p), ('dw', C.c_void_p), ('rowscale', C.c_void_p)] + \
               [(k, C.c_int) for k in ('B', 'IH', 'IW', 'Cs', 'Cin', 'OH', 'OW', 'Cout', 'KH', 'KW', 'SH', 'SW', 'PH', 'PW')]


class JpegInfo(C.Structure):
    _fields_ = [('width', C.c_int), ('height', C.c_int), ('ncomp', C.c_int), ('hmax', C.c_int), ('vmax', C.c_int),
                ('mcus_x', C.c_int), ('mcus_y', C.c_int), ('bh', C.c_int * 3), ('bw', C.c_int * 3), ('reserved', C.c_int),
                ('coef_offset', C.c_int64 * 3), ('coef_count', C.c_int64), ('quant', (C.c_ushort * 64) * 3)]


class JpegDesc(C.Structure):
    _fields_ = [('coefs', C.c_void_p), ('planes', C.c_void_p), ('out', C.c_void_p),
                ('width', C.c_int), ('height', C.c_int), ('ncomp', C.c_int), ('hmax', C.c_int), ('vmax', C.c_int),
                ('bh', C.c_int * 3), ('bw', C.c_int * 3), ('coef_off', C.c_int * 3), ('plane_off', C.c_int * 3),
                ('quant', (C.c_ushort * 64) * 3)]


class ImageDesc(C.Structure):
    _fields_ = [('src', C.c_void_p), ('H', C.c_int), ('W', C.c_int), ('flip', C.c_int), ('gray', C.c_int), ('jitter', C.c_int),
                ('order', C.c_int * 4), ('brightness', C.c_float), ('contrast', C.c_float), ('saturation', C.c_float),
                ('hue', C.c_float), ('reserved', C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f'gpv1_amd: HIP kernel library not found at {_LIB_PATH}. Build it with '
                f'`python -c "import __graft_entry__ as g; g.build()"` (make -C gpv-1_amd/csrc). '
                f'There is no CPU/eager fallback by design.')
        _LIB = C.CDLL(_LIB_PATH)
        _LIB.gpv_abi_version.restype = C.c_int
        if _LIB.gpv_abi_version() != 1:
            raise RuntimeError('gpv1_amd: libgpv_hip.so ABI version mismatch')
    return _LIB


EXPORTS = ['gpv_abi_version', 'gpv_build_id', 'gpv_set_option', 'gpv_gemm', 'gpv_conv2d', 'gpv_conv2d_mask_bits_ok', 'gpv_conv1x1_dual_bits', 'gpv_conv1x1_chain_bits', 'gpv_image_to_nhwc4', 'gpv_maxpool3x3s2',
           'gpv_attention_fwd', 'gpv_attention_bwd', 'gpv_attention_qkv_fwd', 'gpv_layernorm_fwd', 'gpv_layernorm_bwd', 'gpv_layernorm_pos_fwd', 'gpv_layernorm_bwd2', 'gpv_linear_layernorm_fwd',
           'gpv_softmax_ce', 'gpv_roi_weights', 'gpv_add', 'gpv_add_rowbcast', 'gpv_colsum', 'gpv_cast',
           'gpv_cast_rowscale_t', 'gpv_prep_conv_weight', 'gpv_embedding', 'gpv_dropout',
           'gpv_relevance_condition', 'gpv_adamw', 'gpv_sumsq', 'gpv_clip_scale', 'gpv_act_fwd', 'gpv_act_bwd', 'gpv_set_seed_device',
           'gpv_gemm_tt_group', 'gpv_gemm_tt_group_ws', 'gpv_cast_transpose_group', 'gpv_stem_pool', 'gpv_image_pipeline', 'gpv_conv1x1_dual', 'gpv_conv1x1_chain',
           'gpv_conv_wgrad_group', 'gpv_jpeg_parse', 'gpv_jpeg_decode', 'gpv_argmax_rows', 'gpv_ln_linear_rows', 'gpv_attention_row_proj',
           'gpv_layernorm_bwd_blocks', 'gpv_layernorm_bwd3', 'gpv_colsum_fold_group', 'gpv_argmax_rows_embed']


def build_id():
    """gpv_build_id: sha256 prefix of the sources the loaded library was compiled from"""
    buf = C.create_string_buffer(64)
    _chk(lib().gpv_build_id(buf, C.c_int(64)), 'gpv_build_id')
    return buf.value.decode()


OPT_GLDS, OPT_GLDS_LAUNCHES, OPT_SKINNY, OPT_GLDS_WGRAD, OPT_PIPE, OPT_PIPE_LAUNCHES, OPT_C1S, OPT_C3S, OPT_C3S_LAUNCHES = 0, 1, 2, 3, 4, 5, 6, 7, 8
OPT_GEMV, OPT_GEMV_LAUNCHES = 9, 10
OPT_ATTN_BWD1, OPT_ATTN_BWD1_LAUNCHES, OPT_C1S_LAUNCHES, OPT_WG8, OPT_WG8_LAUNCHES, OPT_W8L, OPT_PIPE_SMALL, OPT_WG8H, OPT_C3_HALO, OPT_C3_HALO_LAUNCHES = 11, 12, 13, 14, 15, 16, 17, 18, 19, 20


GEMM_NO_PIPE_SMALL = 2      # gpv_gemm_args.flags: GPV_GEMM_NO_PIPE_SMALL
_TL = threading.local()


class gemm_flags:
    """with hip.gemm_flags(GEMM_NO_PIPE_SMALL): ... -- flags or-ed into every gpv_gemm call THIS THREAD issues inside the block (and
    into the launches a graph capture inside it records).  Per call and per thread: the autograd thread of a running backward, a
    data-loader thread or a second model never see it (round 5 flipped the process-wide GPV_OPT_PIPE_SMALL around every eval
    forward)."""

    def __init__(self, flags):
        self.flags = flags

    def __enter__(self):
        self.prev = getattr(_TL, 'gemm_flags', 0)
        _TL.gemm_flags = self.prev | self.flags
        return self

    def __exit__(self, *exc):
        _TL.gemm_flags = self.prev
        return False


_OPT_SET = {}          # option -> the value this process last set (the library's defaults are not mirrored here)


def set_option(option, value):
    """gpv_set_option: kernel-selection knob (tests / tuning); returns the previous value"""
    _OPT_SET[option] = value
    return lib().gpv_set_option(C.c_int(option), C.c_int(value))


def get_option_cached(option):
    """the value this process last set for `option` through set_option (None: never set -- the library's default)"""
    return _OPT_SET.get(option)


class option:
    """with hip.option(OPT_X, v): ... -- a kernel-selection option for the launches (or the graph capture) inside the block"""

    def __init__(self, opt, value):
        self.opt, self.value = opt, value

    def __enter__(self):
        self.prev = set_option(self.opt, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.opt, self.prev)
        return False


_SEED_DEV = None


def set_seed_device(t):
    """gpv_set_seed_device: lend the library one device int64 word as the dropout seed epoch (None: back to plain seeds).
    The tensor is kept alive here for as long as it is installed."""
    global _SEED_DEV
    if t is not None and (t.dtype != torch.int64 or t.numel() != 1 or not t.is_cuda):
        raise TypeError('seed epoch must be a 1-element int64 device tensor')
    _chk(lib().gpv_set_seed_device(_p(t)), 'gpv_set_seed_device')
    _SEED_DEV = t


def dcode(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f'gpv1_amd: unsupported dtype {t.dtype}')


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('gpv1_amd: tensors must live on the GPU (no CPU path exists)')
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    # current HIP stream of the current device (honours torch.cuda.stream(...) / graph capture).  The raw getter
    # costs 0.3 us, torch.cuda.current_stream().cuda_stream 2.6 us -- per launch, 1500 launches per step.
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(err, what):
    if err != 0:
        raise RuntimeError(f'gpv1_amd: {what} failed with hipError {err}')


def _f32(t):
    if t is not None and t.dtype != torch.float32:
        raise TypeError('expected fp32 tensor')
    return t


_WS = {}
WS_MAX = 256 << 20
_WS_RETIRED = []          # outgrown workspaces, kept alive for the graphs that captured them


def _workspace(device, nbytes):
    """split-reduction scratch (include/gpv_hip.h, gpv_gemm_args.workspace): one buffer per (device, stream), grown on
    demand up to WS_MAX, shared by all launches of that stream.  A launch that would need more gets the buffer as it is
    and the library falls back to fp32 atomics for it."""
    need = min(nbytes, WS_MAX)
    key = (device, _raw_stream(torch.cuda.current_device()) if _raw_stream is not None else 0)
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            # captured graphs (training bodies, inference, decode) have the old buffer's address baked in and keep replaying into
            # it: it must stay allocated for as long as they may run.  At most 64 + 128 MB are parked this way (the buffer doubles
            # up to WS_MAX).
            _WS_RETIRED.append(ws)
            need = max(need, min(2 * ws.numel(), WS_MAX))
        ws = _WS[key] = torch.empty(max(need, 64 << 20), device=device, dtype=torch.uint8)
    return ws


def gemm(A, B, Cm, M, N, K, lda, ldb, ldc, layoutA=KMAJOR, layoutB=KMAJOR, batch=1, sA=0, sB=0, sC=0,
         alpha=1.0, rowscale=None, bias=None, res=None, ldr=0, sR=0, relu_mask=None, ldm=0, act=ACT_NONE,
         drop_p=0.0, seed=0, accumulate=False, split_k=1, a_rowsum=None, kpad_finite=False):
    a = GemmArgs()
    a.A, a.B, a.C = _p(A), _p(B), _p(Cm)
    a.M, a.N, a.K, a.batch = M, N, K, batch
    a.lda, a.ldb, a.ldc, a.sA, a.sB, a.sC = lda, ldb, ldc, sA, sB, sC
    a.layoutA, a.layoutB = layoutA, layoutB
    if A.dtype != B.dtype:
        raise TypeError('gemm: A and B dtypes differ')
    a.dtype_in, a.dtype_out = dcode(A), dcode(Cm)
    a.alpha = alpha
    if rowscale is not None or bias is not None:
        check_epilogue_extents('gemm', M, N, rowscale, bias)
    a.rowscale, a.bias = _p(_f32(rowscale)), _p(_f32(bias))
    if res is not None and res.dtype != Cm.dtype:
        raise TypeError('gemm: residual dtype must equal output dtype')
    if relu_mask is not None and relu_mask.dtype != Cm.dtype:
        raise TypeError('gemm: mask dtype must equal output dtype')
    a.res, a.ldr, a.sR = _p(res), ldr, sR
    a.relu_mask, a.ldm = _p(relu_mask), ldm
    a.act, a.drop_p, a.seed = act, drop_p, seed
    a.accumulate, a.split_k = int(accumulate), split_k
    a.a_rowsum = _p(_f32(a_rowsum))
    a.flags = (1 if kpad_finite else 0) | getattr(_TL, 'gemm_flags', 0)      # GPV_GEMM_KPAD_FINITE | the calling thread's per-call flags
    if accumulate and batch == 1:                      # the library may split the reduction (further) when it has scratch
        # (the direct-to-LDS weight-gradient kernel sizes its own split: 512 blocks of 128x128 fp32 partials = 32 MiB)
        ws = _workspace(A.device, max(max(split_k, 8) * M * N * 4, 512 * 128 * 128 * 4))
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    _chk(lib().gpv_gemm(C.byref(a), _stream()), 'gpv_gemm')


def tt_group_ok(dy, x, dw, M, N, K, lda, ldb, ldc):
    """can this weight gradient (dW[M,N] += dy[K,M]^T x[K,N], bf16 operands, fp32 dW) ride in a grouped launch?"""
    return (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32 and M % 128 == 0 and N % 128 == 0
            and lda % 8 == 0 and ldb % 8 == 0 and ldc % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
            and dw.data_ptr() % 16 == 0 and K * max(lda, ldb) < (1 << 30))


def gemm_tt_group(problems):
    """gpv_gemm_tt_group: problems = [(dy, x, dw, bias_grad | None, M, N, K, lda, ldb, ldc), ...]; longest reductions first so that
    the grid's tail is made of short workgroups"""
    if not problems:
        return
    problems = sorted(problems, key=lambda q: -q[6])
    arr = (TTProblem * len(problems))()
    for i, (dy, x, dw, bg, M, N, K, lda, ldb, ldc) in enumerate(problems):
        a = arr[i]
        a.A, a.B, a.C, a.a_rowsum = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (bg.data_ptr() if bg is not None else None)
        a.M, a.N, a.K, a.lda, a.ldb, a.ldc = M, N, K, lda, ldb, ldc
    # the workspace only serves the 256 x 256 eight-phase path (GPV_OPT_W8L, off by default; the tuning build may switch it on
    # from the environment): without it every problem falls through to gpv_gemm_tt_group and NULL is legal -- no 256 MB grown per
    # (device, stream) for nothing
    if _OPT_SET.get(OPT_W8L, 0) or os.environ.get('GPV_W8L', '0') not in ('', '0'):
        ws = _workspace(problems[0][0].device, WS_MAX)
        _chk(lib().gpv_gemm_tt_group_ws(arr, C.c_int(len(problems)), _p(ws), C.c_int64(ws.numel()), _stream()), 'gpv_gemm_tt_group_ws')
    else:
        _chk(lib().gpv_gemm_tt_group_ws(arr, C.c_int(len(problems)), None, C.c_int64(0), _stream()), 'gpv_gemm_tt_group_ws')


def check_epilogue_extents(what, rows, cols, rowscale, bias):
    """The C ABI takes plain pointers: it cannot see that an epilogue vector is shorter than what the kernels index.  This mirror has the
    tensors, so it refuses here, before any launch.  rowscale is indexed by the OUTPUT ROW (gpv_gemm: rowscale[m], m < M; gpv_conv2d modes
    0 / 1: one factor per output PIXEL, B * OH * OW of them -- NOT the per-channel BatchNorm scale, which gpv_cast_rowscale_t folds into the
    weight copy; mode 2: one per Cout = the rows of dw), bias by the output column (N / Cout).  Found when tools/tune_gemms_bs1.py passed a
    [Cout] vector as a forward convolution's rowscale: the kernel read 19200 floats from a 64-float tensor -- a memory access fault."""
    if rowscale is not None and rowscale.numel() < rows:
        raise ValueError('%s: rowscale has %d elements, the epilogue indexes it by output row: %d needed' % (what, rowscale.numel(), rows))
    if bias is not None and bias.numel() < cols:
        raise ValueError('%s: bias has %d elements, the epilogue indexes it by output column: %d needed' % (what, bias.numel(), cols))


def _conv_args(mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, rowscale=None, bias=None,
               res=None, relu_mask=None, act=ACT_NONE, split_k=0, y_mask_bits=None, relu_mask_bits=None):
    a = ConvArgs()
    a.mode = mode
    a.x, a.w, a.y = _p(x), _p(w), _p(y)
    a.B, a.IH, a.IW, a.Cs, a.Cin, a.OH, a.OW, a.Cout = B, IH, IW, Cs, Cin, OH, OW, Cout
    a.KH, a.KW, a.SH, a.SW, a.PH, a.PW = KH, KW, SH, SW, PH, PW
    a.dtype_in, a.dtype_out = dcode(x), dcode(y)
    if x.dtype != w.dtype:
        raise TypeError('conv2d: operand dtypes differ')
    if rowscale is not None or bias is not None:
        check_epilogue_extents('conv2d mode %d' % mode, Cout if mode == 2 else B * OH * OW, KH * KW * Cin if mode == 2 else Cout, rowscale, bias)
    a.rowscale, a.bias = _p(_f32(rowscale)), _p(_f32(bias))
    for t in (res, relu_mask):
        if t is not None and t.dtype != y.dtype:
            raise TypeError('conv2d: res/mask dtype must equal output dtype')
    a.res, a.relu_mask = _p(res), _p(relu_mask)
    a.act, a.split_k = act, split_k
    for t in (y_mask_bits, relu_mask_bits):           # one-bit ReLU masks: int32 [pixels, Cout / 32] (gpv_conv_args.y_mask_bits / relu_mask_bits)
        if t is not None and (t.dtype != torch.int32 or not t.is_contiguous() or t.numel() * 32 != B * OH * OW * Cout):
            raise TypeError('conv2d: mask bits are a contiguous int32 [pixels, Cout / 32] tensor')
    a.y_mask_bits, a.relu_mask_bits = _p(y_mask_bits), _p(relu_mask_bits)
    return a


def conv2d_mask_bits_ok(*args, **kw):
    """gpv_conv2d_mask_bits_ok: would gpv_conv2d serve this call's y_mask_bits / relu_mask_bits?  (same arguments as conv2d; no launch)"""
    return bool(lib().gpv_conv2d_mask_bits_ok(C.byref(_conv_args(*args, **kw))))


def conv2d(mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, rowscale=None, bias=None,
           res=None, relu_mask=None, act=ACT_NONE, split_k=0, y_mask_bits=None, relu_mask_bits=None):
    a = _conv_args(mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, rowscale, bias, res, relu_mask, act, split_k,
                   y_mask_bits, relu_mask_bits)
    if mode == 2:                                      # wgrad: the library picks the split; lend it the shared scratch
        ws = _workspace(x.device, WS_MAX)
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    elif mode == 0 and B * OH * OW <= 8192 and KH * KW * Cin >= 1024:
        # forward over a few thousand pixels with a long reduction (inference at batch 1): the library may split it
        ws = _workspace(x.device, 16 * B * OH * OW * Cout * 4)
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    _chk(lib().gpv_conv2d(C.byref(a), _stream()), 'gpv_conv2d')


def conv_wgrad_group(problems):
    """gpv_conv_wgrad_group: problems = [(x, dy, dw, rowscale, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW), ...],
    bf16 x / dy, fp32 dw (accumulated into)"""
    if not problems:
        return
    arr = (ConvWgradProblem * len(problems))()
    for i, q in enumerate(problems):
        x, dy, dw, scale = q[:4]
        if x.dtype != torch.bfloat16 or dy.dtype != torch.bfloat16 or dw.dtype != torch.float32:
            raise TypeError('conv_wgrad_group: bf16 operands, fp32 gradients')
        a = arr[i]
        a.x, a.dy, a.dw, a.rowscale = _p(x), _p(dy), _p(dw), _p(_f32(scale))
        (a.B, a.IH, a.IW, a.Cs, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.SH, a.SW, a.PH, a.PW) = q[4:]
    ws = _workspace(problems[0][0].device, WS_MAX)
    _chk(lib().gpv_conv_wgrad_group(arr, C.c_int(len(problems)), _p(ws), C.c_int64(ws.numel()), _stream()), 'gpv_conv_wgrad_group')


def _attn_args(q, k, v, o, strides, B, H, Sq, Sk, dh, scale, kpm, causal, drop_p, seed, lse):
    a = AttnArgs()
    a.q, a.k, a.v, a.o = _p(q), _p(k), _p(v), _p(o)
    (a.q_bs, a.q_rs), (a.k_bs, a.k_rs), (a.v_bs, a.v_rs), (a.o_bs, a.o_rs) = strides
    a.B, a.H, a.Sq, a.Sk, a.dh = B, H, Sq, Sk, dh
    a.scale = scale
    if kpm is not None and kpm.dtype != torch.uint8:
        raise TypeError('key padding mask must be uint8')
    a.kpm, a.causal = _p(kpm), int(causal)
    a.drop_p, a.seed = drop_p, seed
    a.lse = _p(_f32(lse))
    a.dtype = dcode(q)
    return a


def attention_fwd(q, k, v, o, strides, B, H, Sq, Sk, dh, scale, kpm=None, causal=False, drop_p=0.0, seed=0,
                  lse=None):
    """strides = ((q_bs,q_rs),(k_bs,k_rs),(v_bs,v_rs),(o_bs,o_rs)) in elements."""
    a = _attn_args(q, k, v, o, strides, B, H, Sq, Sk, dh, scale, kpm, causal, drop_p, seed, lse)
    _chk(lib().gpv_attention_fwd(C.byref(a), _stream()), 'gpv_attention_fwd')


def attention_qkv_fwd(xp, x, w, bias, q, k, v, o, strides, B, H, S, scale, kpm=None, drop_p=0.0, seed=0, lse=None):
    """gpv_attention_qkv_fwd: q / k / v (column slices of the projection buffers) are WRITTEN; xp, x: [B * S, 256] rows"""
    if xp.stride() != x.stride() or xp.shape != x.shape or x.stride(-1) != 1 or xp.dtype != x.dtype:
        raise ValueError('attention_qkv_fwd: xp and x must share one row layout (the library takes ONE pair of strides for both)')
    a = _attn_args(q, k, v, o, strides, B, H, S, S, 32, scale, kpm, False, drop_p, seed, lse)
    _chk(lib().gpv_attention_qkv_fwd(C.byref(a), _p(xp), _p(x), C.c_int64(S * x.stride(0)), C.c_int64(x.stride(0)), _p(w), _p(_f32(bias)),
                                     _stream()), 'gpv_attention_qkv_fwd')


def attention_bwd(q, k, v, o, dout, dq, dk, dv, strides, do_strides, B, H, Sq, Sk, dh, scale, kpm=None,
                  causal=False, drop_p=0.0, seed=0, lse=None):
    a = _attn_args(q, k, v, o, strides, B, H, Sq, Sk, dh, scale, kpm, causal, drop_p, seed, lse)
    a.dout = _p(dout)
    a.do_bs, a.do_rs = do_strides
    a.dq, a.dk, a.dv = _p(dq), _p(dk), _p(dv)
    _chk(lib().gpv_attention_bwd(C.byref(a), _stream()), 'gpv_attention_bwd')


def linear_layernorm_fwd(a, w, bias, x, gamma, beta, s, y, mean, rstd, rows, eps, drop_p=0.0, seed=0, pos=None, y2=None):
    """gpv_linear_layernorm_fwd: s = a w^T + bias (written), y = LayerNorm(x + dropout(s)) * gamma + beta, y2 = y + pos rows; width 256, bf16"""
    if (pos is None) != (y2 is None):
        raise ValueError('pos and y2 come together')
    if any(t.dtype != torch.bfloat16 for t in (a, w, x, s, y)):
        raise TypeError('linear_layernorm_fwd: bf16 activations and weight')
    _chk(lib().gpv_linear_layernorm_fwd(_p(a), _p(w), _p(_f32(bias)), _p(x), _p(_f32(gamma)), _p(_f32(beta)), _p(s), _p(y), _p(_f32(mean)),
                                        _p(_f32(rstd)), C.c_int(rows), C.c_int(w.shape[1]), C.c_int(w.shape[0]), C.c_float(eps), C.c_float(drop_p), C.c_uint64(seed),
                                        _p(pos), C.c_int(0 if pos is None else pos.numel() // w.shape[0]), _p(y2), _stream()), 'gpv_linear_layernorm_fwd')


def layernorm_fwd(x, s, gamma, beta, y, mean, rstd, rows, cols, eps, drop_p=0.0, seed=0, pos=None, y2=None):
    """pos [pos_rows, cols] + y2 [rows, cols] (both or neither): second output y2 = y + pos[row % pos_rows] (gpv_layernorm_pos_fwd)"""
    if (pos is None) != (y2 is None):
        raise ValueError('layernorm_fwd: pos and y2 come together')
    if pos is not None and (pos.dtype != x.dtype or y2.dtype != x.dtype or pos.numel() % cols != 0):
        raise TypeError('layernorm_fwd: pos / y2 in the activation dtype, whole rows')
    _chk(lib().gpv_layernorm_pos_fwd(_p(x), _p(s), _p(_f32(gamma)), _p(_f32(beta)), _p(y), _p(mean), _p(rstd),
                                     C.c_int(rows), C.c_int(cols), C.c_float(eps), C.c_float(drop_p), C.c_uint64(seed),
                                     _p(pos), C.c_int(0 if pos is None else pos.numel() // cols), _p(y2),
                                     C.c_int(dcode(x)), _stream()), 'gpv_layernorm_pos_fwd')


def layernorm_bwd(dy, x, s, gamma, mean, rstd, dx, ds, dgamma, dbeta, rows, cols, drop_p=0.0, seed=0, dy2=None, partials=None):
    """dy2: a second gradient of the same output (came back through the forward's y2), summed on load.
    partials (instead of dgamma / dbeta): fp32 [layernorm_bwd_blocks(rows, cols), 2 * cols], one row of partial column sums per
    workgroup, added to dgamma / dbeta later by colsum_fold_group"""
    if dy2 is not None and dy2.dtype != dy.dtype:
        raise TypeError('layernorm_bwd: dy2 dtype')
    if partials is not None and (partials.dtype != torch.float32 or partials.numel() < layernorm_bwd_blocks(rows, cols) * 2 * cols):
        raise ValueError('layernorm_bwd: partials too small')
    _chk(lib().gpv_layernorm_bwd3(_p(dy), _p(dy2), _p(x), _p(s), _p(_f32(gamma)), _p(mean), _p(rstd), _p(dx), _p(ds),
                                  _p(_f32(dgamma)), _p(_f32(dbeta)), _p(partials), C.c_int(rows), C.c_int(cols),
                                  C.c_float(drop_p), C.c_uint64(seed), C.c_int(dcode(x)), _stream()),
         'gpv_layernorm_bwd3')


def layernorm_bwd_blocks(rows, cols):
    """gpv_layernorm_bwd_blocks: rows of the partials buffer layernorm_bwd(..., partials=) fills for this shape"""
    n = lib().gpv_layernorm_bwd_blocks(C.c_int(rows), C.c_int(cols))
    if n <= 0:
        raise ValueError('layernorm_bwd_blocks: shape refused')
    return n


def colsum_fold_group(problems):
    """gpv_colsum_fold_group: problems = [(partials, out0, out1, nblk, cols), ...]: out0 += column sums of partials[:, :cols],
    out1 += those of partials[:, cols:], all problems in one launch"""
    if not problems:
        return
    arr = (FoldProblem * len(problems))()
    for i, (part, o0, o1, nblk, cols) in enumerate(problems):
        a = arr[i]
        a.partials, a.out0, a.out1, a.nblk, a.cols = part.data_ptr(), _f32(o0).data_ptr(), _f32(o1).data_ptr(), nblk, cols
    _chk(lib().gpv_colsum_fold_group(arr, C.c_int(len(problems)), _stream()), 'gpv_colsum_fold_group')


def softmax_ce(logits, ld, target, loss, dlogits, gscale, rows, V):
    _chk(lib().gpv_softmax_ce(_p(logits), C.c_int64(ld), _p(target), _p(loss), _p(dlogits), _p(gscale),
                              C.c_int(rows), C.c_int(V), C.c_int(dcode(logits)), _stream()), 'gpv_softmax_ce')


def image_to_nhwc4(img, out, B, H, W, pad, Hp, Wp):
    _chk(lib().gpv_image_to_nhwc4(_p(_f32(img)), _p(out), B, H, W, pad, Hp, Wp, dcode(out), _stream()),
         'gpv_image_to_nhwc4')


def maxpool3x3s2(x, y, B, H, W, Cc, OH, OW):
    _chk(lib().gpv_maxpool3x3s2(_p(x), _p(y), B, H, W, Cc, OH, OW, dcode(x), _stream()), 'gpv_maxpool3x3s2')


def image_pipeline(descs_dev, B, scratch, grey_sum, out, OH, OW, pad, Hp, Wp):
    """gpv_image_pipeline: descs_dev = uint8 device tensor holding B packed ImageDesc structs"""
    _chk(lib().gpv_image_pipeline(_p(descs_dev), B, _p(scratch), _p(_f32(grey_sum)), _p(out), OH, OW, pad, Hp, Wp, dcode(out), _stream()),
         'gpv_image_pipeline')


class JpegUnsupported(ValueError):
    """a JPEG flavour outside the decoder's scope (progressive, arithmetic coding, 12 bit, CMYK, exotic sampling): convert the file"""


def jpeg_parse(data, coefs=None):
    """gpv_jpeg_parse (host side, releases the GIL): data = bytes; coefs = None (header pass) or a writable int16 buffer / numpy
    array / CPU tensor of at least info.coef_count elements.  -> JpegInfo"""
    info = JpegInfo()
    fn = lib().gpv_jpeg_parse
    fn.argtypes = [C.c_char_p, C.c_int64, C.POINTER(JpegInfo), C.c_void_p, C.c_int64]
    if coefs is None:
        ptr, cap = None, 0
    elif torch.is_tensor(coefs):
        if coefs.is_cuda or coefs.dtype != torch.int16:
            raise TypeError('jpeg_parse: coefs must be a CPU int16 tensor (pinned for the upload)')
        ptr, cap = coefs.data_ptr(), coefs.numel()
    else:
        ptr, cap = coefs.ctypes.data, coefs.size
    err = fn(data, len(data), C.byref(info), ptr, cap)
    if err == 801:
        raise JpegUnsupported('gpv_jpeg_parse: not a baseline 8-bit 1- or 3-component JPEG with 4:4:4 / 4:2:2 / 4:2:0 sampling')
    if err:
        raise ValueError(f'gpv_jpeg_parse: malformed JPEG (hipError {err})')
    return info


def jpeg_decode(descs_dev, B, max_blocks, max_pixels):
    """gpv_jpeg_decode: descs_dev = uint8 device tensor holding B packed JpegDesc structs"""
    _chk(lib().gpv_jpeg_decode(_p(descs_dev), B, int(max_blocks), C.c_int64(int(max_pixels)), _stream()), 'gpv_jpeg_decode')


def conv1x1_dual(a1, w1, a2, w2, bias, y, B, OH, OW, K1, IH2, IW2, K2, s2, N, act=ACT_RELU, y_mask_bits=None):
    """gpv_conv1x1_dual(_bits): y = act(a1 . w1^T + a2(stride s2) . w2^T + bias) (+ the one-bit ReLU mask of y); returns False when the
    shape is not one the kernel takes (hipErrorNotSupported) -- the caller then runs the two convolutions"""
    if not all(t.dtype == torch.bfloat16 for t in (a1, w1, a2, w2, y)):
        return False
    if y_mask_bits is not None:
        if y_mask_bits.dtype != torch.int32 or not y_mask_bits.is_contiguous() or y_mask_bits.numel() * 32 != B * OH * OW * N:
            raise TypeError('conv1x1_dual: mask bits are a contiguous int32 [pixels, N / 32] tensor')
        err = lib().gpv_conv1x1_dual_bits(_p(a1), _p(w1), _p(a2), _p(w2), _p(_f32(bias)), _p(y), B, OH, OW, K1, IH2, IW2, K2, s2, N, act, _p(y_mask_bits),
                                          _stream())
    else:
        err = lib().gpv_conv1x1_dual(_p(a1), _p(w1), _p(a2), _p(w2), _p(_f32(bias)), _p(y), B, OH, OW, K1, IH2, IW2, K2, s2, N, act, _stream())
    if err == 801:
        return False
    _chk(err, 'gpv_conv1x1_dual')
    return True


def conv1x1_chain(a1, w1, a2, w2, s2, res, bias, y, wn, bias_n, z, B, OH, OW, z_mask_bits=None):
    """gpv_conv1x1_chain(_bits): y = relu(a1 . w1^T (+ a2[::s2] . w2^T) (+ res) + bias), z = relu(y . wn^T + bias_n) in one launch (+ the
    one-bit ReLU mask of z); False when the shape is not one the kernel takes -- the caller then runs the convolutions one by one"""
    ts = [t for t in (a1, w1, a2, w2, res, y, wn, z) if t is not None]
    if not all(t.dtype == torch.bfloat16 for t in ts):
        return False
    K1, N, N2 = a1.shape[-1], y.shape[-1], z.shape[-1]
    K2 = 0 if a2 is None else a2.shape[-1]
    IH2, IW2 = (a2.shape[1], a2.shape[2]) if a2 is not None else (OH, OW)
    if z_mask_bits is not None:
        if z_mask_bits.dtype != torch.int32 or not z_mask_bits.is_contiguous() or z_mask_bits.numel() * 32 != B * OH * OW * N2:
            raise TypeError('conv1x1_chain: mask bits are a contiguous int32 [pixels, N2 / 32] tensor')
        err = lib().gpv_conv1x1_chain_bits(_p(a1), _p(w1), K1, _p(a2), _p(w2), K2, IH2, IW2, s2, _p(res), _p(_f32(bias)), _p(y), B, OH, OW, N,
                                           _p(wn), _p(_f32(bias_n)), _p(z), N2, _p(z_mask_bits), _stream())
    else:
        err = lib().gpv_conv1x1_chain(_p(a1), _p(w1), K1, _p(a2), _p(w2), K2, IH2, IW2, s2, _p(res), _p(_f32(bias)), _p(y), B, OH, OW, N,
                                      _p(wn), _p(_f32(bias_n)), _p(z), N2, _stream())
    if err == 801:
        return False
    _chk(err, 'gpv_conv1x1_chain')
    return True


def ffn_fused_fwd(x, w1, b1, w2, b2, gamma, beta, h, y, out, mean, rstd, M, D, F, eps, drop_p=0.0, seed1=0, seed2=0, pos=None, out2=None):
    """gpv_ffn_fused_fwd: LayerNorm(x + dropout(linear2(dropout(relu(linear1(x)))))) in one launch (h, y, mean, rstd stored for the
    backward); False when the shape / dtype is not one the kernel takes -- the caller then runs gemm, gemm, layernorm_fwd"""
    if not hasattr(lib(), 'gpv_ffn_fused_fwd'):          # tuning build only (csrc/Makefile TUNE_SRCS)
        return False
    ts = [t for t in (x, w1, w2, h, y, out, pos, out2) if t is not None]
    if not all(t.dtype == torch.bfloat16 and t.is_contiguous() for t in ts) or b1 is None or b2 is None:
        return False
    err = lib().gpv_ffn_fused_fwd(_p(x), _p(w1), _p(_f32(b1)), _p(w2), _p(_f32(b2)), _p(_f32(gamma)), _p(_f32(beta)), _p(h), _p(y), _p(out),
                                  _p(mean), _p(rstd), C.c_int(M), C.c_int(D), C.c_int(F), C.c_float(eps), C.c_float(drop_p), C.c_uint64(seed1), C.c_uint64(seed2),
                                  _p(pos), C.c_int(0 if pos is None else pos.numel() // D), _p(out2), _stream())
    if err == 801:
        return False
    _chk(err, 'gpv_ffn_fused_fwd')
    return True


def stem_pool(x, w, shift, y, B, Hp, Wp, CH, CW, PH, PW):
    """gpv_stem_pool: conv 7x7/2 + FrozenBN shift + ReLU + max-pool 3x3/2 in one launch (bf16)"""
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or y.dtype != torch.bfloat16:
        raise TypeError('stem_pool: bf16 only')
    _chk(lib().gpv_stem_pool(_p(x), _p(w), _p(_f32(shift)), _p(y), B, Hp, Wp, CH, CW, PH, PW, _stream()), 'gpv_stem_pool')


def roi_weights(boxes, wgt, n_roi, H, W, ldw):
    _chk(lib().gpv_roi_weights(_p(_f32(boxes)), _p(wgt), n_roi, H, W, C.c_int64(ldw), dcode(wgt), _stream()),
         'gpv_roi_weights')


def add(a, b, y, n):
    _chk(lib().gpv_add(_p(a), _p(b), _p(y), C.c_int64(n), dcode(a), _stream()), 'gpv_add')


def add_rowbcast(a, b, y, rows_total, rows_b, cols):
    _chk(lib().gpv_add_rowbcast(_p(a), _p(b), _p(y), C.c_int64(rows_total), C.c_int64(rows_b), cols, dcode(a),
                                _stream()), 'gpv_add_rowbcast')


def colsum(x, out, rows, cols, ld):
    _chk(lib().gpv_colsum(_p(x), _p(_f32(out)), rows, cols, C.c_int64(ld), dcode(x), _stream()), 'gpv_colsum')


def cast(src, dst, n):
    _chk(lib().gpv_cast(_p(src), _p(dst), C.c_int64(n), dcode(src), dcode(dst), _stream()), 'gpv_cast')


def cast_rowscale_t(src, scale, dst, dstT, rows, cols):
    d = dst if dst is not None else dstT
    _chk(lib().gpv_cast_rowscale_t(_p(_f32(src)), _p(scale), _p(dst), _p(dstT), rows, cols, dcode(d), _stream()),
         'gpv_cast_rowscale_t')


class TCProblem(C.Structure):
    """include/gpv_hip.h: gpv_tc_problem"""
    _fields_ = [('src', C.c_void_p), ('dstT', C.c_void_p), ('rows', C.c_int), ('cols', C.c_int)]


def cast_transpose_group(items):
    """items: (src fp32 [rows, cols] contiguous, dstT [cols, rows] contiguous in the compute dtype) -- one launch per 128"""
    if not items:
        return
    arr = (TCProblem * len(items))()
    for t, (src, dstT) in zip(arr, items):
        _f32(src)
        rows, cols = src.shape
        assert src.is_contiguous() and dstT.is_contiguous() and tuple(dstT.shape) == (cols, rows)
        t.src, t.dstT, t.rows, t.cols = src.data_ptr(), dstT.data_ptr(), rows, cols
    _chk(lib().gpv_cast_transpose_group(arr, C.c_int(len(items)), C.c_int(dcode(items[0][1])), _stream()), 'gpv_cast_transpose_group')


def prep_conv_weight(src, scale, wf, wd, Cout, T, Cin):
    d = wf if wf is not None else wd
    _chk(lib().gpv_prep_conv_weight(_p(_f32(src)), _p(scale), _p(wf), _p(wd), Cout, T, Cin, dcode(d), _stream()),
         'gpv_prep_conv_weight')


def embedding(table, ids, out, n_ids, dim):
    _chk(lib().gpv_embedding(_p(table), _p(ids), _p(out), C.c_int64(n_ids), dim, dcode(table), dcode(out),
                             _stream()), 'gpv_embedding')


def dropout(x, y, n, p, seed):
    _chk(lib().gpv_dropout(_p(x), _p(y), C.c_int64(n), C.c_float(p), C.c_uint64(seed), dcode(x), _stream()),
         'gpv_dropout')


def relevance_condition(x, logits, tokens, y, rows, dim):
    _chk(lib().gpv_relevance_condition(_p(x), _p(_f32(logits)), _p(_f32(tokens)), _p(y), rows, dim, dcode(x),
                                       _stream()), 'gpv_relevance_condition')


def adamw(p, g, m, v, p_lowp, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale=None, seg_id=None, seg_live=None):
    _chk(lib().gpv_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_lowp), C.c_int64(n), C.c_float(lr), C.c_float(beta1),
                         C.c_float(beta2), C.c_float(eps), C.c_float(wd), C.c_float(bc1), C.c_float(bc2),
                         _p(gscale), _p(seg_id), _p(seg_live), _stream()), 'gpv_adamw')


CLIP_PARTIALS = 1024


def clip_scale(g, max_norm, partial, gscale, pstep=None, live=None):
    """gpv_clip_scale: gscale = min(1, max_norm / (||g|| + 1e-6)) over the contiguous fp32 range g (None: no norm), deterministic;
    pstep += live (int32 per-parameter Adam step counts) in the same launch"""
    n = 0 if g is None else g.numel()
    if g is not None and (not g.is_contiguous() or partial.numel() < CLIP_PARTIALS):
        raise ValueError('clip_scale: contiguous gradient range and >= CLIP_PARTIALS floats of scratch')
    for t in (pstep, live):
        if t is not None and t.dtype != torch.int32:
            raise TypeError('clip_scale: int32 step counts / liveness flags')
    _chk(lib().gpv_clip_scale(_p(_f32(g)), C.c_int64(n), C.c_float(max_norm), _p(_f32(partial)), _p(_f32(gscale)), _p(pstep), _p(live),
                              C.c_int(0 if pstep is None else pstep.numel()), _stream()), 'gpv_clip_scale')


def sumsq(x, n, out):
    _chk(lib().gpv_sumsq(_p(x), C.c_int64(n), _p(out), _stream()), 'gpv_sumsq')


def act_fwd(x, y, n, act):
    _chk(lib().gpv_act_fwd(_p(x), _p(y), C.c_int64(n), act, dcode(x), _stream()), 'gpv_act_fwd')


def argmax_rows(x, addend, out0=None, out1=None, table=None, pos_row=None, xnext=None):
    """x [rows, V] (row pitch x.stride(0), unit column stride), addend fp32 [V] or None; out0 / out1: int64 tensors whose element
    r * stride(0) receives row r's pick (views such as ids[:, t + 1] work).  table [V, D] (+ pos_row [D]) -> xnext [rows, D]:
    the picked entries' rows (+ the position row), gpv_argmax_rows_embed"""
    rows, V = x.shape
    assert x.stride(1) == 1 and (addend is None or (addend.dtype == torch.float32 and addend.is_contiguous()))
    for o in (out0, out1):
        assert o is None or (o.dtype == torch.int64 and o.dim() == 1 and o.shape[0] == rows)
    D, ldt = 0, 0
    if table is not None:
        D, ldt = table.shape[1], table.stride(0)
        assert table.dtype == x.dtype and table.stride(1) == 1 and table.shape[0] >= V
        assert xnext is not None and xnext.dtype == x.dtype and xnext.is_contiguous() and tuple(xnext.shape) == (rows, D)
        assert pos_row is None or (pos_row.dtype == x.dtype and pos_row.is_contiguous() and pos_row.numel() == D)
    _chk(lib().gpv_argmax_rows_embed(_p(x), C.c_int64(x.stride(0)), _p(addend), rows, V, dcode(x),
                                     _p(out0), C.c_int64(out0.stride(0) if out0 is not None else 0),
                                     _p(out1), C.c_int64(out1.stride(0) if out1 is not None else 0),
                                     _p(table), C.c_int64(ldt), _p(pos_row if table is not None else None),
                                     _p(xnext if table is not None else None), C.c_int(D), _stream()), 'gpv_argmax_rows_embed')


LN_LINEAR_MAX_ROWS, LN_LINEAR_MAX_COLS = 4, 1024


def ln_linear_rows(x, s, gamma, beta, eps, xn, Wm, bias, y, ldy, rows, N, K, act=ACT_NONE, s_partial=None, s_bias=None):
    """xn = LayerNorm(x + s) * gamma + beta ; y = act(xn Wm^T + bias)  (rows <= 4, K <= 1024; Wm [N, K] rows of pitch Wm.stride(0)).
    s_partial (fp32 [rows, parts, K], from attention_row_proj) + s_bias instead of s: s = sum over parts + bias"""
    if x.dtype != Wm.dtype or x.dtype != y.dtype or x.dtype != xn.dtype or (s is not None and s.dtype != x.dtype):
        raise TypeError('ln_linear_rows: dtypes differ')
    parts = 0
    if s_partial is not None:
        assert s is None and s_partial.dtype == torch.float32 and s_partial.is_contiguous() and s_partial.shape[0] == rows and s_partial.shape[2] == K
        parts = s_partial.shape[1]
    _chk(lib().gpv_ln_linear_rows(_p(x), _p(s), _p(_f32(gamma)), _p(_f32(beta)), C.c_float(eps), _p(xn), _p(Wm), C.c_int64(Wm.stride(0)),
                                  _p(_f32(bias)), _p(y), C.c_int64(ldy), rows, N, K, act, dcode(x), _p(s_partial), parts, _p(_f32(s_bias)),
                                  _stream()), 'gpv_ln_linear_rows')


ROW_PROJ_MAX_KEYS = 256


def attention_row_proj(q, q_bs, k, k_bs, k_rs, v, v_bs, v_rs, Wo, partial, B, H, Sk, dh, scale):
    """gpv_attention_row_proj: one query row per sequence over Sk keys, out-projection folded in -> partial fp32 [B, H, H*dh]"""
    if q.dtype != k.dtype or q.dtype != v.dtype or q.dtype != Wo.dtype or partial.dtype != torch.float32:
        raise TypeError('attention_row_proj: dtypes')
    _chk(lib().gpv_attention_row_proj(_p(q), C.c_int64(q_bs), _p(k), C.c_int64(k_bs), C.c_int64(k_rs), _p(v), C.c_int64(v_bs), C.c_int64(v_rs),
                                      _p(Wo), C.c_int64(Wo.stride(0)), _p(partial), B, H, Sk, dh, C.c_float(scale), dcode(q), _stream()),
         'gpv_attention_row_proj')


def act_bwd(dy, ref, dx, n, act, alpha=1.0):
    _chk(lib().gpv_act_bwd(_p(dy), _p(ref), _p(dx), C.c_int64(n), act, C.c_float(alpha), dcode(dy), _stream()),
         'gpv_act_bwd')


def _install_debug_sync():
    """GPV_DEBUG_SYNC=1: time (HIP events) and print every entry-point call (tensor shapes, scalars) and synchronise
    after it: per-call timings, and an asynchronous GPU fault dies right after the launch that caused it.  Debug aid only."""
    import functools
    import sys
    g = globals()

    def desc(v):
        if torch.is_tensor(v):
            return f'{str(v.dtype)[6:]}{list(v.shape)}s{list(v.stride())}'
        if isinstance(v, (tuple, list)):
            return '(' + ','.join(desc(x) for x in v) + ')'
        return repr(v)

    def wrap(name, fn):
        @functools.wraps(fn)
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            torch.cuda.synchronize()
            print('[gpv-hip] %8.1f us' % (e0.elapsed_time(e1) * 1e3), name, ' '.join(desc(x) for x in a),
                  ' '.join(f'{n}={desc(x)}' for n, x in k.items()), file=sys.stderr, flush=True)
            return r
        return inner
    for name, fn in list(g.items()):
        if callable(fn) and not name.startswith('_') and getattr(fn, '__module__', None) == __name__ \
                and name not in ('lib', 'dcode') and not isinstance(fn, type):
            g[name] = wrap(name, fn)


if os.environ.get('GPV_DEBUG_SYNC') == '1':
    _install_debug_sync()

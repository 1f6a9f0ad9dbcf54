"""ctypes mirror of include/wrs_debug.h: test hooks of libwrs_mi355x.so (kernel-level GEMM / attention entries on
device pointers, teacher-forced and re-synchronised decode).  Used by tests/ only; nothing here is part of the drop-in
boundary."""
import ctypes as C

import numpy as np

from . import _lib as L

GEMM_AUTO, GEMM_PP2, GEMM_TILE128, GEMM_TILE128_PACKED, GEMM_RING = 0, 1, 2, 3, 4
GEMM_NAMES = {0: "auto", 1: "pp2_256x256", 2: "tile128", 3: "tile128_packedA", 4: "ring<4,8>"}
PATH_TILED, PATH_DEC, PATH_SKINNY_F32, PATH_PRING = 0, 1, 2, 3
PRING_RESID, PRING_F16, PRING_GELU_PACKED, PRING_QKV = 0, 1, 2, 3


class DebugGemmC(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("path", C.c_int32), ("force_kernel", C.c_int32), ("kernel_used", C.c_int32),
                ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("a", C.c_void_p), ("a_packed", C.c_int32),
                ("lda", C.c_int64), ("rows_per_batch", C.c_int64), ("batch_stride", C.c_int64), ("packed_rows", C.c_int64),
                ("w", C.c_void_p), ("n_w", C.c_int64), ("k_w", C.c_int64), ("conv_c", C.c_int32),
                ("bias", C.c_void_p), ("scale", C.c_float), ("scale_n_end", C.c_int32), ("act", C.c_int32),
                ("residual", C.c_void_p), ("posemb", C.c_void_p),
                ("c", C.c_void_p), ("c_f32", C.c_int32), ("c_packed", C.c_int32),
                ("ldc", C.c_int64), ("c_rows_per_batch", C.c_int64), ("c_batch_rows", C.c_int64), ("c_row_off", C.c_int64),
                ("ksplit", C.c_int32), ("slab_stride", C.c_int64),
                ("pring_mode", C.c_int32), ("pring_bm", C.c_int32), ("kcache", C.c_void_p), ("vcache", C.c_void_p),
                ("n_ctx", C.c_int32), ("cache_pos", C.c_int32), ("phys_mul", C.c_int32), ("pring_workgroups", C.c_int32),
                ("pack_k", C.c_void_p), ("pack_v", C.c_void_p),
                ("pack_T", C.c_int32), ("pack_H", C.c_int32), ("pack_k_col0", C.c_int32), ("pack_v_col0", C.c_int32)]


class DebugAttnSelfC(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("n_rows", C.c_int32), ("rows_per_audio", C.c_int32), ("H", C.c_int32), ("D", C.c_int32),
                ("n_ctx", C.c_int32), ("cache_pos", C.c_int32),
                ("qkv", C.c_void_p), ("qkv_bias", C.c_void_p), ("scale", C.c_float),
                ("kc", C.c_void_p), ("vc", C.c_void_p), ("anc", C.c_void_p), ("out_packed", C.c_void_p), ("q16_mode", C.c_int32)]


class DebugAttnCrossC(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("n_audio", C.c_int32), ("G", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("T", C.c_int32),
                ("q", C.c_void_p), ("q_bias", C.c_void_p), ("q_scale", C.c_float),
                ("k", C.c_void_p), ("v", C.c_void_p), ("out_packed", C.c_void_p), ("q16_mode", C.c_int32)]


P = C.POINTER
# every symbol include/wrs_debug.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "wrs_debug_gemm_f16": (C.c_int, [P(DebugGemmC)]),
    "wrs_debug_attn_self_run": (C.c_int, [P(DebugAttnSelfC)]),
    "wrs_debug_attn_cross_run": (C.c_int, [P(DebugAttnCrossC)]),
    "wrs_debug_decode_forced": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wrs_debug_decode_forced_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wrs_debug_decode_resync": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, P(L.DecodeResultC)]),
    "wrs_debug_decode_candidates": (C.c_int, [C.c_void_p, P(C.c_int32), P(C.c_int32), P(C.c_int32), P(P(C.c_float)), P(P(C.c_int32))]),
    "wrs_debug_sequencer_run": (C.c_int, [P(L.TokenIdsC), P(L.DecodeOptionsC), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                         P(L.DecodeResultC), C.c_void_p, C.c_void_p, C.c_void_p]),
    "wrs_debug_transcribe_bookkeeping": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                                   C.c_void_p, P(L.TranscribeOutputC)]),
    "wrs_debug_model_set_parity_f32": (C.c_int, [C.c_void_p, C.c_int]),
    "wrs_debug_model_step_paths": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "wrs_debug_set_encoder_stop": (C.c_int, [C.c_void_p, C.c_int]),
    "wrs_debug_read_encoder": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
}

_bound = False


def lib():
    global _bound
    if not L.IS_DEBUG_LIB:
        raise ImportError("the include/wrs_debug.h hooks live in libwrs_mi355x_debug.so only: set WRS_DEBUG_LIB=1 before importing "
                          "whisper_rs_amd (tests/conftest.py and __graft_entry__.smoke() do)")
    l = L.load()
    if not _bound:
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return l


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pa_index(rows, K, device=None):
    """torch index tensor that maps a row-major [rows, K] matrix to the packed-activation order of csrc/kernels.h
    `pa_off`: packed.flatten()[pa_off(m, k)] = x[m, k].  rows % 16 == 0, K % 32 == 0."""
    import torch
    m = torch.arange(rows, device=device).view(-1, 1)
    k = torch.arange(K, device=device).view(1, -1)
    ksteps = K // 32
    return (((m // 16) * ksteps + (k // 32)) * 64 + ((k % 32) // 8) * 16 + (m % 16)) * 8 + (k % 8)


def pack_activations(x, rows_pad):
    """x [M, K] (torch, any float dtype) -> fp16 packed-activation buffer of rows_pad rows (zero padded)."""
    import torch
    M, K = x.shape
    full = torch.zeros(rows_pad, K, dtype=torch.float16, device=x.device)
    full[:M] = x.to(torch.float16)
    out = torch.empty(rows_pad * K, dtype=torch.float16, device=x.device)
    out[pa_index(rows_pad, K, x.device).flatten()] = full.flatten()
    return out


def unpack_activations(buf, M, K):
    """inverse of pack_activations: packed fp16 buffer -> [M, K] fp32."""
    rows_pad = buf.numel() // K
    return buf[pa_index(rows_pad, K, buf.device).flatten()].view(rows_pad, K)[:M].float()


def decode_resync(task, mel, forced_tokens=None, forced_sources=None, forced_scores=None):
    """wrs_debug_decode_resync: the first len(forced_*) steps follow the given decisions, then the run is free."""
    from .api import _f32
    l = lib()
    mel = _f32(mel)
    if mel.ndim == 2:
        mel = mel[None]
    n = 0 if forced_tokens is None else len(forced_tokens)
    res = L.DecodeResultC()
    if n:
        ft = np.ascontiguousarray(forced_tokens, dtype=np.int32)
        fs = np.ascontiguousarray(forced_sources, dtype=np.int32)
        fl = np.ascontiguousarray(forced_scores, dtype=np.float32)
        L.check(l.wrs_debug_decode_resync(task._h, _p(mel), mel.shape[0], n, _p(ft), _p(fs), _p(fl), C.byref(res)))
    else:
        L.check(l.wrs_debug_decode_resync(task._h, _p(mel), mel.shape[0], 0, None, None, None, C.byref(res)))
    return task._collect(res)


def decode_candidates(task):
    """(logprobs [steps, rows, k], ids [steps, rows, k]) the device proposed at every step of the last resync run."""
    l = lib()
    ns, nr, k = C.c_int32(), C.c_int32(), C.c_int32()
    plp, pid = P(C.c_float)(), P(C.c_int32)()
    L.check(l.wrs_debug_decode_candidates(task._h, C.byref(ns), C.byref(nr), C.byref(k), C.byref(plp), C.byref(pid)))
    shape = (ns.value, nr.value, k.value)
    if ns.value == 0:
        return np.zeros(shape, np.float32), np.zeros(shape, np.int32)
    return np.ctypeslib.as_array(plp, shape=shape).copy(), np.ctypeslib.as_array(pid, shape=shape).copy()


def decode_forced_features(task, feats, forced_tokens, forced_sources):
    """wrs_debug_decode_forced_features: teacher-forced decode from encoder output [n, 1500, D]."""
    from .api import _f32
    l = lib()
    feats = _f32(feats)
    ft = np.ascontiguousarray(forced_tokens, dtype=np.int32)
    fs = np.ascontiguousarray(forced_sources, dtype=np.int32)
    steps, rows = ft.shape
    m = task.options.token_extract_mode
    k = m.beam_size + 1 if m.kind == "beam_search" else 1
    lp = np.empty((steps, rows, k), np.float32)
    ids = np.empty((steps, rows, k), np.int32)
    ns = np.empty(feats.shape[0], np.float64)
    L.check(l.wrs_debug_decode_forced_features(task._h, _p(feats), feats.shape[0], steps, _p(ft), _p(fs), _p(lp), _p(ids), _p(ns)))
    return lp, ids, ns


def parity_f32_model(dims, weights, device=0):
    """a model in f32 parity mode (include/wrs_debug.h): fp32 everywhere, for the bit-exact-ids demonstration."""
    from .api import Whisper
    m = Whisper(dims, device=device)
    L.check(lib().wrs_debug_model_set_parity_f32(m._h, 1))
    m.load(weights)
    return m


STEP_TILED, STEP_PRING, STEP_FUSED, STEP_SELF_MFMA, STEP_QKV_DIRECT = 1, 2, 4, 8, 16


def step_paths(model):
    """OR of the forms of Model::decoder_step this model has taken since it was created (include/wrs_debug.h)"""
    v = C.c_int32(0)
    L.check(lib().wrs_debug_model_step_paths(model._h, C.byref(v)))
    return v.value

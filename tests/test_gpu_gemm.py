"""GPU parity of every GEMM kernel variant of the hot path AT THE SHAPES bench.py RUNS THEM (large-v2: D 1280, 4D 5120,
3D 3840, vocabulary 51865 padded to 51872, 256 windows -> 384 000 encoder rows / 1280 decoder rows), through the library's
own weight packer and launchers (wrs_debug_gemm_f16, include/wrs_debug.h).

Reference: torch fp32 matmul on the SAME fp16-rounded inputs (an independent implementation: rocBLAS), plus the
epilogue in torch.  These kernels replace the `nn::Linear` / `conv1d` call sites of reference src/model.rs:158-206,
279-292,417-441,490-522.  Tolerances (written out below): fp32 outputs 1e-4 of the output scale (accumulation order
only), fp16 outputs 1.5e-3 of the output scale (one fp16 rounding).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from whisper_rs_amd import debug as DBG
from whisper_rs_amd import _lib as L

pytestmark = pytest.mark.gpu

D = 1280
TOL_F32 = 1e-4   # |c - ref|.max() / |ref|.max(), fp32 output
TOL_F16 = 1.5e-3  # fp16 output (half-ulp of fp16 is 4.9e-4 relative)


def _dev():
    return torch.device("cuda", 0)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device=_dev())
    g.manual_seed(seed)
    return (scale * torch.randn(shape, generator=g, device=_dev(), dtype=torch.float32)).to(torch.float16)


def _gelu(x):
    return torch.nn.functional.gelu(x, approximate="none")


def _run(g):
    torch.cuda.synchronize()
    L.check(DBG.lib().wrs_debug_gemm_f16(C.byref(g)))
    return g.kernel_used


def _rel_err(got, ref):
    return float((got - ref).abs().max() / ref.abs().max())


def _tiled(a16, w32, M, N, K, *, force=DBG.GEMM_AUTO, packed=False, bias=None, scale=1.0, scale_n_end=0, act=0, residual=None,
           posemb=None, c_f32=True, c_packed=False, ldc=None, ksplit=1, n_w=None, c_rows=None, a_desc=None, c_desc=None, conv_c=0):
    """one launch through wrs_debug_gemm_f16; returns (C as fp32 [c_rows, ldc] or slabs, kernel id)."""
    dev = _dev()
    g = DBG.DebugGemmC()
    g.device_id, g.path, g.force_kernel = 0, DBG.PATH_TILED, force
    g.M, g.N, g.K = M, N, K
    keep = []
    if packed:
        rows_pad = -(-M // 128) * 128
        ap = DBG.pack_activations(a16, rows_pad)
        keep.append(ap)
        g.a, g.a_packed, g.packed_rows = ap.data_ptr(), 1, rows_pad
    else:
        g.a, g.a_packed = a16.data_ptr(), 0
        if a_desc:
            g.lda, g.rows_per_batch, g.batch_stride = a_desc
        else:
            g.lda = K
    g.w, g.n_w, g.k_w, g.conv_c = w32.data_ptr(), (n_w or N), (3 * conv_c if conv_c else w32.shape[1]), conv_c
    g.bias = bias.data_ptr() if bias is not None else None
    g.scale, g.scale_n_end, g.act = scale, scale_n_end, act
    g.posemb = posemb.data_ptr() if posemb is not None else None
    ldc = ldc or N
    c_rows = c_rows or M
    if ksplit > 1:
        out = torch.zeros(ksplit, c_rows, ldc, dtype=torch.float32, device=dev)
        g.slab_stride = c_rows * ldc
    elif c_packed:
        out = torch.zeros((-(-M // 128) * 128) * N, dtype=torch.float16, device=dev)
    else:
        out = torch.zeros(c_rows, ldc, dtype=torch.float32 if c_f32 else torch.float16, device=dev)
    if residual is not None:  # the model aliases residual and C (x += ...): do the same
        out.copy_(residual)
        g.residual = out.data_ptr()
    g.c, g.c_f32, g.c_packed, g.ldc, g.ksplit = out.data_ptr(), int(c_f32), int(c_packed), ldc, ksplit
    if c_desc:
        g.c_rows_per_batch, g.c_batch_rows, g.c_row_off = c_desc
    used = _run(g)
    if c_packed:
        out = DBG.unpack_activations(out, M, N)
    return out.float(), used


def _ref(a16, w32, bias=None, scale=1.0, scale_n_end=0, act=0, residual=None, posemb=None):
    r = a16.float() @ w32.t()
    if bias is not None:
        r = r + bias
    if scale_n_end:
        r[:, :scale_n_end] *= scale
    if act:
        r = _gelu(r)
    if posemb is not None:
        r = r + posemb
    if residual is not None:
        r = r + residual
    return r


# ------------------------------------------------------------------------------------------------ encoder shapes
# (name, N, K, epilogue) of the encoder / cross-K/V GEMMs of large-v2 (model.cpp::encoder_from_mel_t, load_cross_kv)
ENC_SHAPES = [
    ("qkv", 3 * D, D, dict(bias=True, scale=True, f16=True)),
    ("attn_out", D, D, dict(bias=True, residual=True)),
    ("mlp1", 4 * D, D, dict(bias=True, act=True, f16=True)),
    ("mlp2", D, 4 * D, dict(bias=True, residual=True)),
    ("cross_kv", 2 * D, D, dict(bias=True, scale=True, f16=True)),
]


@pytest.mark.parametrize("force", [DBG.GEMM_AUTO, DBG.GEMM_PP2, DBG.GEMM_TILE128], ids=["auto", "pp2", "tile128"])
@pytest.mark.parametrize("name,N,K,epi", ENC_SHAPES, ids=[s[0] for s in ENC_SHAPES])
@pytest.mark.parametrize("M", [12000, 1280])
def test_encoder_gemm_shapes(gpu, M, name, N, K, epi, force):
    """row-major A.  M = 12 000 = 8 windows x 1500 (not a multiple of 128 or 256: edge tiles), M = 1280."""
    a = _rand((M, K), 1)
    w = _rand((N, K), 2, K ** -0.5).float()
    bias = _rand((N,), 3, 0.1).float() if epi.get("bias") else None
    res = _rand((M, N), 4).float() if epi.get("residual") else None
    scale, sne = (64 ** -0.25, (2 * D if N == 3 * D else D)) if epi.get("scale") else (1.0, 0)
    got, used = _tiled(a, w, M, N, K, force=force, bias=bias, scale=scale, scale_n_end=sne, act=int(bool(epi.get("act"))),
                       residual=res, c_f32=not epi.get("f16"))
    ref = _ref(a, w, bias, scale, sne, epi.get("act"), res)
    err = _rel_err(got, ref)
    print(f"enc {name} M={M} N={N} K={K} kernel={DBG.GEMM_NAMES[used]} rel_err={err:.2e}")
    if force != DBG.GEMM_AUTO:
        assert used == force
    assert err < (TOL_F16 if epi.get("f16") else TOL_F32)


def test_encoder_gemm_full_bench_rows(gpu):
    """M = 384 000 = 256 windows x 1500 rows, the row count of the benchmark pass (attn_out shape: bias + residual, fp32)."""
    M, N, K = 384000, D, D
    a = _rand((M, K), 11)
    w = _rand((N, K), 12, K ** -0.5).float()
    bias = _rand((N,), 13, 0.1).float()
    res = _rand((M, N), 14).float()
    got, used = _tiled(a, w, M, N, K, bias=bias, residual=res)
    assert used == DBG.GEMM_PP2, "the benchmark runs this shape on the 256x256 ping-pong kernel"
    ref = _ref(a, w, bias, residual=res)
    err = _rel_err(got, ref)
    print(f"enc attn_out M={M}: rel_err={err:.2e}")
    assert err < TOL_F32
    del got, ref, res
    # mlp1 at the same row count on a 64 000-row slice of distinct data placed at the END of a 384 000-row problem is
    # equivalent to the tile walk of the last bands; run the full mlp1 shape at 64 000 rows (fp16 + GELU epilogue)
    M2 = 64000
    w1 = _rand((4 * D, D), 15, D ** -0.5).float()
    b1 = _rand((4 * D,), 16, 0.1).float()
    got, used = _tiled(a[:M2], w1, M2, 4 * D, D, bias=b1, act=1, c_f32=False)
    assert used == DBG.GEMM_PP2
    err = _rel_err(got, _ref(a[:M2], w1, b1, act=1))
    print(f"enc mlp1 M={M2}: rel_err={err:.2e}")
    assert err < TOL_F16


@pytest.mark.parametrize("force", [DBG.GEMM_AUTO, DBG.GEMM_TILE128], ids=["auto", "tile128"])
def test_conv_as_gemm_overlapping_rows(gpu, force):
    """conv1 (k3, pad 1) and conv2 (k3, stride 2, pad 1) as implicit GEMMs over overlapping rows of a zero-padded
    time-major activation (reference model.rs:421-431), at D = 1280, 4 windows, vs torch conv1d."""
    nb, Tm, T = 4, 3000, 1500
    g = torch.Generator(device=_dev())
    g.manual_seed(5)
    mel = torch.randn(nb, 80, Tm, generator=g, device=_dev()).to(torch.float16)
    w1 = _rand((D, 80, 3), 6, 240 ** -0.5).float()
    b1 = _rand((D,), 7, 0.1).float()
    # A = [nb, 3002, 80] time-major with one zero row each side (launch_mel_prep's layout)
    a = torch.zeros(nb, 3002, 80, dtype=torch.float16, device=_dev())
    a[:, 1:3001] = mel.permute(0, 2, 1)
    a = torch.cat([a.flatten(), torch.zeros(256, dtype=torch.float16, device=_dev())])
    # c1 = [nb, 3001, D] fp16, row 0 of every window stays zero (conv2's left padding)
    got, used = _tiled(a, w1.flatten(1), nb * Tm, D, 256, force=force, bias=b1, act=1, c_f32=False, n_w=D, conv_c=80,
                       a_desc=(80, Tm, 3002 * 80), c_rows=nb * 3001, c_desc=(Tm, 3001, 1))
    ref1 = _gelu(torch.nn.functional.conv1d(mel.float(), w1, b1, padding=1))          # [nb, D, 3000]
    c1 = got.view(nb, 3001, D)
    assert float(c1[:, 0].abs().max()) == 0.0
    err = _rel_err(c1[:, 1:], ref1.permute(0, 2, 1))
    print(f"conv1 kernel={DBG.GEMM_NAMES[used]} rel_err={err:.2e}")
    assert err < TOL_F16
    # conv2 on the fp16 c1 the library produced: rows overlap (lda = 2D < K = 3D)
    c1h = torch.cat([got.to(torch.float16).flatten(), torch.zeros(4 * D, dtype=torch.float16, device=_dev())])
    w2 = _rand((D, D, 3), 8, (3 * D) ** -0.5).float()
    b2 = _rand((D,), 9, 0.1).float()
    pos = _rand((T, D), 10).float()
    got2, used2 = _tiled(c1h, w2.flatten(1), nb * T, D, 3 * D, force=force, bias=b2, act=1, posemb=pos, n_w=D, conv_c=D,
                         a_desc=(2 * D, T, 3001 * D), c_desc=(T, T, 0))
    x1 = c1[:, 1:].to(torch.float16).float().permute(0, 2, 1)                          # [nb, D, 3000] as stored
    ref2 = _gelu(torch.nn.functional.conv1d(x1, w2, b2, stride=2, padding=1)).permute(0, 2, 1) + pos
    err2 = _rel_err(got2.view(nb, T, D), ref2)
    print(f"conv2 kernel={DBG.GEMM_NAMES[used2]} rel_err={err2:.2e}")
    assert err2 < TOL_F32


# ------------------------------------------------------------------------------------------------ decode-step shapes
def _dec_split(M, N, K):
    """the split-K factor Model::decoder_step picks for the slab GEMMs at >= 480 rows (model.cpp gemm_slab)."""
    tiles = -(-M // 128) * -(-N // 128)
    split = 1
    while tiles * split * 2 <= 256 and (K // 64) % (split * 2) == 0 and K // 64 // (split * 2) >= 10:
        split *= 2
    return split


DEC_SHAPES = [("qkv", 3 * D, D), ("attn_out", D, D), ("mlp2", D, 4 * D)]


@pytest.mark.parametrize("force", [DBG.GEMM_AUTO, DBG.GEMM_TILE128_PACKED, DBG.GEMM_RING], ids=["auto", "tile128p", "ring"])
@pytest.mark.parametrize("name,N,K", DEC_SHAPES, ids=[s[0] for s in DEC_SHAPES])
@pytest.mark.parametrize("M", [1280, 500, 1535])
def test_decode_step_slab_gemms(gpu, M, name, N, K, force):
    """packed activations, fp32 slabs, the split-K the decode step uses (2 for the N = D shapes at 1280 rows: the
    LDS-transposed slab store) and split 1; 1280 rows = 256 windows x beam 5, 500 rows = edge row tile, 1535 rows = the
    bench's 307 windows x beam 5 with the slab padded to the consumer's row stride (1600 rows, as Model::decoder_step
    allocates it): the last row tile then also goes through the LDS-transposed store and writes its 128th row into the padding."""
    a = _rand((M, K), 21)
    w = _rand((N, K), 22, K ** -0.5).float()
    ref = a.float() @ w.t()
    for ksplit in sorted({1, _dec_split(M, N, K), 2}):
        got, used = _tiled(a, w, M, N, K, force=force, packed=True, ksplit=ksplit, c_rows=1600 if M == 1535 else None)
        got = got[..., :M, :]
        if ksplit > 1:
            got = got.sum(0)
        err = _rel_err(got, ref)
        print(f"dec {name} M={M} N={N} K={K} ksplit={ksplit} kernel={DBG.GEMM_NAMES[used]} rel_err={err:.2e}")
        if force != DBG.GEMM_AUTO:
            assert used == force
        elif M == 1280 and N == D and ksplit == 2:
            assert used == DBG.GEMM_RING, "the benchmark runs the N = D decode GEMMs on the ring kernel"
        assert err < TOL_F32


@pytest.mark.parametrize("force", [DBG.GEMM_AUTO, DBG.GEMM_RING], ids=["auto", "ring"])
@pytest.mark.parametrize("M", [1280, 500, 1535])
def test_decode_step_mlp1_packed_gelu(gpu, M, force):
    """MLP-1 of the decode step at >= 480 rows: bias + exact GELU + fp16 store in packed-activation order."""
    N, K = 4 * D, D
    a = _rand((M, K), 31)
    w = _rand((N, K), 32, K ** -0.5).float()
    bias = _rand((N,), 33, 0.1).float()
    got, used = _tiled(a, w, M, N, K, force=force, packed=True, bias=bias, act=1, c_f32=False, c_packed=True, ldc=N)
    err = _rel_err(got, _ref(a, w, bias, act=1))
    print(f"dec mlp1 M={M} kernel={DBG.GEMM_NAMES[used]} rel_err={err:.2e}")
    assert err < TOL_F16


@pytest.mark.parametrize("M", [1280, 500])
def test_decode_step_logits_gemm(gpu, M):
    """tied output projection (reference model.rs:521) at >= 480 rows: V = 51865 weight rows, logits row pitch 51872."""
    V, ld, K = 51865, 51872, D
    a = _rand((M, K), 41)
    w = _rand((V, K), 42, 1.5 * K ** -0.5).float()
    got, used = _tiled(a, w, M, ld, K, packed=True, ldc=ld, n_w=V)
    assert used == DBG.GEMM_TILE128_PACKED
    ref = a.float() @ w.t()
    err = _rel_err(got[:, :V], ref)
    print(f"dec logits M={M} rel_err={err:.2e}")
    assert err < TOL_F32
    assert float(got[:, V:].abs().max()) == 0.0  # padded columns come from zero weight rows


@pytest.mark.parametrize("M", [5, 40, 160, 320, 475])
def test_decode_step_small_row_paths(gpu, M):
    """fewer than 480 rows: the latency-optimised split-K decode GEMM (every shape of a layer) and the skinny fp32
    GEMM of the logits."""
    for name, N, K in DEC_SHAPES + [("mlp1", 4 * D, D)]:
        a = _rand((M, K), 51)
        w = _rand((N, K), 52, K ** -0.5).float()
        rows_pad = max(-(-M // 128) * 128, 640)
        ap = DBG.pack_activations(a, rows_pad)
        slab = torch.zeros(16 * 640 * N, dtype=torch.float32, device=_dev())  # <= 10 slabs x m_pad <= 480 rows x N
        g = DBG.DebugGemmC()
        g.device_id, g.path, g.M, g.N, g.K = 0, DBG.PATH_DEC, M, N, K
        g.a, g.a_packed, g.packed_rows = ap.data_ptr(), 1, rows_pad
        g.w, g.n_w, g.k_w = w.data_ptr(), N, K
        g.c, g.c_f32, g.ldc = slab.data_ptr(), 1, N
        _run(g)
        m_pad = g.slab_stride // N
        got = slab[: g.ksplit * g.slab_stride].view(g.ksplit, m_pad, N).sum(0)[:M]
        err = _rel_err(got, a.float() @ w.t())
        print(f"dec(small) {name} M={M} N={N} K={K} slabs={g.ksplit} rel_err={err:.2e}")
        assert err < TOL_F32
    V, ld, K = 51865, 51872, D
    a = _rand((M, K), 61)
    w = _rand((V, K), 62, 1.5 * K ** -0.5).float()
    rows_pad = -(-M // 128) * 128
    ap = DBG.pack_activations(a, max(rows_pad, 640))
    out = torch.zeros(M, ld, dtype=torch.float32, device=_dev())
    g = DBG.DebugGemmC()
    g.device_id, g.path, g.M, g.N, g.K = 0, DBG.PATH_SKINNY_F32, M, V, K
    g.a, g.a_packed, g.packed_rows = ap.data_ptr(), 1, max(rows_pad, 640)
    g.w, g.n_w, g.k_w = w.data_ptr(), V, K
    g.c, g.c_f32, g.ldc = out.data_ptr(), 1, ld
    _run(g)
    err = _rel_err(out[:, :V], a.float() @ w.t())
    print(f"dec(small) logits M={M} rel_err={err:.2e}")
    assert err < TOL_F32


# ------------------------------------------------------------------------------------------------ persistent ring GEMM
def _pring(a16, w32, M, N, K, mode, bm, *, bias=None, scale=1.0, scale_n_end=0, x0=None, ldc=None, qkv=None):
    """one launch of the persistent ring GEMM (csrc/gemm_pring.hip) through wrs_debug_gemm_f16."""
    dev = _dev()
    g = DBG.DebugGemmC()
    g.device_id, g.path, g.M, g.N, g.K = 0, DBG.PATH_PRING, M, N, K
    rows_pad = -(-M // 128) * 128
    ap = DBG.pack_activations(a16, rows_pad)
    g.a, g.a_packed, g.packed_rows = ap.data_ptr(), 1, rows_pad
    g.w, g.n_w, g.k_w = w32.data_ptr(), N, K
    g.bias = bias.data_ptr() if bias is not None else None
    g.scale, g.scale_n_end = scale, scale_n_end
    g.pring_mode, g.pring_bm = mode, bm
    keep = []
    if mode == DBG.PRING_RESID:
        out = x0.clone()
        g.ldc = N
    elif mode == DBG.PRING_GELU_PACKED:
        out = torch.zeros(rows_pad * N, dtype=torch.float16, device=dev)
        g.ldc = N
    elif mode == DBG.PRING_QKV:
        d, n_ctx, cache_pos, phys_mul = N // 3, qkv["n_ctx"], qkv["cache_pos"], qkv["phys_mul"]
        out = torch.zeros(M, d, dtype=torch.float16, device=dev)
        g.ldc = d
        g.kcache, g.vcache = qkv["kc"].data_ptr(), qkv["vc"].data_ptr()
        g.n_ctx, g.cache_pos, g.phys_mul = n_ctx, cache_pos, phys_mul
    else:
        out = torch.zeros(M, ldc or N, dtype=torch.float16, device=dev)
        g.ldc = ldc or N
    g.c = out.data_ptr()
    _run(g)
    if mode == DBG.PRING_GELU_PACKED:
        out = DBG.unpack_activations(out, M, N)
    return out.float(), g.pring_workgroups


@pytest.mark.parametrize("bm", [64, 128])
@pytest.mark.parametrize("M", [1280, 500, 2560])
@pytest.mark.parametrize("K", [D, 4 * D], ids=["K1280", "K5120"])
def test_pring_residual_epilogue(gpu, M, K, bm):
    """out-projection / MLP-2 of the decode step: x += A W^T + bias in place on the fp32 residual stream (N = D)."""
    a = _rand((M, K), 71)
    w = _rand((D, K), 72, K ** -0.5).float()
    bias = _rand((D,), 73, 0.1).float()
    x0 = _rand((M, D), 74).float()
    got, wgs = _pring(a, w, M, D, K, DBG.PRING_RESID, bm, bias=bias, x0=x0)
    ref = x0 + a.float() @ w.t() + bias
    err = _rel_err(got, ref)
    print(f"pring RESID M={M} K={K} bm={bm} workgroups={wgs} rel_err={err:.2e}")
    assert err < TOL_F32


@pytest.mark.parametrize("M", [1280, 500])
def test_pring_query_and_mlp1_epilogues(gpu, M):
    a = _rand((M, D), 81)
    w = _rand((D, D), 82, D ** -0.5).float()
    bias = _rand((D,), 83, 0.1).float()
    scale = 64 ** -0.25
    got, wgs = _pring(a, w, M, D, D, DBG.PRING_F16, 64, bias=bias, scale=scale, scale_n_end=D)
    err = _rel_err(got, (a.float() @ w.t() + bias) * scale)
    print(f"pring F16 (cross query) M={M} workgroups={wgs} rel_err={err:.2e}")
    assert err < TOL_F16
    w1 = _rand((4 * D, D), 84, D ** -0.5).float()
    b1 = _rand((4 * D,), 85, 0.1).float()
    for bm in (128, 64):
        got, wgs = _pring(a, w1, M, 4 * D, D, DBG.PRING_GELU_PACKED, bm, bias=b1)
        err = _rel_err(got, _gelu(a.float() @ w1.t() + b1))
        print(f"pring GELU_PACKED (MLP-1) M={M} bm={bm} workgroups={wgs} rel_err={err:.2e}")
        assert err < TOL_F16


@pytest.mark.parametrize("M,cache_pos", [(1280, 17), (500, 0), (1280, 447)])
def test_pring_qkv_epilogue_writes_cache(gpu, M, cache_pos):
    """QKV projection of the decode step: q (bias, scale) -> fp16 [M, D]; k (scale) and v (bias) -> the K/V cache rows at the
    step's position (Cached::forward, model.rs:83-87); nothing else in the cache is touched."""
    n_ctx = 448
    a = _rand((M, D), 91)
    w = _rand((3 * D, D), 92, D ** -0.5).float()
    bias = _rand((3 * D,), 93, 0.1).float()
    bias[D:2 * D] = 0.0  # the key projection has no bias (model.rs:138-141)
    scale = 64 ** -0.25
    kc = _rand((M, n_ctx, D), 94, 1.0).to(torch.float16)
    vc = _rand((M, n_ctx, D), 95, 1.0).to(torch.float16)
    kc0, vc0 = kc.clone(), vc.clone()
    got_q, wgs = _pring(a, w, M, 3 * D, D, DBG.PRING_QKV, 128, bias=bias, scale=scale, scale_n_end=2 * D,
                        qkv=dict(kc=kc, vc=vc, n_ctx=n_ctx, cache_pos=cache_pos, phys_mul=1))
    ref = a.float() @ w.t() + bias
    ref[:, :2 * D] *= scale
    assert _rel_err(got_q, ref[:, :D]) < TOL_F16
    assert _rel_err(kc[:, cache_pos].float(), ref[:, D:2 * D]) < TOL_F16
    assert _rel_err(vc[:, cache_pos].float(), ref[:, 2 * D:]) < TOL_F16
    kc0[:, cache_pos], vc0[:, cache_pos] = kc[:, cache_pos], vc[:, cache_pos]
    assert torch.equal(kc, kc0) and torch.equal(vc, vc0)
    print(f"pring QKV M={M} cache_pos={cache_pos} workgroups={wgs}: q/k/v ok")


@pytest.mark.parametrize("M,cache_pos", [(1535, 17), (1280, 0), (500, 447)])
def test_tiled_qkv_epilogue_writes_cache(gpu, M, cache_pos):
    """The form the benchmark runs at >= 480 rows: the QKV projection on the 128x128 packed-A kernel, its epilogue writing q
    (bias, scale) as fp16 [M, D] and the new k (scale) / v (bias) rows into the K/V cache at the step's position, with 16-byte
    stores of 8 consecutive columns (two column groups exchanged between lane rows).  1535 rows = 307 windows x beam 5: the last
    row tile is partial; nothing else in the cache may change."""
    n_ctx = 448
    a = _rand((M, D), 191)
    w = _rand((3 * D, D), 192, D ** -0.5).float()
    bias = _rand((3 * D,), 193, 0.1).float()
    bias[D:2 * D] = 0.0  # the key projection has no bias (model.rs:138-141)
    scale = 64 ** -0.25
    kc = _rand((M, n_ctx, D), 194, 1.0).to(torch.float16)
    vc = _rand((M, n_ctx, D), 195, 1.0).to(torch.float16)
    kc0, vc0 = kc.clone(), vc.clone()
    dev = _dev()
    g = DBG.DebugGemmC()
    g.device_id, g.path, g.force_kernel = 0, DBG.PATH_TILED, DBG.GEMM_AUTO
    g.M, g.N, g.K = M, 3 * D, D
    rows_pad = -(-M // 128) * 128
    ap = DBG.pack_activations(a, rows_pad)
    g.a, g.a_packed, g.packed_rows = ap.data_ptr(), 1, rows_pad
    g.w, g.n_w, g.k_w = w.data_ptr(), 3 * D, D
    g.bias, g.scale, g.scale_n_end = bias.data_ptr(), scale, 2 * D
    q = torch.zeros(M, D, dtype=torch.float16, device=dev)
    g.c, g.c_f32, g.c_packed, g.ldc, g.ksplit = q.data_ptr(), 0, 0, D, 1
    g.kcache, g.vcache, g.n_ctx, g.cache_pos, g.phys_mul = kc.data_ptr(), vc.data_ptr(), n_ctx, cache_pos, 1
    used = _run(g)
    # 4 x 30 tiles at 500 rows give a CU at most one workgroup: the launcher takes the ring kernel (same epilogue code)
    assert used == (DBG.GEMM_RING if M == 500 else DBG.GEMM_TILE128_PACKED)
    ref = a.float() @ w.t() + bias
    ref[:, :2 * D] *= scale
    assert _rel_err(q.float(), ref[:, :D]) < TOL_F16
    assert _rel_err(kc[:, cache_pos].float(), ref[:, D:2 * D]) < TOL_F16
    assert _rel_err(vc[:, cache_pos].float(), ref[:, 2 * D:]) < TOL_F16
    keep = torch.ones(n_ctx, dtype=torch.bool, device=dev)
    keep[cache_pos] = False
    assert torch.equal(kc[:, keep], kc0[:, keep]) and torch.equal(vc[:, keep], vc0[:, keep]), "a cache position other than the step's was written"


def _pack_k_ref(k, n_audio, T, H):
    """packed K of csrc/attention.hip: Kp [audio][head][Tp/16][2][64 lanes][8], (kt, ks, l, j) = K[kt*16 + (l & 15)][ks*32 + 8*(l >> 4) + j]."""
    Tp = -(-T // 32) * 32
    kk = torch.zeros(n_audio, Tp, H, 64, dtype=k.dtype, device=k.device)
    kk[:, :T] = k.view(n_audio, T, H, 64)
    kk = kk.permute(0, 2, 1, 3).reshape(n_audio, H, Tp // 16, 16, 2, 4, 8)        # [a, h, kt, key, ks, chunk, j]
    return kk.permute(0, 1, 2, 4, 5, 3, 6).reshape(n_audio, H, Tp // 16, 2, 64, 8)  # lane = chunk*16 + key


def _pack_v_ref(v, n_audio, T, H):
    """packed V: Vp [audio][head][Tp/32][4][64 lanes][8], (kb, dt, l, j) = V[kb*32 + kperm(l >> 4, j)][dt*16 + (l & 15)],
    kperm(q, j) = j < 4 ? 4q + j : 16 + 4q + (j - 4)."""
    Tp = -(-T // 32) * 32
    vv = torch.zeros(n_audio, Tp, H, 64, dtype=v.dtype, device=v.device)
    vv[:, :T] = v.view(n_audio, T, H, 64)
    vv = vv.permute(0, 2, 1, 3).reshape(n_audio, H, Tp // 32, 2, 4, 4, 4, 16)      # [a, h, kb, half, q, r, dt, d]: key = 16*half + 4*q + r
    return vv.permute(0, 1, 2, 6, 4, 7, 3, 5).reshape(n_audio, H, Tp // 32, 4, 64, 8)  # lane = q*16 + d, j = 4*half + r


@pytest.mark.parametrize("with_q", [False, True], ids=["cross_kv", "encoder_qkv"])
def test_kv_projection_packed_by_the_epilogue(gpu, with_q):
    """K / V projections of the encoder (N = 3D: q stays row-major) and of the decoder's cross-attention (N = 2D) at large-v2 width:
    the 256x256 kernel writes K through its 16-byte exchanged stores and V on the operand-swapped instantiation straight into the
    packed per-(audio, head) formats the attention kernels read.  Checked against torch fp32 matmul packed by an independent
    restatement of the formats (csrc/attention.hip header); 8 audios x 1500 keys: 16-row tiles straddle audios, the last tile is partial."""
    n_audio, T, H = 8, 1500, 20
    M, K = n_audio * T, D
    N = (3 if with_q else 2) * D
    k0 = D if with_q else 0
    a = _rand((M, K), 201)
    w = _rand((N, K), 202, K ** -0.5).float()
    bias = _rand((N,), 203, 0.1).float()
    scale = 64 ** -0.25
    dev = _dev()
    Tp = -(-T // 32) * 32
    kp = torch.zeros(n_audio * H * Tp * 64, dtype=torch.float16, device=dev)
    vp = torch.zeros_like(kp)
    q = torch.zeros(M, N, dtype=torch.float16, device=dev)
    g = DBG.DebugGemmC()
    g.device_id, g.path, g.force_kernel = 0, DBG.PATH_TILED, DBG.GEMM_AUTO
    g.M, g.N, g.K = M, N, K
    g.a, g.a_packed, g.lda = a.data_ptr(), 0, K
    g.w, g.n_w, g.k_w = w.data_ptr(), N, K
    g.bias, g.scale, g.scale_n_end = bias.data_ptr(), scale, k0 + D   # q and k are scaled, v is not (model.rs:169-172)
    g.c, g.c_f32, g.c_packed, g.ldc, g.ksplit = q.data_ptr(), 0, 0, N, 1
    g.pack_k, g.pack_v, g.pack_T, g.pack_H, g.pack_k_col0, g.pack_v_col0 = kp.data_ptr(), vp.data_ptr(), T, H, k0, k0 + D
    used = _run(g)
    assert used == DBG.GEMM_PP2
    ref = a.float() @ w.t() + bias
    ref[:, :k0 + D] *= scale
    kref = _pack_k_ref(ref[:, k0:k0 + D].contiguous(), n_audio, T, H)
    vref = _pack_v_ref(ref[:, k0 + D:].contiguous(), n_audio, T, H)
    assert _rel_err(kp.view_as(kref).float(), kref) < TOL_F16
    assert _rel_err(vp.view_as(vref).float(), vref) < TOL_F16
    if with_q:
        assert _rel_err(q[:, :D].float(), ref[:, :D]) < TOL_F16
        assert not q[:, D:].any(), "the K / V columns must not be written to the row-major buffer any more"

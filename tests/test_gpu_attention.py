"""GPU parity of the two decode attention kernels at the benchmark's shapes (large-v2: H = 20 heads of 64, beam 5,
256 audios -> 1280 decoder rows; cross keys T = 1500, self length up to 447), on caller data through
wrs_debug_attn_cross_run / wrs_debug_attn_self_run (include/wrs_debug.h).

Reference: the reference's attention formula (src/model.rs:169-201: scale (D/H)^-0.25 on q and k, softmax(q k^T) v) in
torch fp32 on the same fp16-rounded operands.  Self-attention additionally replays `Cached::update_cache`
(model.rs:53-62) as the ancestry table: row r at position p reads the physical row anc[r][p].
Tolerance: outputs are fp16 (packed-activation order): 2e-3 of the output scale.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from whisper_rs_amd import debug as DBG
from whisper_rs_amd import _lib as L

pytestmark = pytest.mark.gpu
TOL = 2e-3


def _dev():
    return torch.device("cuda", 0)


def _randn(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device=_dev())
    g.manual_seed(seed)
    return (scale * torch.randn(shape, generator=g, device=_dev(), dtype=torch.float32)).to(dtype)


@pytest.mark.parametrize("q16_mode", [0, 1], ids=["slab_q", "fp16_q"])
@pytest.mark.parametrize("n_audio,G", [(256, 5), (3, 1), (7, 5), (2, 16)])
def test_cross_attention_kernel(gpu, n_audio, G, q16_mode):
    H, D, T = 20, 1280, 1500
    R = n_audio * G
    q = _randn((R, D), 1)                      # un-biased query projection (fp32 slab)
    qb = _randn((D,), 2, 0.1)
    scale = 64 ** -0.25
    k = _randn((n_audio, T, D), 3, 0.5, torch.float16)   # as load_cross_kv leaves them (scale / bias already applied)
    v = _randn((n_audio, T, D), 4, 1.0, torch.float16)
    out = torch.zeros((-(-R // 128) * 128) * D, dtype=torch.float16, device=_dev())
    a = DBG.DebugAttnCrossC()
    a.device_id, a.n_audio, a.G, a.H, a.D, a.T = 0, n_audio, G, H, D, T
    a.q, a.q_bias, a.q_scale = q.data_ptr(), qb.data_ptr(), scale
    a.k, a.v, a.out_packed, a.q16_mode = k.data_ptr(), v.data_ptr(), out.data_ptr(), q16_mode
    torch.cuda.synchronize()
    L.check(DBG.lib().wrs_debug_attn_cross_run(C.byref(a)))
    got = DBG.unpack_activations(out, R, D)
    qq = ((q + qb) * scale).to(torch.float16).float().view(n_audio, G, H, 64).permute(0, 2, 1, 3)   # [a, H, G, 64]
    kk = k.float().view(n_audio, T, H, 64).permute(0, 2, 3, 1)                                       # [a, H, 64, T]
    vv = v.float().view(n_audio, T, H, 64).permute(0, 2, 1, 3)
    ref = (qq @ kk).softmax(-1) @ vv                                                                  # [a, H, G, 64]
    ref = ref.permute(0, 2, 1, 3).reshape(R, D)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"cross-attention n_audio={n_audio} G={G}: rel_err={err:.2e}")
    assert err < TOL


@pytest.mark.parametrize("q16_mode", [0, 1, 2, 3], ids=["slab_qkv", "fp16_q", "mfma_dedup", "mfma_dedup_fp16_q"])
@pytest.mark.parametrize("n_rows,G,cache_pos", [(1280, 5, 447), (1280, 5, 112), (1280, 5, 0), (1280, 5, 1), (30, 5, 31), (30, 5, 32),
                                                (30, 5, 33), (12, 1, 200), (21, 7, 64), (16, 8, 100), (1280, 5, 224)])
def test_self_attention_kernel(gpu, n_rows, G, cache_pos, q16_mode):
    H, D, n_ctx = 20, 1280, 448
    scale = 64 ** -0.25
    qkv = _randn((n_rows, 3 * D), 11)
    bias = _randn((3 * D,), 12, 0.1)
    kc = _randn((n_rows, n_ctx, D), 13, 0.5, torch.float16)
    vc = _randn((n_rows, n_ctx, D), 14, 1.0, torch.float16)
    kc0, vc0 = kc.clone(), vc.clone()
    # beams share history inside their audio: physical row of (row, pos) is a random row of the same audio,
    # with long shared runs like a real beam search (ancestors converge towards the past)
    g = torch.Generator(device="cpu")
    g.manual_seed(15)
    n_groups = n_rows // G
    anc = torch.empty(n_groups, G, n_ctx, dtype=torch.int32)
    cur = torch.arange(G).expand(n_groups, G).clone()
    base = (torch.arange(n_groups) * G).view(-1, 1)
    for p in range(n_ctx - 1, -1, -1):
        if p % 7 == 0:
            cur = torch.gather(cur, 1, torch.randint(0, G, (n_groups, G), generator=g))
        anc[:, :, p] = (base + cur).to(torch.int32)
    anc = anc.view(n_rows, n_ctx).contiguous()
    anc_d = anc.to(_dev())
    out = torch.zeros((-(-n_rows // 128) * 128) * D, dtype=torch.float16, device=_dev())
    a = DBG.DebugAttnSelfC()
    a.device_id, a.n_rows, a.rows_per_audio, a.H, a.D, a.n_ctx, a.cache_pos = 0, n_rows, G, H, D, n_ctx, cache_pos
    a.qkv, a.qkv_bias, a.scale = qkv.data_ptr(), bias.data_ptr(), scale
    a.kc, a.vc, a.anc, a.out_packed, a.q16_mode = kc.data_ptr(), vc.data_ptr(), anc_d.data_ptr(), out.data_ptr(), q16_mode
    torch.cuda.synchronize()
    L.check(DBG.lib().wrs_debug_attn_self_run(C.byref(a)))
    got = DBG.unpack_activations(out, n_rows, D)
    x = qkv + bias
    qn = x[:, :D] * scale
    if q16_mode in (1, 3):
        qn = qn.to(torch.float16).float()  # the query reaches the kernel as fp16 in this form
    kn = (x[:, D:2 * D] * scale).to(torch.float16)
    vn = x[:, 2 * D:].to(torch.float16)
    # the new token's k/v are appended at cache_pos of the row's own physical row; nothing else is touched.  (The kernel
    # rounds (x + b) * scale to fp16 in one step — v_fma_mix — where torch rounds to fp32 first: exact ties may differ by
    # one fp16 ulp, measured on 1 element in 38 400.)
    torch.testing.assert_close(kc[:, cache_pos].float(), kn.float(), rtol=1e-3, atol=1e-7)
    torch.testing.assert_close(vc[:, cache_pos].float(), vn.float(), rtol=1e-3, atol=1e-7)
    assert float((kc[:, cache_pos] != kn).float().mean()) < 1e-3
    kc0[:, cache_pos], vc0[:, cache_pos] = kc[:, cache_pos], vc[:, cache_pos]
    assert torch.equal(kc, kc0) and torch.equal(vc, vc0)
    L_ = cache_pos + 1
    idx = anc_d[:, :L_].long().clone()
    idx[:, cache_pos] = torch.arange(n_rows, device=_dev())
    pos = torch.arange(L_, device=_dev()).expand(n_rows, L_)
    K = kc0[idx, pos].float().view(n_rows, L_, H, 64).permute(0, 2, 3, 1)   # [r, H, 64, L]
    V = vc0[idx, pos].float().view(n_rows, L_, H, 64).permute(0, 2, 1, 3)   # [r, H, L, 64]
    Q = qn.view(n_rows, H, 1, 64)
    ref = ((Q @ K).softmax(-1) @ V).reshape(n_rows, D)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"self-attention rows={n_rows} G={G} len={L_}: rel_err={err:.2e}")
    assert err < TOL

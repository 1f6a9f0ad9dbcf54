"""GPU parity: fused SuppressBlank/SuppressTokens/TimestampTokens + log_softmax + top-k kernel
(reference decode/filter/*.rs, extractor/beam_search.rs:37,62-65) vs the oracle's op-for-op filters.
Index work is bit-exact (token ids, -inf pattern); log-probs within 2e-5."""
import numpy as np
import pytest
import torch

from helpers import gpu_model, oracle_ids, oracle_decode_options
from whisper_rs_amd.consts import MODEL_DIMS, TokenIds
from oracle import decode as OD

pytestmark = pytest.mark.gpu
V = 51864


def _oracle_filtered(task, logits, tokens, sample_begin):
    lg = torch.from_numpy(logits.copy())
    task.prompt_ref[0].sample_begin = sample_begin
    for f in task.filters:
        f.apply(lg, torch.from_numpy(tokens))
    return lg


def _histories(rng, n_rows, sample_begin, n_sampled, ids):
    toks = rng.integers(0, 50000, size=(n_rows, sample_begin + n_sampled)).astype(np.int64)
    toks[:, sample_begin - 1] = ids.sot
    for r in range(n_rows):  # sprinkle timestamp tokens to hit every pairing rule
        for c in range(sample_begin, sample_begin + n_sampled):
            if rng.random() < 0.5:
                toks[r, c] = ids.ts_begin + rng.integers(0, 1501)
    return toks


@pytest.mark.parametrize("timestamps,suppress_blank,mit", [(True, True, 1.0), (True, False, None), (False, True, None)])
@pytest.mark.parametrize("n_sampled", [0, 1, 2, 5])
def test_filter_topk_vs_oracle(gpu, timestamps, suppress_blank, mit, n_sampled):
    rng = np.random.default_rng(100 + n_sampled)
    ids = TokenIds.english()
    n_rows, sample_begin, k = 10, 3, 6
    gm = gpu_model(gpu, "nano")
    opts = gpu.DecodeOptions(timestamps=timestamps, suppress_blank=suppress_blank, max_initial_timestamp=mit, max_audio=2)
    gt = gpu.DecodeTask(gm, ids, opts)
    oo = oracle_decode_options(timestamps=timestamps, suppress_blank=suppress_blank, max_initial_timestamp=mit)
    from helpers import oracle_model
    ot = OD.DecodeTask(oracle_model("nano"), oracle_ids(), oo)
    logits = (rng.standard_normal((n_rows, V)) * 3).astype(np.float32)
    # make the timestamp-mass rule fire on some rows and not on others
    logits[::2, ids.ts_begin:] += 2.5
    logits[1::2, ids.ts_begin:] -= 2.5
    tokens = _histories(rng, n_rows, sample_begin, n_sampled, ids)
    want_f = _oracle_filtered(ot, logits, tokens, sample_begin)
    want_lp = want_f.log_softmax(-1, dtype=torch.float32)
    wl, wi = want_lp.topk(k, dim=-1)
    lp, idx, filt = gt.filter_topk(logits, tokens, sample_begin, k, want_filtered=True)
    np.testing.assert_array_equal(np.isneginf(filt), np.isneginf(want_f.numpy()))
    np.testing.assert_array_equal(filt[~np.isneginf(filt)], want_f.numpy()[~np.isneginf(filt)])
    np.testing.assert_array_equal(idx, wi.numpy())
    np.testing.assert_allclose(lp, wl.numpy(), atol=2e-5, rtol=0)
    gt.close()


def test_filter_topk_greedy_k1_and_few_candidates(gpu):
    """first step with max_initial_timestamp 0.0: exactly one allowed token; k=6 must return it then -inf."""
    rng = np.random.default_rng(3)
    ids = TokenIds.english()
    gm = gpu_model(gpu, "nano")
    gt = gpu.DecodeTask(gm, ids, gpu.DecodeOptions(max_initial_timestamp=0.0, max_audio=2))
    logits = rng.standard_normal((2, V)).astype(np.float32)
    tokens = np.full((2, 1), ids.sot, np.int64)
    lp, idx = gt.filter_topk(logits, tokens, 1, 6)
    assert (idx[:, 0] == ids.ts_begin).all() and (idx[:, 1:] == -1).all()
    np.testing.assert_allclose(lp[:, 0], 0.0, atol=1e-6)
    assert np.isneginf(lp[:, 1:]).all()
    lp1, idx1 = gt.filter_topk(logits, tokens, 1, 1)
    assert (idx1[:, 0] == ids.ts_begin).all()
    gt.close()

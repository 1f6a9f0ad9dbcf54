"""GPU parity: AudioEncoder / TextDecoder forward (reference src/model.rs) vs the CPU oracle.

Floating point.  The HIP path multiplies in fp16 with fp32 accumulation, fp32 residual stream,
fp32 LayerNorm/softmax; weights are fp16-representable so both sides see identical inputs.
Tolerances (stated in each assert) are on LayerNorm-ed / logit outputs of O(1) magnitude."""
import numpy as np
import pytest
import torch

from helpers import gpu_model, oracle_model, mel_windows
from whisper_rs_amd.consts import MODEL_DIMS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims_name,n", [("nano", 1), ("micro", 2), ("tiny.en", 1)])
def test_encoder_forward_vs_oracle(gpu, dims_name, n):
    mel = mel_windows(n)
    want = oracle_model(dims_name).encoder.forward(torch.from_numpy(mel)).numpy()
    got = gpu_model(gpu, dims_name).encoder_forward(mel)
    assert got.shape == want.shape
    err = np.abs(got - want)
    # fp16 activations: ~1e-3 relative per GEMM; outputs are LayerNorm-ed (unit scale)
    assert err.max() < 4e-2, f"{dims_name}: max abs err {err.max()} (mean {err.mean()})"
    assert err.mean() < 3e-3, f"{dims_name}: mean abs err {err.mean()}"


def test_encoder_rejects_bad_shape(gpu):
    with pytest.raises(AssertionError):
        gpu_model(gpu, "nano").encoder_forward(np.zeros((1, 80, 2999), np.float32))


@pytest.mark.parametrize("dims_name,n_batch", [("nano", 3), ("micro", 5)])
def test_decoder_forward_teacher_forced(gpu, dims_name, n_batch):
    """forward_ext with a 4-token prefix then single tokens with the reference's offset = loop index
    (decode/mod.rs:244), including an update_cache re-index (model.rs:53-62) in between."""
    dims = MODEL_DIMS[dims_name]
    om, gm = oracle_model(dims_name), gpu_model(gpu, dims_name)
    rng = np.random.default_rng(5)
    xa = rng.standard_normal((n_batch, dims.n_audio_ctxs, dims.n_text_states)).astype(np.float32)
    toks = rng.integers(0, dims.n_vocab, size=(n_batch, 9)).astype(np.int64)
    om.clear_cache()
    gm.clear_cache()
    xa_t = torch.from_numpy(xa)
    want0 = om.decoder.forward_ext(torch.from_numpy(toks[:, :4]), xa_t, 0).numpy()
    got0 = gm.decoder_forward_ext(toks[:, :4], xa, 0)
    scale = np.abs(want0).max()
    assert np.abs(got0 - want0).max() < 2e-2 * scale, f"prefill logits err {np.abs(got0 - want0).max()} (scale {scale})"
    src = rng.permutation(n_batch).astype(np.int64)
    for i in range(1, 6):
        if i == 3:
            om.update_cache(torch.from_numpy(src))
            gm.update_cache(src)
        t = toks[:, 3 + i:4 + i]
        want = om.decoder.forward_ext(torch.from_numpy(t), xa_t, i).numpy()
        got = gm.decoder_forward_ext(t, None, i)
        err = np.abs(got - want).max()
        assert err < 2e-2 * scale, f"step {i}: logits err {err} (scale {scale})"
        assert (got.argmax(-1) == want.argmax(-1)).mean() >= 0.5
    om.clear_cache()
    gm.clear_cache()


def test_decoder_forward_errors(gpu):
    gm = gpu_model(gpu, "nano")
    gm.clear_cache()
    dims = MODEL_DIMS["nano"]
    with pytest.raises(gpu.WrsError):  # out-of-range token
        gm.decoder_forward_ext(np.array([[dims.n_vocab]], np.int64), np.zeros((1, 1500, dims.n_text_states), np.float32), 0)
    with pytest.raises(gpu.WrsError):  # positional overflow (reference panics slicing positional_embedding)
        gm.decoder_forward_ext(np.zeros((1, 2), np.int64), np.zeros((1, 1500, dims.n_text_states), np.float32), dims.n_text_ctxs - 1)
    gm.clear_cache()


def test_weights_file_loader_equals_set_tensor(gpu, tmp_path):
    """`vars.load(file)` (examples/transcribe.rs:37): a safetensors file of the VarStore-named tensors (F32 and F16 mixed,
    plus a tensor the architecture does not have) gives the same model as wrs_model_set_tensor."""
    from safetensors.numpy import save_file
    from helpers import weights
    w = weights("nano")
    on_disk = {k: (v.astype(np.float16) if v.ndim == 2 and "embedding" not in k else v) for k, v in w.items()}
    on_disk["encoder.positional_embedding"] = np.zeros((1500, MODEL_DIMS["nano"].n_audio_states), np.float32)  # OpenAI buffer: ignored
    path = tmp_path / "nano.safetensors"
    save_file(on_disk, str(path))
    m = gpu.Whisper(MODEL_DIMS["nano"])
    assert m.load_file(path) == len(w)
    ref = gpu.Whisper(MODEL_DIMS["nano"], {k: (v.astype(np.float16).astype(np.float32) if v.ndim == 2 and "embedding" not in k else v) for k, v in w.items()})
    mel = mel_windows(1)
    np.testing.assert_array_equal(m.encoder_forward(mel), ref.encoder_forward(mel))
    m2 = gpu.Whisper(MODEL_DIMS["nano"])
    with pytest.raises(gpu.WrsError):
        m2.load_file(tmp_path / "missing.safetensors")
    m.close(); ref.close(); m2.close()

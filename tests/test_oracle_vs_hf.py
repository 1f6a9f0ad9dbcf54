"""CPU: the reference has no tests for its model code, so the oracle's network is cross-checked against an
independent implementation of the same architecture: transformers' WhisperModel (local code, random-init
config, NO download) loaded with the same synthetic weights under HF's parameter names."""
import numpy as np
import pytest
import torch

from helpers import oracle_model, weights, mel_windows
from whisper_rs_amd.consts import MODEL_DIMS


def _hf_model(dims_name):
    tr = pytest.importorskip("transformers")
    d = MODEL_DIMS[dims_name]
    cfg = tr.WhisperConfig(vocab_size=d.n_vocab, num_mel_bins=80, d_model=d.n_audio_states, encoder_layers=d.n_audio_layers,
                           encoder_attention_heads=d.n_audio_heads, decoder_layers=d.n_text_layers, decoder_attention_heads=d.n_text_heads,
                           encoder_ffn_dim=4 * d.n_audio_states, decoder_ffn_dim=4 * d.n_text_states, max_source_positions=d.n_audio_ctxs,
                           max_target_positions=d.n_text_ctxs, activation_function="gelu", dropout=0.0, attention_dropout=0.0,
                           activation_dropout=0.0, pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1)
    m = tr.WhisperModel(cfg).eval()
    w = {k: torch.from_numpy(v) for k, v in weights(dims_name).items()}
    from oracle.model import sinusoids
    sd = {}

    def attn(src, dst):
        sd[dst + ".q_proj.weight"], sd[dst + ".q_proj.bias"] = w[src + ".query.weight"], w[src + ".query.bias"]
        sd[dst + ".k_proj.weight"] = w[src + ".key.weight"]
        sd[dst + ".v_proj.weight"], sd[dst + ".v_proj.bias"] = w[src + ".value.weight"], w[src + ".value.bias"]
        sd[dst + ".out_proj.weight"], sd[dst + ".out_proj.bias"] = w[src + ".out.weight"], w[src + ".out.bias"]

    for name in ("conv1", "conv2"):
        sd[f"encoder.{name}.weight"], sd[f"encoder.{name}.bias"] = w[f"encoder.{name}.weight"], w[f"encoder.{name}.bias"]
    sd["encoder.embed_positions.weight"] = sinusoids(d.n_audio_ctxs, d.n_audio_states)
    for i in range(d.n_audio_layers):
        s, t = f"encoder.blocks.{i}", f"encoder.layers.{i}"
        attn(s + ".attn", t + ".self_attn")
        for a, b in ((".attn_ln", ".self_attn_layer_norm"), (".mlp_ln", ".final_layer_norm")):
            sd[t + b + ".weight"], sd[t + b + ".bias"] = w[s + a + ".weight"], w[s + a + ".bias"]
        sd[t + ".fc1.weight"], sd[t + ".fc1.bias"] = w[s + ".mlp.0.weight"], w[s + ".mlp.0.bias"]
        sd[t + ".fc2.weight"], sd[t + ".fc2.bias"] = w[s + ".mlp.2.weight"], w[s + ".mlp.2.bias"]
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = w["encoder.ln_post.weight"], w["encoder.ln_post.bias"]
    sd["decoder.embed_tokens.weight"] = w["decoder.token_embedding.weight"]
    sd["decoder.embed_positions.weight"] = w["decoder.positional_embedding"]
    for i in range(d.n_text_layers):
        s, t = f"decoder.blocks.{i}", f"decoder.layers.{i}"
        attn(s + ".attn", t + ".self_attn")
        attn(s + ".cross_attn", t + ".encoder_attn")
        for a, b in ((".attn_ln", ".self_attn_layer_norm"), (".cross_attn_ln", ".encoder_attn_layer_norm"), (".mlp_ln", ".final_layer_norm")):
            sd[t + b + ".weight"], sd[t + b + ".bias"] = w[s + a + ".weight"], w[s + a + ".bias"]
        sd[t + ".fc1.weight"], sd[t + ".fc1.bias"] = w[s + ".mlp.0.weight"], w[s + ".mlp.0.bias"]
        sd[t + ".fc2.weight"], sd[t + ".fc2.bias"] = w[s + ".mlp.2.weight"], w[s + ".mlp.2.bias"]
    sd["decoder.layer_norm.weight"], sd["decoder.layer_norm.bias"] = w["decoder.ln.weight"], w["decoder.ln.bias"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in k for k in missing), missing  # Whisper's key projection has no bias (model.rs:138-141)
    return m


@pytest.mark.parametrize("dims_name", ["micro"])
def test_oracle_matches_independent_whisper_implementation(dims_name):
    d = MODEL_DIMS[dims_name]
    hf = _hf_model(dims_name)
    om = oracle_model(dims_name)
    mel = torch.from_numpy(mel_windows(1))
    toks = torch.tensor([[50257, 50363, 1234, 4321, 50400]])
    with torch.no_grad():
        enc_o = om.encoder.forward(mel)
        enc_h = hf.encoder(mel).last_hidden_state
        np.testing.assert_allclose(enc_o.numpy(), enc_h.numpy(), atol=2e-4)
        om.clear_cache()
        logits_o = om.decoder.forward_ext(toks, enc_o, 0)
        dec_h = hf.decoder(input_ids=toks, encoder_hidden_states=enc_h).last_hidden_state
        logits_h = dec_h @ hf.decoder.embed_tokens.weight.T
        np.testing.assert_allclose(logits_o.numpy(), logits_h.numpy(), atol=5e-4)
        # incremental decoding with the Cached K/V path == full-sequence recompute
        om.clear_cache()
        inc = [om.decoder.forward_ext(toks[:, :2], enc_o, 0)[:, -1]]
        for i in range(2, toks.shape[1]):
            inc.append(om.decoder.forward_ext(toks[:, i:i + 1], enc_o, i)[:, -1])
        om.clear_cache()
        np.testing.assert_allclose(torch.stack(inc, 1).numpy(), logits_o[:, 1:].numpy(), atol=5e-4)

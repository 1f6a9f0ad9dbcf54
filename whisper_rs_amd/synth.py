"""Seeded synthetic inputs (weights, audio) shared by tests, bench.py and the oracle.

There are no checkpoints or audio files in the build environment, so every parity
and benchmark run uses inputs regenerated from a seed.  Tensor names are the
reference's VarStore paths (= OpenAI state_dict keys; SURVEY.md Appendix A,
reference src/model.rs:354-402,453-484).

All weight values are rounded to fp16-representable numbers so that the fp32 CPU
oracle and the fp16 HIP path see *identical inputs* (the north-star's parity
condition); the arithmetic then differs only by activation rounding.
"""
import zlib
import numpy as np

from .consts import ModelDims, N_MELS, N_SAMPLES, SAMPLE_RATE


def weight_manifest(dims: ModelDims):
    """[(name, shape)] in a fixed order; every name `Whisper::new` registers."""
    out = []
    D, Dt = dims.n_audio_states, dims.n_text_states

    def attn(prefix, d):
        out.append((f"{prefix}.query.weight", (d, d)))
        out.append((f"{prefix}.query.bias", (d,)))
        out.append((f"{prefix}.key.weight", (d, d)))  # no bias: model.rs:133-144
        out.append((f"{prefix}.value.weight", (d, d)))
        out.append((f"{prefix}.value.bias", (d,)))
        out.append((f"{prefix}.out.weight", (d, d)))
        out.append((f"{prefix}.out.bias", (d,)))

    def ln(prefix, d):
        out.append((f"{prefix}.weight", (d,)))
        out.append((f"{prefix}.bias", (d,)))

    def block(prefix, d, cross):
        attn(f"{prefix}.attn", d)
        ln(f"{prefix}.attn_ln", d)
        if cross:
            attn(f"{prefix}.cross_attn", d)
            ln(f"{prefix}.cross_attn_ln", d)
        out.append((f"{prefix}.mlp.0.weight", (4 * d, d)))
        out.append((f"{prefix}.mlp.0.bias", (4 * d,)))
        out.append((f"{prefix}.mlp.2.weight", (d, 4 * d)))
        out.append((f"{prefix}.mlp.2.bias", (d,)))
        ln(f"{prefix}.mlp_ln", d)

    out.append(("encoder.conv1.weight", (D, N_MELS, 3)))
    out.append(("encoder.conv1.bias", (D,)))
    out.append(("encoder.conv2.weight", (D, D, 3)))
    out.append(("encoder.conv2.bias", (D,)))
    for i in range(dims.n_audio_layers):
        block(f"encoder.blocks.{i}", D, False)
    ln("encoder.ln_post", D)
    out.append(("decoder.token_embedding.weight", (dims.n_vocab, Dt)))
    out.append(("decoder.positional_embedding", (dims.n_text_ctxs, Dt)))
    for i in range(dims.n_text_layers):
        block(f"decoder.blocks.{i}", Dt, True)
    ln("decoder.ln", Dt)
    return out


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _fp16_round(a):
    return a.astype(np.float16).astype(np.float32)


def make_tensor(name, shape, seed=0, eot_ramp=0.0, emb_gain=1.0):
    """One synthetic weight tensor (float32 values that are exactly fp16-representable)."""
    g = _rng(seed, name)
    n = int(np.prod(shape))
    if name.endswith("_ln.weight") or name.endswith("ln_post.weight") or name == "decoder.ln.weight":
        a = 1.0 + 0.1 * g.standard_normal(n, dtype=np.float32)
    elif name.endswith("_ln.bias") or name.endswith("ln_post.bias") or name == "decoder.ln.bias":
        a = 0.1 * g.standard_normal(n, dtype=np.float32)
    elif name.endswith(".bias"):
        a = 0.02 * g.standard_normal(n, dtype=np.float32)
    elif name == "decoder.token_embedding.weight":
        # logits = LN(x) . E^T have std ~ emb_gain * 1.5: a flat-ish but not degenerate next-token
        # distribution (top-6 log-probs a few tenths of a nat apart), so beam candidates are rarely
        # closer than the fp16 activation noise of the HIP path (tests/test_gpu_decode.py)
        a = (1.5 * emb_gain / np.sqrt(shape[1])) * g.standard_normal(n, dtype=np.float32)
    elif name == "decoder.positional_embedding":
        a = (0.5 / np.sqrt(shape[1])) * g.standard_normal(n, dtype=np.float32)
    else:
        fan_in = int(np.prod(shape[1:]))
        a = g.standard_normal(n, dtype=np.float32) / np.float32(np.sqrt(fan_in))
        if ".cross_attn.query.weight" in name or ".cross_attn.key.weight" in name:
            a *= 3.0  # peaked cross-attention: the decode then really depends on the audio window
    return _fp16_round(a.reshape(shape))


def make_weights(dims: ModelDims, seed=0, emb_gain=1.0, force_eot_after=None):
    """dict name -> float32 ndarray.

    `force_eot_after=k` makes the synthetic model emit <|endoftext|> around decode
    position k (random weights never do): the positional embedding gets a ramp
    along a fixed direction u and the EOT embedding row is aligned with u, so the
    EOT logit grows with position.  Exercises the finished/patience path of the
    beam search (reference beam_search.rs:88-146).
    """
    w = {name: make_tensor(name, shape, seed, emb_gain=emb_gain) for name, shape in weight_manifest(dims)}
    if force_eot_after is not None:
        from .consts import TokenIds
        ids = TokenIds.for_vocab(dims.n_vocab)
        D = dims.n_text_states
        u = _rng(seed, "eot_direction").standard_normal(D, dtype=np.float32)
        u /= np.linalg.norm(u)
        pos = w["decoder.positional_embedding"]
        ramp = np.clip((np.arange(dims.n_text_ctxs, dtype=np.float32) - force_eot_after) / 2.0, -1.0, 1.0)
        pos = pos + 6.0 * ramp[:, None] * u[None, :]
        w["decoder.positional_embedding"] = _fp16_round(pos)
        emb = w["decoder.token_embedding.weight"]
        emb[ids.eot] = _fp16_round(2.5 * emb_gain * u)
    return w


def make_audio(n_windows=1, seed=1234, n_samples=N_SAMPLES, index0=0):
    """[n_windows, n_samples] float32: 0.1*N(0,1) + two sines (220 Hz, 1 kHz; amp 0.2)
    with a slow per-window amplitude envelope so the log-mel is not flat.
    index0: position of the first window in a longer recording — make_audio(n, seed + k, index0=k) are exactly windows
    k .. k+n-1 of make_audio(k + n, seed) (ranks of a sharded job generate only their own windows)."""
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    out = np.empty((n_windows, n_samples), dtype=np.float32)
    for j in range(n_windows):
        i = index0 + j
        g = _rng(seed + j, "audio")
        noise = (0.05 + 0.03 * (i % 5)) * g.standard_normal(n_samples, dtype=np.float32)
        f1, f2 = 150.0 + 97.0 * (i % 13), 700.0 + 310.0 * (i % 11)  # distinct spectra per window
        env = 0.5 + 0.5 * np.sin(2 * np.pi * (0.2 + 0.05 * (i % 7)) * t)
        out[j] = noise + (0.2 * env * (np.sin(2 * np.pi * f1 * t) + np.sin(2 * np.pi * f2 * t))).astype(np.float32)
    return out

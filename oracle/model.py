"""ORACLE (test infrastructure) — Whisper model.  See oracle/__init__.py.

Restates reference src/model.rs op for op with the same ATen CPU operators `tch`
binds (F.linear, F.layer_norm, F.conv1d, gelu("none"), softmax, matmul).
Weights are a dict name -> float32 torch tensor keyed by the reference's VarStore
paths.
"""
import math
import torch
import torch.nn.functional as F


class Cached:
    """model.rs:37-98 `Cached<nn::Linear>`: inner Linear + concatenating K/V cache."""

    def __init__(self, w, b, threshold, memo_long_inputs=False):
        self.w, self.b, self.threshold = w, b, threshold
        self.cache = None
        # memo_long_inputs: cross-attention inputs (len 1500 > threshold) always take the
        # "save as-is" branch (:89-90), i.e. the reference recomputes Linear(xa) at every decode
        # step.  The result is the same tensor every time, so memoising it is
        # numerically identical; the faithful recompute is kept for the timed CPU baseline.
        self.memo_long_inputs = memo_long_inputs
        self._memo = None  # (input tensor object, Linear(input))

    def forward(self, xs):
        if self.memo_long_inputs and xs.shape[1] > self.threshold:
            if self._memo is None or self._memo[0] is not xs:
                self._memo = (xs, F.linear(xs, self.w, self.b))
            self.cache = self._memo[1]
            return self.cache
        out = F.linear(xs, self.w, self.b)  # :79
        if self.cache is not None and xs.shape[1] <= self.threshold:  # :83-87
            out = torch.cat([self.cache, out], dim=1)
        self.cache = out  # :95
        return out

    def update_cache(self, src):  # :53-62
        if self.memo_long_inputs and self._memo is not None and self.cache is self._memo[1]:
            # cross-attention K/V: the reference gathers this tensor too, and overwrites it with Linear(xa) at the next forward
            # (:89-95) without ever reading the gathered rows; with the memo in place the gather is skipped as well
            return
        if self.cache is not None:
            self.cache = self.cache[src]

    def clear_cache(self):  # :64-66
        self.cache = None
        self._memo = None


def sinusoids(length, channels, max_timescale=10000.0):
    """model.rs:102-112 (all float32)."""
    assert channels % 2 == 0
    inc = torch.tensor(math.log(max_timescale), dtype=torch.float32) / (channels // 2 - 1)
    inc = torch.tensor(float(inc), dtype=torch.float32)  # f32::ln(..)/ (n as f32) evaluated in f32
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    scaled = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([scaled.sin(), scaled.cos()], dim=1)


class MultiHeadAttention:
    """model.rs:115-217."""

    def __init__(self, w, prefix, n_heads, threshold, memo_cross=False):
        self.n_heads = n_heads
        self.qw, self.qb = w[f"{prefix}.query.weight"], w[f"{prefix}.query.bias"]
        self.key = Cached(w[f"{prefix}.key.weight"], None, threshold, memo_cross)  # bias: false (:138-141)
        self.value = Cached(w[f"{prefix}.value.weight"], w[f"{prefix}.value.bias"], threshold, memo_cross)
        self.ow, self.ob = w[f"{prefix}.out.weight"], w[f"{prefix}.out.bias"]

    def forward_ext(self, xs, xa=None, mask=None):  # :158-206
        q = F.linear(xs, self.qw, self.qb)
        src = xs if xa is None else xa
        k = self.key.forward(src)
        v = self.value.forward(src)
        n_batch, n_ctx, n_state = q.shape
        scale = float((n_state // self.n_heads)) ** -0.25  # :169 (f64 scalar)
        q = q.view(n_batch, q.shape[1], self.n_heads, -1).permute(0, 2, 1, 3) * scale
        k = k.view(k.shape[0], k.shape[1], self.n_heads, -1).permute(0, 2, 3, 1) * scale
        v = v.view(v.shape[0], v.shape[1], self.n_heads, -1).permute(0, 2, 1, 3)
        qk = q.matmul(k)
        if mask is not None:
            qk = qk + mask[:n_ctx, :n_ctx]  # :194-196 (broadcasts against [.., n_ctx, Lk]; Lk == n_ctx or n_ctx == 1)
        w = qk.softmax(dim=-1)
        wv = w.matmul(v).permute(0, 2, 1, 3).flatten(2)
        return F.linear(wv, self.ow, self.ob)

    def update_cache(self, src):
        self.value.update_cache(src)
        self.key.update_cache(src)

    def clear_cache(self):
        self.value.clear_cache()
        self.key.clear_cache()


class ResidualAttentionBlock:
    """model.rs:235-336.  LayerNorm eps = 1e-5 (tch nn::LayerNormConfig default)."""

    def __init__(self, w, prefix, n_state, n_heads, cross, threshold, memo_cross=False):
        self.n_state = n_state
        self.attn = MultiHeadAttention(w, f"{prefix}.attn", n_heads, threshold)
        self.attn_ln = (w[f"{prefix}.attn_ln.weight"], w[f"{prefix}.attn_ln.bias"])
        self.cross_attn = MultiHeadAttention(w, f"{prefix}.cross_attn", n_heads, threshold, memo_cross) if cross else None
        self.cross_attn_ln = (w[f"{prefix}.cross_attn_ln.weight"], w[f"{prefix}.cross_attn_ln.bias"]) if cross else None
        self.mlp0 = (w[f"{prefix}.mlp.0.weight"], w[f"{prefix}.mlp.0.bias"])
        self.mlp2 = (w[f"{prefix}.mlp.2.weight"], w[f"{prefix}.mlp.2.bias"])
        self.mlp_ln = (w[f"{prefix}.mlp_ln.weight"], w[f"{prefix}.mlp_ln.bias"])

    def _ln(self, x, p):
        return F.layer_norm(x, (self.n_state,), p[0], p[1], 1e-5)

    def forward_ext(self, xs, xa=None, mask=None):  # :301-319
        x = xs + self.attn.forward_ext(self._ln(xs, self.attn_ln), None, mask)
        if self.cross_attn is not None:
            x = x + self.cross_attn.forward_ext(self._ln(x, self.cross_attn_ln), xa, None)
        h = F.linear(self._ln(x, self.mlp_ln), *self.mlp0)
        h = F.gelu(h, approximate="none")
        x = x + F.linear(h, *self.mlp2)
        return x

    def update_cache(self, src):
        self.attn.update_cache(src)
        if self.cross_attn is not None:
            self.cross_attn.update_cache(src)

    def clear_cache(self):
        self.attn.clear_cache()
        if self.cross_attn is not None:
            self.cross_attn.clear_cache()


class AudioEncoder:
    """model.rs:345-441."""

    def __init__(self, w, n_ctxs, n_states, n_heads, n_layers):
        self.w = w
        self.n_states = n_states
        self.blocks = [ResidualAttentionBlock(w, f"encoder.blocks.{i}", n_states, n_heads, False, n_ctxs)
                       for i in range(n_layers)]
        self.position_emb = sinusoids(n_ctxs, n_states)  # computed, not loaded (:395)

    def forward(self, xs):  # :420-440; xs [B, 80, 3000]
        w = self.w
        xs = F.gelu(F.conv1d(xs, w["encoder.conv1.weight"], w["encoder.conv1.bias"], padding=1), approximate="none")
        xs = F.gelu(F.conv1d(xs, w["encoder.conv2.weight"], w["encoder.conv2.bias"], stride=2, padding=1),
                    approximate="none").permute(0, 2, 1)
        assert tuple(xs.shape[1:]) == tuple(self.position_emb.shape), "incorrect audio shape"
        xs = xs + self.position_emb
        for b in self.blocks:
            xs = b.forward_ext(xs)
        return F.layer_norm(xs, (self.n_states,), w["encoder.ln_post.weight"], w["encoder.ln_post.bias"], 1e-5)

    def update_cache(self, src):
        for b in self.blocks:
            b.update_cache(src)

    def clear_cache(self):
        for b in self.blocks:
            b.clear_cache()


class TextDecoder:
    """model.rs:444-535."""

    def __init__(self, w, n_vocab, n_ctxs, n_states, n_heads, n_layers, memo_cross=False):
        self.w = w
        self.n_states = n_states
        self.blocks = [ResidualAttentionBlock(w, f"decoder.blocks.{i}", n_states, n_heads, True, n_ctxs, memo_cross)
                       for i in range(n_layers)]
        self.mask = torch.full((n_ctxs, n_ctxs), float("-inf")).triu_(1)  # :474-476

    def forward_ext(self, tokens, xa, offset):  # :490-522
        w = self.w
        x = F.embedding(tokens, w["decoder.token_embedding.weight"]) + \
            w["decoder.positional_embedding"][offset:offset + tokens.shape[-1]]
        for b in self.blocks:
            x = b.forward_ext(x, xa, self.mask)
        x = F.layer_norm(x, (self.n_states,), w["decoder.ln.weight"], w["decoder.ln.bias"], 1e-5)
        return x.matmul(w["decoder.token_embedding.weight"].transpose(0, 1))

    def update_cache(self, src):
        for b in self.blocks:
            b.update_cache(src)

    def clear_cache(self):
        for b in self.blocks:
            b.clear_cache()


class Whisper:
    """model.rs:538-604.  `dims` is any object with the ModelDims fields."""

    def __init__(self, weights, dims, memo_cross=True):
        w = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)).to(torch.float32) for k, v in weights.items()}
        self.dims = dims
        self.encoder = AudioEncoder(w, dims.n_audio_ctxs, dims.n_audio_states, dims.n_audio_heads, dims.n_audio_layers)
        self.decoder = TextDecoder(w, dims.n_vocab, dims.n_text_ctxs, dims.n_text_states, dims.n_text_heads,
                                   dims.n_text_layers, memo_cross)

    def forward_ext(self, mel, tokens):
        return self.decoder.forward_ext(tokens, self.encoder.forward(mel), 0)

    def update_cache(self, src):
        self.decoder.update_cache(src)
        self.encoder.update_cache(src)

    def clear_cache(self):
        self.decoder.clear_cache()
        self.encoder.clear_cache()

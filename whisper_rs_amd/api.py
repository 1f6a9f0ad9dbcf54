"""Host-side mirror of the reference crate's public API over the C-ABI (include/wrs.h).

Names, argument meaning and error behaviour follow laptou/whisper-rs:
  audio.mel_filter / pad_or_trim / log_mel_spectrogram    src/audio.rs:175,213,230
  ModelDims, Whisper{forward/update_cache/clear_cache}     src/model.rs:546-604
  TokenExtractMode, DecodeOptions, DecodeOutput, DecodeTask src/decode/mod.rs:95-419
  TranscribePrompt, TranscribeOptions, TranscribeTask       src/transcribe.rs:25-243
numpy arrays stand where the reference has `tch::Tensor`.  All arithmetic happens in
libwrs_mi355x.so on a gfx950 device; nothing here computes.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .consts import ModelDims, TokenIds, NON_SPEECH_TOKENS, N_FRAMES, N_MELS, N_SAMPLES


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return L.load().wrs_device_count()


# ---------------------------------------------------------------------------- audio
class audio:
    """reference `whisper::audio` (src/audio.rs)."""

    @staticmethod
    def load_audio(path) -> np.ndarray:
        """`audio::load_audio(path)` (audio.rs:20-173) for WAV files: mono f32 at 16 kHz."""
        lib = L.load()
        p, n = C.POINTER(C.c_float)(), C.c_int64()
        L.check(lib.wrs_load_audio(str(path).encode(), C.byref(p), C.byref(n)))
        try:
            return np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
        finally:
            lib.wrs_audio_free(p)

    @staticmethod
    def resample(x: np.ndarray, rate_in: int, rate_out: int = 16000) -> np.ndarray:
        lib = L.load()
        x = _f32(x)
        p, n = C.POINTER(C.c_float)(), C.c_int64()
        L.check(lib.wrs_resample(_ptr(x), x.shape[0], rate_in, rate_out, C.byref(p), C.byref(n)))
        try:
            return np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
        finally:
            lib.wrs_audio_free(p)

    @staticmethod
    def mel_filter() -> np.ndarray:
        out = np.empty((N_MELS, 201), dtype=np.float32)
        L.check(L.load().wrs_mel_filter(_ptr(out)))
        return out

    @staticmethod
    def pad_or_trim(x: np.ndarray, length: int) -> np.ndarray:
        x = _f32(x)
        rows = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
        out = np.empty(x.shape[:-1] + (length,), dtype=np.float32)
        L.check(L.load().wrs_pad_or_trim(_ptr(x), rows, x.shape[-1], length, _ptr(out)))
        return out

    @staticmethod
    def log_mel_spectrogram(audio_: np.ndarray, mel_filter: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
        """audio [n] -> [80, n//160]  (or [b, n] -> [b, 80, n//160], each row its own recording)."""
        a = _f32(audio_)
        single = a.ndim == 1
        if single:
            a = a[None]
        b, n = a.shape
        out = np.empty((b, N_MELS, n // 160), dtype=np.float32)
        filt = None if mel_filter is None else _f32(mel_filter)
        L.check(L.load().wrs_log_mel_spectrogram(device, _ptr(a), b, n, None if filt is None else _ptr(filt), _ptr(out),
                                                 L.WRS_MEM_HOST))
        return out[0] if single else out


# ---------------------------------------------------------------------------- tokenizer
def _take_text(ptr):
    """UTF-8 text of a library-owned char* (or None)."""
    return None if not ptr else C.string_at(ptr).decode("utf-8")


class Tokenizer:
    """reference `Tokenizer` (src/tokenize.rs:111-321): GPT-2 byte-level BPE + Whisper special ids, host-only."""

    def __init__(self, handle):
        self._lib = L.load()
        self._h = handle
        ids = L.TokenIdsC()
        L.check(self._lib.wrs_tokenizer_token_ids(self._h, C.byref(ids)))
        self.ids = TokenIds(**{k: getattr(ids, k) for k, _ in L.TokenIdsC._fields_})

    @staticmethod
    def new_from_bytes(data: bytes) -> "Tokenizer":
        h = C.c_void_p()
        buf = C.create_string_buffer(data, len(data))
        L.check(L.load().wrs_tokenizer_create_from_bytes(buf, len(data), C.byref(h)))
        return Tokenizer(h)

    @staticmethod
    def new_from_file(path: str) -> "Tokenizer":
        h = C.c_void_p()
        L.check(L.load().wrs_tokenizer_create_from_file(str(path).encode(), C.byref(h)))
        return Tokenizer(h)

    def encode(self, text: str) -> List[int]:
        raw = text.encode("utf-8")
        cap = max(16, len(raw) + 8)
        out = (C.c_int32 * cap)()
        n = C.c_int32()
        L.check(self._lib.wrs_tokenizer_encode(self._h, raw, len(raw), out, cap, C.byref(n)))
        assert n.value <= cap
        return list(out[:n.value])

    def decode(self, tokens) -> str:
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        p, n = C.c_void_p(), C.c_size_t()
        L.check(self._lib.wrs_tokenizer_decode(self._h, _ptr(t) if t.size else None, t.size, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p.value, n.value).decode("utf-8")
        finally:
            self._lib.wrs_string_free(p)

    def non_speech_tokens(self) -> List[int]:
        out = (C.c_int32 * 256)()
        n = C.c_int32()
        L.check(self._lib.wrs_tokenizer_non_speech_tokens(self._h, out, 256, C.byref(n)))
        return list(out[:n.value])

    def token_to_id(self, token: str) -> Optional[int]:
        i = self._lib.wrs_tokenizer_token_to_id(self._h, token.encode("utf-8"))
        return None if i < 0 else i

    def language_token(self, code: str) -> Optional[int]:
        i = self._lib.wrs_tokenizer_language_token(self._h, code.encode("utf-8"))
        return None if i < 0 else i

    def vocab_size(self) -> int:
        return self._lib.wrs_tokenizer_vocab_size(self._h)

    def sequence_sot(self) -> List[int]:
        """`Tokenizer::sequence_sot()` (tokenize.rs:209-213)."""
        return [self.ids.sot]

    def close(self):
        if self._h:
            self._lib.wrs_tokenizer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------- model
class Whisper:
    """reference `Whisper` (src/model.rs:538-604).  Weights are given as a dict of VarStore paths
    (what `vars.load("weights.ot")` reads, examples/transcribe.rs:37)."""

    def __init__(self, dims: ModelDims, weights: Optional[dict] = None, device: int = 0):
        self._lib = L.load()
        self.dims = dims
        self._h = C.c_void_p()
        d = L.ModelDimsC(**dims.as_dict())
        L.check(self._lib.wrs_model_create(C.byref(d), device, C.byref(self._h)))
        if weights is not None:
            self.load(weights)

    def load(self, weights: dict):
        for name, w in weights.items():
            self.set_tensor(name, w)
        L.check(self._lib.wrs_model_finalize(self._h))

    def set_tensor(self, name, w):
        if hasattr(w, "data_ptr"):  # torch tensor (host or device), fp32 or fp16
            import torch
            assert w.is_contiguous() and w.dtype in (torch.float32, torch.float16)
            shape = (C.c_int64 * w.dim())(*w.shape)
            L.check(self._lib.wrs_model_set_tensor(self._h, name.encode(), L.WRS_F32 if w.dtype == torch.float32 else L.WRS_F16,
                                                   shape, w.dim(), C.c_void_p(w.data_ptr()),
                                                   L.WRS_MEM_DEVICE if w.is_cuda else L.WRS_MEM_HOST))
            return
        w = np.ascontiguousarray(w)
        assert w.dtype in (np.float32, np.float16)
        shape = (C.c_int64 * w.ndim)(*w.shape)
        L.check(self._lib.wrs_model_set_tensor(self._h, name.encode(), L.WRS_F32 if w.dtype == np.float32 else L.WRS_F16, shape,
                                               w.ndim, _ptr(w), L.WRS_MEM_HOST))

    def finalize(self):
        L.check(self._lib.wrs_model_finalize(self._h))

    def load_file(self, path) -> int:
        """`vars.load("weights.ot")` (examples/transcribe.rs:37) from a safetensors file of the same named tensors;
        returns the number of tensors taken and finalizes the model."""
        n = C.c_int32()
        L.check(self._lib.wrs_model_load_safetensors(self._h, str(path).encode(), C.byref(n)))
        L.check(self._lib.wrs_model_finalize(self._h))
        return n.value

    def device(self):
        return self._lib.wrs_model_device(self._h)

    def encoder_forward(self, mel: np.ndarray) -> np.ndarray:
        """`AudioEncoder::forward` (model.rs:417-441): [b, 80, 3000] -> [b, 1500, D]."""
        mel = _f32(mel)
        assert mel.ndim == 3 and mel.shape[1:] == (N_MELS, N_FRAMES), "incorrect audio shape"
        out = np.empty((mel.shape[0], self.dims.n_audio_ctxs, self.dims.n_audio_states), dtype=np.float32)
        L.check(self._lib.wrs_encoder_forward(self._h, _ptr(mel), mel.shape[0], _ptr(out)))
        return out

    def decoder_forward_ext(self, tokens: np.ndarray, xa: Optional[np.ndarray], offset: int) -> np.ndarray:
        """`TextDecoder::forward_ext(tokens, xa, offset)` (model.rs:490-522): [b, s] -> logits [b, s, V].
        xa=None reuses the cross-attention K/V of the previous call."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int64)
        b, s = tokens.shape
        out = np.empty((b, s, self.dims.n_vocab), dtype=np.float32)
        xa_ = None if xa is None else _f32(xa)
        L.check(self._lib.wrs_decoder_forward(self._h, _ptr(tokens), b, s, None if xa_ is None else _ptr(xa_), int(offset), _ptr(out)))
        return out

    def update_cache(self, sources_indices):
        src = np.ascontiguousarray(sources_indices, dtype=np.int64)
        L.check(self._lib.wrs_model_update_cache(self._h, _ptr(src), src.shape[0]))

    def clear_cache(self):
        L.check(self._lib.wrs_model_clear_cache(self._h))

    def close(self):
        if self._h:
            self._lib.wrs_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------- decode
@dataclass
class TokenExtractMode:
    """reference enum `TokenExtractMode` (decode/extractor/mod.rs:12-16)."""
    kind: str = "beam_search"  # "greedy" | "beam_search"
    beam_size: int = 5
    patience: float = 1.0
    group_size: int = 1  # Greedy { group_sze }

    @staticmethod
    def Greedy(group_size=1):
        return TokenExtractMode("greedy", group_size=group_size)

    @staticmethod
    def BeamSearch(beam_size=5, patience=1.0):
        return TokenExtractMode("beam_search", beam_size=beam_size, patience=patience)


@dataclass
class DecodeOptions:
    """reference `DecodeOptions` (decode/mod.rs:95-106); defaults = examples/transcribe.rs:44-60."""
    sample_len: Optional[int] = None
    token_extract_mode: TokenExtractMode = field(default_factory=TokenExtractMode)
    len_penalty: Optional[float] = None
    max_initial_timestamp: Optional[float] = 1.0
    timestamps: bool = True
    suppress_blank: bool = True
    suppress_non_speech: bool = True
    suppress_tokens: Optional[List[int]] = None
    prompt: Optional[Sequence[int]] = None
    # build-side knobs (not in the reference)
    position_offset_quirk: bool = True   # keep the reference's decoder offset = loop index
    max_audio: int = 1                   # windows per run()
    non_speech_tokens: Sequence[int] = NON_SPEECH_TOKENS
    sot_sequence_tail: Optional[Sequence[int]] = None  # tokens after <|startoftranscript|>: [lang, task] for multilingual models


@dataclass
class DecodeOutput:
    """reference `DecodeOutput` (decode/mod.rs:109-119).  text needs the BPE tokenizer."""
    tokens: np.ndarray
    avg_logprob: float
    no_speech_prob: float
    sum_logprob: float
    audio_features_device: int  # device pointer to [1500, D] fp32
    text: Optional[str] = None  # filled when the task has a tokenizer (decode/mod.rs:404)


def _fill_decode_options(o: DecodeOptions, keep: list) -> L.DecodeOptionsC:
    c = L.DecodeOptionsC()
    c.sample_len = o.sample_len if o.sample_len is not None else 0
    m = o.token_extract_mode
    if m.kind == "greedy":
        c.extract_mode, c.group_size, c.patience = L.WRS_EXTRACT_GREEDY, m.group_size, 1.0
    else:
        c.extract_mode, c.group_size, c.patience = L.WRS_EXTRACT_BEAM_SEARCH, m.beam_size, m.patience
    c.has_len_penalty = o.len_penalty is not None
    c.len_penalty = o.len_penalty or 0.0
    c.has_max_initial_timestamp = o.max_initial_timestamp is not None
    c.max_initial_timestamp = o.max_initial_timestamp or 0.0
    c.timestamps, c.suppress_blank, c.suppress_non_speech = o.timestamps, o.suppress_blank, o.suppress_non_speech
    st = np.asarray(o.suppress_tokens or [], dtype=np.int32)
    ns = np.asarray(o.non_speech_tokens, dtype=np.int32)
    keep += [st, ns]
    c.suppress_tokens = st.ctypes.data_as(C.POINTER(C.c_int32)) if st.size else None
    c.n_suppress_tokens = st.size
    c.non_speech_tokens = ns.ctypes.data_as(C.POINTER(C.c_int32)) if ns.size else None
    c.n_non_speech_tokens = ns.size
    c.position_offset_quirk = o.position_offset_quirk
    c.max_audio = o.max_audio
    tail = np.asarray(list(o.sot_sequence_tail or []), dtype=np.int32)
    keep.append(tail)
    c.sot_sequence_tail = tail.ctypes.data_as(C.POINTER(C.c_int32)) if tail.size else None
    c.n_sot_sequence_tail = tail.size
    return c


def _ids_c(ids: TokenIds) -> L.TokenIdsC:
    return L.TokenIdsC(**{k: getattr(ids, k) for k, _ in L.TokenIdsC._fields_})


class DecodeTask:
    """reference `DecodeTask` (decode/mod.rs:24-419)."""

    def __init__(self, model: Whisper, tokenizer_ids: TokenIds, options: DecodeOptions):
        self._lib = L.load()
        self.model, self.ids, self.options = model, tokenizer_ids, options
        self._keep = []
        c = _fill_decode_options(options, self._keep)
        ids = _ids_c(tokenizer_ids)
        self._h = C.c_void_p()
        L.check(self._lib.wrs_decode_task_create(model._h, C.byref(ids), C.byref(c), C.byref(self._h)))
        if options.prompt is not None:
            self.set_prompt(options.prompt)

    def set_prompt(self, prompt):
        p = np.ascontiguousarray(prompt if prompt is not None else [], dtype=np.int64)
        L.check(self._lib.wrs_decode_task_set_prompt(self._h, _ptr(p) if p.size else None, p.size))

    def set_tokenizer(self, tokenizer: Optional["Tokenizer"]):
        """the reference passes Rc<Tokenizer> to DecodeTask::new (decode/mod.rs:122-126); with it outputs carry text."""
        self._tokenizer = tokenizer
        L.check(self._lib.wrs_decode_task_set_tokenizer(self._h, tokenizer._h if tokenizer else None))

    def _collect(self, res):
        outs = []
        for i in range(res.n_outputs):
            o = res.outputs[i]
            toks = np.ctypeslib.as_array(o.tokens, shape=(o.n_tokens,)).copy() if o.n_tokens > 0 else np.zeros(0, np.int64)
            outs.append(DecodeOutput(toks, o.avg_logprob, o.no_speech_prob, o.sum_logprob, o.audio_features_device or 0, _take_text(o.text)))
        self.n_steps = res.n_steps
        self._lib.wrs_decode_result_free(C.byref(res))
        return outs

    def run(self, mel: np.ndarray) -> List[DecodeOutput]:
        """`DecodeTask::run(&mel)`: mel [80, 3000] (one window, as the reference) or [n, 80, 3000]."""
        mel = _f32(mel)
        if mel.ndim == 2:
            mel = mel[None]
        assert mel.shape[1:] == (N_MELS, N_FRAMES), "2d mel spectrogram"
        res = L.DecodeResultC()
        L.check(self._lib.wrs_decode_task_run(self._h, _ptr(mel), mel.shape[0], L.WRS_MEM_HOST, C.byref(res)))
        return self._collect(res)

    def run_features(self, feats: np.ndarray) -> List[DecodeOutput]:
        feats = _f32(feats)
        res = L.DecodeResultC()
        L.check(self._lib.wrs_decode_task_run_features(self._h, _ptr(feats), feats.shape[0], L.WRS_MEM_HOST, C.byref(res)))
        return self._collect(res)

    def run_audio(self, audio_: np.ndarray) -> List[DecodeOutput]:
        a = _f32(audio_)
        assert a.ndim == 2 and a.shape[1] == N_SAMPLES
        res = L.DecodeResultC()
        L.check(self._lib.wrs_decode_task_run_audio(self._h, _ptr(a), a.shape[0], L.WRS_MEM_HOST, C.byref(res)))
        return self._collect(res)

    def read_audio_features(self, audio_index: int, n_ctx: int, n_state: int) -> np.ndarray:
        """host copy of `DecodeOutput.audio_features` (decode/mod.rs:110) of one window of the last decoded batch"""
        out = np.empty((n_ctx, n_state), np.float32)
        L.check(self._lib.wrs_decode_task_read_audio_features(self._h, audio_index, _ptr(out), out.size))
        return out

    def encode_async(self, audio_, n_audio: Optional[int] = None):
        """queue mel + encoder + cross-K/V of a window batch on the side stream (overlapped mode).  audio_: numpy
        [n, 480000] (host) or a device pointer (int) with n_audio given; a device buffer must outlive the batch."""
        if isinstance(audio_, int):
            L.check(self._lib.wrs_decode_task_encode_async(self._h, C.c_void_p(audio_), n_audio, L.WRS_MEM_DEVICE))
            return
        a = _f32(audio_)
        assert a.ndim == 2 and a.shape[1] == N_SAMPLES
        L.check(self._lib.wrs_decode_task_encode_async(self._h, _ptr(a), a.shape[0], L.WRS_MEM_HOST))
        # host buffers are copied by an asynchronous H2D on the side stream: keep `a` alive until run_pending
        self._keep_async = getattr(self, "_keep_async", []) + [a]

    def run_pending(self) -> List[DecodeOutput]:
        res = L.DecodeResultC()
        L.check(self._lib.wrs_decode_task_run_pending(self._h, C.byref(res)))
        if getattr(self, "_keep_async", None):
            self._keep_async.pop(0)
        return self._collect(res)

    def run_encoded(self, n_audio: int) -> List[DecodeOutput]:
        """decode the windows whose encoder output is still resident (after detect_language / a previous run)."""
        res = L.DecodeResultC()
        L.check(self._lib.wrs_decode_task_run_encoded(self._h, n_audio, C.byref(res)))
        return self._collect(res)

    def detect_language(self, mel: np.ndarray, first_language_id: int, n_languages: int = 99):
        """OpenAI whisper `detect_language` (the reference only declares Task::LanguageId): returns
        (language token id per window [n], probabilities [n, n_languages])."""
        mel = _f32(mel)
        if mel.ndim == 2:
            mel = mel[None]
        n = mel.shape[0]
        ids = np.empty(n, np.int32)
        probs = np.empty((n, n_languages), np.float32)
        L.check(self._lib.wrs_decode_task_detect_language(self._h, _ptr(mel), n, L.WRS_MEM_HOST, first_language_id, n_languages,
                                                          _ptr(ids), _ptr(probs)))
        return ids, probs

    def set_languages(self, language_ids):
        a = np.ascontiguousarray(language_ids if language_ids is not None else [], dtype=np.int32)
        L.check(self._lib.wrs_decode_task_set_languages(self._h, _ptr(a) if a.size else None, a.size))

    def run_audio_device(self, dev_ptr: int, n_audio: int) -> List[DecodeOutput]:
        """audio windows already resident in HBM ([n_audio, 480000] f32)."""
        res = L.DecodeResultC()
        L.check(self._lib.wrs_decode_task_run_audio(self._h, C.c_void_p(dev_ptr), n_audio, L.WRS_MEM_DEVICE, C.byref(res)))
        return self._collect(res)

    def trace(self):
        """(tokens [steps, rows], sources [steps, rows], sum_logprobs [steps, rows]) of the last run."""
        ns, nr = C.c_int32(), C.c_int32()
        pt, ps, pl = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)()
        L.check(self._lib.wrs_decode_task_trace(self._h, C.byref(ns), C.byref(nr), C.byref(pt), C.byref(ps), C.byref(pl)))
        shape = (ns.value, nr.value)
        if ns.value == 0:
            return np.zeros(shape, np.int32), np.zeros(shape, np.int32), np.zeros(shape, np.float32)
        return (np.ctypeslib.as_array(pt, shape=shape).copy(), np.ctypeslib.as_array(ps, shape=shape).copy(),
                np.ctypeslib.as_array(pl, shape=shape).copy())

    def filter_topk(self, logits, tokens, sample_begin, k, want_filtered=False):
        """LogitFilter chain + extractor log_softmax/top-k on device (test hook, wrs_filter_topk)."""
        logits = _f32(logits)
        tokens = np.ascontiguousarray(tokens, dtype=np.int64)
        n, V = logits.shape
        lp = np.empty((n, k), np.float32)
        ids = np.empty((n, k), np.int32)
        filt = np.empty((n, V), np.float32) if want_filtered else None
        L.check(self._lib.wrs_filter_topk(self._h, _ptr(logits), _ptr(tokens), n, tokens.shape[1], sample_begin, k, _ptr(lp),
                                          _ptr(ids), _ptr(filt) if want_filtered else None))
        return (lp, ids, filt) if want_filtered else (lp, ids)

    def run_forced(self, mel, forced_tokens, forced_sources):
        """test hook (wrs_debug_decode_forced): follow the given per-step beam decisions [steps, rows] and return the
        device's raw per-step candidates (logprobs [steps, rows, k], ids [steps, rows, k], no_speech [n_audio])."""
        mel = _f32(mel)
        if mel.ndim == 2:
            mel = mel[None]
        ft = np.ascontiguousarray(forced_tokens, dtype=np.int32)
        fs = np.ascontiguousarray(forced_sources, dtype=np.int32)
        steps, rows = ft.shape
        m = self.options.token_extract_mode
        k = m.beam_size + 1 if m.kind == "beam_search" else 1
        lp = np.empty((steps, rows, k), np.float32)
        ids = np.empty((steps, rows, k), np.int32)
        ns = np.empty(mel.shape[0], np.float64)
        fn = self._lib.wrs_debug_decode_forced
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.check(fn(self._h, _ptr(mel), mel.shape[0], steps, _ptr(ft), _ptr(fs), _ptr(lp), _ptr(ids), _ptr(ns)))
        return lp, ids, ns

    def set_profiling(self, enabled: bool):
        L.check(self._lib.wrs_decode_task_set_profiling(self._h, int(enabled)))

    def last_run_times(self):
        a, b, c, d = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        L.check(self._lib.wrs_decode_task_last_run_times(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(encoder_ms=a.value, prefill_ms=b.value, steps_ms=c.value, host_ms=d.value)

    def set_cross_timing(self, enabled: bool, layer_stride: int = 8):
        L.check(self._lib.wrs_decode_task_set_cross_timing(self._h, int(enabled), layer_stride))

    def cross_timing(self):
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        L.check(self._lib.wrs_decode_task_get_cross_timing(self._h, C.byref(ms), C.byref(n), C.byref(b)))
        return dict(ms=ms.value, launches=n.value, bytes=b.value)

    def stage_times(self):
        n = C.c_int32()
        buf = (L.StageTimeC * 64)()
        L.check(self._lib.wrs_decode_task_stage_times(self._h, buf, 64, C.byref(n)))
        return [dict(name=buf[i].name.decode(), ms=buf[i].ms, launches=buf[i].launches, bytes=buf[i].bytes, flops=buf[i].flops)
                for i in range(min(n.value, 64))]

    def close(self):
        if self._h:
            self._lib.wrs_decode_task_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WindowPool:
    """Work queue over independent 30-s windows (include/wrs.h, window pool): one worker per model replica — replicas on
    different GPUs, or several on one GPU — batches of `options.max_audio` windows, no collective."""

    def __init__(self, models: Sequence[Whisper], tokenizer_ids: TokenIds, options: DecodeOptions, overlapped: bool = True):
        self._lib = L.load()
        self.models = list(models)
        self._keep = []
        c = _fill_decode_options(options, self._keep)
        ids = _ids_c(tokenizer_ids)
        arr = (C.c_void_p * len(self.models))(*[m._h for m in self.models])
        self._h = C.c_void_p()
        L.check(self._lib.wrs_window_pool_create(arr, len(self.models), C.byref(ids), C.byref(c), int(overlapped), C.byref(self._h)))

    def set_tokenizer(self, tokenizer: Optional["Tokenizer"]):
        self._tokenizer = tokenizer
        L.check(self._lib.wrs_window_pool_set_tokenizer(self._h, tokenizer._h if tokenizer else None))

    def run_audio(self, audio_, n_windows: Optional[int] = None) -> List[DecodeOutput]:
        """audio_: numpy [n, 480000] (host), or a device pointer (int) with n_windows when all replicas share that GPU."""
        res = L.DecodeResultC()
        if isinstance(audio_, int):
            L.check(self._lib.wrs_window_pool_run_audio(self._h, C.c_void_p(audio_), n_windows, L.WRS_MEM_DEVICE, C.byref(res)))
        else:
            a = _f32(audio_)
            assert a.ndim == 2 and a.shape[1] == N_SAMPLES
            L.check(self._lib.wrs_window_pool_run_audio(self._h, _ptr(a), a.shape[0], L.WRS_MEM_HOST, C.byref(res)))
        return self._collect(res)

    def run_mel(self, mel: np.ndarray) -> List[DecodeOutput]:
        """one long recording's log-mel spectrogram [80, n_frames] (whole-file normalisation, as the reference): fixed-stride
        3000-frame windows dealt to the pool's workers; outputs in window order."""
        mel = _f32(mel)
        assert mel.ndim == 2 and mel.shape[0] == N_MELS
        res = L.DecodeResultC()
        L.check(self._lib.wrs_window_pool_run_mel(self._h, _ptr(mel), mel.shape[1], C.byref(res)))
        return self._collect(res)

    def _collect(self, res):
        self.n_steps = res.n_steps
        outs = []
        for i in range(res.n_outputs):
            o = res.outputs[i]
            toks = np.ctypeslib.as_array(o.tokens, shape=(o.n_tokens,)).copy() if o.n_tokens > 0 else np.zeros(0, np.int64)
            outs.append(DecodeOutput(toks, o.avg_logprob, o.no_speech_prob, o.sum_logprob, o.audio_features_device or 0, _take_text(o.text)))
        self._lib.wrs_decode_result_free(C.byref(res))
        return outs

    def close(self):
        if self._h:
            self._lib.wrs_window_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------- transcribe
@dataclass
class TranscribePrompt:
    """reference enum `TranscribePrompt` (transcribe.rs:25-29); Text needs the BPE tokenizer."""
    pretokenized: Optional[List[int]] = None
    condition_on_prev_text: bool = True
    text: Optional[str] = None

    @staticmethod
    def Pretokenized(tokens):
        return TranscribePrompt(list(tokens), True)

    @staticmethod
    def Text(text):
        return TranscribePrompt(None, True, text)

    @staticmethod
    def None_(condition_on_prev_text=True):
        return TranscribePrompt(None, condition_on_prev_text)


@dataclass
class TranscribeOptions:
    """reference `TranscribeOptions` (transcribe.rs:32-50)."""
    decode: DecodeOptions = field(default_factory=DecodeOptions)
    prompt: TranscribePrompt = field(default_factory=TranscribePrompt)
    tokenizer_ids: TokenIds = field(default_factory=TokenIds.english)
    tokenizer: Optional[Tokenizer] = None  # reference `TranscribeOptions.tokenizer` (transcribe.rs:49); gives text outputs


@dataclass
class TranscribeOutputSegment:
    seek: int
    start_time: float
    end_time: float
    start_token: int
    end_token: int
    text: Optional[str] = None


@dataclass
class TranscribeOutput:
    tokens: np.ndarray
    segments: List[TranscribeOutputSegment]
    windows: list  # (seek, avg_logprob, no_speech_prob)
    text: Optional[str] = None
    truncated: bool = False  # the loop stopped before the end of the audio (zero seek advance / window limit)


class TranscribeTask:
    """reference `TranscribeTask` (transcribe.rs:13-244)."""

    def __init__(self, model: Whisper, options: TranscribeOptions):
        self._lib = L.load()
        self.model, self.options = model, options
        self._keep = []
        c = L.TranscribeOptionsC()
        c.decode = _fill_decode_options(options.decode, self._keep)
        p = options.prompt
        if p.pretokenized is not None:
            arr = np.ascontiguousarray(p.pretokenized, dtype=np.int64)
            self._keep.append(arr)
            c.prompt_tokens = arr.ctypes.data_as(C.POINTER(C.c_int64))
            c.n_prompt_tokens = arr.size
        c.condition_on_prev_text = p.condition_on_prev_text
        ids = _ids_c(options.tokenizer.ids if options.tokenizer is not None else options.tokenizer_ids)
        self._h = C.c_void_p()
        L.check(self._lib.wrs_transcribe_task_create(model._h, C.byref(ids), C.byref(c), C.byref(self._h)))
        if options.tokenizer is not None:
            L.check(self._lib.wrs_transcribe_task_set_tokenizer(self._h, options.tokenizer._h))
        if p.text is not None:
            L.check(self._lib.wrs_transcribe_task_set_prompt_text(self._h, p.text.encode("utf-8")))

    def _collect(self, out):
        toks = np.ctypeslib.as_array(out.tokens, shape=(out.n_tokens,)).copy() if out.n_tokens else np.zeros(0, np.int64)
        segs = [TranscribeOutputSegment(s.seek, s.start_time, s.end_time, s.start_token, s.end_token, _take_text(s.text))
                for s in (out.segments[i] for i in range(out.n_segments))]
        wins = [(out.window_seek[i], out.window_avg_logprob[i], out.window_no_speech_prob[i]) for i in range(out.n_windows)]
        text = _take_text(out.text)
        truncated = bool(out.truncated)
        self._lib.wrs_transcribe_output_free(C.byref(out))
        return TranscribeOutput(toks, segs, wins, text, truncated)

    def run(self, audio_: np.ndarray) -> TranscribeOutput:
        a = _f32(audio_)
        out = L.TranscribeOutputC()
        L.check(self._lib.wrs_transcribe_task_run(self._h, _ptr(a), a.shape[0], C.byref(out)))
        return self._collect(out)

    def run_mel(self, mel: np.ndarray) -> TranscribeOutput:
        """the long-form loop on a log-mel spectrogram [80, n_frames] (run_inner after transcribe.rs:112)."""
        mel = _f32(mel)
        assert mel.ndim == 2 and mel.shape[0] == N_MELS
        out = L.TranscribeOutputC()
        L.check(self._lib.wrs_transcribe_task_run_mel(self._h, _ptr(mel), mel.shape[1], C.byref(out)))
        return self._collect(out)

    def run_chunked(self, audio_: np.ndarray, batch: int) -> TranscribeOutput:
        a = _f32(audio_)
        out = L.TranscribeOutputC()
        L.check(self._lib.wrs_transcribe_task_run_chunked(self._h, _ptr(a), a.shape[0], batch, C.byref(out)))
        return self._collect(out)

    def close(self):
        if self._h:
            self._lib.wrs_transcribe_task_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

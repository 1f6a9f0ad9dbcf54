"""ORACLE (test infrastructure) — window decode.  See oracle/__init__.py.

Restates reference src/decode/mod.rs, decode/filter/*.rs, decode/extractor/*.rs,
decode/sequencer/max_likelihood.rs.  PARITY UNPINNED by the reference (it has no
tests for this code); deterministic where the reference is not:

  * beam candidates: the reference keeps them in a HashMap with a randomised
    hasher and stable-sorts by score, so the order of equal-score candidates,
    the admission order of newly finished sequences and the order of the final
    candidate list are random.  Here: candidates keep first-insertion order, a
    duplicate sequence overwrites (score, source) in place (beam_search.rs:77),
    the sort is stable by score descending, newly finished sequences are admitted
    in that sorted order, and final candidates are listed in admission order
    (finished first, then live beams by descending score).  This equals the
    reference whenever the reference's own result is not hasher-dependent.
"""
from dataclasses import dataclass, field
from typing import List, Optional
import math
import torch

from .audio import CHUNK_LENGTH, N_FRAMES

NEG_INF = float("-inf")


@dataclass
class TokenIds:
    eot: int = 50256
    sot: int = 50257
    translate: int = 50357
    transcribe: int = 50358
    startoflm: int = 50359
    startofprev: int = 50360
    no_speech: int = 50361
    no_timestamps: int = 50362
    ts_begin: int = 50363
    space: int = 220


class DecodePrompt:
    """decode/mod.rs:46-92."""

    def __init__(self, prompt: Optional[torch.Tensor], ids: TokenIds, sot_tail=()):
        # tokenize.rs:209-213 sequence_sot() == [sot]; sot_tail = OpenAI's multilingual [lang, task] (not in the reference)
        seq_sot = [ids.sot] + list(sot_tail)
        if prompt is not None:
            initial = torch.cat([torch.tensor([ids.startofprev], dtype=torch.int64), prompt.to(torch.int64),
                                 torch.tensor(seq_sot, dtype=torch.int64)])
        else:
            initial = torch.tensor(seq_sot, dtype=torch.int64)
        self.initial_tokens = initial
        self.sample_begin = initial.shape[0]
        self.sot_idx = int((initial == ids.sot).nonzero()[0, 0])


# ---------------------------------------------------------------- logit filters
class SuppressBlank:
    """filter/suppress_blanks.rs:15-43."""

    def __init__(self, ids, prompt_ref):
        self.idx = torch.tensor([ids.space, ids.eot], dtype=torch.int64)
        self.prompt_ref = prompt_ref

    def apply(self, logits, tokens):
        if tokens.shape[1] == self.prompt_ref[0].sample_begin:
            logits.index_fill_(1, self.idx, NEG_INF)


class SuppressTokens:
    """filter/suppress_tokens.rs:10-23."""

    def __init__(self, token_ids):
        self.idx = torch.tensor(list(token_ids), dtype=torch.int64)

    def apply(self, logits, tokens):
        logits.index_fill_(1, self.idx, NEG_INF)


class TimestampTokens:
    """filter/timestamp_tokens.rs:33-97 (OpenAI's ApplyTimestampRules, Dec-2022)."""

    def __init__(self, ids, prompt_ref, max_initial_timestamp_index):
        self.ids, self.prompt_ref, self.mit = ids, prompt_ref, max_initial_timestamp_index

    def apply(self, logits, tokens):
        sample_begin = self.prompt_ref[0].sample_begin
        logits[:, self.ids.no_timestamps] = NEG_INF  # :37-41
        tsb = self.ids.ts_begin
        for k in range(tokens.shape[0]):  # :47-66
            seq = tokens[k, sample_begin:]
            last_was_ts = seq.shape[0] >= 1 and int(seq[-1]) >= tsb
            second_last_was_ts = seq.shape[0] < 2 or int(seq[-2]) >= tsb
            if last_was_ts:
                if second_last_was_ts:
                    logits[k, tsb:] = NEG_INF
                else:
                    logits[k, :self.ids.eot] = NEG_INF
        if tokens.shape[1] == sample_begin:  # :68-78
            logits[:, :tsb] = NEG_INF
            if self.mit is not None:
                last_allowed = tsb + self.mit
                logits[:, last_allowed + 1:] = NEG_INF
        logprobs = logits.log_softmax(dim=-1, dtype=torch.float32)  # :82
        for k in range(tokens.shape[0]):  # :83-96
            ts_lp = float(logprobs[k, tsb:].logsumexp(dim=-1))
            max_text_lp = float(logprobs[k, :tsb].max())
            if ts_lp > max_text_lp:
                logits[k, :tsb] = NEG_INF


# ---------------------------------------------------------------- extractors
class BeamSearchTokenExtractor:
    """extractor/beam_search.rs:10-211."""

    def __init__(self, beam_size, eot, patience=1.0):
        self.beam_size, self.eot, self.patience = beam_size, eot, patience
        self.finished_sequences = None
        self.trace = []  # per step: (tokens [B], source [B], scores [B]) — test aid, not in the reference

    def reset(self):
        self.finished_sequences = None
        self.trace = []

    def group_size(self):
        return self.beam_size

    def update(self, model, tokens, logits, sum_logprobs):
        B, cur = tokens.shape
        n_audio = B // self.beam_size
        if self.finished_sequences is None:
            self.finished_sequences = [dict() for _ in range(n_audio)]  # insertion-ordered
        logprobs = logits.log_softmax(dim=-1, dtype=torch.float32)  # :37
        lp, tk = logprobs.topk(self.beam_size + 1, dim=-1, largest=True, sorted=True)  # :62-65 (per row)
        return self.update_from_candidates(model, tokens, lp, tk, sum_logprobs)

    def update_from_candidates(self, model, tokens, lp, tk, sum_logprobs):
        """the rest of `update` given each row's sorted top-(beam+1) (log-prob, token) lists (test aid: lets the host
        sequencer of the build be checked against this one on arbitrary candidate lists, without a model)."""
        B = tokens.shape[0]
        n_audio = B // self.beam_size
        if self.finished_sequences is None:
            self.finished_sequences = [dict() for _ in range(n_audio)]
        next_tokens, source_indices, finished_all = beam_select(tokens.tolist(), sum_logprobs, lp, tk, self.beam_size, self.eot)
        new_tokens = torch.tensor(next_tokens, dtype=torch.int64)  # :109
        src = torch.tensor(source_indices, dtype=torch.int64)
        if model is not None:
            model.update_cache(src)  # :114
        max_candidates = int(_round_half_away(self.patience * self.beam_size))  # :122 (f32::round)
        for prev, new in zip(self.finished_sequences, finished_all):  # :124-138
            for seq, score in new.items():
                if len(prev) >= max_candidates:
                    break
                prev[seq] = score
        completed = all(len(s) >= max_candidates for s in self.finished_sequences)  # :141-146
        self.trace.append((new_tokens[:, -1].clone(), src.clone(), sum_logprobs.clone()))
        return new_tokens, completed

    def finalize(self, tokens, sum_logprobs):
        """:151-206.  tokens [n_audio, n_group, len]; returns (list of list of 1-D tensors, list of list of f64)."""
        for i, seqs in enumerate(self.finished_sequences):
            if len(seqs) < self.beam_size:  # :165-178
                order = torch.argsort(sum_logprobs[i], dim=-1, descending=False, stable=True).tolist()
                for j in reversed(order):
                    seq = tuple(tokens[i, j].tolist()) + (self.eot,)
                    seqs[seq] = float(sum_logprobs[i, j])
                    if len(seqs) >= self.beam_size:
                        break
        toks = [[torch.tensor(s, dtype=torch.int64) for s in seqs.keys()] for seqs in self.finished_sequences]
        lps = [list(seqs.values()) for seqs in self.finished_sequences]
        return toks, lps


def beam_select(prefixes, sum_logprobs, cand_lp, cand_tk, beam_size, eot):
    """STEP 1 + STEP 2 of BeamSearchTokenExtractor::update (beam_search.rs:49-107) given each row's sorted
    top-(beam+1) candidates.  Mutates sum_logprobs in place like the reference (:91-95).
    Returns (next_tokens, source_indices, finished_per_audio)."""
    n_audio = len(prefixes) // beam_size
    next_tokens, source_indices, finished_all = [], [], []
    for i in range(n_audio):
        cand = {}  # sequence tuple -> [score(f64), source]; first-insertion order, overwrite in place (:77)
        for j in range(beam_size):
            idx = i * beam_size + j
            prefix = tuple(prefixes[idx])
            base = float(sum_logprobs[idx])  # double_value of the f32 tensor (:71)
            for c in range(cand_lp.shape[1]):
                if int(cand_tk[idx, c]) < 0:
                    continue
                seq = prefix + (int(cand_tk[idx, c]),)
                cand[seq] = [base + float(cand_lp[idx, c]), idx]
        order = sorted(cand.items(), key=lambda kv: -kv[1][0])  # stable, score descending (:86)
        finished = {}
        saved = 0
        for seq, (score, source) in order:  # :87-104
            if seq[-1] == eot:
                finished[seq] = score
            else:
                sum_logprobs[len(next_tokens)] = score  # stored as f32 (:91-95)
                next_tokens.append(list(seq))
                source_indices.append(source)
                saved += 1
                if saved == beam_size:
                    break
        finished_all.append(finished)
    return next_tokens, source_indices, finished_all


class GreedyTokenExtractor:
    """extractor/greedy.rs:8-66.  Unreachable in the reference (DecodeTask::new hits todo!() for
    Greedy, finalize is todo!()).  Semantics here = `update` as written (:17-51) with
    temperature 0, `finalize` = the commented-out body (:54-58: pad one EOT).  `completed` follows
    OpenAI (all rows' LAST token is EOT); the literal `tokens.eq(eot).all()` (:46) can never be true
    because column 0 is <|startoftranscript|>, and since rows that reached EOT keep emitting EOT with
    no logprob added (:38-43), stopping early does not change any output."""

    def __init__(self, eot, group_size=1):
        self.eot, self._group = eot, group_size
        self.trace = []

    def reset(self):
        self.trace = []

    def group_size(self):
        return self._group

    def update(self, model, tokens, logits, sum_logprobs):
        next_tokens = logits.argmax(dim=-1)  # :25
        logprobs = logits.log_softmax(dim=-1, dtype=torch.float32)  # :31
        cur = logprobs[torch.arange(logprobs.shape[0]), next_tokens]
        last_eot = tokens[:, -1] == self.eot
        sum_logprobs += cur * (~last_eot)  # :38
        next_tokens = torch.where(last_eot, torch.full_like(next_tokens, self.eot), next_tokens)  # :40-44
        tokens = torch.cat([tokens, next_tokens[:, None]], dim=-1)
        completed = bool((tokens[:, -1] == self.eot).all())
        self.trace.append((next_tokens.clone(), torch.arange(tokens.shape[0]), sum_logprobs.clone()))
        return tokens, completed

    def finalize(self, tokens, sum_logprobs):
        toks = torch.nn.functional.pad(tokens, (0, 1), value=self.eot)
        return ([[toks[i, j] for j in range(toks.shape[1])] for i in range(toks.shape[0])],
                [[float(x) for x in sum_logprobs[i]] for i in range(sum_logprobs.shape[0])])


def _round_half_away(x):
    """f32::round / f64::round (half away from zero), unlike Python's banker's rounding."""
    return math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1)


class MaximumLikelihoodRanker:
    """sequencer/max_likelihood.rs:13-34."""

    def __init__(self, length_penalty=None):
        self.length_penalty = length_penalty

    def rank(self, tokens, sum_logprobs):
        out = []
        for seqs, lps in zip(tokens, sum_logprobs):
            scores = []
            for t, lp in zip(seqs, lps):
                length = t.shape[0]
                penalty = ((5.0 + length) / 6.0) ** self.length_penalty if self.length_penalty is not None else float(length)
                scores.append(lp / penalty if penalty != 0 else (math.copysign(math.inf, lp) if lp != 0 else math.nan))
            out.append(int(torch.tensor(scores, dtype=torch.float64).argmax()))
        return out


# ---------------------------------------------------------------- the task
@dataclass
class DecodeOptions:
    """decode/mod.rs:95-106 (task omitted: only Transcribe exists, tokenize.rs:143-147)."""
    sample_len: Optional[int] = None
    beam_size: Optional[int] = 5      # TokenExtractMode::BeamSearch{beam_size, patience}; None => Greedy
    patience: float = 1.0
    greedy_group_size: int = 1
    len_penalty: Optional[float] = None
    max_initial_timestamp: Optional[float] = 1.0
    timestamps: bool = True
    suppress_blank: bool = True
    suppress_non_speech: bool = True
    suppress_tokens: Optional[List[int]] = None
    prompt: Optional[torch.Tensor] = None
    non_speech_tokens: tuple = ()
    position_offset_quirk: bool = True  # decoder offset = loop index i (decode/mod.rs:244)
    sot_sequence_tail: tuple = ()       # beyond the reference: OpenAI's [lang, task] after sot (BASELINE config 4)


@dataclass
class DecodeOutput:
    """decode/mod.rs:109-119 (text omitted: BPE text decode is a 'next' row)."""
    audio_features: torch.Tensor
    tokens: torch.Tensor
    avg_logprob: float
    no_speech_prob: float
    sum_logprob: float = 0.0  # test aid


class DecodeTask:
    """decode/mod.rs:24-419."""

    def __init__(self, model, ids: TokenIds, options: DecodeOptions):
        self.model, self.ids, self.opt = model, ids, options
        self.sample_len = options.sample_len if options.sample_len is not None else model.dims.n_text_ctxs // 2  # :129-131
        if options.beam_size is None:
            self.extractor = GreedyTokenExtractor(ids.eot, options.greedy_group_size)
        else:
            self.extractor = BeamSearchTokenExtractor(options.beam_size, ids.eot, options.patience)  # :135-143
        self.ranker = MaximumLikelihoodRanker(options.len_penalty)
        self.prompt_ref = [DecodePrompt(options.prompt, ids, options.sot_sequence_tail)]
        self.languages = None  # per-audio language token (OpenAI decoding.py: tokens[:, sot_index + 1] = lang_tokens)
        self.filters = []
        if options.suppress_blank:  # :158-164
            self.filters.append(SuppressBlank(ids, self.prompt_ref))
        if options.suppress_tokens is not None or options.suppress_non_speech:  # :166-174
            st = list(options.suppress_tokens or [])
            if options.suppress_non_speech:
                st.extend(options.non_speech_tokens)
            self.filters.append(SuppressTokens(st))
        if options.timestamps:  # :176-190
            precision = CHUNK_LENGTH / model.dims.n_audio_ctxs
            mit = None if options.max_initial_timestamp is None else int(_round_half_away(options.max_initial_timestamp / precision))
            self.filters.append(TimestampTokens(ids, self.prompt_ref, mit))
        self.logits_trace = None  # set to [] to record post-filter logits per step (test aid)

    def set_prompt(self, prompt):  # :206-218
        if prompt is not None and prompt.shape[0] == 0:
            prompt = None
        self.prompt_ref[0] = DecodePrompt(prompt, self.ids, self.opt.sot_sequence_tail)

    def detect_language(self, mel, first_language_id, n_languages=99):
        """OpenAI whisper decoding.py detect_language (PARITY UNPINNED by the reference, which only declares
        Task::LanguageId, tokenize.rs:135-139): logits of [sot] at position 0, every non-language token masked,
        argmax + softmax."""
        with torch.no_grad():
            feats = self.model.encoder.forward(mel)
            self.model.clear_cache()
            x = torch.full((feats.shape[0], 1), self.ids.sot, dtype=torch.int64)
            logits = self.model.decoder.forward_ext(x, feats, 0)[:, 0]
            self.model.clear_cache()
            lang = logits[:, first_language_id:first_language_id + n_languages].float()
            return lang.argmax(dim=-1) + first_language_id, lang.softmax(dim=-1)

    def main_loop(self, audio_features, tokens):  # :220-282
        n_batch = tokens.shape[0]
        sum_logprobs = torch.zeros(n_batch, dtype=audio_features.dtype)
        no_speech_probs = [math.nan] * n_batch
        prompt = self.prompt_ref[0]
        for i in range(self.sample_len):
            toks = tokens
            if toks.shape[-1] > prompt.sample_begin:
                toks = toks[:, -1:]
            offset = i if self.opt.position_offset_quirk else (0 if i == 0 else tokens.shape[-1] - 1)
            logits = self.model.decoder.forward_ext(toks, audio_features, offset)  # :244
            if i == 0:  # :250-255
                probs = logits[:, prompt.sot_idx].softmax(dim=-1)
                no_speech_probs = [float(x) for x in probs[:, self.ids.no_speech]]
            logits = logits[:, -1].clone()
            for f in self.filters:
                f.apply(logits, tokens)
            if self.logits_trace is not None:
                self.logits_trace.append(logits.clone())
            tokens, completed = self.extractor.update(self.model, tokens, logits, sum_logprobs)
            if completed or tokens.shape[-1] > self.model.dims.n_text_ctxs:
                break
        self.model.clear_cache()
        return tokens, sum_logprobs, no_speech_probs

    def run(self, mel):  # :284-289; mel [80, 3000]
        with torch.no_grad():
            return self.run_inner(mel.unsqueeze(0))

    def run_batch(self, mel):
        """Not in the reference's public API (run() forces n_audio = 1, :288) but its internals are
        batch-general (:292-347); this entry exposes them for batched-window parity."""
        with torch.no_grad():
            return self.run_inner(mel)

    def run_inner(self, mel):  # :291-418
        n_audio, _, n_frames = mel.shape
        assert n_frames == N_FRAMES
        audio_features = self.model.encoder.forward(mel)
        return self.run_from_features(audio_features)

    def run_from_features(self, audio_features):
        n_audio = audio_features.shape[0]
        prompt = self.prompt_ref[0]
        repeated = prompt.initial_tokens.repeat(n_audio, 1)
        if self.languages is not None:
            repeated[:, prompt.sot_idx + 1] = torch.as_tensor(self.languages, dtype=torch.int64)[:n_audio]
        n_group = self.extractor.group_size()
        feats = audio_features.repeat_interleave(n_group, dim=0)  # :312
        repeated = repeated.repeat_interleave(n_group, dim=0)
        self.model.clear_cache()  # :324
        self.extractor.reset()
        tokens, sum_logprobs, no_speech = self.main_loop(feats, repeated)
        no_speech = no_speech[::n_group]
        tokens = tokens.reshape(n_audio, n_group, -1)
        sum_logprobs = sum_logprobs.reshape(n_audio, n_group)
        cands, lps = self.extractor.finalize(tokens, sum_logprobs)  # :355
        sliced = []
        for seqs in cands:  # :356-370
            row = []
            for t in seqs:
                end = int((t == self.ids.eot).nonzero()[0, 0])
                row.append(t[prompt.sample_begin:end])
            sliced.append(row)
        selected = self.ranker.rank(sliced, lps)  # :376
        results = []
        for i in range(n_audio):
            t = sliced[i][selected[i]]
            lp = float(lps[i][selected[i]])
            results.append(DecodeOutput(audio_features=audio_features[i], tokens=t,
                                        avg_logprob=lp / (t.shape[0] + 1.0),  # :394-397
                                        no_speech_prob=float(no_speech[i]), sum_logprob=lp))
        return results

"""ORACLE (test infrastructure) — long-form transcribe loop.  See oracle/__init__.py.

Restates reference src/transcribe.rs:110-243.  PARITY UNPINNED by the reference
(no tests).  Text fields are omitted (BPE decode is a 'next' row).
"""
from dataclasses import dataclass
from typing import List, Optional
import torch

from . import audio as A
from .decode import DecodeTask, DecodeOptions


@dataclass
class Segment:
    """transcribe.rs:59-67 (text omitted)."""
    seek: int
    start_time: float
    end_time: float
    start_token: int
    end_token: int


@dataclass
class TranscribeOutput:
    tokens: torch.Tensor
    segments: List[Segment]
    windows: list  # per window: (seek, segment token tensor, avg_logprob, no_speech_prob) — test aid


class TranscribeTask:
    def __init__(self, model, ids, options: DecodeOptions, prompt_tokens: Optional[List[int]] = None,
                 condition_on_prev_text: bool = True):
        """prompt_tokens = TranscribePrompt::Pretokenized; None = TranscribePrompt::None{condition_on_prev_text}
        (transcribe.rs:25-29,142-147: any explicit prompt implies conditioning)."""
        self.model, self.ids = model, ids
        opts = DecodeOptions(**{**options.__dict__, "prompt": None})  # :88-89
        self.decode_task = DecodeTask(model, ids, opts)
        self.prompt_tokens = prompt_tokens
        self.condition_on_prev_text = True if prompt_tokens is not None else condition_on_prev_text

    def run(self, audio: torch.Tensor) -> TranscribeOutput:
        with torch.no_grad():
            return self.run_inner(audio)

    def run_inner(self, audio):
        filt = A.mel_filter()  # :111
        mel = A.log_mel_spectrogram(audio, filt)  # :112 (whole file, global max)
        return self.run_mel(mel)

    def run_mel(self, mel):
        """run_inner from :113 on, given the spectrogram (test aid: lets the loop run on the reference's own
        spectrogram fixture, whose WAV the reference does not ship)."""
        n_frames = mel.shape[1]
        QUANTUM = A.HOP_LENGTH / A.SAMPLE_RATE  # :117
        input_stride = A.N_FRAMES // self.model.dims.n_audio_ctxs  # :120
        time_precision = input_stride * QUANTUM  # :122
        seek = 0
        tokens = torch.tensor(self.prompt_tokens or [], dtype=torch.int64)  # :126-140
        tsb = self.ids.ts_begin
        segments, windows = [], []
        while seek < n_frames:  # :151
            seg_mel = A.pad_or_trim(mel[:, seek:], A.N_FRAMES)  # :152
            segment_duration = float(A.CHUNK_LENGTH)
            if self.condition_on_prev_text:
                self.decode_task.set_prompt(tokens)  # :155-158
            result = self.decode_task.run(seg_mel)[0]  # :161-162
            seg = result.tokens
            windows.append((seek, seg.clone(), result.avg_logprob, result.no_speech_prob))
            ts = seg >= tsb  # :167
            consecutive = (torch.where(ts[:-1] & ts[1:])[0] + 1).tolist()  # :168-176
            ts_offset = seek * QUANTUM
            if len(consecutive) > 0:  # :181-210
                last_slice = 0
                for cur in consecutive:
                    sl = seg[last_slice:cur]
                    start_pos = int(sl[0]) - tsb
                    end_pos = int(sl[-1]) - tsb
                    off = tokens.shape[0]
                    segments.append(Segment(seek, ts_offset + start_pos * time_precision,
                                            ts_offset + end_pos * time_precision, off + last_slice + 1, off + cur))
                    last_slice = cur
                last_ts_pos = int(seg[last_slice - 1]) - tsb
                seek += last_ts_pos * input_stride  # :206
                tokens = torch.cat([tokens, seg[:last_slice + 1]])  # :207-210
            else:  # :211-233
                stamps = seg[ts.nonzero().flatten()]
                # the reference evaluates timestamps.i(-1) before the emptiness check and panics on an
                # empty tensor; a guard is the only sane restatement.
                if stamps.shape[0] > 0 and int(stamps[-1]) != tsb:
                    segment_duration = (int(stamps[-1]) - tsb) * time_precision
                segments.append(Segment(seek, ts_offset, ts_offset + segment_duration, 0, seg.shape[0]))
                seek += A.N_FRAMES  # :231
                tokens = torch.cat([tokens, seg])
            if seek_guard(seek, windows):
                break
        return TranscribeOutput(tokens, segments, windows)


def seek_guard(seek, windows, limit=10000):
    """The reference loops forever if a window's last consecutive timestamp is <|0.00|> (seek += 0).
    Stop when a window made no progress so a synthetic model cannot hang the test-suite
    (the HIP host loop has the same guard)."""
    return len(windows) >= limit or windows[-1][0] == seek

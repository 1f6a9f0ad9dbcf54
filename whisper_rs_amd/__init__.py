"""whisper-rs_amd: MI355X-native (gfx950) Whisper transcription hot path behind a C-ABI
(include/wrs.h -> libwrs_mi355x.so), with the host-side mirror of laptou/whisper-rs's API.

The product path has no CPU fallback: every compute entry point goes through the HIP library and
fails loudly (ImportError / WrsError) if it is missing or no gfx950 device is present.
"""
from .consts import (SAMPLE_RATE, N_FFT, N_MELS, HOP_LENGTH, CHUNK_LENGTH, N_SAMPLES, N_FRAMES, ModelDims, MODEL_DIMS,
                     TokenIds, NON_SPEECH_TOKENS)
from .api import (audio, Whisper, Tokenizer, TokenExtractMode, DecodeOptions, DecodeOutput, DecodeTask, WindowPool, TranscribePrompt,
                  TranscribeOptions, TranscribeOutput, TranscribeOutputSegment, TranscribeTask, device_count)
from ._lib import WrsError, load as load_library

__all__ = [n for n in dir() if not n.startswith("_")]

"""Constants of the Whisper hot path, with the reference site each one restates.

Audio constants: reference src/audio.rs:12-18.
Special token ids: reference src/tokenize.rs:113-131,179-192 evaluated on
tokenizer/gpt2.json (SURVEY.md §8(a) rows 20 and 28 list the derived values).
Standard model sizes: SURVEY.md §8 dims table (the reference is generic over
ModelDims, src/model.rs:546-557; examples/transcribe.rs:25-35 hard-codes base.en).
"""
from dataclasses import dataclass, asdict

SAMPLE_RATE = 16000
N_FFT = 400
N_MELS = 80
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH  # 3000
N_FREQS = N_FFT // 2 + 1  # 201


@dataclass(frozen=True)
class ModelDims:
    """Mirror of reference `ModelDims` (src/model.rs:546-557); field names identical."""
    n_vocab: int
    n_audio_ctxs: int
    n_audio_states: int
    n_audio_heads: int
    n_audio_layers: int
    n_text_ctxs: int
    n_text_states: int
    n_text_heads: int
    n_text_layers: int

    def as_dict(self):
        return asdict(self)


def _dims(v, d, h, l):
    return ModelDims(v, 1500, d, h, l, 448, d, h, l)


MODEL_DIMS = {
    "tiny.en": _dims(51864, 384, 6, 4),
    "base.en": _dims(51864, 512, 8, 6),
    "small": _dims(51865, 768, 12, 12),
    "medium": _dims(51865, 1024, 16, 24),
    "large-v2": _dims(51865, 1280, 20, 32),
    # test-only shapes (same architecture, fewer/smaller layers so the CPU oracle is quick)
    "micro": _dims(51864, 128, 2, 2),
    "micro.ml": _dims(51865, 128, 2, 2),  # multilingual vocabulary layout
    "nano": _dims(51864, 64, 1, 1),
    # large-v2's width, heads and vocabulary with 2 layers: every kernel runs at the benchmark's shapes (D 1280, 20 heads,
    # 4D 5120, V 51865) while the CPU oracle still finishes in seconds
    "wide": _dims(51865, 1280, 20, 2),
}


@dataclass(frozen=True)
class TokenIds:
    """Special ids of the reference `Tokenizer` (src/tokenize.rs:113-131).

    `english()` are the values the reference computes from tokenizer/gpt2.json
    (eot 50256 is in the file; 106 added specials follow; timestamps start at
    last_special+1).  `multilingual()` is OpenAI's layout (all +1), which the
    reference does not implement — provided for configs 4/5.
    """
    eot: int
    sot: int
    translate: int
    transcribe: int
    startoflm: int
    startofprev: int
    no_speech: int
    no_timestamps: int
    ts_begin: int
    space: int = 220  # tokenizer.encode(" ")[0], reference suppress_blanks.rs:22-28

    @staticmethod
    def english():
        return TokenIds(eot=50256, sot=50257, translate=50357, transcribe=50358,
                        startoflm=50359, startofprev=50360, no_speech=50361,
                        no_timestamps=50362, ts_begin=50363)

    @staticmethod
    def multilingual():
        return TokenIds(eot=50257, sot=50258, translate=50358, transcribe=50359,
                        startoflm=50360, startofprev=50361, no_speech=50362,
                        no_timestamps=50363, ts_begin=50364)

    @staticmethod
    def for_vocab(n_vocab):
        return TokenIds.english() if n_vocab == 51864 else TokenIds.multilingual()


# reference Tokenizer::non_speech_tokens() (src/tokenize.rs:215-307) evaluated on
# tokenizer/gpt2.json: 96 entries, 84 unique, sorted.
NON_SPEECH_TOKENS = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93,
    357, 366, 438, 532, 685, 705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377,
    1391, 1635, 1782, 1875, 2162, 2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855,
    6329, 7203, 9609, 9959, 10563, 10786, 11420, 11709, 11907, 13163, 13697, 13700,
    14808, 15306, 16410, 16791, 17992, 19203, 19510, 20724, 22305, 22935, 27007, 30109,
    30420, 33409, 34949, 40283, 40493, 40549, 47282, 49146,
)

"""CPU: the C++ tokenizer (csrc/tokenizer.cpp behind wrs_tokenizer_*) against the `tokenizers` package — the Python
binding of the same Rust crate the reference wraps (reference src/tokenize.rs:142-312, Cargo.lock:1296) — on

  * the reference's own tokenizer/gpt2.json (skipped where /root/reference is absent), with the reference's special
    tokens added the way `Tokenizer::new_from_bytes` adds them, and
  * a small byte-level BPE trained inside the test (needs nothing from the reference).

Bar: ids and decoded strings identical."""
import json
import os
import random

import pytest

import whisper_rs_amd as W
from whisper_rs_amd.consts import NON_SPEECH_TOKENS, TokenIds

tokenizers = pytest.importorskip("tokenizers")
REF_JSON = "/root/reference/tokenizer/gpt2.json"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LANGS = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi iw uk el ms cs ro da hu ta no th ur hr bg lt la mi ml cy sk te fa "
         "lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn "
         "mt sa lb my bo tl mg as tt haw ln ha ba jw su").split()
SPECIALS = (["<|startoftranscript|>"] + [f"<|{l}|>" for l in LANGS] +
            ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"])

TEXTS = [
    "", " ", "  ", "   a", "a   ", "Hello, world!", " Hello world", "It's we're they'll I'd you've I'm don't 'tis 'Twas",
    "12345 67.89 1,000,000 ٣٤٥ ⅓ ²", "tabs\tand\nnewlines\r\n\r\n  mixed \t \n x", "trailing spaces   ", "\n\n\n", " \n \n",
    "naïve café façade Ærøskøbing Ελληνικά русский язык "
    "中文字符 日本語のテキスト 한국어",
    "emoji \U0001F600\U0001F603 \U0001F44D\U0001F3FD flags \U0001F1FA\U0001F1F8 zero‍width",
    "♪♪♪ ( SPEAKING FOREIGN LANGUAGE ) [DAVID] Hey there, -- --- ((( )))",
    "「引用」『二重』 ♩ ♪ ♫ ♬ ♭ ♮ ♯",
    "<|endoftext|>", "a<|endoftext|>b", "<|startoftranscript|><|en|><|transcribe|><|notimestamps|> text<|endoftext|>",
    "<|notaspecial|> <| en |> <|en|", "under_score camelCase snake_case kebab-case 'quoted' \"double\"",
    "x" * 300, " ".join(["word"] * 200),
    " nbsp emspace　ideographic linesepnel ogham narrow math",
    "combining é å ́start", "math ∑∫√≠≤≥ ±×÷ € £ ¥ © ® ™",
    "\x00\x01\x1f control \x7f \x1c\x1d\x1e",
]


def _hf(json_bytes):
    tk = tokenizers.Tokenizer.from_str(json_bytes.decode("utf-8"))
    tk.add_special_tokens([tokenizers.AddedToken(s, special=True) for s in SPECIALS])  # tokenize.rs:154-168
    return tk


POOLS = [
    "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", "0123456789", "      \t\n", ".,;:!?'\"()[]{}<>|-_/\\@#$%^&*+=~`",
    "äöüßéèêñçøåæœ", "αβγδεζηθ",
    "абвгдежз", "中文日本語한국어", "♪♫♯☺★",
    "  　​ ", "\U0001F600\U0001F44D\U0001F3FD\U0001F3B5", "'s't're've'm'll'd",
    "١٢²½Ⅰ〇", "́̈⃣️",
]


def _random_text(rng, n):
    return "".join(rng.choice(rng.choice(POOLS)) for _ in range(n))


def _check_pair(cpp, hf, seed, n_random=300):
    for t in TEXTS:
        assert cpp.encode(t) == hf.encode(t, add_special_tokens=True).ids, repr(t)
    rng = random.Random(seed)
    for i in range(n_random):
        t = _random_text(rng, rng.randint(1, 60))
        ids = hf.encode(t, add_special_tokens=True).ids
        assert cpp.encode(t) == ids, repr(t)
        assert cpp.decode(ids) == hf.decode(ids, skip_special_tokens=False), repr(t)
    # decode of arbitrary id sequences: split UTF-8 sequences (-> U+FFFD), specials, ids past the vocabulary (dropped)
    V = hf.get_vocab_size(with_added_tokens=True)
    for i in range(n_random):
        ids = [rng.randrange(0, V + 200) for _ in range(rng.randint(0, 40))]
        assert cpp.decode(ids) == hf.decode(ids, skip_special_tokens=False), ids


@pytest.mark.skipif(not os.path.exists(REF_JSON), reason="reference tokenizer file not present on this machine")
def test_gpt2_json_matches_tokenizers_crate():
    data = open(REF_JSON, "rb").read()
    cpp, hf = W.Tokenizer.new_from_bytes(data), _hf(data)
    # special ids the reference computes (tokenize.rs:170-197)
    assert cpp.ids == TokenIds.english()
    for s in SPECIALS + ["<|endoftext|>"]:
        assert cpp.token_to_id(s) == hf.token_to_id(s), s
    assert cpp.ids.ts_begin == hf.token_to_id("<|notimestamps|>") + 1 == cpp.vocab_size()
    assert cpp.language_token("en") == 50258 and cpp.language_token("su") == 50356 and cpp.language_token("xx") is None
    assert cpp.sequence_sot() == [50257]
    _check_pair(cpp, hf, seed=0, n_random=1500)
    # non_speech_tokens(): same procedure as tokenize.rs:215-307 through the C++ encoder == frozen list in consts.py
    ns = cpp.non_speech_tokens()
    assert ns == sorted(ns) and len(ns) == 96 and sorted(set(ns)) == sorted(set(NON_SPEECH_TOKENS))
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer_gpt2_cases.json")))
    for c in cases["encode"]:
        assert cpp.encode(c["text"]) == c["ids"], c["text"]
    for c in cases["decode"]:
        assert cpp.decode(c["ids"]) == c["text"], c["ids"]
    # file loader == bytes loader
    f = W.Tokenizer.new_from_file(REF_JSON)
    assert f.encode("same thing") == cpp.encode("same thing")


def _train_small_bpe():
    from tokenizers import models, pre_tokenizers, decoders, trainers
    tk = tokenizers.Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    rng = random.Random(7)
    corpus = [_random_text(rng, 80) for _ in range(400)] + TEXTS
    tr = trainers.BpeTrainer(vocab_size=600, special_tokens=["<|endoftext|>"], initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                             show_progress=False)
    tk.train_from_iterator(corpus, tr)
    return tk.to_str().encode("utf-8")


def test_trained_bpe_matches_tokenizers_crate():
    data = _train_small_bpe()
    cpp, hf = W.Tokenizer.new_from_bytes(data), _hf(data)
    assert cpp.ids.sot == hf.token_to_id("<|startoftranscript|>")
    assert cpp.ids.eot == hf.token_to_id("<|endoftext|>")
    assert cpp.ids.ts_begin == hf.get_vocab_size(with_added_tokens=True)
    _check_pair(cpp, hf, seed=1, n_random=200)


def test_tokenizer_errors():
    with pytest.raises(W.WrsError):
        W.Tokenizer.new_from_bytes(b"{not json")
    with pytest.raises(W.WrsError):
        W.Tokenizer.new_from_bytes(b'{"model": {"type": "WordPiece", "vocab": {}, "merges": []}}')
    with pytest.raises(W.WrsError):
        W.Tokenizer.new_from_file("/nonexistent/tokenizer.json")


def test_synthetic_full_size_tokenizer_layout():
    """the corpus-free 50k-entry tokenizer the GPU text tests use has the reference's id layout and round-trips."""
    from helpers import synthetic_tokenizer_json
    data = synthetic_tokenizer_json()
    cpp, hf = W.Tokenizer.new_from_bytes(data), _hf(data)
    assert cpp.ids.eot == 50256 and cpp.ids.sot == 50257 and cpp.ids.ts_begin == 50363
    _check_pair(cpp, hf, seed=2, n_random=100)

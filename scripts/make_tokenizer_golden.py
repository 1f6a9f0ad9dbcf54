"""Writes tests/golden/tokenizer_gpt2_cases.json: (text -> ids) and (ids -> text) pairs produced by the `tokenizers`
package (Python binding of the Rust crate the reference wraps, Cargo.lock:1296) on the reference's tokenizer/gpt2.json
with the reference's special tokens added (src/tokenize.rs:154-168).  Run in the build container only (needs
/root/reference); the fixture holds inputs and expected outputs, not the vocabulary."""
import json
import os
import sys

import tokenizers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_tokenizer_cpu import SPECIALS, REF_JSON  # noqa: E402

TEXTS = [
    "And so my fellow Americans, ask not what your country can do for you, ask what you can do for your country.",
    " The quick brown fox jumps over the lazy dog.", "It's 9:30 p.m. -- we're late!", "  two  spaces  and\ttabs\n",
    "café naïve 中文 日本語 \U0001F600", "<|startoftranscript|><|notimestamps|> hi<|endoftext|>", "♪♪♪",
]
IDS = [
    [50257, 50363, 464, 2068, 7586, 21831, 50463, 50256],
    [50360, 15496, 50257, 50363, 50563],
    [447, 247, 447, 5008, 5008, 50258, 60000],
    [10263, 120, 254, 10263],
]


def main():
    tk = tokenizers.Tokenizer.from_file(REF_JSON)
    tk.add_special_tokens([tokenizers.AddedToken(s, special=True) for s in SPECIALS])
    out = {"generator": "scripts/make_tokenizer_golden.py", "tokenizers_version": tokenizers.__version__,
           "encode": [{"text": t, "ids": tk.encode(t, add_special_tokens=True).ids} for t in TEXTS],
           "decode": [{"ids": i, "text": tk.decode(i, skip_special_tokens=False)} for i in IDS]}
    path = os.path.join(ROOT, "tests", "golden", "tokenizer_gpt2_cases.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()

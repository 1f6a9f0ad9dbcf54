"""Re-derive the special ids and non-speech suppress list the reference computes at
run time (src/tokenize.rs:142-193, 215-307) from its tokenizer/gpt2.json, and check
them against whisper_rs_amd/consts.py.  Runs only where /root/reference exists."""
import sys, json
sys.path.insert(0, ".")
from tokenizers import Tokenizer, AddedToken
from whisper_rs_amd.consts import TokenIds, NON_SPEECH_TOKENS

path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tokenizer/gpt2.json"
tok = Tokenizer.from_file(path)
LANGS = 99  # len(LANGUAGES) in src/tokenize.rs:7-108
specials = ["<|startoftranscript|>"] + [f"<|lang{i}|>" for i in range(LANGS)] + [
    "<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
tok.add_special_tokens([AddedToken(s, special=True) for s in specials])
ids = {s: tok.token_to_id(s) for s in specials}
got = dict(eot=tok.token_to_id("<|endoftext|>"), sot=ids["<|startoftranscript|>"], translate=ids["<|translate|>"],
           transcribe=ids["<|transcribe|>"], startoflm=ids["<|startoflm|>"], startofprev=ids["<|startofprev|>"],
           no_speech=ids["<|nospeech|>"], no_timestamps=ids["<|notimestamps|>"], ts_begin=ids["<|notimestamps|>"] + 1,
           space=tok.encode(" ", add_special_tokens=True).ids[0])
print(got)
assert got == TokenIds.english().__dict__, (got, TokenIds.english())

symbols = list("\"#()*+/:;<=>@[\\]^_`{|}~「」『』") + ["<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "♪♪", "♪♪♪"]
res = [tok.encode(" -").ids[0], tok.encode(" '").ids[0]]
for s in symbols:
    for t in (s, " " + s):
        e = tok.encode(t).ids
        if len(e) == 1:
            res.append(e[0])
for s in "♩♪♫♬♭♮♯":
    for t in (s, " " + s):
        res.append(tok.encode(t).ids[0])
res.sort()
print(len(res), len(set(res)))
assert tuple(sorted(set(res))) == NON_SPEECH_TOKENS
print("OK: consts.py matches the reference tokenizer file")

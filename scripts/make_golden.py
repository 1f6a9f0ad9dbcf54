#!/usr/bin/env python3
"""Generate tests/golden/*.json: outputs of the CPU oracle (oracle/) on seeded synthetic inputs, plus summary
statistics of the reference's own mel-spectrogram fixture.  Inputs are regenerated from seeds by
whisper_rs_amd.synth, so only outputs are stored.  Run from the repo root in the build container:

    python scripts/make_golden.py [--reference /root/reference]

The reference's two test fixtures are data and are copied verbatim: test/data/mel-filter-8x8.csv and
test/data/mel-spectrogram.csv (the expected log-mel of a real speech clip, 80 x 1100; its input WAV is not shipped by the
reference, so it is used as INPUT to the encoder / decoder / transcribe parity tests, tests/test_gpu_real_speech.py).
Decode cases store, besides the oracle's per-step trace, each row's 12 best (id, log-prob) of the filtered distribution
of every step, so that a near-tie split of the fp16 device path can be priced in the oracle's own scores without
running the oracle (tests/parity.py)."""
import argparse, dataclasses, json, os, shutil, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from whisper_rs_amd import synth
from whisper_rs_amd.consts import MODEL_DIMS
from oracle import audio as OA, decode as OD, transcribe as OT
from helpers import oracle_model, oracle_ids, oracle_decode_options, mel_windows

ap = argparse.ArgumentParser()
ap.add_argument("--reference", default="/root/reference")
args = ap.parse_args()
G = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)

# 1. reference fixtures
ref_csv = os.path.join(args.reference, "test/data/mel-spectrogram.csv")
if os.path.exists(ref_csv):
    shutil.copyfile(ref_csv, os.path.join(G, "mel-spectrogram.csv"))
    m = np.loadtxt(ref_csv, delimiter=",", dtype=np.float64)
    json.dump({"source": "reference test/data/mel-spectrogram.csv (expected output of test_mel_spectrogram, audio.rs:281-301)",
               "shape": list(m.shape), "min": m.min(), "max": m.max(), "mean": m.mean(), "std": m.std(),
               "corner_4x8": m[:4, :8].tolist()}, open(os.path.join(G, "reference_mel_spectrogram_stats.json"), "w"), indent=1)

# 2. oracle log-mel of the seeded synthetic windows: checksum + samples
a = synth.make_audio(2, seed=1234)
mel = np.stack([OA.log_mel_spectrogram(torch.from_numpy(x), OA.mel_filter()).numpy() for x in a])
idx = (np.arange(0, 80, 7)[:, None], np.arange(0, 3000, 211)[None, :])
json.dump({"input": "synth.make_audio(2, seed=1234)", "sum": [float(m.astype(np.float64).sum()) for m in mel],
           "min": [float(m.min()) for m in mel], "max": [float(m.max()) for m in mel],
           "rows": idx[0].ravel().tolist(), "cols": idx[1].ravel().tolist(),
           "samples": [m[idx].tolist() for m in mel]}, open(os.path.join(G, "oracle_log_mel.json"), "w"))

# 3. oracle encoder samples (micro dims)
om = oracle_model("micro")
enc = om.encoder.forward(torch.from_numpy(mel_windows(1))).numpy()[0]
eidx = (np.arange(0, 1500, 137)[:, None], np.arange(0, 128, 17)[None, :])
json.dump({"model": "micro seed 0", "input": "mel_windows(1)", "rows": eidx[0].ravel().tolist(), "cols": eidx[1].ravel().tolist(),
           "samples": enc[eidx].tolist(), "abs_sum": float(np.abs(enc).astype(np.float64).sum())}, open(os.path.join(G, "oracle_encoder_micro.json"), "w"))

# 4. oracle decode results + traces
cases = {}
def run_case(name, dims_name, n_audio, opts, prompt=None, **wkw):
    m = oracle_model(dims_name, **wkw)
    t = OD.DecodeTask(m, oracle_ids(MODEL_DIMS[dims_name].n_vocab), oracle_decode_options(**opts))
    if prompt is not None:
        t.set_prompt(torch.tensor(prompt, dtype=torch.int64))
    t.logits_trace = []
    outs = t.run_batch(torch.from_numpy(mel_windows(n_audio)))
    tr = t.extractor.trace
    top = [lg.log_softmax(-1, dtype=torch.float32).topk(12, dim=-1) for lg in t.logits_trace]
    cases[name] = {"dims": dims_name, "n_audio": n_audio, "options": {k: v for k, v in opts.items()}, "prompt": prompt, "weights_kw": wkw,
                   "tokens": [o.tokens.tolist() for o in outs], "sum_logprob": [o.sum_logprob for o in outs],
                   "avg_logprob": [o.avg_logprob for o in outs], "no_speech_prob": [o.no_speech_prob for o in outs],
                   "n_steps": len(tr), "trace_tokens": [x[0].tolist() for x in tr], "trace_sources": [x[1].tolist() for x in tr],
                   "trace_sum_logprobs": [[float(v) for v in x[2]] for x in tr],
                   "initial_tokens": t.prompt_ref[0].initial_tokens.tolist(),
                   "cand_ids": [tp.indices.tolist() for tp in top],
                   "cand_lp": [[[None if v == float("-inf") else round(float(v), 6) for v in row] for row in tp.values.tolist()] for tp in top]}
run_case("nano_beam5", "nano", 1, dict(sample_len=16, beam_size=5))
run_case("micro_beam5_x2", "micro", 2, dict(sample_len=24, beam_size=5))
run_case("micro_greedy_x2", "micro", 2, dict(sample_len=20, beam_size=None))
run_case("micro_forced_eot", "micro", 2, dict(sample_len=40, beam_size=5), force_eot_after=10)
run_case("micro_prompt", "micro", 1, dict(sample_len=16, beam_size=5), prompt=[50364, 1000, 2000, 50400, 50400, 777])
run_case("micro_beam3_patience2_notimestamps", "micro", 1, dict(sample_len=12, beam_size=3, patience=2.0, timestamps=False))
json.dump(cases, open(os.path.join(G, "oracle_decode_cases.json"), "w"))

# 5. oracle transcribe (nano, 50 s)
audio = synth.make_audio(1, seed=77, n_samples=16000 * 50)[0]
out = {}
for cond in (False, True):
    t = OT.TranscribeTask(oracle_model("nano"), oracle_ids(), oracle_decode_options(sample_len=10), None, cond)
    r = t.run(torch.from_numpy(audio))
    out[str(cond)] = {"tokens": r.tokens.tolist(), "seeks": [w[0] for w in r.windows],
                      "segments": [dataclasses.asdict(s) for s in r.segments]}
json.dump({"model": "nano seed 0", "input": "synth.make_audio(1, seed=77, n_samples=800000)", "condition_on_prev_text": out},
          open(os.path.join(G, "oracle_transcribe_nano.json"), "w"))
print("golden fixtures written to", G, [f for f in os.listdir(G)])

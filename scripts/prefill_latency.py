#!/usr/bin/env python3
"""prefill latency of one window with a long prompt (what condition_on_prev_text feeds every window after the first,
reference transcribe.rs:155-158): large-v2, beam 5, prompt of N tokens.  Run once as is (batched causal prefill) and once
with WRS_PREFILL_SEQUENTIAL=1 (one decoder step per prompt token, the round-1 behaviour).
    python scripts/prefill_latency.py [n_prompt_tokens] [n_windows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import os
os.environ.setdefault("WRS_DEBUG_LIB", "1")  # test build: include/wrs_debug.h hooks + experiment switches
import whisper_rs_amd as W
from whisper_rs_amd import synth
from bench import make_weights_device

n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dims = W.MODEL_DIMS["large-v2"]
dev = torch.device("cuda", 0)
model = W.Whisper(dims)
wts = make_weights_device(dims, dev, seed=0)
torch.cuda.synchronize()
for name, t in wts.items():
    model.set_tensor(name, t)
model.finalize()
del wts
ids = W.TokenIds.for_vocab(dims.n_vocab)
task = W.DecodeTask(model, ids, W.DecodeOptions(sample_len=16, max_audio=n_win))
rng = np.random.default_rng(0)
task.set_prompt([int(x) for x in rng.integers(0, 50000, size=n_prompt)])
audio = synth.make_audio(n_win, seed=1)
for it in range(3):
    t0 = time.perf_counter()
    outs = task.run_audio(audio)
    dt = time.perf_counter() - t0
    print(f"mode={'sequential' if os.environ.get('WRS_PREFILL_SEQUENTIAL') == '1' else 'batched'} windows={n_win} prompt={n_prompt} "
          f"run {it}: total {dt * 1e3:.1f} ms, split {task.last_run_times()}, first tokens {outs[0].tokens[:6].tolist()}")

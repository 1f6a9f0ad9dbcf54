"""Experiment: do two model replicas on ONE GPU, each decoding its own window batch from its own host thread / stream,
overlap one replica's MFMA-bound encoder with the other's HBM-bound decode steps?"""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
import torch
import os
os.environ.setdefault("WRS_DEBUG_LIB", "1")  # test build: include/wrs_debug.h hooks + experiment switches
import whisper_rs_amd as W
from whisper_rs_amd import synth
from bench import make_weights_device

R = int(sys.argv[1]); batch = int(sys.argv[2]); steps = int(sys.argv[3]); stagger = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
dims = W.MODEL_DIMS["large-v2"]; ids = W.TokenIds.for_vocab(dims.n_vocab)
dev = torch.device("cuda", 0)
w = make_weights_device(dims, dev, seed=0)
reps = []
for r in range(R):
    m = W.Whisper(dims, device=0)
    for name, t in w.items():
        m.set_tensor(name, t)
    m.finalize()
    task = W.DecodeTask(m, ids, W.DecodeOptions(sample_len=224, max_audio=batch, token_extract_mode=W.TokenExtractMode.BeamSearch(5, 1.0)))
    audio = torch.from_numpy(synth.make_audio(batch, seed=1234 + 1000 * r)).to(dev)
    reps.append((m, task, audio))
del w
torch.cuda.empty_cache()
torch.cuda.synchronize()
print("free GB", torch.cuda.mem_get_info()[0] / 1e9, flush=True)

def work(r, n, delay):
    time.sleep(delay)
    m, task, audio = reps[r]
    for _ in range(n):
        task.run_audio_device(audio.data_ptr(), batch)

def run(n):
    th = [threading.Thread(target=work, args=(r, n, stagger * r)) for r in range(R)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0

run(1)
el = run(steps)
print(f"replicas {R} x batch {batch}, {steps} steps each, stagger {stagger}: {el:.3f} s -> xRT {R * batch * steps * 30.0 / el:.1f}", flush=True)
print({k: round(v, 1) for k, v in reps[0][1].last_run_times().items()})

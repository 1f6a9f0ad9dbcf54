import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import os
os.environ.setdefault("WRS_DEBUG_LIB", "1")  # test build: include/wrs_debug.h hooks + experiment switches
import whisper_rs_amd as W
from helpers import gpu_model, oracle_model, oracle_ids, oracle_decode_options, mel_windows
from oracle import decode as OD
np.set_printoptions(linewidth=200, precision=5, suppress=True)
name, n_audio, sl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mel = mel_windows(n_audio)
om = oracle_model(name)
ot = OD.DecodeTask(om, oracle_ids(), oracle_decode_options(sample_len=sl, beam_size=5))
ot.logits_trace = []
oo = ot.run_batch(torch.from_numpy(mel))
tr = ot.extractor.trace
gm = gpu_model(W, name)
gt = W.DecodeTask(gm, W.TokenIds.english(), W.DecodeOptions(sample_len=sl, max_audio=n_audio))
go = gt.run(mel)
t, s, l = gt.trace()
for st in range(min(len(tr), t.shape[0])):
    same = np.array_equal(t[st], tr[st][0].numpy()) and np.array_equal(s[st], tr[st][1].numpy())
    print("step", st, "same" if same else "DIFF")
    if not same or st < 2:
        print("  oracle tok", tr[st][0].numpy(), "src", tr[st][1].numpy()); print("  oracle slp", tr[st][2].numpy())
        print("  gpu    tok", t[st], "src", s[st]); print("  gpu    slp", l[st])
    if not same:
        # oracle candidates at this step: top-6 per beam
        lp = ot.logits_trace[st].log_softmax(-1)
        v, i = lp.topk(6, -1)
        prev = tr[st - 1][2].numpy() if st > 0 else np.zeros(lp.shape[0])
        for b in range(lp.shape[0]):
            print("   beam", b, "cand", [(int(i[b, c]), round(float(prev[b] + v[b, c]), 4)) for c in range(6)])
        break

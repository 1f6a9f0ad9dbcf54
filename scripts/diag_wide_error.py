"""Diagnosis: per-step distribution of |device candidate log-prob - oracle's| on the wide model (D = 1280, 2 layers), 96 audios x
beam 5 teacher-forced — does the fp16 path's error grow with the number of cached positions?
    python scripts/diag_wide_error.py [steps 48]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import os
os.environ.setdefault("WRS_DEBUG_LIB", "1")  # test build: include/wrs_debug.h hooks + experiment switches
import whisper_rs_amd as W
from whisper_rs_amd import debug as DBG
from whisper_rs_amd.consts import MODEL_DIMS, TokenIds
import parity as P
from helpers import gpu_model, oracle_model, oracle_ids, oracle_decode_options
from oracle import decode as OD

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
g = torch.Generator(); g.manual_seed(7)
feats = torch.randn(n, 1500, 1280, generator=g).to(torch.float16).to(torch.float32).numpy()
om = oracle_model("wide")
ot = OD.DecodeTask(om, oracle_ids(MODEL_DIMS["wide"].n_vocab), oracle_decode_options(sample_len=steps, beam_size=5))
ot.logits_trace = []
with torch.no_grad():
    oouts = ot.run_from_features(torch.from_numpy(feats))
view = P.OracleView.from_task(ot, oouts, 5)
gm = gpu_model(W, "wide")
task = W.DecodeTask(gm, TokenIds.multilingual(), W.DecodeOptions(sample_len=steps, max_audio=n))
glp, gid, gns = DBG.decode_forced_features(task, feats, view.tok, view.src)
print("step  max_err   p99.9     p99      median   max|lp|  (err of the top-1 candidate: max, median)")
for s in range(glp.shape[0]):
    ids = torch.from_numpy(gid[s].astype(np.int64)); lp = torch.from_numpy(glp[s])
    want = view.olp(s).gather(1, ids.clamp(min=0))
    ok = (ids >= 0) & torch.isfinite(want) & torch.isfinite(lp)
    d = (lp - want).abs()[ok].numpy()
    d1 = (lp[:, 0] - want[:, 0]).abs().numpy()
    print(f"{s:4d}  {d.max():.2e}  {np.quantile(d, 0.999):.2e}  {np.quantile(d, 0.99):.2e}  {np.median(d):.2e}  {float(lp[ok].abs().max()):6.2f}   {d1.max():.2e} {np.median(d1):.2e}")

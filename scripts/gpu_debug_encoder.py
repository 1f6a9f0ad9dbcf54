import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import os
os.environ.setdefault("WRS_DEBUG_LIB", "1")  # test build: include/wrs_debug.h hooks + experiment switches
import whisper_rs_amd as W
from whisper_rs_amd import synth
from whisper_rs_amd.consts import ModelDims
from oracle import model as OM
from helpers import mel_windows
mel = mel_windows(1)
for (D, H, L) in [(128, 2, 4), (192, 3, 1), (256, 4, 1), (384, 6, 1), (384, 6, 2), (512, 8, 1)]:
    dims = ModelDims(51864, 1500, D, H, L, 448, D, H, 1)
    w = synth.make_weights(dims, seed=0)
    want = OM.Whisper(w, dims).encoder.forward(torch.from_numpy(mel)).numpy()
    gm = W.Whisper(dims, w)
    got = gm.encoder_forward(mel)
    err = np.abs(got - want)
    print(D, H, L, "max", err.max(), "mean", err.mean(), "worst row", int(err.max(-1).argmax()), "worst col", int(err.max(1).argmax()), flush=True)
    gm.close()

import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, torch.nn.functional as F
import os
os.environ.setdefault("WRS_DEBUG_LIB", "1")  # test build: include/wrs_debug.h hooks + experiment switches
import whisper_rs_amd as W
from whisper_rs_amd import synth, _lib
from whisper_rs_amd.consts import ModelDims
from oracle import model as OM
from helpers import mel_windows
lib = _lib.load()
lib.wrs_debug_set_encoder_stop.argtypes = [C.c_void_p, C.c_int]
lib.wrs_debug_read_encoder.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
mel = mel_windows(1)
D, H, L = int(sys.argv[1]), int(sys.argv[2]), 1
dims = ModelDims(51864, 1500, D, H, L, 448, D, H, 1)
w = synth.make_weights(dims, seed=0)
tw = {k: torch.from_numpy(v) for k, v in w.items()}
gm = W.Whisper(dims, w)
def read(which, shape):
    out = np.empty(shape, np.float32)
    assert lib.wrs_debug_read_encoder(gm._h, which, out.ctypes.data_as(C.c_void_p), out.size) == 0
    return out
def run(stop):
    lib.wrs_debug_set_encoder_stop(gm._h, stop)
    try:
        gm.encoder_forward(mel)
    except Exception as e:
        print("err", e)
def rep(name, got, want):
    err = np.abs(got - want)
    print(f"{name:12s} max {err.max():.5f} mean {err.mean():.6f} scale {np.abs(want).max():.3f} worst {np.unravel_index(err.argmax(), err.shape)}", flush=True)
x = torch.from_numpy(mel)
c1 = F.gelu(F.conv1d(x, tw["encoder.conv1.weight"], tw["encoder.conv1.bias"], padding=1))
run(1); g = read(5, (3001, D)); rep("conv1", g[1:], c1[0].T.numpy())
c2 = F.gelu(F.conv1d(c1, tw["encoder.conv2.weight"], tw["encoder.conv2.bias"], stride=2, padding=1)).permute(0, 2, 1) + OM.sinusoids(1500, D)
run(2); rep("conv2+pos", read(0, (1500, D)), c2[0].numpy())
p = "encoder.blocks.0"
ln1 = F.layer_norm(c2, (D,), tw[p + ".attn_ln.weight"], tw[p + ".attn_ln.bias"], 1e-5)
run(3); rep("ln1", read(1, (1500, D)), ln1[0].numpy())
sc = 64 ** -0.25
q = F.linear(ln1, tw[p + ".attn.query.weight"], tw[p + ".attn.query.bias"]) * sc
k = F.linear(ln1, tw[p + ".attn.key.weight"]) * sc
v = F.linear(ln1, tw[p + ".attn.value.weight"], tw[p + ".attn.value.bias"])
run(4); g = read(2, (1500, 3 * D)); rep("q", g[:, :D], q[0].numpy()); rep("k", g[:, D:2 * D], k[0].numpy()); rep("v", g[:, 2 * D:], v[0].numpy())
qh = q.view(1, 1500, H, 64).permute(0, 2, 1, 3); kh = k.view(1, 1500, H, 64).permute(0, 2, 3, 1); vh = v.view(1, 1500, H, 64).permute(0, 2, 1, 3)
att = (qh @ kh).softmax(-1) @ vh
att = att.permute(0, 2, 1, 3).flatten(2)
run(6); rep("attn", read(3, (1500, D)), att[0].numpy())
x1 = c2 + F.linear(att, tw[p + ".attn.out.weight"], tw[p + ".attn.out.bias"])
run(7); rep("x+out", read(0, (1500, D)), x1[0].numpy())
ln2 = F.layer_norm(x1, (D,), tw[p + ".mlp_ln.weight"], tw[p + ".mlp_ln.bias"], 1e-5)
run(8); rep("ln2", read(1, (1500, D)), ln2[0].numpy())
h = F.gelu(F.linear(ln2, tw[p + ".mlp.0.weight"], tw[p + ".mlp.0.bias"]))
run(9); rep("mlp1", read(4, (1500, 4 * D)), h[0].numpy())
x2 = x1 + F.linear(h, tw[p + ".mlp.2.weight"], tw[p + ".mlp.2.bias"])
run(10); rep("x+mlp2", read(0, (1500, D)), x2[0].numpy())

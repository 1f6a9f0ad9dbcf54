"""Converts a Whisper checkpoint to the safetensors file `wrs_model_load_safetensors` reads.

  python scripts/convert_checkpoint.py base.en.pt weights.safetensors [--f16]

Input: an OpenAI checkpoint ({"dims": ..., "model_state_dict": ...}) or a bare state_dict, loaded with
torch.load(weights_only=True) (nothing from the file is executed).  The tensor names are kept: they are the VarStore paths
the reference creates.  Prints the `ModelDims` to pass to wrs_model_create."""
import argparse
import json
import sys

import torch
from safetensors.torch import save_file


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--f16", action="store_true", help="store matrices as fp16 (the MFMA path converts them anyway)")
    a = ap.parse_args()
    ck = torch.load(a.src, map_location="cpu", weights_only=True)
    sd = ck["model_state_dict"] if isinstance(ck, dict) and "model_state_dict" in ck else ck
    out = {}
    for k, v in sd.items():
        if not torch.is_tensor(v):
            continue
        v = v.detach().contiguous()
        keep32 = v.dim() == 1 or "ln" in k or k.endswith("positional_embedding")
        out[k] = v.to(torch.float32) if (keep32 or not a.f16) else v.to(torch.float16)
    save_file(out, a.dst)
    if isinstance(ck, dict) and "dims" in ck:
        d = ck["dims"]
        print(json.dumps({"n_vocab": d["n_vocab"], "n_audio_ctxs": d["n_audio_ctx"], "n_audio_states": d["n_audio_state"],
                          "n_audio_heads": d["n_audio_head"], "n_audio_layers": d["n_audio_layer"], "n_text_ctxs": d["n_text_ctx"],
                          "n_text_states": d["n_text_state"], "n_text_heads": d["n_text_head"], "n_text_layers": d["n_text_layer"]}))
    print(f"wrote {len(out)} tensors to {a.dst}", file=sys.stderr)


if __name__ == "__main__":
    main()

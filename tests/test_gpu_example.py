"""GPU: the C++ counterpart of the reference's examples/transcribe.rs runs end to end on files — WAV in (load_audio),
tokenizer.json in, text and segments out — through nothing but the C-ABI."""
import os
import subprocess
import wave

import numpy as np
import pytest

from helpers import synthetic_tokenizer_json
from whisper_rs_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_transcribe_on_files(gpu, tmp_path):
    exe = os.path.join(ROOT, "examples", "transcribe")
    assert os.path.exists(exe), "examples/transcribe is built by `make` / __graft_entry__.build()"
    audio = synth.make_audio(1, seed=9, n_samples=16000 * 35)[0]
    wav = tmp_path / "in.wav"
    with wave.open(str(wav), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.clip(np.round(audio * 32767), -32768, 32767).astype("<i2").tobytes())
    tok = tmp_path / "tokenizer.json"
    tok.write_bytes(synthetic_tokenizer_json())
    r = subprocess.run([exe, str(wav), str(tok)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "loaded audio (560000 samples)" in r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].startswith("transcribed audio:") and lines[1].startswith("text: ")
    assert sum(1 for l in lines if l.startswith("  segment seek=")) >= 2  # 35 s -> at least two windows
    # same program without files: synthetic everything, ids only
    r2 = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and "text:" not in r2.stdout

"""GPU parity: DecodeTask::run (reference src/decode/mod.rs:284-419) — encoder, sampling loop with the
fused filters, beam search / greedy, ranking — vs the CPU oracle on identical seeded inputs.

Bar (tests/parity.py): token ids and beam source indices bit-exact at every step and in the final ranking; every
candidate log-prob the device proposes within CAND_ERR of the oracle's; cumulative scores within DRIFT_TOL.  The HIP
path multiplies in fp16, so on synthetic (random-weight) models two beam candidates are occasionally closer than that
noise: such a split must be a near-tie IN THE ORACLE'S OWN SCORES (margin <= 2 * CAND_ERR + 2 * measured drift, recorded
in the test output and gpurun_out/parity_splits.jsonl), after which the device is re-synchronised with the oracle and
the comparison continues to the last step, finalize and rank.  In the f32 parity mode (test_gpu_parity_f32.py) no split
is allowed at all."""
import numpy as np
import pytest
import torch

import parity as P
from helpers import gpu_model, oracle_model, oracle_ids, oracle_decode_options, mel_windows
from whisper_rs_amd import debug as DBG
from whisper_rs_amd.consts import MODEL_DIMS, TokenIds
from oracle import decode as OD

pytestmark = pytest.mark.gpu
LP_TOL = P.DRIFT_TOL


def _oracle_run(dims_name, mel, oopts, prompt=None, languages=None, **wkw):
    """oracle decode of mel [n, 80, 3000] with the post-filter logits of every step recorded -> (OracleView, task)."""
    om = oracle_model(dims_name, **wkw)
    task = OD.DecodeTask(om, oracle_ids(MODEL_DIMS[dims_name].n_vocab), oopts)
    if prompt is not None:
        task.set_prompt(torch.tensor(prompt, dtype=torch.int64))
    task.languages = languages
    task.logits_trace = []
    outs = task.run_batch(torch.from_numpy(mel))
    G = oopts.beam_size if oopts.beam_size is not None else 1
    return P.OracleView.from_task(task, outs, G), task


def _free_run(case, gtask, mel, view, G, exact=False):
    return P.run_with_resync(case, gtask, mel, view, G, gtask.ids.eot, exact=exact)


@pytest.mark.parametrize("dims_name,n_audio,sample_len", [("nano", 1, 16), ("micro", 2, 24), ("micro", 3, 40)])
def test_beam_search_decode_vs_oracle(gpu, dims_name, n_audio, sample_len):
    mel = mel_windows(n_audio)
    view, _ = _oracle_run(dims_name, mel, oracle_decode_options(sample_len=sample_len, beam_size=5))
    gm = gpu_model(gpu, dims_name)
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=sample_len, max_audio=n_audio))
    _free_run(f"{dims_name}_beam5_x{n_audio}", task, mel, view, 5)
    # second run on the same task must reproduce itself exactly (cache/ancestry reset)
    gouts = task.run(mel)
    gouts2 = task.run(mel)
    for a, b in zip(gouts, gouts2):
        np.testing.assert_array_equal(a.tokens, b.tokens)
        assert a.sum_logprob == b.sum_logprob
    task.close()


def test_beam_search_decode_with_mfma_self_attention(gpu, monkeypatch):
    """the beam-de-duplicating MFMA self-attention (csrc/attn_self_mfma.hip: what the benchmark runs; by default from 64
    audios per step) on a 3-audio beam-5 decode and a 2-audio beam-3 decode with a length penalty: free-running vs the oracle."""
    from helpers import switched_models
    with switched_models(monkeypatch, WRS_SELF_MFMA_MIN_AUDIO=1):  # read when the model is created
        mel = mel_windows(3)
        view, _ = _oracle_run("micro", mel, oracle_decode_options(sample_len=40, beam_size=5))
        gm = gpu_model(gpu, "micro")
        task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=40, max_audio=3))
        _free_run("micro_beam5_x3_mfma_self", task, mel, view, 5)
        assert DBG.step_paths(gm) & DBG.STEP_SELF_MFMA, "the MFMA self-attention did not run"
        task.close()
        mel = mel_windows(2)
        view, _ = _oracle_run("nano", mel, oracle_decode_options(sample_len=24, beam_size=3, len_penalty=0.8))
        gm = gpu_model(gpu, "nano")
        task = gpu.DecodeTask(gm, TokenIds.english(),
                              gpu.DecodeOptions(sample_len=24, max_audio=2, token_extract_mode=gpu.TokenExtractMode.BeamSearch(3, 1.0), len_penalty=0.8))
        _free_run("nano_beam3_x2_mfma_self", task, mel, view, 3)
        assert DBG.step_paths(gm) & DBG.STEP_SELF_MFMA
        task.close()


def test_greedy_equals_beam1(gpu):
    """greedy (unreachable in the reference) == greedy.rs update + one EOT pad; token-equal to beam_size 1."""
    mel = mel_windows(2)
    gm = gpu_model(gpu, "micro")
    tg = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=20, max_audio=2, token_extract_mode=gpu.TokenExtractMode.Greedy()))
    tb = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=20, max_audio=2, token_extract_mode=gpu.TokenExtractMode.BeamSearch(1, 1.0)))
    og, ob = tg.run(mel), tb.run(mel)
    for a, b in zip(og, ob):
        np.testing.assert_array_equal(a.tokens, b.tokens)
        assert abs(a.sum_logprob - b.sum_logprob) < 1e-3
    view, _ = _oracle_run("micro", mel, oracle_decode_options(sample_len=20, beam_size=None))
    _free_run("micro_greedy_x2", tg, mel, view, 1)
    tg.close(); tb.close()


def test_forced_eot_finishes_early(gpu):
    """weights built so <|endoftext|> wins around position 10: exercises finished/patience bookkeeping
    (beam_search.rs:88-146) and finalize (151-206)."""
    kw = dict(force_eot_after=10)
    mel = mel_windows(2)
    view, _ = _oracle_run("micro", mel, oracle_decode_options(sample_len=40, beam_size=5), **kw)
    gm = gpu_model(gpu, "micro", **kw)
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=40, max_audio=2))
    assert view.tok.shape[0] < 40, "fixture did not finish early in the oracle; adjust force_eot_after"
    _free_run("micro_forced_eot", task, mel, view, 5)
    task.close()


def test_prompt_conditioning_and_position_quirk(gpu):
    """set_prompt ([startofprev, prompt.., sot], decode/mod.rs:46-92) + the reference's offset = loop index (Q2)."""
    mel = mel_windows(1)
    prompt = [50364, 1000, 2000, 50400, 50400, 777]
    for quirk in (True, False):
        view, _ = _oracle_run("micro", mel, oracle_decode_options(sample_len=16, beam_size=5, position_offset_quirk=quirk), prompt=prompt)
        gm = gpu_model(gpu, "micro")
        task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=16, position_offset_quirk=quirk))
        task.set_prompt(prompt)
        _free_run(f"micro_prompt_quirk{int(quirk)}", task, mel, view, 5)
        task.close()


def test_long_prompt_prefill(gpu):
    """a 200-token prompt (what condition_on_prev_text feeds the second window of a file, transcribe.rs:155-158) goes
    through the batched causal prefill (decode/mod.rs:236-245: all sample_begin tokens in one forward pass)."""
    mel = mel_windows(2)
    rng = np.random.default_rng(11)
    prompt = [int(x) for x in rng.integers(0, 50000, size=200)]
    view, _ = _oracle_run("micro", mel, oracle_decode_options(sample_len=12, beam_size=5), prompt=prompt)
    task = gpu.DecodeTask(gpu_model(gpu, "micro"), TokenIds.english(), gpu.DecodeOptions(sample_len=12, max_audio=2))
    task.set_prompt(prompt)
    _free_run("micro_prompt200_x2", task, mel, view, 5)
    task.close()


def test_options_matrix(gpu):
    mel = mel_windows(1)
    gm = gpu_model(gpu, "nano")
    for kw_g, kw_o in [
        (dict(timestamps=False), dict(timestamps=False)),
        (dict(suppress_non_speech=False, suppress_tokens=[11, 13, 50257]), dict(suppress_non_speech=False, suppress_tokens=[11, 13, 50257])),
        (dict(len_penalty=1.0, max_initial_timestamp=None), dict(len_penalty=1.0, max_initial_timestamp=None)),
        (dict(token_extract_mode=gpu.TokenExtractMode.BeamSearch(3, 2.0)), dict(beam_size=3, patience=2.0)),
    ]:
        view, _ = _oracle_run("nano", mel, oracle_decode_options(sample_len=12, **kw_o))
        task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=12, **kw_g))
        _free_run("nano_options_" + "_".join(sorted(kw_o)), task, mel, view, kw_o.get("beam_size", 5))
        task.close()


def test_run_audio_matches_run_mel(gpu):
    from whisper_rs_amd import synth
    a = synth.make_audio(2, seed=1234)
    gm = gpu_model(gpu, "nano")
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=10, max_audio=2))
    o1 = task.run_audio(a)
    o2 = task.run(gpu.audio.log_mel_spectrogram(a))
    for x, y in zip(o1, o2):
        np.testing.assert_array_equal(x.tokens, y.tokens)
    task.close()


def test_audio_features_copy_out(gpu):
    """`DecodeOutput.audio_features` (decode/mod.rs:110,399): the encoder output of every window of the decoded batch can be
    copied to the host and equals AudioEncoder::forward of the same spectrogram; once the encoder workspace belongs to a
    later batch (another run, or a batch queued with encode_async) the device pointer is NULL and the copy is refused."""
    from whisper_rs_amd import synth
    d = MODEL_DIMS["nano"]
    gm = gpu_model(gpu, "nano")
    mel = mel_windows(2)
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=4, max_audio=2))
    outs = task.run(mel)
    assert all(o.audio_features_device for o in outs)
    got = np.stack([task.read_audio_features(i, d.n_audio_ctxs, d.n_audio_states) for i in range(2)])
    with pytest.raises(gpu.WrsError):
        task.read_audio_features(2, d.n_audio_ctxs, d.n_audio_states)
    np.testing.assert_array_equal(got, gm.encoder_forward(mel))
    with pytest.raises(gpu.WrsError):  # encoder_forward above re-used the workspace (same values, but not this batch's any more)
        task.read_audio_features(0, d.n_audio_ctxs, d.n_audio_states)
    a = synth.make_audio(2, seed=77)
    task.encode_async(a)
    task.encode_async(a)
    stale = task.run_pending()  # the second batch's encoder owns the workspace
    assert all(not o.audio_features_device for o in stale)
    with pytest.raises(gpu.WrsError):
        task.read_audio_features(0, d.n_audio_ctxs, d.n_audio_states)
    live = task.run_pending()
    assert all(o.audio_features_device for o in live)
    assert np.isfinite(task.read_audio_features(1, d.n_audio_ctxs, d.n_audio_states)).all()
    task.close()


def test_capacity_and_errors(gpu):
    gm = gpu_model(gpu, "nano")
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=4, max_audio=1))
    with pytest.raises(gpu.WrsError):
        task.run(mel_windows(2))
    with pytest.raises(gpu.WrsError):
        task.set_prompt([60000])
    task.close()
    with pytest.raises(gpu.WrsError):
        gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(token_extract_mode=gpu.TokenExtractMode.BeamSearch(9, 1.0)))


# ------------------------------------------------------------------------------------------------
# Teacher-forced verification: the device follows the ORACLE's beam decisions at every step, so every step of a long
# decode is checked in ONE device run (the free-running comparison above re-runs the device after each near-tie).
def _teacher_forced_check(gpu, dims_name, n_audio, sample_len, min_exact_frac, feats=None, ids=None, okw=None, gkw=None, cand_tol=P.CAND_ERR,
                          f32_mode=False, quantile_bounds=None, **wkw):
    """feats: encoder outputs [n_audio, 1500, D] (decoder-only run through wrs_debug_decode_forced_features) instead of
    the synthetic mel windows.  okw / gkw: extra oracle / device decode options."""
    ids = ids or TokenIds.for_vocab(MODEL_DIMS[dims_name].n_vocab)
    om = oracle_model(dims_name, **wkw)
    ot = OD.DecodeTask(om, oracle_ids(MODEL_DIMS[dims_name].n_vocab), oracle_decode_options(sample_len=sample_len, beam_size=5, **(okw or {})))
    ot.logits_trace = []
    with torch.no_grad():
        if feats is None:
            mel = mel_windows(n_audio)
            oouts = ot.run_batch(torch.from_numpy(mel))
        else:
            oouts = ot.run_from_features(torch.from_numpy(feats))
    view = P.OracleView.from_task(ot, oouts, 5)
    if f32_mode:
        from helpers import weights
        gm = DBG.parity_f32_model(MODEL_DIMS[dims_name], weights(dims_name, 0, **wkw))
    else:
        gm = gpu_model(gpu, dims_name, **wkw)
    task = gpu.DecodeTask(gm, ids, gpu.DecodeOptions(sample_len=sample_len, max_audio=n_audio, **(gkw or {})))
    if feats is None:
        glp, gid, gns = task.run_forced(mel, view.tok, view.src)
    else:
        glp, gid, gns = DBG.decode_forced_features(task, feats, view.tok, view.src)
    task.close()
    if f32_mode:
        gm.close()
    exact, steps, _ = P.teacher_forced_check(f"{dims_name}_tf_x{n_audio}{'_f32' if f32_mode else ''}", glp, gid, gns, view, 5, ids.eot, min_exact_frac, cand_tol,
                                             quantile_bounds=quantile_bounds)
    return exact, steps


@pytest.mark.parametrize("dims_name,n_audio,sample_len", [("micro", 2, 48), ("tiny.en", 2, 32)])
def test_teacher_forced_steps_vs_oracle(gpu, dims_name, n_audio, sample_len):
    _teacher_forced_check(gpu, dims_name, n_audio, sample_len, 0.95)


def test_teacher_forced_with_early_eot(gpu):
    _teacher_forced_check(gpu, "micro", 2, 40, 0.9, force_eot_after=12)  # 28 selections only: 2 near-ties are 7 %


def test_large_batch_gemm_path_teacher_forced(gpu, monkeypatch):
    """decode steps with many rows (>= 480) run their GEMMs on the tiled kernel instead of the split-K one;
    force that path at test size and hold it to the same teacher-forced bar."""
    from helpers import switched_models
    with switched_models(monkeypatch, WRS_DEC_BIG_GEMM_ROWS=1):  # the threshold is read when the model is created
        _teacher_forced_check(gpu, "micro", 2, 32, 0.95)
        assert DBG.step_paths(gpu_model(gpu, "micro")) & DBG.STEP_TILED, "the tiled-GEMM decode path did not run"


def test_overlapped_mode_equals_sequential(gpu):
    """wrs_decode_task_encode_async / run_pending (encoder of the next batch on a side stream, two cross-K/V buffers)
    return exactly what run_audio returns, batch after batch, in FIFO order."""
    from whisper_rs_amd import synth
    gm = gpu_model(gpu, "micro")
    batches = [synth.make_audio(2, seed=500 + 10 * i) for i in range(4)]
    ref_task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=14, max_audio=2))
    want = [ref_task.run_audio(b) for b in batches]
    ref_task.close()
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=14, max_audio=2))
    got = []
    task.encode_async(batches[0])
    for i in range(len(batches)):
        if i + 1 < len(batches):
            task.encode_async(batches[i + 1])
        got.append(task.run_pending())
    for w, g in zip(want, got):
        for a, b in zip(w, g):
            np.testing.assert_array_equal(a.tokens, b.tokens)
            assert a.sum_logprob == b.sum_logprob and a.no_speech_prob == b.no_speech_prob
    # misuse is refused, and the sequential entry still works afterwards
    with pytest.raises(gpu.WrsError):
        task.run_pending()
    task.encode_async(batches[0]); task.encode_async(batches[1])
    with pytest.raises(gpu.WrsError):
        task.encode_async(batches[2])
    with pytest.raises(gpu.WrsError):
        task.run_audio(batches[0])
    a = task.run_pending(); b = task.run_pending()
    np.testing.assert_array_equal(a[0].tokens, want[0][0].tokens)
    np.testing.assert_array_equal(b[1].tokens, want[1][1].tokens)
    again = task.run_audio(batches[2])
    np.testing.assert_array_equal(again[0].tokens, want[2][0].tokens)
    task.close()


def test_window_pool_equals_sequential(gpu):
    """wrs_window_pool_*: worker threads (one per model replica) pulling window batches from a shared queue return, in
    window order, exactly what a single task returns for each window batch (uneven last batch included)."""
    from whisper_rs_amd import synth
    import helpers
    audio = synth.make_audio(7, seed=900)
    opts = gpu.DecodeOptions(sample_len=12, max_audio=2)
    gm = gpu_model(gpu, "micro")
    task = gpu.DecodeTask(gm, TokenIds.english(), opts)
    want = []
    for i in range(0, 7, 2):
        want += task.run_audio(audio[i:i + 2])
    task.close()
    replicas = [gpu.Whisper(MODEL_DIMS["micro"], helpers.weights("micro")) for _ in range(3)]
    seq_pool = gpu.WindowPool(replicas[:2], TokenIds.english(), opts, overlapped=False)
    for w, g in zip(want, seq_pool.run_audio(audio)):
        np.testing.assert_array_equal(w.tokens, g.tokens)
    seq_pool.close()
    pool = gpu.WindowPool(replicas, TokenIds.english(), opts)
    for _ in range(2):  # a pool is reusable
        got = pool.run_audio(audio)
        assert len(got) == 7
        for w, g in zip(want, got):
            np.testing.assert_array_equal(w.tokens, g.tokens)
            assert w.sum_logprob == g.sum_logprob and w.no_speech_prob == g.no_speech_prob
    one = pool.run_audio(audio[:1])  # fewer batches than workers: the idle workers just return
    assert len(one) == 1 and len(one[0].tokens) > 0
    with pytest.raises(gpu.WrsError):
        gpu.WindowPool([replicas[0], replicas[0]], TokenIds.english(), opts)
    pool.close()
    for r in replicas:
        r.close()

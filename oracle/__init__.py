"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A CPU restatement of the hot path of laptou/whisper-rs (reference mounted at
/root/reference, never shipped): log_mel_spectrogram -> AudioEncoder ->
TextDecoder steps -> logit filters -> beam/greedy extractor -> ranker ->
transcribe loop.  The reference has no arithmetic of its own: every op is a
`tch` 0.10 call into libtorch ATen (Cargo.lock:1225-1227).  This oracle issues the
same ATen CPU operators through the local `torch` (2.10 CPU), in the order the
reference's Rust issues them; each function cites the file:line it follows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product path (whisper_rs_amd/) never
imports it and fails loudly if the HIP library is missing.

PINNING STATUS:
  * mel_filter           — PINNED by the reference's own fixture
                            test/data/mel-filter-8x8.csv (audio.rs:266-279),
                            copied to tests/golden/ as data.
  * log_mel_spectrogram  — the reference fixture test/data/mel-spectrogram.csv
                            needs test/data/jfk_resampled.wav, which the reference
                            does not ship; only its invariants (range exactly 2.0,
                            min/max) can be checked.  PARITY UNPINNED beyond that.
  * encoder / decoder / filters / beam search / ranker / transcribe —
                            the reference has no tests or fixtures for these:
                            PARITY UNPINNED by the reference.  Cross-checked
                            instead against an independent implementation of the
                            same network (transformers' WhisperModel with the same
                            random weights; tests/test_oracle_vs_hf.py).
The reference (Rust) cannot be built here: no cargo/rustc, `tch` is a git
dependency — so there is no oracle/_ref.
"""

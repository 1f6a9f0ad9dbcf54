#!/usr/bin/env python3
"""Headline benchmark: xRT (audio-seconds / wall-second) of the Whisper hot path, large-v2,
beam 5, fp16 MFMA, synthetic 30-s windows, on N MI355X of one node (BASELINE.json `metric`).

A "step" is one pass of the whole hot path over one batch of windows already resident in HBM:
log-mel -> AudioEncoder -> cross-K/V projection -> 224 decode steps (beam 5, SuppressBlank /
SuppressTokens / ApplyTimestampRules filters, beam search) -> ranking.  Windows are independent
units: every rank owns a model replica and its own batch, there is no collective
on the data path ("scaling": "weak"); torch.distributed is used only for the barrier and the
max-over-ranks of the timed region.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_ACHIEVABLE_GBS = 6300.0  # what a device copy sustains
MFMA_PEAK_F16 = 2.5e15  # dense fp16 MFMA peak, same guide
PMC_FILE = "profiles/r03_pmc_fetch_write_b307.json"


def end_to_end_roofline(dims, n_audio, beam, n_steps, sample_begin=1):
    """SURVEY.md section 8(d) formula: seconds one pass of `n_audio` windows needs if the decode steps stream their
    algorithmic bytes at the HBM rate and the encoder / cross-K/V projection run at the dense MFMA peak.
    bytes(step at cached length l) = W_step + A * KV_cross + A * G * l * KV_tok + A * G * V * 4."""
    D, L, V, T = dims.n_text_states, dims.n_text_layers, dims.n_vocab, dims.n_audio_ctxs
    w_step = 2.0 * (14.0 * D * D * L + V * D)
    kv_cross = 2.0 * 2 * T * D * L
    kv_tok = 2.0 * 2 * D * L
    rows = n_audio * beam
    dec_bytes = 0.0
    for i in range(n_steps):  # step i reads sample_begin + i cached positions (step 0 = the prompt pass on one row per audio)
        r = n_audio if i == 0 else rows
        dec_bytes += w_step + n_audio * kv_cross + r * (sample_begin + i) * kv_tok + r * V * 4.0
    De, Le = dims.n_audio_states, dims.n_audio_layers
    enc_flops = n_audio * (2.0 * 3000 * 240 * De + 2.0 * T * 3 * De * De + Le * (24.0 * T * De * De + 4.0 * T * T * De))
    ckv_flops = n_audio * 4.0 * T * D * D * L
    mel_bytes = n_audio * (480000 * 4.0 + 80 * 3000 * 2.0)
    out = {}
    for name, bw in (("achievable_6.3TBs", HBM_ACHIEVABLE_GBS * 1e9), ("spec_8TBs", HBM_PEAK_GBS * 1e9)):
        t = (dec_bytes + mel_bytes) / bw + (enc_flops + ckv_flops) / MFMA_PEAK_F16
        out[name] = n_audio * 30.0 / t
    return out, dec_bytes, enc_flops + ckv_flops


def make_weights_device(dims, device, seed=0):
    """Random-init weights of the named architecture generated on the GPU (there are no checkpoints).
    Same distributions as whisper_rs_amd.synth.make_tensor; values are fp16-representable."""
    from whisper_rs_amd.synth import weight_manifest
    g = torch.Generator(device=device)
    out = {}
    for i, (name, shape) in enumerate(weight_manifest(dims)):
        g.manual_seed(seed * 100003 + i)
        r = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        if name.endswith("_ln.weight") or name.endswith("ln_post.weight") or name == "decoder.ln.weight":
            w = 1.0 + 0.1 * r
        elif name.endswith("_ln.bias") or name.endswith("ln_post.bias") or name == "decoder.ln.bias":
            w = 0.1 * r
        elif name.endswith(".bias"):
            w = 0.02 * r
        elif name == "decoder.token_embedding.weight":
            w = (1.5 / shape[1] ** 0.5) * r
        elif name == "decoder.positional_embedding":
            w = (0.5 / shape[1] ** 0.5) * r
        else:
            fan_in = int(np.prod(shape[1:]))
            w = r / fan_in ** 0.5
            if ".cross_attn.query.weight" in name or ".cross_attn.key.weight" in name:
                w = w * 3.0
        out[name] = w.to(torch.float16).to(torch.float32).contiguous()
    return out


def cpu_baseline(dims, weights, audio0, first_tokens, beam, sample_len, n_decode_steps=2):
    """CPU restatement of the reference (oracle/, fp32 ATen CPU ops = what `tch` binds), timed on this host on a
    bounded sample of the SAME workload and the SAME inputs the GPU decoded: the weights of the benchmarked model
    (copied from the device), window 0 of rank 0, and the token ids the device chose for it.  ONE window through log-mel
    and the encoder, then `n_decode_steps` single-token decode steps for one audio x `beam` rows, extrapolated linearly
    to `sample_len` steps.  Two variants: the reference as written (cross-attention K/V recomputed at every step,
    model.rs:79) and `kvcached` (that redundancy removed), so the GPU/CPU ratio is not inflated by the
    reference's ~99 % redundant decode FLOPs.  Reported, not the target."""
    from oracle import model as OM, audio as OA
    threads = torch.get_num_threads()
    w = {k: v.detach().to("cpu", torch.float32) for k, v in weights.items()}
    toks = torch.tensor([[int(t)] for t in first_tokens[:1]] * beam, dtype=torch.int64)
    out = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        mel = OA.log_mel_spectrogram(torch.from_numpy(audio0), OA.mel_filter())[None]
        t_mel = time.perf_counter() - t0
        for variant, memo in (("reference", False), ("kvcached", True)):
            m = OM.Whisper(w, dims, memo_cross=memo)
            if variant == "reference":
                t0 = time.perf_counter()
                xa = m.encoder.forward(mel)
                t_enc = time.perf_counter() - t0
            m.clear_cache()
            xa5 = xa.repeat_interleave(beam, 0)
            m.decoder.forward_ext(torch.full((beam, 1), 0, dtype=torch.int64), xa5, 0)  # step 0 (fills the caches), not timed
            t0 = time.perf_counter()
            for i in range(1, n_decode_steps + 1):
                logits = m.decoder.forward_ext(toks, xa5, i)
                logits[:, -1].log_softmax(-1).topk(beam + 1)
            out[variant] = (time.perf_counter() - t0) / n_decode_steps
            del m
    total = t_mel + t_enc + sample_len * out["reference"]
    total_kv = t_mel + t_enc + sample_len * out["kvcached"]
    return {"value": 30.0 / total, "unit": "xRT (audio-sec/wall-sec)", "cores": threads, "kind": "port",
            "inputs": "the benchmarked model's weights (copied from the device), window 0 of rank 0, the device's first token",
            "sample": f"1 window: log-mel ({t_mel:.2f} s) + encoder ({t_enc:.2f} s) + {n_decode_steps} of {sample_len} beam-{beam} decode steps "
                      f"({out['reference']:.2f} s/step, cross-K/V recomputed per step as the reference does) extrapolated linearly; fp32 ATen CPU",
            "kvcached": {"value": 30.0 / total_kv, "unit": "xRT (audio-sec/wall-sec)",
                         "sample": f"same sample with the cross-attention K/V computed once per window ({out['kvcached']:.3f} s/step)"}}


def window_seed(rank, batch=1000):
    """BASELINE config 5: one long synthetic recording cut at the fixed 30-s stride; window w is generated from seed
    1234 + w wherever it is needed.  Rank r of a job owns windows r*batch .. (r+1)*batch - 1 (independent units, no
    data-path collective): rank_windows(r, batch) is exactly those windows."""
    return 1234 + rank * batch


def rank_windows(rank, batch, n_samples=480000):
    from whisper_rs_amd import synth
    return synth.make_audio(batch, seed=window_seed(rank, batch), n_samples=n_samples, index0=rank * batch)


def deal_windows(n_windows, world, batch):
    """window indices per rank: consecutive batches of `batch` windows dealt round-robin (batch b -> rank b % world)."""
    per_rank = [[] for _ in range(world)]
    for b0 in range(0, n_windows, batch):
        per_rank[(b0 // batch) % world] += list(range(b0, min(n_windows, b0 + batch)))
    return per_rank


def gather_transcript(local_windows, local_tokens, n_windows, world, device, max_len=448):
    """results return by WINDOW INDEX and are concatenated on the host in window order (SURVEY.md section 8(e);
    reference concatenation transcribe.rs:231-236).  local_tokens[i] = token ids of window local_windows[i].  Every rank
    contributes an int32 block of (index, length, ids) rows (all_gather: the transcript is a few KB per window, this is
    result collection, not a data-path collective); returns the per-window token arrays in window order on every rank."""
    import torch.distributed as dist
    use_dist = dist.is_available() and dist.is_initialized()
    dev = device if device is not None else "cpu"
    cap = torch.tensor([len(local_windows)], dtype=torch.int64, device=dev)
    if use_dist:
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    blk = torch.full((max(int(cap.item()), 1), max_len + 2), -1, dtype=torch.int32)
    for i, (w, t) in enumerate(zip(local_windows, local_tokens)):
        t = np.asarray(t, dtype=np.int64)[:max_len]
        blk[i, 0], blk[i, 1] = int(w), len(t)
        blk[i, 2:2 + len(t)] = torch.from_numpy(t.astype(np.int32))
    if use_dist:
        blk = blk.to(dev)
        parts = [torch.empty_like(blk) for _ in range(world)]
        dist.all_gather(parts, blk)
        parts = [p.cpu() for p in parts]
    else:
        parts = [blk]
    out = [None] * n_windows
    for p in parts:
        for row in p:
            w = int(row[0])
            if w >= 0:
                assert 0 <= w < n_windows and out[w] is None, f"window {w} returned twice or out of range"
                out[w] = row[2:2 + int(row[1])].numpy().astype(np.int64)
    assert all(o is not None for o in out), "a window is missing from the gathered transcript"
    return out


def max_over_ranks(elapsed, world, device):
    """timed-region length of the job = the slowest rank's (all_reduce MAX; the only collective in the bench)."""
    import torch.distributed as dist
    if world <= 1 and not (dist.is_available() and dist.is_initialized()):
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_xrt(steps, batch, world, elapsed):
    """audio-seconds decoded by ALL ranks in the timed region / wall seconds (weak scaling: batch is per rank)."""
    return steps * batch * 30.0 * world / elapsed


def bench_lanes(args, W, synth, models, ids, dims, dev, rank, world, use_dist):
    """--lanes L: L model replicas on this GPU behind one window pool (worker thread per replica, overlapped mode each)."""
    import torch.distributed as dist
    opts = W.DecodeOptions(sample_len=args.sample_len, max_audio=args.batch, token_extract_mode=W.TokenExtractMode.BeamSearch(args.beam, 1.0))
    pool = W.WindowPool(models, ids, opts, overlapped=not args.no_overlap)
    n_win = args.lanes * args.batch
    audio = torch.from_numpy(rank_windows(rank, n_win)).to(dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pool.run_audio(audio.data_ptr(), n_win)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pool.run_audio(audio.data_ptr(), n_win)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    if rank == 0:
        print(json.dumps({
            "metric": "xRT (audio-sec/wall-sec), large-v2 beam=5 fp16" if args.model == "large-v2" else f"xRT (audio-sec/wall-sec), {args.model} beam={args.beam} fp16",
            "value": round(whole_job_xrt(args.steps, n_win, world, elapsed), 2), "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.model} beam={args.beam}, {args.lanes} lanes x {args.batch} independent synthetic 30-s windows per GPU per step "
                                   f"(window pool: one model replica + worker thread per lane), sample_len {args.sample_len}",
                       "windows_per_step_per_gpu": n_win, "lanes": args.lanes, "beam": args.beam, "sample_len": args.sample_len,
                       "sharding": f"windows x{world}, no collective"},
            "roofline": None}), flush=True)
    pool.close()
    if use_dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large-v2")
    ap.add_argument("--batch", type=int, default=307, help="30-s windows per step per GPU (independent units; 307 x beam 5 = 1535 decoder rows = 12 row tiles of 128: the "
                    "largest batch whose decode GEMMs still fit one round of workgroups — 480 MLP-1 tiles on 512 slots; 320 windows take two rounds and lose 5 %)")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--sample-len", type=int, default=224)
    ap.add_argument("--lanes", type=int, default=1, help="model replicas per GPU, each decoding its own batches from its own worker thread (wrs_window_pool); "
                    "--batch is then windows per lane and a step is lanes x batch windows.  Not the default: per-launch timing of a kernel "
                    "is not meaningful while several lanes share the GPU, so no roofline object is reported")
    ap.add_argument("--no-overlap", action="store_true", help="sequential passes: do not queue the next step's mel/encoder/cross-K/V on the side stream while this step decodes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-lib", action="store_true", help="load libwrs_mi355x_debug.so (honours the WRS_* experiment switches) instead of the product library")
    ap.add_argument("--no-shard-check", action="store_true", help="N > 1: skip rank 0's re-decode of another rank's windows (one extra pass after the timed region)")
    ap.add_argument("--stage-profile", action="store_true", help="extra eager pass with per-kernel HIP-event timing (stderr)")
    ap.add_argument("--cross-stride", type=int, default=8, help="sample every n-th layer's cross-attention launch for the roofline")
    args = ap.parse_args()
    # the benchmark runs the PRODUCT library (no test hooks, no experiment switches); --debug-lib loads the test build, whose
    # environment switches select the measured-and-not-kept variants (A/B runs of scripts/r03_batch*.sh)
    os.environ["WRS_DEBUG_LIB"] = "1" if args.debug_lib else "0"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("WRS_BENCH_FORCE_DIST") == "1"
    if use_dist:  # launched by torch.distributed.run: one rank per GPU, RCCL only for barrier + MAX
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a banner ("Hostname : ...", "Librccl path : ...") on STDOUT when its communicator is created (at init or
        # at the first collective): keep stdout to the one JSON line by pointing fd 1 at stderr until that has happened
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import whisper_rs_amd as W
    from whisper_rs_amd import synth
    dims = W.MODEL_DIMS[args.model]
    ids = W.TokenIds.for_vocab(dims.n_vocab)
    weights = make_weights_device(dims, dev, seed=0)
    torch.cuda.synchronize()  # wrs_model_set_tensor reads device pointers on the library's own stream: the producer must be done
    models = []
    for _ in range(max(1, args.lanes)):
        mdl = W.Whisper(dims, device=local_rank)
        for name, t in weights.items():
            mdl.set_tensor(name, t)
        mdl.finalize()
        models.append(mdl)
    model = models[0]
    want_cpu = (not args.no_cpu_baseline) and world == 1 and args.lanes == 1
    if not want_cpu:
        del weights
    torch.cuda.empty_cache()
    if args.lanes > 1:
        return bench_lanes(args, W, synth, models, ids, dims, dev, rank, world, use_dist)
    task = W.DecodeTask(model, ids, W.DecodeOptions(sample_len=args.sample_len, max_audio=args.batch,
                                                    token_extract_mode=W.TokenExtractMode.BeamSearch(args.beam, 1.0)))
    # synthetic 30-s windows, distinct per rank, resident in HBM before the timed region
    # this rank's windows of the long recording (window w <- seed 1234 + w), resident in HBM before the timed region
    audio_host = rank_windows(rank, args.batch)
    audio = torch.from_numpy(audio_host).to(dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    overlap = not args.no_overlap
    if overlap:
        # Overlapped passes: while step i decodes (HBM-bound), step i+1's log-mel + encoder + cross-K/V projection
        # (MFMA-bound) run on the library's side stream.  Every timed step still performs exactly one encoder pass and
        # one full decode of `batch` windows: the encoder of the first timed step is queued before the timed region,
        # the one queued by the last timed step finishes inside it (the closing barrier synchronises the device).
        task.encode_async(audio.data_ptr(), args.batch)

        def one_step():
            task.encode_async(audio.data_ptr(), args.batch)
            return task.run_pending()
    else:
        def one_step():
            return task.run_audio_device(audio.data_ptr(), args.batch)

    for _ in range(args.warmup):
        outs = one_step()
    task.set_cross_timing(True, args.cross_stride)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    cross = task.cross_timing()
    split = task.last_run_times()
    overlap_drained = False
    task.set_cross_timing(False, args.cross_stride)
    n_steps_decoded = task.n_steps
    elapsed_local = elapsed
    elapsed = max_over_ranks(elapsed, world, dev)
    # per-rank pass times (for the scaling curve) and the transcript of the recording in window order: every rank's
    # results go to rank 0 BY WINDOW INDEX (no data-path collective: a few KB per window after the timed region)
    per_rank_s = [elapsed_local]
    if use_dist:
        t = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        per_rank_s = [float(x.item()) for x in parts]
    n_windows = world * args.batch
    my_windows = list(range(rank * args.batch, (rank + 1) * args.batch))
    transcript = gather_transcript(my_windows, [o.tokens for o in outs], n_windows, world, dev if use_dist else None)
    shard_check = None
    if world > 1 and rank == 0 and not args.no_shard_check:
        # "equal to a 1-rank run": rank 0 re-decodes the LAST rank's windows itself and compares the token ids
        other = world - 1
        a2 = torch.from_numpy(rank_windows(other, args.batch)).to(dev)
        torch.cuda.synchronize()
        if overlap:
            task.run_pending()  # drain the batch queued by the last timed step
            overlap_drained = True
        again = task.run_audio_device(a2.data_ptr(), args.batch)
        shard_check = all(np.array_equal(o.tokens, transcript[other * args.batch + j]) for j, o in enumerate(again))
        assert shard_check, "a window decoded by another rank differs from rank 0's own decode of it"
        del a2

    if overlap and not overlap_drained:
        task.run_pending()  # drain the batch queued by the last timed step (outside the timed region)
    if args.stage_profile and rank == 0:
        task.set_profiling(True)
        one_step()
        st = task.stage_times()
        task.set_profiling(False)
        tot = sum(s["ms"] for s in st)
        for s in sorted(st, key=lambda s: -s["ms"]):
            gbs = s["bytes"] / s["ms"] / 1e6 if s["ms"] > 0 and s["bytes"] else 0
            tfs = s["flops"] / s["ms"] / 1e9 if s["ms"] > 0 and s["flops"] else 0
            print(f"[stage] {s['name']:22s} {s['ms']:9.2f} ms {100 * s['ms'] / tot:5.1f}% launches {s['launches']:6d} "
                  f"avg {1e3 * s['ms'] / max(1, s['launches']):8.1f} us  {gbs:8.1f} GB/s {tfs:8.1f} TFLOP/s", file=sys.stderr)

    if rank == 0:
        value = whole_job_xrt(args.steps, args.batch, world, elapsed)
        # roofline of the dominant kernel: decode cross-attention streams every audio's packed K and V once
        # per launch: algorithmic bytes = batch * 2 (K,V) * 1500 * D * 2 B (KV_cross / L)
        rl = None
        if cross["launches"] > 0:
            avg_ms = cross["ms"] / cross["launches"]
            bytes_per_launch = cross["bytes"] / cross["launches"]
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            # HBM traffic per launch from the PMC pass committed under profiles/ (FETCH_SIZE, doubled as the gfx950 guide
            # prescribes for wide coalesced reads; WRITE_SIZE is negligible for this kernel), scaled to this batch
            traffic = None
            try:
                pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))["cross_attention_decode"]
                traffic = pmc["fetch_bytes_corrected_per_launch"] * args.batch / pmc["batch"] if args.model == "large-v2" else None
            except Exception:
                pass
            e2e, dec_bytes, enc_flops = end_to_end_roofline(dims, args.batch, args.beam, n_steps_decoded)
            rl = {"bound": "hbm", "kernel": "attn_mfma_kernel<true> (decode cross-attention)", "achieved": round(achieved, 1),
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                  "traffic_source": f"{PMC_FILE}: rocprofv3 --pmc FETCH_SIZE pass of this workload (x2 gfx950 correction), scaled to this batch; NOT a live counter",
                  "avg_launch_us": round(avg_ms * 1e3, 2), "launches_sampled": cross["launches"],
                  "bytes_per_launch": bytes_per_launch,
                  # whole path against the SURVEY section 8(d) formula at this run's windows / beam / steps, per GPU
                  "end_to_end": {"xrt_per_gpu": round(value / world, 2),
                                 "xrt_roofline": round(e2e["achievable_6.3TBs"], 1), "frac": round(value / world / e2e["achievable_6.3TBs"], 4),
                                 "xrt_roofline_hbm_spec": round(e2e["spec_8TBs"], 1), "frac_hbm_spec": round(value / world / e2e["spec_8TBs"], 4),
                                 "formula": "sum over steps of (W_step + A*KV_cross + rows*l*KV_tok + rows*V*4) / HBM rate + (encoder + cross-K/V projection FLOP) / 2.5 PFLOP/s",
                                 "decode_bytes_per_pass": dec_bytes, "mfma_flops_per_pass": enc_flops}}
        line = {
            "metric": "xRT (audio-sec/wall-sec), large-v2 beam=5 fp16" if args.model == "large-v2" else f"xRT (audio-sec/wall-sec), {args.model} beam={args.beam} fp16",
            "value": round(value, 2), "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.model} beam={args.beam} patience=1 timestamps+suppress filters, {args.batch} independent "
                                   f"synthetic 30-s windows per GPU per step, sample_len {args.sample_len} ({n_steps_decoded} decode steps taken), "
                                   f"mel+encoder+decoder+beam search end to end, random-init weights",
                       "windows_per_step_per_gpu": args.batch, "beam": args.beam, "sample_len": args.sample_len, "sharding": f"windows x{world}, no collective",
                       "overlap": "next step's encoder on a side stream during this step's decode" if overlap else "none"},
            "roofline": rl,
            "last_step_split_ms": {k: round(v, 2) for k, v in split.items()},
            "per_rank_xrt": [round(args.steps * args.batch * 30.0 / t, 2) for t in per_rank_s],
            "transcript": {"windows": n_windows, "tokens": int(sum(len(t) for t in transcript)), "gathered_by": "window index, rank 0",
                           "other_rank_windows_equal_rank0_decode": shard_check},
            "hbm_free_gb_after_run": round(torch.cuda.mem_get_info(dev)[0] / 1e9, 1),
            "library": os.path.basename(W._lib.LIB_PATH),
        }
        if want_cpu:
            line["cpu_baseline"] = cpu_baseline(dims, weights, audio_host[0], outs[0].tokens, args.beam, args.sample_len)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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


def cpu_baseline(dims, beam, sample_len, n_decode_steps=2):
    """CPU restatement of the reference (oracle/, fp32 ATen CPU ops = what `tch` binds), timed on this
    host on a bounded sample: ONE window through the encoder plus `n_decode_steps` single-token decode
    steps for one audio x `beam` rows with the reference's per-step cross-K/V recomputation
    (model.rs:79), extrapolated linearly to `sample_len` steps.  Reported, not the target."""
    from oracle import model as OM
    from whisper_rs_amd.synth import weight_manifest
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    w = {}
    for name, shape in weight_manifest(dims):
        fan = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        w[name] = torch.randn(shape) / fan ** 0.5 if len(shape) > 1 else (torch.ones(shape) if name.endswith("weight") else torch.zeros(shape))
    m = OM.Whisper(w, dims, memo_cross=False)
    mel = torch.randn(1, 80, 3000)
    with torch.no_grad():
        t0 = time.perf_counter()
        xa = m.encoder.forward(mel)
        t_enc = time.perf_counter() - t0
        m.clear_cache()
        xa5 = xa.repeat_interleave(beam, 0)
        toks = torch.randint(0, dims.n_vocab, (beam, 1))
        m.decoder.forward_ext(toks, xa5, 0)  # step 0 (fills the caches), not timed
        t0 = time.perf_counter()
        for i in range(1, n_decode_steps + 1):
            logits = m.decoder.forward_ext(toks, xa5, i)
            logits[:, -1].log_softmax(-1).topk(beam + 1)
        t_step = (time.perf_counter() - t0) / n_decode_steps
    total = t_enc + sample_len * t_step
    return {"value": 30.0 / total, "unit": "xRT (audio-sec/wall-sec)", "cores": threads, "kind": "port",
            "sample": f"1 window: encoder ({t_enc:.2f} s) + {n_decode_steps} of {sample_len} beam-{beam} decode steps "
                      f"({t_step:.2f} s/step, cross-K/V recomputed per step as the reference does) extrapolated linearly; fp32 ATen CPU"}


def window_seed(rank):
    """every rank decodes its own, distinct synthetic windows (independent units, no data-path collective)."""
    return 1234 + 1000 * rank


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
    audio = torch.from_numpy(synth.make_audio(n_win, seed=window_seed(rank))).to(dev)
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
    ap.add_argument("--batch", type=int, default=256, help="30-s windows per step per GPU (independent units; 256 x beam 5 = 1280 decoder rows)")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--sample-len", type=int, default=224)
    ap.add_argument("--lanes", type=int, default=1, help="model replicas per GPU, each decoding its own batches from its own worker thread (wrs_window_pool); "
                    "--batch is then windows per lane and a step is lanes x batch windows.  Not the default: per-launch timing of a kernel "
                    "is not meaningful while several lanes share the GPU, so no roofline object is reported")
    ap.add_argument("--no-overlap", action="store_true", help="sequential passes: do not queue the next step's mel/encoder/cross-K/V on the side stream while this step decodes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-profile", action="store_true", help="extra eager pass with per-kernel HIP-event timing (stderr)")
    ap.add_argument("--cross-stride", type=int, default=8, help="sample every n-th layer's cross-attention launch for the roofline")
    args = ap.parse_args()

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
        dist.init_process_group("nccl", device_id=dev)

    import whisper_rs_amd as W
    from whisper_rs_amd import synth
    dims = W.MODEL_DIMS[args.model]
    ids = W.TokenIds.for_vocab(dims.n_vocab)
    weights = make_weights_device(dims, dev, seed=0)
    models = []
    for _ in range(max(1, args.lanes)):
        mdl = W.Whisper(dims, device=local_rank)
        for name, t in weights.items():
            mdl.set_tensor(name, t)
        mdl.finalize()
        models.append(mdl)
    model = models[0]
    del weights
    torch.cuda.empty_cache()
    if args.lanes > 1:
        return bench_lanes(args, W, synth, models, ids, dims, dev, rank, world, use_dist)
    task = W.DecodeTask(model, ids, W.DecodeOptions(sample_len=args.sample_len, max_audio=args.batch,
                                                    token_extract_mode=W.TokenExtractMode.BeamSearch(args.beam, 1.0)))
    # synthetic 30-s windows, distinct per rank, resident in HBM before the timed region
    audio = torch.from_numpy(synth.make_audio(args.batch, seed=window_seed(rank))).to(dev)
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
    if overlap:
        task.run_pending()  # drain the batch queued by the last timed step (outside the timed region)
    task.set_cross_timing(False, args.cross_stride)
    n_steps_decoded = task.n_steps
    elapsed = max_over_ranks(elapsed, world, dev)

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
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_b256.json")))["cross_attention_decode"]
                traffic = pmc["fetch_bytes_corrected_per_launch"] * args.batch / pmc["batch"] if args.model == "large-v2" else None
            except Exception:
                pass
            rl = {"bound": "hbm", "kernel": "attn_mfma_kernel<true> (decode cross-attention)", "achieved": round(achieved, 1),
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                  "avg_launch_us": round(avg_ms * 1e3, 2), "launches_sampled": cross["launches"],
                  "bytes_per_launch": bytes_per_launch}
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
            "hbm_free_gb_after_run": round(torch.cuda.mem_get_info(dev)[0] / 1e9, 1),
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(dims, args.beam, args.sample_len)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

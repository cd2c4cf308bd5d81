#!/usr/bin/env python
"""Headline benchmark: real-time factor (audio-seconds / wall-second) and end-to-end latency of
`TextToSpeech.tts_with_preset(..., preset='standard')` on N MI355X GPUs (BASELINE.json metric).

A "step" is one complete utterance: 256 autoregressive candidates (sharded N/gpus per rank) x M mel
tokens -> CLVP ranking (one all_gather) -> latent re-pass -> 200 diffusion iterations (cond_free) ->
UnivNet.  No checkpoints exist offline, so the four networks are built at the reference's
hyper-parameters (tortoise/api.py:217-236) with seeded synthetic weights; the stop token is
suppressed and the decode length fixed at M (default 200 -> 9.28 s of 24 kHz audio) exactly as
SURVEY.md §8(d) prescribes, because random weights never emit a meaningful end-of-speech.

Launching: `python bench.py --gpus N` is enough - with no WORLD_SIZE in the environment the script starts its N ranks itself
(torch.distributed.run, rendezvous on 127.0.0.1, a free port) and rank 0 prints the line; under an external
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses the ranks it was given.  For N > 1 the line also
carries `replica_rtf` (N utterances, one per GPU, no data-path collective: where N GPUs pay for this pipeline) next to the
sharded-latency `value`, and `rccl_ranks_seen` (ranks that answered an all_gather over the active backend).

One JSON line on rank 0.  Extra legs inside the same command (rank 0, N == 1):
  roofline      one additional un-captured step with every kernel launch bracketed by HIP events on its
                launch stream; reports the dominant kernel class against the gfx950 roofline
  cpu_baseline  the CPU oracle (reference algorithm, fp32) timed on the host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16


def synthetic_weights(seed=1234):
    from tortoise_tts_amd import weights as W
    from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig
    ar_cfg = ARConfig()
    sds = {
        "autoregressive": W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(ar_cfg), seed), ar_cfg),
        "clvp": W.synthetic_state_dict(W.clvp_manifest(CLVPConfig()), seed + 1),
        "diffusion": W.synthetic_state_dict(W.diffusion_manifest(DiffusionConfig()), seed + 2),
        "vocoder": W.fold_weight_norm(W.synthetic_state_dict(W.vocoder_manifest(VocoderConfig()), seed + 3)),
    }
    return sds


def synthetic_prompt(seed=0, text_tokens=54):
    """Seeded stand-in prompt of the benchmark's SHAPE (the full-width goldens of tests/golden/full_*.npz were generated on it:
    oracle/make_golden_full.py).  bench.py itself times bench_prompt()."""
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(1, 255, (text_tokens,), generator=g)  # do_tts.py default sentence -> 54 BPE ids
    auto = torch.randn(1, 1024, generator=g) * 0.5             # stands in for voices/cond_latent_example/pat.pth
    diff = torch.randn(1, 2048, generator=g) * 0.5
    return text, (auto, diff)


def bench_prompt():
    """The prompt SURVEY.md 8(d) prescribes: the default sentence of tortoise/do_tts.py:12 as 54 BPE ids (the reference's tokenizer.json,
    basic cleaners) and the reference's example voice latents voices/cond_latent_example/pat.pth - committed as
    tests/golden/bench_prompt.npz by oracle/make_bench_prompt.py (the GPU box has no /root/reference)."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "bench_prompt.npz"))
    return torch.from_numpy(z["ids"]).long(), (torch.from_numpy(z["auto"]).float(), torch.from_numpy(z["diffusion"]).float())


def cpu_baseline(sds, text, latents, preset_kw, M, cores):
    """Reference algorithm (oracle/, fp32) on the host cores, bounded sample, linear extrapolation.  The thread count is
    swept on the dominant stage (the KV-cached AR step: GEMV-sized work that oversubscribed threads slow down ~4x) and the
    best one is used for every stage; `cores` in the result is the thread count actually used."""
    from oracle import tortoise_oracle as O
    from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig
    import torch.nn.functional as F
    ar_cfg, clvp_cfg, d_cfg, v_cfg = ARConfig(), CLVPConfig(), DiffusionConfig(), VocoderConfig()
    auto, diffc = latents
    tt = F.pad(text.int()[None], (0, 1))
    N, iters = preset_kw["num_autoregressive_samples"], preset_kw["diffusion_iterations"]
    Bc = 16  # reference default AR batch on a >=14 GB device (api.py:156-157)
    AR_STEPS, CLVP_C, PAIRS = 10, 4, 3  # bounded sample: cached AR steps, CLVP candidates, conditioned / conditioning-free denoiser pairs
    with torch.no_grad():
        sd = sds["autoregressive"]
        torch.set_num_threads(min(cores, 16))
        prefix = O.ar_prefix(sd, ar_cfg, auto, tt)
        lg0, kv0 = O.ar_prefill(sd, ar_cfg, prefix, Bc)
        # the cached step is timed at the MEAN context of the decode (prefix + M / 2 generated tokens), not right after the prefill: the
        # per-layer caches are extended with stand-in rows of the same shape (the timing does not depend on their values)
        ctx_extra = max(0, M // 2 - 1)
        g_kv = torch.Generator().manual_seed(9)
        kv0 = [(torch.cat([k, torch.randn(k.shape[0], k.shape[1], ctx_extra, k.shape[3], generator=g_kv) * k.std()], dim=2),
                torch.cat([v, torch.randn(v.shape[0], v.shape[1], ctx_extra, v.shape[3], generator=g_kv) * v.std()], dim=2)) for k, v in kv0]
        tok = torch.zeros(Bc, dtype=torch.long)
        sweep = {}
        for th in sorted({t for t in (8, 16, 32, 64) if t <= cores} | {min(cores, 8)}):
            torch.set_num_threads(th)
            lg, kv = O.ar_step(sd, ar_cfg, tok, ctx_extra + 1, kv0)  # warm the thread pool
            t0 = time.perf_counter()
            for s_ in range(3):
                lg, kv = O.ar_step(sd, ar_cfg, tok, ctx_extra + s_ + 2, kv)
                O.warp_logits(lg, torch.zeros(Bc, 60, dtype=torch.long))
            sweep[th] = (time.perf_counter() - t0) / 3
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        # the sample proper: AR_STEPS cached steps on the best thread count (the sweep above only picks it)
        lg, kv = O.ar_step(sd, ar_cfg, tok, ctx_extra + 1, kv0)
        t0 = time.perf_counter()
        for s_ in range(AR_STEPS):
            lg, kv = O.ar_step(sd, ar_cfg, tok, ctx_extra + s_ + 2, kv)
            O.warp_logits(lg, torch.zeros(Bc, 60, dtype=torch.long))
        sweep[best] = (time.perf_counter() - t0) / AR_STEPS
        t_step = sweep[best]
        t0 = time.perf_counter()
        O.ar_prefill(sd, ar_cfg, prefix, Bc)
        t_pf = time.perf_counter() - t0
        ar_total = (N / Bc) * (t_pf + (M - 1) * t_step)
        codes = torch.randint(0, 8192, (1, M))
        ccodes = torch.randint(0, 8192, (CLVP_C, M))
        t0 = time.perf_counter()
        O.clvp_score(sds["clvp"], clvp_cfg, tt.long(), ccodes)
        clvp_total = (time.perf_counter() - t0) / CLVP_C * N
        t0 = time.perf_counter()
        lat = O.ar_latents(sd, ar_cfg, auto, tt, codes)
        lat_total = time.perf_counter() - t0
        S = M * 4 * 24000 // 22050
        dsd = sds["diffusion"]
        t0 = time.perf_counter()
        emb = O.diffusion_timestep_independent(dsd, d_cfg, lat, diffc, S)
        t_ti = time.perf_counter() - t0
        x = torch.randn(1, 100, S)
        t0 = time.perf_counter()
        for tv in (3900, 2000, 100)[:PAIRS]:
            ts = torch.tensor([tv])
            O.diffusion_forward(dsd, d_cfg, x, ts, emb, False)
            O.diffusion_forward(dsd, d_cfg, x, ts, emb, True)
        t_pair = (time.perf_counter() - t0) / PAIRS
        diff_total = t_ti + iters * (t_pair if preset_kw.get("cond_free", True) else t_pair / 2)
        t0 = time.perf_counter()
        O.univnet_inference(sds["vocoder"], v_cfg, torch.randn(1, 100, S), torch.randn(1, 64, S + 10))
        voc_total = time.perf_counter() - t0
    total = ar_total + clvp_total + lat_total + diff_total + voc_total
    audio_s = S * 256 / 24000.0
    stages_s = {"ar": ar_total, "clvp": clvp_total, "latents": lat_total, "diffusion": diff_total, "vocoder": voc_total}
    # calibration of this port against the reference's OWN classes (oracle/calibrate_cpu_baseline.py, run where /root/reference exists: same
    # bounded sample, same threads): per-stage t_reference / t_oracle; the utterance factor below weights them with THIS run's stage times
    ref_ratio = None
    try:
        import json as _json
        cal = _json.load(open(os.path.join(ROOT, "profiles", "r05_cpu_reference_vs_oracle.json")))
        rr = cal["ratio_reference_over_oracle"]
        ref_total = sum(stages_s[k] * rr[k] for k in stages_s)
        ref_ratio = {"per_stage": {k: round(rr[k], 4) for k in stages_s}, "utterance": round(ref_total / total, 4),
                     "value_reference_equivalent": audio_s / ref_total,
                     "source": "profiles/r05_cpu_reference_vs_oracle.json (reference nn.Modules vs this oracle, %d threads, build container)" % cal["threads"]}
    except Exception as ex:  # the calibration file travels with the repository; without it the line says so
        ref_ratio = {"error": "calibration file unreadable: %s" % ex}
    return {"value": audio_s / total, "unit": "audio-s/wall-s", "cores": best, "host_cores": cores, "kind": "port", "reference_ratio": ref_ratio,
            "what": "the CPU ORACLE (oracle/tortoise_oracle.py: fp32 restatement of the reference algorithm, pinned against the reference's modules), "
                    "not the reference's own classes - /root/reference does not exist on the GPU box; bounded sample, extrapolated",
            "context": "AR step timed at the MEAN decode context (prefix + M / 2 cached tokens); lines of rounds <= 2 timed it right after the prefill",
            "latency_s_extrapolated": total,
            "ar_step_s_by_threads": {str(k): round(v, 4) for k, v in sweep.items()},
            "sample": (f"oracle fp32 on {best} threads (best of {sorted(sweep)} on the AR step): AR prefill + {AR_STEPS} cached steps at the mean decode context (prefix + {ctx_extra + 1} tokens) at B={Bc} "
                       f"({t_pf:.2f}s + {t_step:.3f}s/step), CLVP {CLVP_C} of {N} candidates, 1 latent pass, timestep_independent + {PAIRS} cond/uncond "
                       f"denoiser pairs of {iters} ({t_pair:.2f}s each), full UnivNet ({voc_total:.2f}s); stages extrapolated linearly to the full utterance"),
            "stages_s": stages_s}


def dtype_label(per_stage):
    """'bf16' / 'fp16' when every stage runs the same operand type, else e.g. 'bf16(ar,clvp)+fp16(diffusion,vocoder)'."""
    kinds = sorted(set(per_stage.values()))
    if len(kinds) == 1:
        return kinds[0]
    return "+".join("%s(%s)" % (k, ",".join(n for n, v in per_stage.items() if v == k)) for k in kinds)


def source_digest():
    """sha256 over the engine sources: ties a committed PMC summary to the build it was collected on."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, "tortoise_tts_amd", "csrc")
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(root, f), "rb") as fh:
                h.update(f.encode())
                h.update(fh.read())
    return h.hexdigest()[:16]


def roofline_leg(tts, run_step):
    """One extra step with graph replay off and every launch timed (tt_prof_*; single-kernel launchers by the start / end
    timestamps of the dispatch itself, the clock of a rocprofv3 kernel trace): dominant kernel class."""
    import ctypes as C
    from tortoise_tts_amd import engine as E
    lib = E.load_library()
    lib.tt_graph_replay(0)
    lib.tt_prof_enable(1)
    try:
        run_step()
        torch.cuda.synchronize()
    finally:
        lib.tt_prof_enable(0)
        lib.tt_graph_replay(1)
    rows = []
    buf = (C.c_double * 4)()
    for i in range(lib.tt_prof_classes()):
        lib.tt_prof_read(i, buf)
        if buf[0] > 0:
            rows.append({"kernel": lib.tt_prof_class_name(i).decode(), "launches": int(buf[0]), "total_ms": buf[1],
                         "avg_us": 1e3 * buf[1] / buf[0], "flops": buf[2], "bytes": buf[3]})
    rows.sort(key=lambda r: -r["total_ms"])
    if not rows:
        return None, []
    d = rows[0]
    sec = d["total_ms"] / 1e3
    intensity = d["flops"] / max(d["bytes"], 1.0)
    ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    if intensity >= ridge:
        ach = d["flops"] / sec / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS}
    else:
        ach = d["bytes"] / sec / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    traffic, traffic_src, stale = pmc_traffic(d["kernel"])
    roof.update({"timing": "dispatch start/end timestamps (hipExtLaunchKernelGGL event pair), graph replay off", "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": stale, "kernel": d["kernel"], "launches": d["launches"], "avg_launch_us": d["avg_us"],
                 "algorithmic_flops_per_launch": d["flops"] / d["launches"], "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                 "arithmetic_intensity": intensity, "share_of_kernel_time": d["total_ms"] / sum(r["total_ms"] for r in rows)})
    return roof, rows


def stage_rooflines(stages_s, N, M, iters, cond_free, text_tokens):
    """Per-stage achieved fraction of the roofline: SURVEY.md 8(d)'s algorithmic work of each stage / its measured time.
      AR      HBM: per decode step the trunk + head weights once (386.3e6 parameters, 2 bytes) + every candidate's own K/V rows
              (30 layers x 16 heads x 64 x K,V x 2 bytes = 122 880 bytes per cached token) + the shared prefix K/V once
      CLVP    MFMA: 133 GFLOP per candidate at 500 codes -> scaled by M / 500 rows: 500 * (236e6 + 4 * 500 * 768 * 20) at M = 500
      denoiser MFMA: (249.0e6 * S + 13 * 4 * S^2 * 1024) flops per row; 2 rows per iteration with conditioning-free guidance
      UnivNet HBM: inputs + output + weights + the audio-rate activations (the materialised LVC kernels are not algorithmic)."""
    out = {}
    S = M * 4 * 24000 // 22050
    P1 = 1 + text_tokens + 2 + 1
    ar_bytes = sum(386.3e6 * 2 + (N * (t + 1) + P1) * 122880.0 for t in range(M - 1))
    if stages_s.get("ar_s"):
        ach = ar_bytes / stages_s["ar_s"] / 1e9
        out["ar_decode"] = {"bound": "hbm", "algorithmic_bytes": ar_bytes, "seconds": stages_s["ar_s"], "achieved": ach, "unit": "GB/s",
                            "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS}
    clvp_flops = N * M * (236e6 + 4.0 * M * 768 * 20)
    if stages_s.get("clvp_s"):
        ach = clvp_flops / stages_s["clvp_s"] / 1e12
        out["clvp"] = {"bound": "mfma", "algorithmic_flops": clvp_flops, "seconds": stages_s["clvp_s"], "achieved": ach, "unit": "TFLOP/s",
                       "peak": MFMA_PEAK_TFLOPS, "frac": ach / MFMA_PEAK_TFLOPS}
    d_flops = iters * (2 if cond_free else 1) * (249.0e6 * S + 13 * 4.0 * S * S * 1024)
    if stages_s.get("diffusion_s"):
        ach = d_flops / stages_s["diffusion_s"] / 1e12
        out["denoiser"] = {"bound": "mfma", "algorithmic_flops": d_flops, "seconds": stages_s["diffusion_s"], "achieved": ach, "unit": "TFLOP/s",
                           "peak": MFMA_PEAK_TFLOPS, "frac": ach / MFMA_PEAK_TFLOPS}
    if stages_s.get("vocoder_s"):
        # HBM-bound.  Algorithmic bytes = mel + noise in, audio out, the weights once (KernelPredictor matrices in the
        # operand type), and the unavoidable audio-rate activations 32 ch x (8 + 64 + 256) samples per frame x 4 layers x (read + write);
        # the predicted LVC kernels (3 x 24576 x L x 4 B written and re-read by this engine) are NOT algorithmic: a fused design keeps them on chip.
        L = S + 10
        v_bytes = 100 * L * 4 + 64 * L * 4 + 256 * L * 4 + 14.9e6 * 2 + 32 * (8 + 64 + 256) * L * 4 * (4 * 2)
        v_flops = 45e9 * L / 880.0
        ach = v_bytes / stages_s["vocoder_s"] / 1e9
        out["univnet"] = {"bound": "hbm", "algorithmic_bytes": v_bytes, "algorithmic_flops": v_flops, "seconds": stages_s["vocoder_s"],
                          "achieved": ach, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS,
                          "achieved_tflops_fp32_valu": v_flops / stages_s["vocoder_s"] / 1e12,
                          # the audio-rate layers run on v_mfma_f32_32x32x2_f32 (exact f32 at the f32 VECTOR rate, 157.3 TFLOP/s): what actually bounds
                          # this stage (removing the conv-output round trip made it slower), next to the HBM line SURVEY 8(d) assigns it
                          "frac_f32_matrix": v_flops / stages_s["vocoder_s"] / 157.3e12}
    return out


PMC_SUMMARY = os.path.join("profiles", "r06_pmc_bench.json")


def pmc_traffic(kernel_class):
    """HBM-side bytes per launch of one kernel class from the committed PMC summary (profiles/r06_pmc_bench.json, written by
    scripts/pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over this same workload, aggregated
    per kernel class under the names the engine's profiler uses).  Counters cannot be read from inside a timed run, so the
    bench line carries the figure of the last committed PMC pass, corrected as MI355X_MICROARCH.md prescribes for gfx950:
    both counters are in KiB and FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) reads at 64 bytes, so it is
    doubled.  The summary records the source digest of the build it was collected on; a different digest now => stale."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            pmc = json.load(f)
        row = pmc[kernel_class]
        stale = pmc.get("_meta", {}).get("source_digest") != source_digest()
        return (2.0 * row["FETCH_SIZE"]["avg"] + row["WRITE_SIZE"]["avg"]) * 1024.0, "%s[%s]" % (PMC_SUMMARY, kernel_class), stale
    except (OSError, KeyError, ValueError):
        return None, None, None


def stream_bench(args):
    """tortoise.api_fast.TextToSpeech.tts_stream (api_fast.py:311-420) on the engine: one sampled sequence, the per-step latents the
    decode steps file, HiFi-GAN re-decode of the growing prefix per chunk, cross-fade.  A step = one utterance of M mel tokens
    (EOS suppressed), stream_chunk_size 40 and the 60-token first buffer of the reference.  Reports the wall time until the first
    audio chunk is on the host side of the generator and wall-seconds per audio-second (the two figures README.md:34 quotes)."""
    assert torch.cuda.is_available() and args.gpus == 1, "the streaming path is one sequence on one GPU"
    from tortoise_tts_amd import weights as W
    from tortoise_tts_amd.api_fast import TextToSpeech
    from tortoise_tts_amd.config import ARConfig, HifiganConfig
    a_cfg, h_cfg = ARConfig(), HifiganConfig()
    t_build = time.perf_counter()
    sds = {"autoregressive": W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(a_cfg), 1234), a_cfg),
           "hifidecoder": W.synthetic_state_dict(W.hifigan_manifest(h_cfg), 1238)}
    M = args.mel_tokens
    tts = TextToSpeech(state_dicts=sds, dtype=args.dtype or "bf16", max_mel_tokens=max(M, 64), kv_cache=True)
    t_build = time.perf_counter() - t_build
    text, (auto, _) = bench_prompt()

    def run(i):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first, samples, pieces = None, 0, 0
        for chunk in tts.tts_stream(text, conditioning_latents=(auto,), max_mel_tokens=M, use_deterministic_seed=1000 + i,
                                    stream_chunk_size=40, overlap_wav_len=1024, verbose=False):
            chunk = chunk.cpu()  # the caller plays the piece: it has to be on the host
            if first is None:
                first = time.perf_counter() - t0
            samples += int(chunk.shape[0])
            pieces += 1
        return first, samples, pieces, time.perf_counter() - t0

    for i in range(args.warmup):
        run(i)
    firsts, walls, samples, pieces = [], [], 0, 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        f, samples, pieces, w = run(100 + i)
        firsts.append(f)
        walls.append(w)
    dt = time.perf_counter() - t0
    audio_s = samples / 24000.0
    print(json.dumps({
        "metric": "rtf_stream_api_fast", "value": audio_s * args.steps / dt, "unit": "audio-s/wall-s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype or "bf16", "data": "synthetic",
        "first_chunk_latency_s": sum(firsts) / len(firsts), "first_chunk_latency_s_max": max(firsts),
        "wall_per_audio": dt / (audio_s * args.steps), "pieces_per_step": pieces, "audio_seconds_per_step": audio_s,
        "reference_claim": "README.md:34: 0.25-0.3 wall/audio on a 4 GB GPU, < 500 ms to the first chunk with streaming (other hardware; not a BASELINE number)",
        "config": {"workload": f"api_fast.tts_stream: 1 sequence x {M} mel tokens (EOS suppressed), first buffer 60 tokens then every 40, "
                               f"HiFi-GAN re-decode of the prefix per piece, overlap 1024 samples; {pieces} pieces, {audio_s:.2f} s of 24 kHz audio",
                   "weights": "seeded synthetic at the reference hyper-parameters", "parallelism": "1 GPU"},
        "engine_build_s": t_build}))


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch_command(argv, gpus, port):
    """The command `python bench.py --gpus N` re-executes itself with when nobody launched its ranks (the driver's own form for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, gpus):
    """Start the N ranks and relay rank 0's line (the ranks inherit stdout / stderr); returns the launcher's exit status."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if 0 < have < gpus and env.get("TT_DIST_SHARE_DEVICE") != "1":
        print(f"bench.py: --gpus {gpus} but {have} GPU(s) are visible (TT_DIST_SHARE_DEVICE=1 runs the ranks on shared devices over gloo: "
              f"a control-flow test mode whose number means nothing)", file=sys.stderr)
        return 2
    return subprocess.call(self_launch_command(argv, gpus, free_port()), env=env)


def rank_check():
    """`--rank-check`: rendezvous + THE collective of the path on stand-in data, no engines: every rank contributes 2 scores and 2 x 8 codes
    through dist.gather_candidates, rank 0 prints what it saw.  Runs on CPU ranks (gloo) as well: the launcher test of tests/."""
    from tortoise_tts_amd import dist as tdist
    rank, world, local = tdist.init_from_env_or_exit()
    n_loc, M = 2, 8
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    scores = torch.arange(n_loc, dtype=torch.float32, device=dev) + 10.0 * rank
    codes = (torch.arange(n_loc * M, dtype=torch.int32, device=dev).reshape(n_loc, M) + 100 * rank)
    s_all, c_all = tdist.gather_candidates(scores, codes)
    seen, backend = tdist.ranks_seen()
    ok = bool(s_all.numel() == world * n_loc and all(float(s_all[r * n_loc]) == 10.0 * r and int(c_all[r * n_loc, 0]) == 100 * r for r in range(world)))
    tdist.barrier()
    if rank == 0:
        print(json.dumps({"rank_check": ok, "n_gpus": world, "rccl_ranks_seen": seen, "backend": backend,
                          "all_gather_calls": tdist.COLLECTIVE_CALLS.get("all_gather", 0), "device": dev}))
    tdist.barrier()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="standard")
    ap.add_argument("--mel-tokens", type=int, default=200, help="fixed decode length M (SURVEY.md §8d: 200 and 500)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16"],
                    help="MFMA operand type of EVERY stage; default: the engine's per-stage defaults (bf16 for the autoregressive + CLVP "
                         "stages, fp16 for the diffusion decoder + vocoder, which the reference runs in fp32: api.py resolve_stage_dtypes)")
    ap.add_argument("--workload", default="utterance", choices=["utterance", "read", "stream"],
                    help="utterance: one tts_with_preset call per step (BASELINE metric); read: one long-form paragraph per step = 15 chunks "
                         "spread over the GPUs as replicas (BASELINE config #4, tortoise/read.py); stream: one api_fast.tts_stream "
                         "utterance per step (SURVEY 8f-4; the reference's only published numbers: README.md:34)")
    ap.add_argument("--diffusion-iterations", type=int, default=None,
                    help="override the preset's diffusion iterations (profiling passes only: the headline metric uses the preset's own)")
    ap.add_argument("--utterance-batch", type=int, default=None, help="read workload: chunks per shared decode batch (TextToSpeech(utterance_batch=)); "
                    "default 16; 1 = one chunk after the other like tortoise/read.py")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the decode / sampler loops eagerly instead of replaying hipGraphs "
                    "(counter passes only: rocprofv3 --pmc crashes under graph replay); same kernels, same order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--candidates-per-rank", type=int, default=None,
                    help="one-GPU PREDICTION of the sharded job's per-rank critical path: decode and rank only this many candidates (32 = the "
                         "8-GPU share of 'standard'), then the full tail; the line is labelled as such (metric suffix, config.projection)")
    ap.add_argument("--no-replica", action="store_true", help="N > 1: skip the replica-throughput leg (replica_rtf)")
    ap.add_argument("--rank-check", action="store_true", help="rendezvous + the path's one all_gather on stand-in data, no engines (launcher test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # nobody launched the ranks: do it here
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.rank_check:
        sys.exit(rank_check())

    if args.workload == "stream":
        return stream_bench(args)
    from tortoise_tts_amd import dist as tdist
    rank, world, local = tdist.init_from_env_or_exit()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # (a launch whose collectives could not initialise continues on rank 0 alone: the line then reports n_gpus = 1 and says so)
    fell_back = tdist.FALLBACK_SINGLE and world == 1
    assert world == args.gpus or fell_back, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    from tortoise_tts_amd.api import TextToSpeech
    from tortoise_tts_amd.config import PRESETS, BASE_SETTINGS

    if args.no_graph:
        from tortoise_tts_amd import engine as E
        E.load_library().tt_graph_replay(0)
    preset_kw = dict(BASE_SETTINGS)
    preset_kw.update(PRESETS[args.preset])
    extra_kw = {}
    if args.diffusion_iterations is not None:
        preset_kw["diffusion_iterations"] = args.diffusion_iterations
        extra_kw["diffusion_iterations"] = args.diffusion_iterations
    N = preset_kw["num_autoregressive_samples"]
    if args.candidates_per_rank is not None:
        assert world == 1, "--candidates-per-rank is the ONE-GPU prediction of a sharded job's per-rank work"
        assert N % args.candidates_per_rank == 0, f"{args.candidates_per_rank} candidates per rank do not divide the preset's {N}"
        projected_gpus = N // args.candidates_per_rank
        N = args.candidates_per_rank
        preset_kw["num_autoregressive_samples"] = N
        extra_kw["num_autoregressive_samples"] = N
    M = args.mel_tokens
    t_build = time.perf_counter()
    sds = synthetic_weights()
    text, latents = bench_prompt()
    read_mode = args.workload == "read"
    # read: every rank holds a complete engine with the full candidate batch and renders whole chunks (no candidate sharding)
    ubatch = (args.utterance_batch or 16) if read_mode else 1
    replica_leg = world > 1 and not read_mode and not args.no_replica
    tts = TextToSpeech(state_dicts=sds, dtype=args.dtype, max_candidates=N if (read_mode or replica_leg) else N // world, max_mel_tokens=max(M, 32),
                       candidate_sharding=not read_mode, utterance_batch=ubatch)
    t_build = time.perf_counter() - t_build
    S_audio = M * 4 * 24000 // 22050 * 256 / 24000.0  # seconds of audio per fixed-length utterance (api.py:122, hop 256 @ 24 kHz)
    if read_mode:
        from tortoise_tts_amd.longform import read_long_form
        g = torch.Generator().manual_seed(4)
        # tortoise/data/riding_hood.txt -> 15 chunks of 60-110 text tokens; synthetic ids of those lengths
        chunks = [torch.randint(1, 255, (int(torch.randint(60, 111, (1,), generator=g)),), generator=g) for _ in range(15)]

        def run_step(i=0):
            full, _ = read_long_form(tts, chunks, preset=args.preset, conditioning_latents=latents, seed=1000 + i, texts_are_chunks=True,
                                     max_mel_tokens=M, verbose=False)
            return full
        audio_per_step = len(chunks) * S_audio
    else:
        def run_step(i=0):
            return tts.tts_with_preset(text, preset=args.preset, conditioning_latents=latents, max_mel_tokens=M,
                                       use_deterministic_seed=1000 + i, k=1, verbose=False, **extra_kw)
        audio_per_step = S_audio

    wav = None
    for i in range(args.warmup):
        wav = run_step(i)
    tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stage_acc = {}
    for i in range(args.steps):
        wav = run_step(100 + i)
        for k_, v in tts.timings.items():
            stage_acc[k_] = stage_acc.get(k_, 0.0) + v
    tdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = tdist.max_over_ranks(dt)
    audio_s = audio_per_step
    if rank == 0:  # rendered audio lives on rank 0 only
        assert abs(float(wav.shape[-1]) / 24000.0 - audio_s) < 1e-6, (wav.shape, audio_s)
        assert torch.isfinite(wav).all() and wav.abs().max() <= 1.0

    # N > 1: the same engines as REPLICAS - every rank renders its own complete utterance (all N candidates on its GPU, its own seed), no
    # data-path collective; N utterances per step.  This is where N GPUs pay for this pipeline (one utterance does not hold N GPUs' worth
    # of parallel work after the candidates are sharded: the winner's tail is serial).
    replica = None
    ranks_seen, backend = tdist.ranks_seen() if world > 1 else (1, None)
    if replica_leg:
        tts.set_candidate_sharding(False)
        for i in range(args.warmup):
            tts.tts_with_preset(text, preset=args.preset, conditioning_latents=latents, max_mel_tokens=M, use_deterministic_seed=5000 + 97 * rank + i,
                                k=1, verbose=False, **extra_kw)
        tdist.barrier()
        torch.cuda.synchronize()
        t0r = time.perf_counter()
        for i in range(args.steps):
            rw = tts.tts_with_preset(text, preset=args.preset, conditioning_latents=latents, max_mel_tokens=M,
                                     use_deterministic_seed=6000 + 97 * rank + i, k=1, verbose=False, **extra_kw)
        tdist.barrier()
        torch.cuda.synchronize()
        dtr = tdist.max_over_ranks(time.perf_counter() - t0r)
        assert rw is not None and torch.isfinite(rw).all()
        replica = {"value": world * S_audio * args.steps / dtr, "unit": "audio-s/wall-s", "utterances_per_step": world, "ms_per_step": 1e3 * dtr / args.steps,
                   "what": f"{world} independent utterances per step, one per GPU ({N} candidates each on that GPU), no data-path collective"}
        tts.set_candidate_sharding(True)

    roof, breakdown, cpu = None, [], None
    if not args.no_roofline and not read_mode:
        roof, breakdown = roofline_leg(tts, lambda: run_step(999))  # every rank runs it (the step contains a collective)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not read_mode:
        cores = min(os.cpu_count() or 1, 64)
        cpu = cpu_baseline(sds, text, latents, preset_kw, M, cores)

    stages_mean = {k_: v / args.steps for k_, v in stage_acc.items()}
    if roof is not None and not read_mode:
        roof["stages"] = stage_rooflines(stages_mean, N // world, M, preset_kw["diffusion_iterations"], bool(preset_kw.get("cond_free", True)),
                                         int(text.numel()))
    if rank == 0:
        out = {
            "metric": ("rtf_standard_preset" if not read_mode else "rtf_longform_read") + ("_per_rank_projection" if args.candidates_per_rank else ""),
            "value": audio_s * args.steps / dt, "unit": "audio-s/wall-s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "latency_s": dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dtype_label(tts.dtype_names()), "dtype_per_stage": tts.dtype_names(), "overflow_demotions": list(tts.demotions),
            "data": "synthetic",
            "config": {"workload": ("read.py long-form: 15 chunks per step, each = " if read_mode else "") +
                                   f"tts_with_preset('{args.preset}'): {N} AR candidates x {M} mel tokens (EOS suppressed, fixed length), "
                                   f"CLVP top-1, {preset_kw['diffusion_iterations']} diffusion iterations cond_free={preset_kw.get('cond_free', True)}, "
                                   f"UnivNet; prompt = do_tts.py default sentence (54 BPE ids + pad = 55 text tokens) with the reference's pat.pth "
                                   f"voice latents (tests/golden/bench_prompt.npz); {audio_s:.2f} s of 24 kHz audio per step",
                       "weights": "seeded synthetic at the reference hyper-parameters (no checkpoints offline)",
                       **({"projection": f"ONE GPU doing the per-rank work of a {projected_gpus}-GPU candidate-sharded job: {N} of the preset's "
                                          f"{N * projected_gpus} candidates decoded + ranked, then the winner's full tail (no all_gather, tail not split)"}
                          if args.candidates_per_rank else {}),
                       "parallelism": (f"chunk j on rank j % {world} (replicas, complete pipeline per rank), clips sent to rank 0" if read_mode else
                                       f"candidates sharded {N // world}/GPU, 1 all_gather of scores+codes, "
                                       + ("winner's diffusion tail split over ranks 0/1 (one denoiser row each, 1 exchange per step), vocoder on rank 0"
                                          if tts.split_diffusion else "winner rendered on rank 0"))},
            "stages_s_per_step": stages_mean,
            "collective_fallback": bool(fell_back),
            "rccl_ranks_seen": ranks_seen, "collective_backend": backend,
            "all_gather_calls_per_utterance": (tdist.COLLECTIVE_CALLS.get("all_gather", 0) / max(1, args.steps + args.warmup + (0 if args.no_roofline or read_mode else 1))) if world > 1 else 0,
            "replica_rtf": replica["value"] if replica else None, "replica": replica,
            "audio_seconds_per_step": audio_s, "engine_build_s": t_build,
            "roofline": roof, "cpu_baseline": cpu,
            "kernel_breakdown_ms": [{"kernel": r["kernel"], "launches": r["launches"], "total_ms": round(r["total_ms"], 3),
                                     "avg_us": round(r["avg_us"], 2)} for r in breakdown],
        }
        print(json.dumps(out))
    tdist.barrier()


if __name__ == "__main__":
    main()

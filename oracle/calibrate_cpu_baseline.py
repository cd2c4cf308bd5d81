"""TEST / MEASUREMENT INFRASTRUCTURE ONLY - ties bench.py's `cpu_baseline` (the oracle PORT, the only CPU code that can travel to the GPU
box) to the reference's OWN classes: both are timed here, in the build container (needs /root/reference), on the SAME bounded sample with
the same thread count, stage by stage.

    python -m oracle.calibrate_cpu_baseline [--threads 8]   ->   profiles/r05_cpu_reference_vs_oracle.json

Sample (BASELINE 'standard' shapes, api.py:217-236 hyper-parameters, bench.py's seeded synthetic weights and committed prompt):
  AR      UnifiedVoice.inference_model: prefill of the 57-row prefix at batch 16 (autoregressive_batch_size, api.py:156-157) + 10 KV-cached steps
  CLVP    4 candidates x 200 codes (api.py:463)
  latents one teacher-forced pass, 200 codes (api.py:521-524)
  denoiser timestep_independent + 3 conditioned / conditioning-free pairs at S = 870 (diffusion_decoder.py:262-322)
  UnivNet 870 frames (api.py:559)
The per-stage ratio t_reference / t_oracle, weighted by the oracle's extrapolated stage times, gives the factor bench.py reports as
cpu_baseline.reference_ratio."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
from oracle import make_golden as G  # noqa: E402
from oracle import make_golden_full as GF  # noqa: E402
from oracle import tortoise_oracle as O  # noqa: E402
from tortoise_tts_amd.config import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig  # noqa: E402

AR_B, AR_STEPS, CLVP_B, PAIRS, M = 16, 10, 4, 3, 200


def best_of(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), out


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 8))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_cpu_reference_vs_oracle.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    import bench
    ref = ref_shims.import_reference()
    sds = bench.synthetic_weights()
    text, (auto, diffc) = bench.bench_prompt()
    auto, diffc = auto.float(), diffc.float()
    ar_cfg, clvp_cfg, d_cfg, v_cfg = ARConfig(), CLVPConfig(), DiffusionConfig(), VocoderConfig()
    g = torch.Generator().manual_seed(3)
    toks = torch.randint(0, 8192, (AR_STEPS, AR_B), generator=g)
    res = {}

    # ---- AR: reference GPT2InferenceModel (autoregressive.py:45-147) vs oracle
    m = G.build_ref_ar(ref, ar_cfg, sds["autoregressive"])
    tt = F.pad(text.int()[None], (0, 1))                               # api.py:391: the text as tts() hands it on (T = 55)
    t = F.pad(tt.long(), (0, 1), value=m.stop_text_token)              # autoregressive.py:538-541 (inference_speech pads again)
    t, _ = m.build_aligned_inputs_and_targets(t, m.start_text_token, m.stop_text_token)
    emb = torch.cat([auto.reshape(1, 1, -1), m.text_embedding(t) + m.text_pos_embedding(t)], dim=1)
    m.inference_model.store_mel_emb(emb)
    P = emb.shape[1]

    def ref_prefill():
        ids = torch.full((AR_B, P + 1), 1, dtype=torch.long)
        ids[:, -1] = m.start_mel_token
        out = m.inference_model(input_ids=ids, attention_mask=torch.ones_like(ids), use_cache=True, return_dict=True)
        return ids, out.past_key_values, out.logits[:, -1]
    t_ref_pf, (ids, past, lg_ref0) = best_of(ref_prefill)
    def ref_steps():
        ids_, past_, lg_ = ref_prefill()[:2] + (None,)
        t0 = time.perf_counter()
        for s in range(AR_STEPS):
            ids_ = torch.cat([ids_, toks[s][:, None]], dim=1)
            out = m.inference_model(input_ids=toks[s][:, None], past_key_values=past_, attention_mask=torch.ones_like(ids_), use_cache=True, return_dict=True)
            past_ = out.past_key_values
            lg_ = out.logits[:, -1]
        return (time.perf_counter() - t0) / AR_STEPS, lg_
    t_ref_step, lg_ref = min((ref_steps() for _ in range(2)), key=lambda r: r[0])
    sd = sds["autoregressive"]
    prefix = O.ar_prefix(sd, ar_cfg, auto.reshape(1, -1), tt)
    assert prefix.shape == emb.shape and float((prefix - emb).abs().max()) < 1e-5, "the two sides do not see the same prefix"
    t_or_pf, (lg0, kv) = best_of(lambda: O.ar_prefill(sd, ar_cfg, prefix, AR_B))
    def or_steps():
        lg_, kv_ = O.ar_prefill(sd, ar_cfg, prefix, AR_B)
        t0 = time.perf_counter()
        for s in range(AR_STEPS):
            lg_, kv_ = O.ar_step(sd, ar_cfg, toks[s], s + 1, kv_)
        return (time.perf_counter() - t0) / AR_STEPS, lg_
    t_or_step, lg = min((or_steps() for _ in range(2)), key=lambda r: r[0])
    keep = torch.ones(ar_cfg.number_mel_codes, dtype=torch.bool)
    keep[ar_cfg.stop_mel_token] = False  # (the stop logit is suppressed to -1e9 in the benchmark weights: it would dominate any norm)
    agree = float((lg[:, keep] - lg_ref[:, keep]).norm() / lg_ref[:, keep].norm())
    res["ar_prefill"] = {"reference_s": t_ref_pf, "oracle_s": t_or_pf}
    res["ar_step"] = {"reference_s": t_ref_step, "oracle_s": t_or_step, "steps": AR_STEPS, "batch": AR_B, "rel_l2_logit_diff_after_last_step": agree}
    codes = torch.randint(0, 8192, (1, M), generator=g)
    t_ref_lat, _ = best_of(lambda: m(auto.reshape(1, -1), tt.long(), torch.tensor([tt.shape[-1]]), codes.clone(),
                                     torch.tensor([M * m.mel_length_compression]), return_latent=True, clip_inputs=False))
    t_or_lat, lat = best_of(lambda: O.ar_latents(sd, ar_cfg, auto.reshape(1, -1), tt, codes))
    res["latents"] = {"reference_s": t_ref_lat, "oracle_s": t_or_lat}
    del m

    # ---- CLVP (clvp.py:99-135)
    cm = ref.CLVP(dim_text=clvp_cfg.dim, dim_speech=clvp_cfg.dim, dim_latent=clvp_cfg.dim_latent, num_text_tokens=256, text_enc_depth=clvp_cfg.depth,
                  text_seq_len=350, text_heads=clvp_cfg.heads, num_speech_tokens=8192, speech_enc_depth=clvp_cfg.depth, speech_heads=clvp_cfg.heads,
                  speech_seq_len=430, use_xformers=True).eval()
    cm.load_state_dict(sds["clvp"], strict=True)
    ccodes = torch.randint(0, 8192, (CLVP_B, M), generator=g)
    t_ref_clvp, sc_ref = best_of(lambda: cm(tt.long().repeat(CLVP_B, 1), ccodes, return_loss=False))
    t_or_clvp, sc = best_of(lambda: O.clvp_score(sds["clvp"], clvp_cfg, tt.long(), ccodes))
    res["clvp"] = {"reference_s": t_ref_clvp / CLVP_B, "oracle_s": t_or_clvp / CLVP_B, "candidates": CLVP_B, "max_abs_score_diff": float((sc - sc_ref).abs().max())}
    del cm

    # ---- denoiser (diffusion_decoder.py:262-322)
    dm = GF.build_ref_diffusion(ref, d_cfg, sds["diffusion"])
    S = M * 4 * 24000 // 22050
    lat1 = torch.randn(1, M, 1024, generator=g)
    t_ref_ti, code_emb = best_of(lambda: dm.timestep_independent(lat1, diffc, S, False))
    t_or_ti, emb_o = best_of(lambda: O.diffusion_timestep_independent(sds["diffusion"], d_cfg, lat1, diffc, S))
    x = torch.randn(1, 100, S, generator=g)
    tsl = [torch.tensor([v]) for v in (3900, 2000, 100)][:PAIRS]
    def ref_pairs():
        for ts in tsl:
            dm(x, ts, precomputed_aligned_embeddings=code_emb, conditioning_free=False)
            dm(x, ts, precomputed_aligned_embeddings=code_emb, conditioning_free=True)
    def or_pairs():
        for ts in tsl:
            O.diffusion_forward(sds["diffusion"], d_cfg, x, ts, emb_o, False)
            O.diffusion_forward(sds["diffusion"], d_cfg, x, ts, emb_o, True)
    t_ref_pair = best_of(ref_pairs, 2)[0] / len(tsl)
    t_or_pair = best_of(or_pairs, 2)[0] / len(tsl)
    res["timestep_independent"] = {"reference_s": t_ref_ti, "oracle_s": t_or_ti}
    res["denoiser_pair"] = {"reference_s": t_ref_pair, "oracle_s": t_or_pair, "pairs": len(tsl)}
    del dm

    # ---- UnivNet (vocoder.py:155-216, 284-325)
    from tortoise_tts_amd import weights as W
    raw = W.synthetic_state_dict(W.vocoder_manifest(v_cfg), 1234 + 3)
    vm = ref.UnivNetGenerator()
    vm.load_state_dict(raw, strict=True)
    vm.eval(inference=True)
    mel = torch.randn(1, 100, S, generator=g)
    z = torch.randn(1, 64, S + 10, generator=g)
    t_ref_voc, _ = best_of(lambda: vm.inference(mel, z))
    t_or_voc, _ = best_of(lambda: O.univnet_inference(sds["vocoder"], v_cfg, mel, z))
    res["univnet"] = {"reference_s": t_ref_voc, "oracle_s": t_or_voc}

    # whole utterance ('standard': 256 candidates in batches of 16, 200 tokens, 200 iterations cond_free) from either side's unit times
    def utterance(side):
        k = side + "_s"
        ar = (256 / AR_B) * (res["ar_prefill"][k] + (M - 1) * res["ar_step"][k])
        return {"ar": ar, "clvp": 256 * res["clvp"][k], "latents": res["latents"][k],
                "diffusion": res["timestep_independent"][k] + 200 * res["denoiser_pair"][k], "vocoder": res["univnet"][k]}
    ur, uo = utterance("reference"), utterance("oracle")
    out = {"what": "the reference's own nn.Modules vs the oracle port (bench.py cpu_baseline), same bounded sample, same thread count, build container",
           "threads": args.threads, "host_cores": os.cpu_count(), "torch": torch.__version__,
           "unit_times": res,
           "utterance_extrapolated_s": {"reference": ur, "oracle": uo, "reference_total": sum(ur.values()), "oracle_total": sum(uo.values())},
           "ratio_reference_over_oracle": {**{k: ur[k] / uo[k] for k in ur}, "utterance": sum(ur.values()) / sum(uo.values())},
           "note": "AR steps are timed right after the prefill on both sides (contexts 58 .. 67 keys); bench.py times the oracle at the mean decode context"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["ratio_reference_over_oracle"], indent=1))
    print("wrote", args.out)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""How often does a generation over 2 / 4 concurrent row ranges (tt_ar_set_option) differ from the single-range codes?  Used to bisect
the intermittent difference of round 4 between builds (`TORTOISE_MI355X_LIB=<alt .so>`): write-through vs plain split-K slab stores."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import make_golden as G  # noqa: E402
from tortoise_tts_amd import engine as E, stages, weights as W  # noqa: E402
from tortoise_tts_amd.config import ARConfig  # noqa: E402


def main():
    cfg = ARConfig(**G.AR_CFG)
    sd = W.suppress_stop_token(W.synthetic_state_dict(W.ar_manifest(cfg), seed=G.AR_SEED), cfg)
    B, steps, runs = 64, 40, int(os.environ.get("RUNS", "40"))
    cond, text = G.ar_inputs(cfg)
    st = stages.ArStage(sd, cfg, dtype=E.TT_BF16, max_batch=B, max_text=80, max_new_tokens=48, max_latent_candidates=1)
    st.prefill(cond, text)
    base = st.generate(B, steps, seed=11)[0].clone()
    lib = E.load_library()
    for graphs in (1, 0):
        lib.tt_graph_replay(graphs)
        for nsub in (2, 4):
            st.set_option(E.TT_AR_OPT_SUBBATCHES, nsub)
            for look in (1, 6):
                st.set_option(E.TT_AR_OPT_LOOKAHEAD, look)
                bad = 0
                for rep in range(runs):
                    st.prefill(cond, text)
                    bad += 0 if torch.equal(st.generate(B, steps, seed=11)[0], base) else 1
                print("rate %s graphs=%d ranges=%d lookahead=%d: %d of %d generations differ" % (os.environ.get("AB_TAG", "product"), graphs, nsub, look, bad, runs), flush=True)
    lib.tt_graph_replay(1)
    st.close()


if __name__ == "__main__":
    main()

"""TEST / MEASUREMENT INFRASTRUCTURE - generates tests/golden/bench_prompt.npz, the prompt SURVEY.md 8(d) prescribes for bench.py:
the default sentence of tortoise/do_tts.py:12 through the reference's tokenizer.json with the basic cleaners (inflect / unidecode
are absent), and the reference's example voice latents tortoise/voices/cond_latent_example/pat.pth (a real (f32[1,1024],
f32[1,2048]) pair).  Run in the build container (needs /root/reference); the GPU box only reads the committed fixture.
    python -m oracle.make_bench_prompt"""
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shims  # noqa: E402


def do_tts_default_text():
    src = open(os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "do_tts.py")).read()
    m = re.search(r"'--text'.*?default=\"([^\"]+)\"", src)
    assert m, "do_tts.py: default --text not found"
    return m.group(1)


def main():
    from tortoise_tts_amd.text import VoiceBpeTokenizer
    text = do_tts_default_text()
    vocab = os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "data", "tokenizer.json")
    ids = np.asarray(VoiceBpeTokenizer(vocab, use_basic_cleaners=True).encode(text), dtype=np.int32)
    auto, diff = torch.load(os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "voices", "cond_latent_example", "pat.pth"), map_location="cpu")
    assert ids.shape == (54,) and tuple(auto.shape) == (1, 1024) and tuple(diff.shape) == (1, 2048), (ids.shape, auto.shape, diff.shape)
    out = os.path.join(ROOT, "tests", "golden", "bench_prompt.npz")
    np.savez_compressed(out, text=np.frombuffer(text.encode(), dtype=np.uint8), ids=ids, auto=auto.float().numpy(), diffusion=diff.float().numpy())
    print("wrote", out, "ids", ids[:8], "...", "auto std %.3f diffusion std %.3f" % (float(auto.std()), float(diff.std())))


if __name__ == "__main__":
    main()

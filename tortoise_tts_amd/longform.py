"""Long-form reading (BASELINE.json config #4; reference driver: tortoise/read.py:44-99).

The reference splits the text into chunks (`split_and_recombine_text`, utils/text.py), renders them ONE AFTER THE OTHER with
the same seed (read.py:66-71) and concatenates the clips (read.py:87).  Chunks are independent utterances, so on a multi-GPU
node they are spread over the ranks as replicas: chunk j -> rank j % R, every rank runs the complete pipeline
for its chunks on its own GPU (no candidate sharding, no collective on the data path), rank 0 receives the clips point to
point and concatenates them in chunk order.  The audio of a chunk does not depend on R: same seed, same GPU-local pipeline.
With TextToSpeech(utterance_batch=G) a rank renders its chunks G at a time (tts_many): the candidates of G chunks share one decode
batch - per-chunk codes bit-identical to the sequential order - which is what 288 GB of HBM per GPU are for.
"""
import torch

from . import dist as tdist
from .text import split_and_recombine_text


def chunk_owner(j, world):
    return j % world


def read_long_form(tts, text, preset="standard", conditioning_latents=None, voice_samples=None, seed=None, texts_are_chunks=False,
                   **tts_kwargs):
    """tts: a TextToSpeech built with candidate_sharding=False (one complete engine per rank).
    text: the whole text (str; '|' splits it like read.py:46-50, else split_and_recombine_text) or, with texts_are_chunks=True,
    a list of chunks (str or pre-tokenised id sequences).
    Returns (full_audio f32 [1, n] or None, parts: list of per-chunk clips [1, 1, n_j]) on rank 0, (None, None) elsewhere."""
    if getattr(tts, "world", 1) != 1:
        raise ValueError("read_long_form spreads chunks over the ranks: build TextToSpeech(candidate_sharding=False)")
    if texts_are_chunks:
        texts = list(text)
    elif "|" in text:
        texts = text.split("|")
    else:
        texts = split_and_recombine_text(text)
    rank, world = tdist.world()
    if seed is None:  # read.py:54 uses the wall clock; every rank must agree on it
        import time
        seed = int(time.time())
    seed = tdist.broadcast_int(seed)
    mine = {}
    my_chunks = [j for j in range(len(texts)) if chunk_owner(j, world) == rank]
    if getattr(tts, "utterance_batch", 1) > 1 and len(my_chunks) > 1:
        # this rank's chunks share decode batches (TextToSpeech.tts_many): same seed, same per-chunk codes as one after the other
        from .config import BASE_SETTINGS, PRESETS
        settings = dict(BASE_SETTINGS)
        settings.update(PRESETS[preset])
        settings.update(tts_kwargs)
        settings.pop("k", None)
        wavs = tts.tts_many([texts[j] for j in my_chunks], voice_samples=voice_samples, conditioning_latents=conditioning_latents,
                            use_deterministic_seed=seed, **settings)
        mine = {j: w.cpu() for j, w in zip(my_chunks, wavs)}
        my_chunks = []
    for j in my_chunks:
        gen = tts.tts_with_preset(texts[j], voice_samples=voice_samples, conditioning_latents=conditioning_latents, preset=preset, k=1,
                                  use_deterministic_seed=seed, **tts_kwargs)  # read.py:70-71
        mine[j] = gen.cpu()
    parts = tdist.collect_on_rank0(mine, len(texts))
    if parts is None:
        return None, None
    clips = [parts[j] for j in range(len(texts))]
    full = torch.cat([c.squeeze(0) for c in clips], dim=-1)  # read.py:74, 87
    return full, clips

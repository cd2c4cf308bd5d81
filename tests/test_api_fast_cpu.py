"""CPU checks of the streaming surface (tortoise_tts_amd/api_fast.py): handle_chunks against the reference's own method
(tortoise/api_fast.py:275-309, extracted from the source file because the module itself does not import offline), and the
class keeps the reference's signatures."""
import ast
import inspect
import os

import pytest
import torch

from oracle import ref_shims

REF = os.path.join(ref_shims.REFERENCE_ROOT, "tortoise", "api_fast.py")


def reference_method(name):
    tree = ast.parse(open(REF).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TextToSpeech"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    ns = {"torch": torch, "MODELS_DIR": None}  # (a default-argument name of __init__)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "api_fast_excerpt", "exec"), ns)
    return fn, ns[name]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_handle_chunks_matches_the_reference_method():
    from tortoise_tts_amd.api_fast import TextToSpeech
    _, ref_fn = reference_method("handle_chunks")
    g = torch.Generator().manual_seed(0)
    for overlap in (64, 256):
        prev_r = over_r = prev_m = over_m = None
        length = 0
        for grow in (900, 300, 40, 700, 10):  # growing decodes of the latents so far; one piece shorter than the overlap
            length += grow
            wav = torch.randn(length, generator=g)
            cr, prev_r, over_r = ref_fn(None, wav.clone(), prev_r, over_r, overlap)
            cm, prev_m, over_m = TextToSpeech.handle_chunks(wav.clone(), prev_m, over_m, overlap)
            assert torch.equal(cr, cm)
            assert (over_r is None) == (over_m is None) and (over_r is None or torch.equal(over_r, over_m))
            assert prev_r.shape == prev_m.shape  # only its length is used by the next call (the reference cross-fades it in place)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_signatures_follow_the_reference():
    from tortoise_tts_amd.api_fast import TextToSpeech
    for name in ("tts_stream", "tts"):
        fn, _ = reference_method(name)
        ref_args = [a.arg for a in fn.args.args][1:]
        ours = list(inspect.signature(getattr(TextToSpeech, name)).parameters)[1:]
        for a in ref_args:
            assert a in ours, f"{name}: reference parameter {a!r} missing"
    fn, _ = reference_method("__init__")
    ref_args = [a.arg for a in fn.args.args][1:]
    ours = list(inspect.signature(TextToSpeech.__init__).parameters)[1:]
    assert ours[:len(ref_args)] == ref_args

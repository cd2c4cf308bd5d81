"""TEST INFRASTRUCTURE ONLY — import-only shims so the *real* reference modules under
/root/reference can be imported in the build container (CPU, transformers 5.x, no torchaudio).

Nothing here is shipped or measured.  Only oracle/make_golden.py and the
`not gpu` cross-check tests use it, and only when /root/reference exists (it does not
on the GPU box).

Shimmed imports (all import-only; none of the stubbed symbols is ever called on the hot path):
  * torchaudio(+.transforms,.functional)   <- tortoise/models/arch_util.py:8 (TorchMelSpectrogram only)
  * rotary_embedding_torch                 <- tortoise/models/transformer.py (unused: use_xformers=True, api.py:232)
  * transformers.utils.model_parallel_utils <- tortoise/models/autoregressive.py:8 (dead parallelize())
  * transformers.LogitsWarper (base-class NAME only; 5.x merged it into LogitsProcessor) so that the reference's REAL
    tortoise/utils/typical_sampling.py imports (autoregressive.py:10): TypicalLogitsWarper itself is the reference's code
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TORTOISE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "tortoise", "models"))


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    import transformers  # noqa: F401  (must be imported before the stubs below)

    def _stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    ta = _stub("torchaudio")
    ta.transforms = _stub("torchaudio.transforms")
    ta.functional = _stub("torchaudio.functional")

    class _DummyRotary:  # never instantiated on the shipped config
        def __init__(self, *a, **k):
            raise RuntimeError("stub")

    _stub("rotary_embedding_torch", RotaryEmbedding=_DummyRotary, broadcat=None)
    _stub("transformers.utils.model_parallel_utils",
          get_device_map=lambda *a, **k: None, assert_device_map=lambda *a, **k: None)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not hasattr(transformers, "LogitsWarper"):  # 4.31 had LogitsWarper next to LogitsProcessor (same __call__ protocol)
        transformers.LogitsWarper = transformers.LogitsProcessor
    # imported NOW, while the alias is visible: the first `from transformers import <model class>` makes the package swap the module
    # object registered in sys.modules, and a later `from transformers import LogitsWarper` would not find the alias any more
    import tortoise.utils.typical_sampling  # noqa: F401
    _installed = True


def import_reference():
    """Returns a namespace with the reference classes on the hot path."""
    install()
    ns = types.SimpleNamespace()
    from tortoise.models.autoregressive import UnifiedVoice
    from tortoise.models.diffusion_decoder import DiffusionTts
    from tortoise.models.clvp import CLVP
    from tortoise.models.vocoder import UnivNetGenerator
    from tortoise.utils.diffusion import SpacedDiffusion, space_timesteps, get_named_beta_schedule
    ns.UnifiedVoice = UnifiedVoice
    ns.DiffusionTts = DiffusionTts
    ns.CLVP = CLVP
    ns.UnivNetGenerator = UnivNetGenerator
    ns.SpacedDiffusion = SpacedDiffusion
    ns.space_timesteps = space_timesteps
    ns.get_named_beta_schedule = get_named_beta_schedule
    return ns


def import_stream_generator():
    """tortoise/models/stream_generator.py (the reference's in-tree fork of transformers 4.31's sampling loop) under the installed
    transformers: the names its import line pulls in that 5.x no longer exports (beam-search / constraint classes, the
    SampleOutput alias) are stubbed import-only - sample_stream (stream_generator.py:722-1000) touches none of them."""
    install()
    import transformers
    import transformers.generation.utils as GU
    for n in ("DisjunctiveConstraint", "BeamSearchScorer", "PhrasalConstraint", "ConstrainedBeamSearchScorer"):
        if not hasattr(transformers, n):
            setattr(transformers, n, type(n, (), {}))
    for n in ("GenerateOutput", "SampleOutput"):
        if not hasattr(GU, n):
            setattr(GU, n, object)
    import tortoise.models.stream_generator as SG
    return SG


def enable_generate(unified_voice):
    """Give the reference's GPT2InferenceModel back the `generate` it had under transformers 4.31 by mixing the
    INSTALLED GenerationMixin into a test-only subclass (5.x dropped the mixin from PreTrainedModel).

    One compatibility fix-up, nothing else: 5.x hands an empty Cache object to the first decoding step where 4.31
    passed `past_key_values=None`, and the reference's `if past_key_values:` (autoregressive.py:83, 93) would then
    feed only the last token to the prefill.  The subclass makes that first call see "no past", exactly as 4.31 did.
    Used by tests/test_oracle_vs_reference.py and oracle/make_golden.py to pin oracle.ar_sample_loop against a real
    HF `generate(do_sample=True, ...)` run of the reference model."""
    from transformers import GenerationConfig, GenerationMixin
    im = unified_voice.inference_model

    class _GenerativeInferenceModel(type(im), GenerationMixin):
        def prepare_inputs_for_generation(self, input_ids, past_key_values=None, **kwargs):
            empty = (past_key_values is not None and hasattr(past_key_values, "get_seq_length")
                     and past_key_values.get_seq_length() == 0)
            out = super().prepare_inputs_for_generation(input_ids, past_key_values=None if empty else past_key_values, **kwargs)
            if empty and self.kv_cache:
                out["past_key_values"] = past_key_values  # the (empty) cache object the model fills in
            return out

    im.__class__ = _GenerativeInferenceModel
    im.generation_config = GenerationConfig.from_model_config(im.config)
    return unified_voice

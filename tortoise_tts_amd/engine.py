"""ctypes binding of include/tortoise_mi355x.h (the drop-in boundary) and include/tortoise_mi355x_test.h (operator-level test entries).

The library is the product: there is no PyTorch/CPU fallback.  If the shared object is missing
or the device is not gfx950, loading fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TORTOISE_MI355X_LIB") or os.path.join(HERE, "lib", "libtortoise_mi355x.so")  # env: an alternative build of the same ABI

TT_BF16, TT_F16, TT_F32 = 0, 1, 2  # TT_F32: the slow fp32-operand VERIFICATION mode of the AR / CLVP / diffusion / vocoder stages (tests)
DTYPE_NAMES = {TT_BF16: "bf16", TT_F16: "fp16", TT_F32: "fp32"}
TT_AR_OPT_LOOKAHEAD = 4
TT_DIFF_OPT_OVERLAP_PREPASS = 1
TT_DIFF_OPT_FUSED_GN = 2
TTX_FLASH32, TTX_GEMM_P8, TTX_VOC_MFMA, TTX_GEMM_SKINNY, TTX_AR_GEMV = 0, 1, 2, 3, 4  # ttx_kernel_variant families (include/tortoise_mi355x_test.h)


def dtype_code(name):
    """'bf16' | 'fp16' | 'f16' (or an engine code) -> engine dtype code."""
    if isinstance(name, int):
        return name
    try:
        return {"bf16": TT_BF16, "fp16": TT_F16, "f16": TT_F16, "fp32": TT_F32, "f32": TT_F32}[name]
    except KeyError:
        raise ValueError(f"unknown operand type {name!r} (bf16 | fp16; fp32 = the slow verification mode)") from None
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3, 4, 5

vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p  # device pointers are passed as integers


class EngineError(RuntimeError):
    pass


class GptLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_proj", "b_proj", "ln2_g", "ln2_b",
                                  "w_fc", "b_fc", "w_proj2", "b_proj2")]


class ArConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "layers", "model_dim", "heads", "vocab", "start_mel_token", "stop_mel_token",
                                       "mel_pos_len", "max_batch", "max_prefix", "max_new_tokens", "max_full_rows",
                                       "mel_pos_offset", "max_groups")]


class ArWeights(C.Structure):
    _fields_ = [("layers_host", C.POINTER(GptLayer))] + [(n, vp) for n in (
        "lnf_g", "lnf_b", "final_norm_g", "final_norm_b", "w_mel_head", "b_mel_head", "mel_emb", "mel_pos")]


class Sampling(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("repetition_penalty", C.c_float), ("top_k", C.c_int),
                ("seed", C.c_ulonglong), ("row_offset", C.c_int), ("exp_noise", vp), ("group_seeds", C.POINTER(C.c_ulonglong)),
                ("typical_mass", C.c_float)]


class ClvpLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("attn_norm_g", "w_qkv", "w_out", "b_out", "ff_norm_g", "w_ff1", "b_ff1", "w_ff2", "b_ff2")]


class ClvpTower(C.Structure):
    _fields_ = [("layers_host", C.POINTER(ClvpLayer))] + [(n, vp) for n in ("emb", "inv_freq", "norm_g", "norm_b", "w_latent")]


class ClvpConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "dim", "latent_dim", "depth", "heads", "ff_inner", "rot_dim", "max_rows")]


class AttnBlock(C.Structure):
    _fields_ = [(n, vp) for n in ("norm_g", "norm_b", "w_qkv", "b_qkv", "w_proj", "b_proj", "relpos")]


class ResBlock(C.Structure):
    _fields_ = [(n, vp) for n in ("gn1_g", "gn1_b", "w_in", "b_in", "gn2_g", "gn2_b", "w_out", "b_out")]


class DiffConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "channels", "heads", "num_layers", "in_channels", "in_pad", "out_channels",
                                       "latent_channels", "max_seq", "max_codes", "max_steps", "max_batch")]


class DiffWeights(C.Structure):
    _fields_ = [("w_latent_conv", vp), ("b_latent_conv", vp), ("latent_attn_host", C.POINTER(AttnBlock)),
                ("code_norm_g", vp), ("code_norm_b", vp), ("uncond_emb", vp),
                ("w_time1", vp), ("b_time1", vp), ("w_time2", vp), ("b_time2", vp), ("w_emb_all", vp), ("b_emb_all", vp),
                ("res_host", C.POINTER(ResBlock)), ("attn_host", C.POINTER(AttnBlock)),
                ("w_inp", vp), ("b_inp", vp), ("w_integ", vp), ("b_integ", vp), ("out_gn_g", vp), ("out_gn_b", vp),
                ("w_final", vp), ("b_final", vp)]


class DiffStep(C.Structure):
    _fields_ = [("timestep", C.c_int)] + [(n, C.c_float) for n in (
        "min_log", "max_log", "cfk", "sqrt_recip", "sqrt_recipm1", "coef1", "coef2", "nonzero")]


class VocBlock(C.Structure):
    _fields_ = [("w_convt", vp), ("b_convt", vp), ("w_kp_in", vp), ("b_kp_in", vp),
                ("w_kp_res", vp * 6), ("b_kp_res", vp * 6), ("w_kp_kernel", vp), ("b_kp_kernel", vp),
                ("w_kp_bias", vp), ("b_kp_bias", vp), ("w_conv", vp * 4), ("b_conv", vp * 4), ("stride", C.c_int)]


class VocConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "max_frames", "mel_channels", "mel_pad")]


class VocWeights(C.Structure):
    _fields_ = [("w_pre", vp), ("b_pre", vp), ("blocks_host", C.POINTER(VocBlock)), ("w_post", vp), ("b_post", vp)]


class CondConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "ar_dim", "ar_heads", "ar_blocks", "ar_mel", "ar_mel_pad", "diff_channels", "diff_heads",
                                       "diff_blocks", "diff_mel", "diff_mel_pad", "max_frames")]


class CondWeights(C.Structure):
    _fields_ = [("ar_w_init", vp), ("ar_b_init", vp), ("ar_attn_host", C.POINTER(AttnBlock)),
                ("diff_w_c0", vp), ("diff_b_c0", vp), ("diff_w_c1", vp), ("diff_b_c1", vp), ("diff_attn_host", C.POINTER(AttnBlock))]


class CvvpTower(C.Structure):
    _fields_ = [("layers_host", C.POINTER(ClvpLayer)), ("inv_freq", vp), ("norm_g", vp), ("norm_b", vp), ("w_pre0", vp), ("b_pre0", vp),
                ("attn", AttnBlock), ("w_pre2", vp), ("b_pre2", vp), ("w_latent", vp)]


class CvvpConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "dim", "heads", "depth", "rot_dim", "mel_channels", "mel_pad", "max_rows", "max_cond_frames")]


class CvvpWeights(C.Structure):
    _fields_ = [("cond", CvvpTower), ("speech", CvvpTower), ("w_cond0", vp), ("b_cond0", vp), ("w_cond1", vp), ("b_cond1", vp),
                ("speech_emb", vp), ("temperature", vp)]


HIFI_MAX_STAGES = 6


class HifiResBlock(C.Structure):
    _fields_ = [("w1", vp * 3), ("b1", vp * 3), ("w2", vp * 3), ("b2", vp * 3)]


class HifiConfig(C.Structure):
    _fields_ = [("dtype", C.c_int), ("in_channels", C.c_int), ("cond_channels", C.c_int), ("initial_channel", C.c_int),
                ("num_stages", C.c_int), ("up_factor", C.c_int * HIFI_MAX_STAGES), ("num_kernels", C.c_int), ("kernel_size", C.c_int * 3),
                ("num_dilations", C.c_int), ("dilation", C.c_int * 3), ("lrelu_slope", C.c_float), ("max_latents", C.c_int)]


class HifiWeights(C.Structure):
    _fields_ = [("w_pre", vp), ("b_pre", vp), ("w_cond", vp), ("b_cond", vp), ("w_up", vp * HIFI_MAX_STAGES), ("b_up", vp * HIFI_MAX_STAGES),
                ("res_host", C.POINTER(HifiResBlock)), ("w_post", vp), ("b_post", vp)]


# order == tt_struct_size(which)
BOUNDARY_STRUCTS = [GptLayer, ArConfig, ArWeights, Sampling, ClvpLayer, ClvpTower, ClvpConfig, AttnBlock, ResBlock, DiffConfig,
                    DiffWeights, DiffStep, VocBlock, VocConfig, VocWeights, CondConfig, CondWeights, HifiResBlock, HifiConfig, HifiWeights,
                    CvvpTower, CvvpConfig, CvvpWeights]

_i, _f, _sz = C.c_int, C.c_float, C.c_size_t
_PROTOS = {
    "tt_last_error": (C.c_char_p, []),
    "tt_init": (_i, []),
    "tt_abi_version": (_i, []),
    "tt_struct_size": (_sz, [_i]),
    "tt_ar_create": (_i, [C.POINTER(ArConfig), C.POINTER(ArWeights), C.POINTER(vp)]),
    "tt_ar_destroy": (None, [vp]),
    "tt_ar_prefill": (_i, [vp, vp, _i, vp]),
    "tt_ar_prefill_group": (_i, [vp, _i, _i, vp, _i, vp]),
    "tt_ar_get_logits": (_i, [vp, vp, _i, vp]),
    "tt_ar_generate": (_i, [vp, _i, _i, C.POINTER(Sampling), vp, C.POINTER(_i), vp]),
    "tt_ar_generate_chunk": (_i, [vp, _i, _i, _i, _i, C.POINTER(Sampling), vp, C.POINTER(_i), C.POINTER(_i), vp]),
    "tt_ar_begin": (_i, [vp, _i, vp]),
    "tt_ar_decode_step": (_i, [vp, vp, vp]),
    "tt_ar_latents": (_i, [vp, vp, _i, _i, vp, vp]),
    "tt_ar_stream_latents": (_i, [vp, _i, _i, vp, vp]),
    "tt_ar_set_option": (_i, [vp, _i, _i]),
    "tt_ar_guard": (_i, [vp, _i]),
    "tt_ar_stat": (_i, [vp, _i]),
    "tt_diff_stat": (_i, [vp, _i]),
    "tt_diff_set_option": (_i, [vp, _i, _i]),
    "tt_clvp_guard": (_i, [vp, _i]),
    "tt_diff_guard": (_i, [vp, _i]),
    "tt_clvp_create": (_i, [C.POINTER(ClvpConfig), C.POINTER(ClvpTower), C.POINTER(ClvpTower), vp, C.POINTER(vp)]),
    "tt_clvp_destroy": (None, [vp]),
    "tt_clvp_score": (_i, [vp, vp, _i, vp, _i, _i, vp, vp]),
    "tt_clvp_score_groups": (_i, [vp, vp, C.POINTER(_i), _i, vp, _i, _i, vp, vp]),
    "tt_diff_create": (_i, [C.POINTER(DiffConfig), C.POINTER(DiffWeights), C.POINTER(vp)]),
    "tt_diff_destroy": (None, [vp]),
    "tt_diff_condition": (_i, [vp, vp, _i, vp, vp, _i, vp]),
    "tt_diff_get_code_emb": (_i, [vp, vp, vp]),
    "tt_diff_forward": (_i, [vp, vp, _i, _i, vp, vp]),
    "tt_diff_sample": (_i, [vp, vp, vp, C.POINTER(DiffStep), _i, _i, vp, vp]),
    "tt_diff_batch_begin": (_i, [vp, _i, _i, vp]),
    "tt_diff_condition_slot": (_i, [vp, _i, vp, _i, vp, vp, _i, vp]),
    "tt_diff_sample_batch": (_i, [vp, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(DiffStep), _i, _i, C.POINTER(C.c_void_p), vp]),
    "tt_diff_split_begin": (_i, [vp, vp, C.POINTER(DiffStep), _i, _i, vp]),
    "tt_diff_split_forward": (_i, [vp, vp, vp]),
    "tt_diff_split_update": (_i, [vp, vp, vp, vp, vp]),
    "tt_diff_split_end": (_i, [vp]),
    "tt_cond_create": (_i, [C.POINTER(CondConfig), C.POINTER(CondWeights), C.POINTER(vp)]),
    "tt_cond_destroy": (None, [vp]),
    "tt_cond_ar_clip": (_i, [vp, vp, _i, vp, vp]),
    "tt_cond_diff_clip": (_i, [vp, vp, _i, vp, C.POINTER(_i), vp]),
    "tt_hifi_create": (_i, [C.POINTER(HifiConfig), C.POINTER(HifiWeights), C.POINTER(vp)]),
    "tt_hifi_destroy": (None, [vp]),
    "tt_hifi_output_frames": (_i, [_i]),
    "tt_hifi_run": (_i, [vp, vp, _i, vp, vp, C.POINTER(_i), vp]),
    "tt_voc_create": (_i, [C.POINTER(VocConfig), C.POINTER(VocWeights), C.POINTER(vp)]),
    "tt_voc_destroy": (None, [vp]),
    "tt_voc_run": (_i, [vp, vp, _i, vp, vp, vp]),
    "tt_voc_guard": (_i, [vp, _i]),
    "tt_cvvp_create": (_i, [C.POINTER(CvvpConfig), C.POINTER(CvvpWeights), C.POINTER(vp)]),
    "tt_cvvp_destroy": (None, [vp]),
    "tt_cvvp_score": (_i, [vp, vp, _i, _i, vp, _i, _i, vp, vp]),
    "tt_cvvp_guard": (_i, [vp, _i]),
    "tt_prof_enable": (_i, [_i]),
    "tt_graph_replay": (_i, [_i]),
    "tt_prof_classes": (_i, []),
    "tt_prof_class_name": (C.c_char_p, [_i]),
    "tt_prof_read": (_i, [_i, C.POINTER(C.c_double)]),
}
# include/tortoise_mi355x_test.h: operator-level TEST entries + the A/B switch (not part of the boundary a maintainer binds)
_TEST_PROTOS = {
    "ttx_kernel_variant": (_i, [_i, _i]),
    "tt_op_gemm": (_i, [_i, vp, _i, vp, _i, _i, _i, _i, _i, _i, _i, vp, _i, vp, vp, vp, vp]),
    "tt_op_layernorm": (_i, [_i, vp, _i, _i, vp, vp, _f, _i, vp, vp, vp]),
    "tt_op_groupnorm": (_i, [_i, vp, _i, _i, _i, vp, vp, vp, _i, vp, vp, vp, vp]),
    "tt_op_groupnorm_workspace": (_sz, [_i, _i]),
    "tt_op_gn_gemm": (_i, [_i, vp, _i, _i, vp, vp, _i, vp, vp, _i, vp, vp, vp]),
    "tt_op_gn_gemm_workspace": (_sz, [_i, _i]),
    "tt_op_flash_attention": (_i, [_i, vp, vp, vp, vp, _i, _i, _i, _i, _i, vp, vp]),
    "tt_op_decode_attention": (_i, [_i, vp, vp, vp, _i, vp, vp, _i, _i, vp, _i, _i, _i, vp]),
    "tt_op_gemv": (_i, [_i, vp, vp, _i, _i, _i, vp, _i, vp, vp, vp]),
    "tt_op_gemv_ln": (_i, [_i, vp, vp, vp, _f, vp, _i, _i, vp, vp, vp]),
    "tt_op_sample": (_i, [vp, _i, _i, _i, vp, C.POINTER(Sampling), _i, vp, _i, vp, _i, vp]),
    "tt_op_typical_mask": (_i, [vp, _i, _i, _i, vp, _f, _f, vp, vp]),
    "tt_op_conv1d": (_i, [vp, vp, vp, vp, _i, _i, _i, _i, _i, _i, _f, _i, _f, vp]),
    "tt_op_convt1d": (_i, [vp, vp, vp, vp, _i, _i, _i, _f, vp]),
    "tt_op_lvc": (_i, [_i, vp, vp, _i, _i, vp, _i, _i, vp, _i, _i, vp]),
}

_lib = None
_initialised = False


def load_library():
    """dlopen the engine and declare every prototype.  Needs no GPU (ABI checks run on CPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError("MI355X engine library not built: %s is missing. Run `python -m tortoise_tts_amd.build` "
                          "(hipcc, gfx950). There is no fallback path." % LIB_PATH)
    # torch must be loaded first: the engine shares device pointers and streams with PyTorch-ROCm, so
    # both have to bind to the ONE HIP runtime that torch ships (libamdhip64.so.7, resolved by SONAME).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(_PROTOS.items()) + list(_TEST_PROTOS.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for i, st in enumerate(BOUNDARY_STRUCTS):
        want = lib.tt_struct_size(i)
        if C.sizeof(st) != want:
            raise EngineError("ABI mismatch: %s is %d bytes in Python, %d in the library" % (st.__name__, C.sizeof(st), want))
    _lib = lib
    return lib


def init():
    """Load the library and initialise it on the current HIP device (must be gfx950)."""
    global _initialised
    lib = load_library()
    if not _initialised:
        check(lib.tt_init())
        # process-wide kernel A/B switches (diagnostics; the defaults are the measured winners)
        for env, which in (("TT_GEMM_VARIANT", TTX_GEMM_P8), ("TT_FLASH_VARIANT", TTX_FLASH32), ("TT_VOC_VARIANT", TTX_VOC_MFMA), ("TT_GEMM_SKINNY", TTX_GEMM_SKINNY), ("TT_AR_GEMV", TTX_AR_GEMV)):
            if os.environ.get(env, "") != "":
                lib.ttx_kernel_variant(which, int(os.environ[env]))
        _initialised = True
    return lib


def require_gpu(device=None):
    """The torch.device the engine will run on; raises EngineError unless it is a HIP/ROCm GPU (there is no CPU path)."""
    import torch
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != "cuda":
        raise EngineError("TextToSpeech needs an MI355X (gfx950) device; the engine has no CPU path")
    return torch.device(device)


class OperandOverflow(EngineError):
    """A stage's overflow guard tripped: non-finite values downstream of an MFMA operand cast (fp16 saturates at 65504)."""


def guard_count(rc):
    """Return value of a tt_*_guard call: >= 0 is the count, negative an error."""
    if rc < 0:
        check(rc)
    return rc


def check(rc):
    if rc != 0:
        msg = load_library().tt_last_error()
        raise EngineError("tortoise_mi355x error %d: %s" % (rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "engine tensors must be contiguous"
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream

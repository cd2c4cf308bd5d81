"""rocprofv3 kernel name -> the engine profiler's class name (tortoise_tts_amd/csrc/common.hip g_prof_names): shared by the PMC aggregation
(scripts/pmc_bench.sh) and tests/test_pmc_classes.py, which holds it against every kernel name of the committed trace."""
import re


def klass(k):
    """rocprofv3 kernel name -> the engine profiler's class name (tortoise_tts_amd/csrc/common.hip g_prof_names)."""
    if "gemm_gna_kernel" in k:
        return "gemm_gna<32,256,EpiStd,stats>"
    if "gemm_p8_kernel" in k:                              # round 5: the eight-phase 256 x 256 tile reports under the classes of the 16-wave tile it replaces
        return "gemm_glds<256,256,EpiQkvHeads>" if "EpiQkvHeads" in k else "gemm_glds<256,256,EpiStd,1x1>"
    m = re.search(r"gemm_glds_kernelI\w+?Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ENS_\d+(EpiStd|EpiQkvHeads|EpiQkvDecode|EpiGeglu)(\w*?)EELb([01])ELb[01]E", k)
    if m:
        bm, bn, epi, targs, conv = m.groups()
        # EpiStd<T, ACT, STATS, MODE>: the 64x64 1x1 GEMMs with the statistics epilogue (denoiser) are their own class in the engine's profiler
        st = re.match(r"IDF16[b_]Lin?\d+ELi1E", targs) is not None
        if epi == "EpiStd" and bm == "64" and bn == "64" and conv == "0" and st:
            return "gemm_glds<64,64,EpiStd,1x1,stats>"
        if epi == "EpiGeglu":
            return "gemm_glds<%s,%s,EpiStd,1x1>" % (bm, bn)  # (reported with the plain 1x1 class of its tile, as the engine's profiler does)
        return "gemm_glds<%s,%s,%s%s>" % (bm, bn, epi, (",conv" if conv == "1" else ",1x1") if epi == "EpiStd" else "")
    if "gemm_conv3s_kernel" in k:
        return "gemm_glds<128,64,EpiStd,conv>"  # the shared-halo 3-tap kernel reports under the conv class of its tile
    if "gemm_glds_kernel" in k:
        m = re.search(r"gemm_glds_kernel<[^,]+, (\d+), (\d+), \d+, \d+, \d+, tt::(\w+)<[^>]+>, (true|false)", k)
        if m:
            bm, bn, epi, conv = m.groups()
            return "gemm_glds<%s,%s,%s%s>" % (bm, bn, epi, (",conv" if conv == "true" else ",1x1") if epi == "EpiStd" else "")
        return "gemm_glds<?>"
    if "gemv_kernel" in k:
        return "gemv_kernel"
    for pat, name in (("flash32_kernel", "flash_kernel"), ("conv1d_mfma", "conv1d_direct_kernel"), ("lvc_mfma", "lvc_kernel"), ("flash_lds_kernel", "flash_kernel"), ("flash_kernel", "flash_kernel"), ("decode_attn_lds_kernel", "decode_attn_kernel"), ("decode_attn_kernel", "decode_attn_kernel"),
                      ("gn_apply", "gn_apply_kernel(+gn_stats)"), ("gn_stats", "gn_apply_kernel(+gn_stats)"), ("rownorm", "rownorm_kernel"),
                      ("sample_kernel", "sample_kernel"), ("lvc_kernel", "lvc_kernel"), ("conv1d_direct", "conv1d_direct_kernel"), ("convt1d", "convt1d_kernel")):
        if pat in k:
            return name
    return None

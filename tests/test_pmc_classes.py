"""CPU: the kernel-name classifier behind profiles/r*_pmc_*.json knows every engine kernel of the committed kernel trace - a kernel added
to the engine without a class would silently drop out of `roofline.traffic` (round 4: gemm_gna_kernel and the EpiGeglu GEMM did)."""
import csv
import os

from scripts.pmc_classes import klass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF_NAMES = open(os.path.join(ROOT, "tortoise_tts_amd", "csrc", "common.hip")).read()


def test_every_gemm_and_attention_kernel_of_the_trace_has_a_profiler_class():
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r04_final_kernel_stats.csv"))))
    assert len(rows) > 40
    seen = set()
    for r in rows:
        name = r["Name"]
        if not ("tt::" in name or "_ZN2tt" in name):
            continue  # torch / runtime kernels of the engine build, not the engine's
        k = klass(name)
        if any(p in name for p in ("gemm_", "flash_", "decode_attn", "gn_apply", "rownorm", "sample_kernel", "lvc_kernel")):
            assert k is not None and "?" not in k, name
            assert '"%s"' % k in PROF_NAMES, (name, k)  # the class exists in the engine's own profiler table
            seen.add(k)
    assert {"gemm_gna<32,256,EpiStd,stats>", "decode_attn_kernel", "gemm_glds<64,64,EpiStd,1x1>", "gemm_glds<128,64,EpiStd,conv>", "flash_kernel"} <= seen


def test_known_names():
    assert klass("_ZN2tt15gemm_gna_kernelIDF16_Li32ELi256ELi8ELi1ELi2ENS_6EpiStdIDF16_Li0ELi1ELi5EEELb0ELb1EEEvNS_10GemmGnaDevINT5_4ArgsEEE") == "gemm_gna<32,256,EpiStd,stats>"
    assert klass("_ZN2tt16gemm_glds_kernelIDF16bLi256ELi256ELi16ELi4ELi2ENS_8EpiGegluIDF16bEELb0ELb1ELi0EEEvNS_7GemmDevINT5_4ArgsEEE") == "gemm_glds<256,256,EpiStd,1x1>"
    assert klass("_ZN2tt16gemm_glds_kernelIDF16_Li64ELi64ELi4ELi2ELi4ENS_6EpiStdIDF16_Li0ELi1ELi7EEELb0ELb1ELi0EEEvNS_7GemmDevINT5_4ArgsEEE") == "gemm_glds<64,64,EpiStd,1x1,stats>"
    assert klass("void at::native::vectorized_elementwise_kernel<4, ...>") is None
    # round 5
    assert klass("_ZN2tt14flash32_kernelIDF16_Li2EEEvNS_9FlashArgsE") == "flash_kernel"
    assert klass("_ZN2tt18conv1d_mfma_kernelENS_10Conv1dArgsE") == "conv1d_direct_kernel" and klass("_ZN2tt15lvc_mfma_kernelILi256EEEvNS_7LvcArgsE") == "lvc_kernel"
    assert klass("_ZN2tt14gemm_p8_kernelIDF16_NS_6EpiStdIDF16_Li0ELi1ELi7EEEEEvNS_7GemmDevINT0_4ArgsEEE") == "gemm_glds<256,256,EpiStd,1x1>"
    assert klass("_ZN2tt14gemm_p8_kernelIDF16bNS_11EpiQkvHeadsIDF16bEEEEvNS_7GemmDevINT0_4ArgsEEE") == "gemm_glds<256,256,EpiQkvHeads>"
    assert klass("_ZN2tt15gemm_gna_kernelIDF16_Li32ELi256ELi8ELi1ELi2ENS_11EpiQkvHeadsIDF16_EELb0ELb0ELb0EEEvNS_10GemmGnaDevINT5_4ArgsEEE") == "gemm_gna<32,256,EpiStd,stats>"
    for k in ("gemm_glds<256,256,EpiQkvHeads>", "gemm_glds<256,256,EpiStd,1x1>"):
        assert '"%s"' % k in PROF_NAMES

"""In-tree build of the gfx950 engine library (hipcc cross-compiles without a GPU).

`python -m tortoise_tts_amd.build` or `__graft_entry__.build()` produces
tortoise_tts_amd/lib/libtortoise_mi355x.so.  Objects are cached by source hash so an
unchanged file is not recompiled.  There is exactly one target (gfx950); no other arch, no
fallback path.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtortoise_mi355x.so")
KBENCH_LIB = os.path.join(LIBDIR, "libtortoise_kbench.so")  # experiments only (scripts/kbench.py); never loaded by the product
SOURCES = ["common.hip", "gemm.hip", "gemm_bf16.hip", "gemm_f16.hip", "gemm_f32.hip", "gemv.hip", "norm.hip", "attention.hip", "attention_f32.hip", "sampling.hip", "misc.hip", "univnet.hip",
           "gpt2.hip", "clvp.hip", "cvvp.hip", "diffusion.hip", "cond.hip", "hifigan.hip", "vocoder.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-ffp-contract=on",
         "-Werror=extra-tokens", "-Werror=return-type"]  # (a knob inserted as `#endif <rest of the line>` silently drops the rest: round 6 lost a GPU call to one)


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built")


def _digest(paths):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "tortoise_mi355x.h"))
    return hs


def build(force=False, verbose=True, variant=None, defines=()):
    """variant / defines: an alternative library lib/libtortoise_mi355x_<variant>.so compiled with extra -D knobs (csrc/knobs.h)
    for in-situ A/B runs (TORTOISE_MI355X_LIB selects it); the product library is the one built without them."""
    OBJ = os.path.join(CSRC, "build" if not variant else "build_" + variant)
    LIB = os.path.join(LIBDIR, "libtortoise_mi355x.so" if not variant else "libtortoise_mi355x_%s.so" % variant)
    FLAGS = globals()["FLAGS"] + ["-D" + d for d in defines]
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = _headers()
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        stamp = op + ".sha"
        dig = _digest([sp] + headers) + "|" + " ".join(defines)
        objs.append(op)
        if not force and os.path.exists(op) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((sp, op, stamp, dig))

    def compile_one(job):
        sp, op, stamp, dig = job
        cmd = [hipcc] + FLAGS + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (sp, r.stdout, r.stderr))
        with open(stamp, "w") as f:
            f.write(dig)
        return sp

    if jobs:
        if verbose:
            print("[build] compiling %d file(s) for gfx950 ..." % len(jobs), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[build] linked %s" % LIB, file=sys.stderr)
    return LIB


def build_kbench(verbose=True):
    """Kernel experiments / microbenchmarks (csrc/kbench/kbench.hip) linked with the product objects into a SEPARATE library."""
    build(verbose=verbose)
    hipcc = _hipcc()
    src = os.path.join(CSRC, "kbench", "kbench.hip")
    obj = os.path.join(OBJ, "kbench.o")
    r = subprocess.run([hipcc] + FLAGS + ["-DTT_KBENCH", "-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    objs = [os.path.join(OBJ, s_.replace(".hip", ".o")) for s_ in SOURCES] + [obj]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", KBENCH_LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("[build] linked %s" % KBENCH_LIB, file=sys.stderr)
    return KBENCH_LIB


if __name__ == "__main__":
    if "--kbench" in sys.argv:
        build_kbench()
    elif "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        build(variant=name, defines=[a[2:] for a in sys.argv if a.startswith("-D")])
    else:
        build(force="--force" in sys.argv)

#!/usr/bin/env python3
"""Build libgt_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python galerkin-transformer_amd/build.py            # release library
    python galerkin-transformer_amd/build.py --emu      # + debug twin with shuffle-emulated MFMA

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under csrc/.obj/.
"""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT_DIR = os.path.join(HERE, "galerkin_transformer", "_lib")
SOURCES = ["gt_gemm.hip", "gt_gemm_x3.hip", "gt_ops.hip", "gt_resize.hip", "gt_fourier.hip", "gt_fourier16.hip", "gt_dft.hip", "gt_tsmm.hip", "gt_head.hip", "gt_optim.hip", "gt_convw.hip", "gt_ffn.hip"]
HEADERS = [os.path.join(CSRC, "gt_common.h"), os.path.join(CSRC, "gt_gemm_core.h"), os.path.join(INC, "gt_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(emu=False, verbose=False, force=False, tag=None, defines=(), only=None):
    """tag/defines: extra named variants (ablation builds for tools/ablate_gemm.sh), e.g.
    build(tag="_nomfma", defines=["GT_ABL_NOMFMA"])."""
    os.makedirs(OUT_DIR, exist_ok=True)
    tag = tag if tag is not None else ("_emu" if emu else "")
    objdir = os.path.join(CSRC, ".obj" + tag)
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC]
    if emu:
        flags.append("-DGT_EMULATE_MFMA=1")
    flags += ["-D" + d for d in defines]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if only is not None and src not in only:       # variant builds recompile only the files the define touches
            objs.append(os.path.join(CSRC, ".obj", src.replace(".hip", ".o")))
            continue
        if force or _stale(o, [s] + HEADERS):
            jobs.append([_hipcc()] + flags + ["-c", s, "-o", o])
        objs.append(o)
    if jobs:        # one hipcc per source, a few at a time (the big translation units take minutes each)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=int(os.environ.get("GT_BUILD_JOBS", "6"))) as ex:
            list(ex.map(run, jobs))
    lib = os.path.join(OUT_DIR, f"libgt_hip{tag}.so")
    if force or _stale(lib, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--ablate", action="store_true", help="also build the GEMM ablation variants")
    ap.add_argument("--ablate-x3", action="store_true", help="also build the split-operand GEMM ablation variants")
    ap.add_argument("--ablate-head", action="store_true", help="also build the head_bwd16_kernel timing ablations")
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    print(build(False, a.verbose, a.force))
    if a.emu:
        print(build(True, a.verbose, a.force))
    if a.ablate_x3:
        for t, d in (("_x3nomfma", ["GT_ABL_X3_NOMFMA"]), ("_x3nosplit", ["GT_ABL_X3_NOSPLIT_A", "GT_ABL_X3_NOSPLIT_B"]),
                     ("_x3nosplitb", ["GT_ABL_X3_NOSPLIT_B"]), ("_x3nostore", ["GT_ABL_X3_NOSTORE"]),
                     ("_x3onlyload", ["GT_ABL_X3_NOMFMA", "GT_ABL_X3_NOSPLIT_A", "GT_ABL_X3_NOSPLIT_B", "GT_ABL_X3_NOSTORE"])):
            print(build(False, a.verbose, a.force, tag=t, defines=d, only=["gt_gemm_x3.hip"]))
    if a.ablate_head:
        for m in (1, 2, 3, 4, 8, 16, 31):
            print(build(False, a.verbose, a.force, tag=f"_h16abl{m}", defines=[f"H16_ABL={m}"], only=["gt_head.hip"]))
    if a.ablate:
        for t, d in (("_nomfma", ["GT_ABL_NOMFMA"]), ("_noload", ["GT_ABL_NOLOAD"]),
                     ("_nostore", ["GT_ABL_NOSTORE"]), ("_onlymfma", ["GT_ABL_NOLOAD", "GT_ABL_NOSTORE"])):
            print(build(False, a.verbose, a.force, tag=t, defines=d))

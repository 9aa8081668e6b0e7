"""Build libnrms_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m pytorch_news_recommender_amd.build [--force]

The library sits next to the package so it travels to the GPU box with the repo
snapshot; nothing is installed into site-packages and there is no JIT cache.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
BUILD = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libnrms_hip.so")
SOURCES = ["gemm.hip", "gemm_bf16.hip", "embed.hip", "attention.hip", "pool.hip", "wide.hip", "fused16.hip", "fused16_v1.hip", "fused16_bwd.hip", "fused16_v1_bwd.hip", "user64.hip", "segpool.hip", "hier.hip", "empty_seq.hip", "capi.hip"]
HEADERS = ["common.h", "gemm.h", "fused16.h", "fused16_bwd.h", "fused16_v1.h", os.path.join("..", "..", "include", "nrms_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    flags += os.environ.get("NRMS_HIPCC_EXTRA", "").split()      # diagnostic builds only (e.g. -DNRMS_U64_EXPERIMENTS)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(BUILD, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))

"""Build libwefax_hip.so (gfx950) in-tree with hipcc.

    python -m wefax_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to wefax_amd/csrc/build/, the
library to wefax_amd/libwefax_hip.so (git-ignored, shipped to the GPU box).
The whole library is built with -ffp-contract=off: the parity-critical scalar
code (percentile lerp, quantise, Pillow coefficients) must round like
NumPy/Pillow do; the FFT butterflies call fma() explicitly.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libwefax_hip.so")
SOURCES = ["wfx_context.hip", "wfx_fft.hip", "wfx_mrfft.hip", "wfx_stages.hip", "wfx_polyphase.hip", "wfx_ingest.hip", "wfx_fmm.hip", "wfx_api.hip",
           "wfx_comm.hip", "wfx_dist.hip", "wfx_shard.hip", "wfx_synth.hip", "wfx_png.hip"]
HEADERS = [os.path.join(CSRC, "wfx_internal.h"), os.path.join(CSRC, "wfx_dist.h"), os.path.join(CSRC, "wfx_notch.h"), os.path.join(REPO, "include", "wefax_hip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(REPO, "include"), "-I", CSRC]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libwefax_hip.so cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


LAST_BUILD = {"compiled": 0, "up_to_date": 0, "linked": False}


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """Compile what is stale (everything with ``force``) and link.  LAST_BUILD says what this call did."""
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    flags = [*FLAGS, *extra_flags]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([cc, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    link = bool(force or jobs or _stale(LIB, objs))
    if link:
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"])
    LAST_BUILD.update(compiled=len(jobs), up_to_date=len(SOURCES) - len(jobs), linked=link)
    return LIB


if __name__ == "__main__":
    stats = "--pick-stats" in sys.argv      # diagnostic build: cycle stamps in sync_pick_kernel (WFX_DEBUG=1 prints them)
    print(build(force="--force" in sys.argv or stats, verbose=True,
                extra_flags=("-DWFX_PICK_STATS",) if stats else ()))

"""Builds librpo_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librpo_hip.so")
SOURCES = ["gemm.hip", "gemm_ws.hip", "norm.hip", "attn_image.hip", "attn_text.hip", "misc.hip", "preprocess.hip"]
# The measured-slower experiments of rounds 3 / 4 (include/rpo_amd_experimental.h: paired launches, the fused MLP launch,
# the persistent backward chain) are compiled only into a SECOND library, with -DRPO_EXPERIMENTAL, that nothing loads
# unless RPO_EXPERIMENTAL=1 is set (rpo_amd/_lib.py): `python -m rpo_amd.build --experimental`
LIB_EXP = os.path.join(HERE, "build", "librpo_hip_exp.so")
SOURCES_EXP = SOURCES + ["chain.hip"]
# preprocess.hip reproduces Pillow's double-precision coefficient math bit for bit: no FMA contraction
EXTRA_FLAGS = {"preprocess.hip": ["-ffp-contract=off"]}
# -fvisibility=hidden: the dynamic symbol table holds the C ABI of include/rpo_amd.h (which pushes default visibility
# around its declarations) and nothing else -- no mangled helpers, no kernel host stubs (tests/test_host_logic.py)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: librpo_hip.so cannot be built")


def _stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "..", "include", "rpo_amd.h"), os.path.join(HERE, "..", "include", "rpo_amd_experimental.h"),
        os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, experimental: bool = False) -> str:
    lib, sources = (LIB_EXP, SOURCES_EXP) if experimental else (LIB, SOURCES)
    flags = FLAGS + (["-DRPO_EXPERIMENTAL"] if experimental else [])
    if not force and not _stale(lib):
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", "exp") if experimental else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={os.path.join(CSRC, 'exports.map')}",
                        *objs, "-o", lib],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, experimental="--experimental" in sys.argv))

"""In-tree build of libmmdeer_hip.so (hipcc, gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container and on
the GPU box alike.  The .so lands next to this file: it is git-ignored but
travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(PKG_DIR, "libmmdeer_hip.so")
SOURCES = ["gemm_nt.hip", "gemm_nx.hip", "gemm_tt.hip", "gemm_glds.hip", "gemm_ln.hip", "chain.hip", "gemm_tt256.hip", "gemm_nt256.hip", "gemm.hip", "rowops.hip", "attention.hip", "tri_fused.hip", "nig.hip", "optim.hip", "side.hip", "stackb.hip", "stackb_train.hip", "fusions.hip", "comm.hip", "api.hip"]
ARCH = "gfx950"
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs at wave start (gemm_glds.hip)
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmmdeer_hip.so cannot be built")
    return exe


def _newest_dep_mtime() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc", ".hip"))]
    deps.append(os.path.join(os.path.dirname(PKG_DIR), "include", "mmdeer.h"))
    return max(os.path.getmtime(d) for d in deps)


def needs_build() -> bool:
    return not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < _newest_dep_mtime()


def build_stamps() -> str:
    """Diagnostic library with in-kernel s_memtime stamps (tools/gemm_stamps.py); never loaded by the product."""
    hipcc = _hipcc()
    out = os.path.join(PKG_DIR, "libmmdeer_stamps.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    r = subprocess.run([hipcc, *FLAGS, "-DMMDEER_STAMPS", "-shared", "-o", out, *srcs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


def build_variant(name: str, defines=()) -> str:
    """A/B library libmmdeer_var_<name>.so built with extra -D flags (tools/ab_fused.py); never loaded by the product."""
    hipcc = _hipcc()
    out = os.path.join(PKG_DIR, f"libmmdeer_var_{name}.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    r = subprocess.run([hipcc, *FLAGS, *[f"-D{d}" for d in defines], "-shared", "-o", out, *srcs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    # one builder at a time: the N ranks of a multi-GPU launch all get here together when the library is stale, and
    # they share the object directory and the output file
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():       # another process built it while this one waited
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, srcs))
    tmp = LIB_PATH + ".tmp"
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

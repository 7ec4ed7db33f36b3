"""Build libcommonscenes_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo).

The library is the product: there is no CPU fallback.  `build_native()` is what
`__graft_entry__.build()` calls; it cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_NAME = "libcommonscenes_hip.so"
LIB_PATH = PKG_DIR / LIB_NAME
SOURCES = ["cs_plan.hip", "cs_gemm.hip", "cs_gemm_f16x3.hip", "cs_gemm_kw.hip", "cs_mesh.hip", "cs_metrics.hip", "cs_norm.hip", "cs_attention.hip", "cs_attention_f16x3.hip", "cs_ops.hip", "cs_unet.hip", "cs_vqvae.hip"]
HEADERS = [CSRC / "cs_common.h", CSRC / "cs_f16x3.h", CSRC / "cs_mc_tables.h", CSRC / "cs_driver.h", PKG_DIR.parent / "include" / "commonscenes_hip.h"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", "-Wall",
         "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: cannot build libcommonscenes_hip.so")


def _flagstr() -> str:
    return " ".join([*FLAGS, *os.environ.get("CS_EXTRA_HIPCC_FLAGS", "").split()])


def _deps(src: Path, seen=None) -> set:
    """src and every file it includes with quotes, recursively (paths relative to the including file)"""
    import re
    seen = set() if seen is None else seen
    src = src.resolve()
    if src in seen or not src.exists():
        return seen
    seen.add(src)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', src.read_text(), re.M):
        _deps(src.parent / m.group(1), seen)
    return seen


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    stamp = PKG_DIR / "build" / "flags.txt"
    if stamp.exists() and stamp.read_text() != _flagstr():
        return True             # the library on disk came from another flag set (a what-if build): never keep it silently
    t = LIB_PATH.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + HEADERS      # (the flag set is tracked by build/flags.txt)
    return any(d.stat().st_mtime > t for d in deps)


def build_native(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    extra = os.environ.get("CS_EXTRA_HIPCC_FLAGS", "").split()      # debug builds only (e.g. -DCS_ABLATE=1)
    objdir = PKG_DIR / "build"
    objdir.mkdir(exist_ok=True)
    import time
    t0 = time.time()          # outputs are stamped with the build's START: a source edited while hipcc runs stays newer
    procs = []
    objs = []
    # incremental: an object is rebuilt when its source, any header, this file or the flag set is newer / different
    stamp = objdir / "flags.txt"
    flagstr = _flagstr()
    same_flags = stamp.exists() and stamp.read_text() == flagstr
    for s in SOURCES:
        obj = objdir / (s.replace(".hip", ".o"))
        objs.append(str(obj))
        # (per-source dependencies: the quoted includes, followed recursively -- cs_gemm_f16x3.hip takes five minutes and
        # does not include cs_driver.h)
        newest_dep = max(d.stat().st_mtime for d in _deps(CSRC / s))
        if not force and same_flags and obj.exists() and obj.stat().st_mtime > newest_dep:
            continue
        cmd = [hipcc, *FLAGS, *extra, "-c", str(CSRC / s), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
        if verbose and out.strip():
            print(out.decode())
        os.utime(objdir / (s.replace(".hip", ".o")), (t0, t0))
    stamp.write_text(flagstr)
    tmp = LIB_PATH.with_suffix(".so.tmp")
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", str(tmp)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB_PATH)
    os.utime(LIB_PATH, (t0, t0))
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))

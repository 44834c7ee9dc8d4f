"""Builds the in-tree native libraries.

* ``lambda_amd/csrc/liblambda_ext.so`` -- the product: gfx950 HIP kernels + the C ABI of include/lambda_ext.h
  + the C++ host mirror of the reference's extension driver.  Built with ``hipcc --offload-arch=gfx950``
  (cross-compiles without a GPU).
* ``oracle/_build/liblx_oracle.so`` -- TEST INFRASTRUCTURE: the CPU restatement used as the parity checker and
  as bench.py's cpu_baseline leg.  Never linked into the product.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "lambda_amd" / "csrc"
LIB = CSRC / "liblambda_ext.so"
ORACLE_DIR = ROOT / "oracle"
ORACLE_LIB = ORACLE_DIR / "_build" / "liblx_oracle.so"

HIP_SOURCES = ["lx_score.hip", "lx_score_f16.hip", "lx_score_i16.hip", "lx_sweep_mq.hip", "lx_trace.hip", "lx_ckpt.hip", "lx_select.hip", "lx_pack.hip", "lx_prefilter.hip", "lx_level2.hip", "lx_plan_free.hip", "lx_records.hip", "lx_level2_host.cpp", "lx_api.cpp", "lx_host.cpp", "lx_host_batch.cpp", "lx_host_pool.cpp", "host/lx_driver.cpp", "host/lx_output.cpp", "host/lx_translate.cpp"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _product_deps():
    srcs = [CSRC / s for s in HIP_SOURCES if (CSRC / s).exists()]
    return srcs, list(srcs) + sorted(CSRC.glob("*.h")) + sorted((CSRC / "host").glob("*.hpp")) + [ROOT / "include" / "lambda_ext.h"]


def source_id() -> str:
    """SHA-256 (first 16 hex digits) over every source the product library is built from, names included.  Compiled
    into the library (lx_build_id()); tests and build() compare it with the tree so that a stale .so cannot pass."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(_product_deps()[1], key=lambda f: str(f.relative_to(ROOT))):
        h.update(str(f.relative_to(ROOT)).encode() + b"\0")
        h.update(f.read_bytes())
        h.update(b"\0")
    h.update(os.environ.get("LX_EXTRA_DEFINES", "").encode())
    return h.hexdigest()[:16]


_ID_MARK = b"LXBUILDID:"


def library_id(lib: Path = None) -> str | None:
    """The source id compiled into a built library (read from the file, no dlopen), None if absent."""
    lib = lib or LIB
    if not lib.exists():
        return None
    data = lib.read_bytes()
    at = data.find(_ID_MARK)
    return data[at + len(_ID_MARK): at + len(_ID_MARK) + 16].decode("ascii", "replace") if at >= 0 else None


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(d).stat().st_mtime <= t for d in deps)


def build_product(force: bool = False, verbose: bool = False) -> Path:
    srcs, deps = _product_deps()
    sid = source_id()
    if not force and library_id() == sid:
        return LIB  # the library says it was built from exactly this tree (mtimes are not trusted)
    objs = []
    hipcc = _hipcc()
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", f"-I{ROOT / 'include'}", f"-I{CSRC}",
              "-Wall", "-Wno-unused-function", *os.environ.get("LX_EXTRA_DEFINES", "").split()]
    procs = []
    variant = bool(os.environ.get("LX_EXTRA_DEFINES", "").split())
    for s in srcs:
        # (a kernel variant -- LX_EXTRA_DEFINES -- gets objects of its own, always recompiled: an object made with other defines is
        # newer than its source and would otherwise end up in the next default build)
        o = s.with_suffix(".var.o" if variant else ".o")
        objs.append(o)
        is_api = s.name == "lx_api.cpp"  # carries the build id: recompiled whenever anything changed
        if not force and not variant and not is_api and _newer(o, [s] + [d for d in deps if d.suffix in (".h", ".hpp")]):
            continue
        cmd = [hipcc, *common, *([f'-DLX_BUILD_ID="{sid}"'] if is_api else []), "-c", str(s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


CLI = CSRC / "lambda3"


def build_cli(force: bool = False) -> Path:
    """The minimal `lambda3 searchp|searchn` front end (plumbing for BASELINE.json configs[0])."""
    src = CSRC / "host" / "lambda3_main.cpp"
    if not force and _newer(CLI, [src, LIB] + list((CSRC / "host").glob("*.hpp"))):
        return CLI
    cmd = [_hipcc(), "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", f"-I{ROOT / 'include'}", str(src), "-o", str(CLI),
           f"-L{CSRC}", "-llambda_ext", f"-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"CLI build failed:\n{r.stdout}")
    return CLI


def build_oracle(force: bool = False) -> Path:
    srcs = [ORACLE_DIR / "lx_oracle.c", ORACLE_DIR / "lx_oracle_simd.cpp", ORACLE_DIR / "lx_oracle.h"]
    if not force and _newer(ORACLE_LIB, srcs):
        return ORACLE_LIB
    r = subprocess.run(["make", "-C", str(ORACLE_DIR)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle build failed:\n{r.stdout}")
    return ORACLE_LIB


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_product(force=force, verbose=verbose)
    build_cli(force=force)
    build_oracle(force=force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(ORACLE_LIB)

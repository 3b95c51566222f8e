"""Build helpers: the C-ABI runtime library and per-model gfx950 code objects.

* ``build_runtime()``  hipcc -shared csrc/pk_runtime.cpp -> pockit_amd/libpockit_hip.so  (in-tree, so the
  built library travels with the repository snapshot to the GPU box).
* ``compile_model(source)``  generated HIP source -> code object for gfx950 (``hipcc --genco``), cached in
  pockit_amd/_cache/<sha>.hsaco keyed by the source hash.  The generated source is mesh-independent, so
  one code object serves every mesh of a model -- the analogue of the reference's FastFunc cache
  (/root/reference/pockit/base/fastfunc.py:126-131,196-223).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libpockit_hip.so")
CACHE_DIR = os.environ.get("POCKIT_AMD_CACHE", os.path.join(HERE, "_cache"))
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X evaluator needs the ROCm toolchain to build its kernels")
    return exe


def _run(cmd):
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + res.stderr[-4000:])
    return res


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_runtime(force=False):
    """Compile libpockit_hip.so (host C++ against libamdhip64)."""
    srcs = [os.path.join(CSRC, "pk_runtime.cpp"), os.path.join(CSRC, "pk_abi.h"),
            os.path.join(os.path.dirname(HERE), "include", "pockit_hip.h")]
    if force or _stale(LIB_PATH, srcs):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-O2", "-fPIC", "-shared", "-std=c++17", srcs[0], "-o", LIB_PATH])
    return LIB_PATH


# extra device-compile flags (part of the cache key); POCKIT_AMD_HIPCC_FLAGS overrides for experiments
EXTRA_FLAGS = os.environ.get("POCKIT_AMD_HIPCC_FLAGS", "").split()


# leading scalar kernel arguments (pk_cycle's tile list, counts, flags) arrive in SGPRs with the wave
PRELOAD_FLAGS = [] if os.environ.get("POCKIT_AMD_KERNARG_PRELOAD", "1") == "0" else ["-mllvm", "-amdgpu-kernarg-preload-count=4"]


def _kernel_header_hash():
    h = hashlib.sha256()
    for name in ("pk_kernels.hip.h", "pk_abi.h"):
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


# Each code object leaves a one-line record <key>.gen holding the kernel-header hash it was compiled against: a header
# edit orphans every earlier entry, and tools/prune_cache.py drops the orphans before they travel to a GPU lease.
def live_keys():
    """Keys in the cache that were compiled against the CURRENT kernel header."""
    cur, keep = _kernel_header_hash(), set()
    if os.path.isdir(CACHE_DIR):
        for name in os.listdir(CACHE_DIR):
            if name.endswith(".gen"):
                try:
                    with open(os.path.join(CACHE_DIR, name)) as fh:
                        if fh.read().strip() == cur:
                            keep.add(name[:-4])
                except OSError:
                    pass
    return keep


def write_index(keep):
    """Remove the generation records of entries that are gone or stale."""
    if os.path.isdir(CACHE_DIR):
        for name in os.listdir(CACHE_DIR):
            if name.endswith(".gen") and name[:-4] not in keep:
                os.remove(os.path.join(CACHE_DIR, name))


# wall-clock seconds this process spent in hipcc for model code objects (cache misses); bench.py reports it
COMPILE_SECONDS = {"total": 0.0, "count": 0, "last": 0.0}


def compile_model(source: str, fastmath: bool = False, keep_source: bool = True) -> bytes:
    """Return the gfx950 code object of a generated model source (compiling on a cache miss)."""
    key = hashlib.sha256((source + _kernel_header_hash() + str(bool(fastmath)) + " ".join(PRELOAD_FLAGS + EXTRA_FLAGS)).encode()
                         ).hexdigest()[:32]
    os.makedirs(CACHE_DIR, exist_ok=True)
    path = os.path.join(CACHE_DIR, key + ".hsaco")
    gen = os.path.join(CACHE_DIR, key + ".gen")
    if os.path.exists(path) and not os.path.exists(gen):
        try:
            with open(gen, "w") as fh:
                fh.write(_kernel_header_hash())
        except OSError:
            pass
    if not os.path.exists(path):
        import time

        t_start = time.perf_counter()
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "model.hip")
            with open(src, "w") as fh:
                fh.write(source)
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "--genco", f"-I{CSRC}", src, "-o",
                   os.path.join(tmp, "model.hsaco")] + PRELOAD_FLAGS + EXTRA_FLAGS
            if fastmath:  # reassociation subset of fast-math (reference: numba fastmath=True, fastfunc.py:24,35)
                cmd += ["-fassociative-math", "-freciprocal-math", "-fno-signed-zeros", "-fno-trapping-math"]
            _run(cmd)
            # publish atomically through a name of our own: ranks that start cold together (bench.py under torchrun,
            # the two-process test) all compile the same model and must not share a staging file
            fd, staged = tempfile.mkstemp(dir=CACHE_DIR, prefix=key + ".", suffix=".part")
            os.close(fd)
            shutil.copyfile(os.path.join(tmp, "model.hsaco"), staged)
            os.replace(staged, path)
            if keep_source:
                fd, staged = tempfile.mkstemp(dir=CACHE_DIR, prefix=key + ".", suffix=".part")
                os.close(fd)
                shutil.copyfile(src, staged)
                os.replace(staged, os.path.join(CACHE_DIR, key + ".hip"))
            with open(gen, "w") as fh:
                fh.write(_kernel_header_hash())
        COMPILE_SECONDS["last"] = time.perf_counter() - t_start
        COMPILE_SECONDS["total"] += COMPILE_SECONDS["last"]
        COMPILE_SECONDS["count"] += 1
    with open(path, "rb") as fh:
        return fh.read()

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
        # (the compiler's own error lines first: the resource remarks of a large model bury them beyond any tail)
        errs = [ln for ln in res.stderr.splitlines() if "error" in ln.lower() and "remark:" not in ln]
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + "\n".join(errs[:20]) + "\n...\n" + res.stderr[-3000:])
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
PRELOAD_FLAGS = ["-mllvm", "-amdgpu-kernarg-preload-count=4"]


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


def _parse_resource_usage(stderr: str) -> dict:
    """``-Rpass-analysis=kernel-resource-usage`` remarks -> {kernel: {sgpr, vgpr, agpr, scratch, occupancy, sgpr_spill,
    vgpr_spill, lds}} (registers per lane, scratch in bytes per lane)."""
    keys = {"TotalSGPRs": "sgpr", "SGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "LDS Size [bytes/block]": "lds"}
    out, cur = {}, None
    for line in stderr.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass-analysis")[0].strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
            continue
        if cur is None or ":" not in body:
            continue
        name, val = body.rsplit(":", 1)
        if name.strip() in keys:
            try:
                cur[keys[name.strip()]] = int(val.strip())
            except ValueError:
                pass
    return out


def resource_usage(source: str, fastmath: bool = False, extra_flags=()):
    """What the compiler said about every kernel of a model's code object (registers, spills, scratch, occupancy): recorded
    by ``compile_model`` beside the object.  None for an object compiled before the record existed."""
    import json

    path = os.path.join(CACHE_DIR, _key(source, fastmath, extra_flags) + ".res.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


def spills(usage) -> dict:
    """The kernels of ``usage`` that spill vector registers to scratch memory: {kernel: (vgpr_spill, scratch bytes)}.
    (A few bytes of private segment WITHOUT a spilled register -- pk_hessc shows 20 -- are the register allocator's
    emergency slot for scalar spills: the ISA holds no scratch instruction; tools/examples_resources.py lists them.)"""
    return {k: (v.get("vgpr_spill", 0), v.get("scratch", 0)) for k, v in (usage or {}).items() if v.get("vgpr_spill", 0) > 0}


def _key(source: str, fastmath: bool, extra_flags=()) -> str:
    # (``extra_flags``: flags of ONE compile -- evaluator.Evaluator.checked's rebuild; empty: the key of every earlier object)
    return hashlib.sha256((source + _kernel_header_hash() + str(bool(fastmath)) + " ".join(PRELOAD_FLAGS + EXTRA_FLAGS + list(extra_flags))
                           ).encode()).hexdigest()[:32]


# wall-clock seconds this process spent in hipcc for model code objects (cache misses); bench.py reports it
COMPILE_SECONDS = {"total": 0.0, "count": 0, "last": 0.0}


def compile_model(source: str, fastmath: bool = False, keep_source: bool = None, extra_flags=()) -> bytes:
    """Return the gfx950 code object of a generated model source (compiling on a cache miss).  ``keep_source``: also keep
    the generated source beside the object (default: only with POCKIT_AMD_KEEP_SOURCE=1 -- 19 MB for the GPU suite's models)."""
    if keep_source is None:
        keep_source = os.environ.get("POCKIT_AMD_KEEP_SOURCE", "0") == "1"
    import zlib

    key = _key(source, fastmath, extra_flags)
    os.makedirs(CACHE_DIR, exist_ok=True)
    # (stored deflated: a code object of a large model has 1-3 MB, the cache of the GPU suite 140 MB -- 35 MB deflated -- and
    #  the whole cache travels to every GPU lease)
    path = os.path.join(CACHE_DIR, key + ".hsacoz")
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
                   os.path.join(tmp, "model.hsaco"), "-Rpass-analysis=kernel-resource-usage"] + PRELOAD_FLAGS + EXTRA_FLAGS + list(extra_flags)
            if fastmath:  # reassociation subset of fast-math (reference: numba fastmath=True, fastfunc.py:24,35)
                cmd += ["-fassociative-math", "-freciprocal-math", "-fno-signed-zeros", "-fno-trapping-math"]
            res = _run(cmd)
            try:        # the compiler's per-kernel register / spill / scratch report, kept beside the object
                import json

                fd, staged = tempfile.mkstemp(dir=CACHE_DIR, prefix=key + ".", suffix=".part")
                with os.fdopen(fd, "w") as fh:
                    json.dump(_parse_resource_usage(res.stderr), fh)
                os.replace(staged, os.path.join(CACHE_DIR, key + ".res.json"))
            except OSError:
                pass
            # publish atomically through a name of our own: ranks that start cold together (bench.py under torchrun,
            # the two-process test) all compile the same model and must not share a staging file
            fd, staged = tempfile.mkstemp(dir=CACHE_DIR, prefix=key + ".", suffix=".part")
            with open(os.path.join(tmp, "model.hsaco"), "rb") as fin, os.fdopen(fd, "wb") as fout:
                fout.write(zlib.compress(fin.read(), 6))
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
        log = os.environ.get("POCKIT_AMD_COMPILE_LOG")      # (tests/conftest.py: which objects a session had to compile)
        if log:
            try:
                with open(log, "a") as fh:
                    fh.write(f"{key} {COMPILE_SECONDS['last']:.1f}\n")
            except OSError:
                pass
    used = os.environ.get("POCKIT_AMD_USED_LOG")      # (tests/conftest.py on a GPU box, __graft_entry__.build(): the keys a session
    if used:                                              #  was served -- tools/prune_cache.py --keep-list ships exactly those)
        try:
            with open(used, "a") as fh:
                fh.write(key + "\n")
        except OSError:
            pass
    with open(path, "rb") as fh:
        return zlib.decompress(fh.read())

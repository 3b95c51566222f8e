"""The C-ABI library builds, loads and exports every symbol include/pockit_hip.h declares; without a
GPU the product fails loudly (no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re

import pytest

from pockit_amd import hipbuild, runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hipbuild.build_runtime()
    lib = runtime.load_library()
    header = open(os.path.join(ROOT, "include", "pockit_hip.h")).read()
    declared = set(re.findall(r"\b(pk_[a-z_]+)\s*\(", header))
    declared -= {"pk_ctx"}
    assert declared, "no declarations parsed"
    assert declared == set(runtime.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_sizes_match_the_c_abi():
    # csrc/pk_abi.h: PkPhase 30 ints, PkTile 22 ints, PkKind 8 ints, PkItem {int64, double, int32, int32}
    assert runtime.PHASE_DTYPE.itemsize == 30 * 4
    assert runtime.TILE_DTYPE.itemsize == 22 * 4
    assert runtime.KIND_DTYPE.itemsize == 8 * 4
    assert runtime.ITEM_DTYPE.itemsize == 24
    assert runtime.OUTER_DTYPE.itemsize == 40
    assert runtime.ERRIV_DTYPE.itemsize == 48
    assert C.sizeof(runtime.ModelDesc) == 29 * 4


def test_numpy_mirrors_agree_with_the_compiled_structs(tmp_path):
    """sizeof / offsetof of csrc/pk_abi.h and include/pockit_hip.h as g++ lays them out."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sizes.cpp"
    src.write_text(
        '#include <cstdio>\n#include <cstddef>\n#include "pockit_amd/csrc/pk_abi.h"\n#include "include/pockit_hip.h"\n'
        'int main() { std::printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(PkPhase), sizeof(PkTile), '
        "sizeof(PkKind), sizeof(PkItem), sizeof(PkOuter), sizeof(PkErrIv), offsetof(PkErrIv, out_off), "
        "offsetof(PkErrIv, width), sizeof(pk_model_desc), sizeof(pk_problem_desc)); }\n")
    exe = tmp_path / "sizes"
    subprocess.run(["g++", "-I", root, str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [runtime.PHASE_DTYPE.itemsize, runtime.TILE_DTYPE.itemsize, runtime.KIND_DTYPE.itemsize,
            runtime.ITEM_DTYPE.itemsize, runtime.OUTER_DTYPE.itemsize, runtime.ERRIV_DTYPE.itemsize,
            runtime.ERRIV_DTYPE.fields["out_off"][1], runtime.ERRIV_DTYPE.fields["width"][1],
            C.sizeof(runtime.ModelDesc), C.sizeof(runtime.ProblemDesc)]
    assert got == want


def test_no_gpu_means_loud_failure():
    lib = runtime.load_library()
    if lib.pk_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device|pk_create failed"):
        runtime.Context(0)
    import models
    import pockit_amd.radau as radau

    system, _, guess = models.brachistochrone(radau, 2, 3)
    x = models.pack_guess(system, guess)
    with pytest.raises(RuntimeError):
        system.objective(x)


def test_generated_code_cross_compiles_for_gfx950():
    import models
    import pockit_amd.radau as radau
    from pockit_amd.codegen import ModelSource

    system, _, _ = models.two_stage_rocket(radau, 3, 2)
    src = ModelSource(system.plan)
    code = hipbuild.compile_model(src.source)
    assert (code[:4] == b"\x7fELF" or code.startswith(b"__CLANG_OFFLOAD_BUNDLE__")) and len(code) > 1000
    for k in runtime.KERNELS:
        assert k.encode() in code

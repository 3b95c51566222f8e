"""Driver entry points: build() compiles every HIP artefact for gfx950; smoke() runs one tiny
NLP-callback cycle on cuda:0 through the C ABI and checks it against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bench_models():
    from pockit_amd import benchmarks as models
    import pockit_amd.radau as radau

    yield models.brachistochrone(radau, 3, 4)
    yield models.brachistochrone(radau, 200, 8)
    yield models.planar_quadrotor(radau, 2000, 6)
    yield models.two_stage_rocket(radau, 1000, 4)
    yield models.humanoid_wbc(radau, 50, 8)
    import pockit_amd.lobatto as lobatto

    yield models.planar_quadrotor(lobatto, 20, 6)


def build() -> None:
    """Compile the C-ABI runtime (libpockit_hip.so) and pre-compile the gfx950 code objects of the
    benchmark models into pockit_amd/_cache (hipcc cross-compiles without a GPU); import the package."""
    from pockit_amd import hipbuild, runtime
    from pockit_amd.codegen import ModelSource

    hipbuild.build_runtime(force=True)
    lib = runtime.load_library()
    for name in runtime.EXPORTS:
        getattr(lib, name)
    for system, _, _ in _bench_models():
        src = ModelSource(system.plan)
        hipbuild.compile_model(src.source, fastmath=system._fastmath)
    import oracle.radau  # noqa: F401  (the oracle is NumPy; nothing to compile)


def smoke() -> None:
    """One small invocation of the hot path on cuda:0, checked against the oracle."""
    import numpy as np

    from pockit_amd import benchmarks as models
    import oracle.radau
    import pockit_amd.radau as radau

    system, _, guess = models.brachistochrone(radau, 6, 5)
    ref, _, ref_guess = models.brachistochrone(oracle.radau, 6, 5)
    x, lam, sigma = models.bench_inputs(system, guess)

    def close(a, b):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape
        assert np.max(np.abs(a - b)) <= 1e-11 * max(1.0, np.max(np.abs(b))), np.max(np.abs(a - b))

    close(system.objective(x), ref.objective(x))
    close(system.gradient(x), ref.gradient(x))
    close(system.constraints(x), ref.constraints(x))
    close(system.jacobian(x), ref.jacobian(x))
    close(system.hessian(x, lam, sigma), ref.hessian(x, lam, sigma))
    jr, jc = system.jacobianstructure()
    rr, rc = ref.jacobianstructure()
    assert np.array_equal(jr, rr) and np.array_equal(jc, rc)
    print("smoke OK: f, grad f, g, J, H of brachistochrone LGR 6x5 match the oracle on", "cuda:0")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()

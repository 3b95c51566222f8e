"""The benchmark / parity models live in the product package (``pockit_amd.benchmarks``: bench.py and
``__graft_entry__`` must not depend on the test tree); the tests, the golden generator and the tools keep
importing them under this name."""
from pockit_amd.benchmarks import *  # noqa: F401,F403
from pockit_amd.benchmarks import _quadrotor_profiles, _humanoid_hands_numeric  # noqa: F401

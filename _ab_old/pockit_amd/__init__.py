"""MI355X-native evaluator for pockit's NLP-callback hot path.

``pockit_amd.radau`` / ``pockit_amd.lobatto`` expose pockit's ``System`` / ``Phase`` modeling API;
the objects implement the cyipopt ``problem_obj`` protocol with every callback evaluated by
generated + hand-written HIP kernels through the C ABI in ``include/pockit_hip.h``.
There is no CPU evaluation path in this package.
"""
__version__ = "0.1.0"

"""CPU oracle for the HIP backend -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's parity / cpu_baseline leg may import
this package -- plus the test tooling under scripts/ that is the same kind of checker run by
hand (fuzz_gpu.py, fuzz_reduce.py, accuracy_study.py, eig_sweep_histogram.py).  The product
package `nitorch_fastmath_amd` never does (tests/test_abi_host.py enforces it).
"""
from .oracle import *  # noqa: F401,F403

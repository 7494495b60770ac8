"""Run the host-grouping differential tests against a SANITIZER build of csrc/group_host.cpp (argv[1]); started by
tests/test_host_cpu.py::test_host_grouping_under_sanitizers with libasan preloaded.  Test scaffolding, not product code."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sleap_nn_amd import _lib as L

HOST = ("ph_lsap", "ph_toposort_edges", "ph_group_class_peaks", "ph_group_batch")


class Shim:
    """The sanitized host functions; anything else (device code) must not be reached from these tests."""

    def __init__(self, path):
        so = C.CDLL(path)
        for name in HOST:
            fn = getattr(so, name)
            fn.restype, fn.argtypes = L.SIGNATURES[name]
            setattr(self, name, fn)
        self.ph_last_error = so.ph_last_error
        self.ph_last_error.restype = C.c_char_p


L._lib = Shim(sys.argv[1])
from tests import test_host_cpu as T

T.test_lsap_matches_scipy_including_ties()
T.test_lsap_edge_cases()
T.test_toposort_matches_reference_vectors()
for name in ("chain5", "tree6", "chain13", "rev4"):
    T.test_group_batch_matches_reference_on_golden_candidates(name)
T.test_group_batch_matches_oracle_on_random_graphs()
T.test_group_class_peaks_matches_oracle()
print("sanitized host grouping: OK")

"""Shared helpers for the parity tests (configs of BASELINE.json, synthetic weights / inputs)."""
from drivescenegen_amd.configs import (CFG1, CFG2, CFG3, CFG4, CFG4_SMALL, CFG5, DEFAULT3, PARAM_COUNTS,  # noqa: F401
                                       noisy_inputs, synth_weights)


class same_kernels_at_any_batch:
    """Context: switch the small-batch split-K path off (dsg_set_tuning key 19), so that a batch-1 call selects the same
    kernels as the full batch and 'row i of the batch == the batch-1 call on row i' can be asserted BITWISE.  With the
    path on (the default) the batch-1 call contracts K in slices: same values to fp32 round-off, not the same bits."""

    def __enter__(self):
        from drivescenegen_amd import _lib
        _lib.check(_lib.load().dsg_set_tuning(19, 0))

    def __exit__(self, *exc):
        from drivescenegen_amd import _lib
        _lib.load().dsg_set_tuning(19, 1)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max())

"""Shared helpers for the parity tests (configs of BASELINE.json, synthetic weights / inputs)."""
from drivescenegen_amd.configs import (CFG1, CFG2, CFG3, CFG4, CFG4_SMALL, CFG5, DEFAULT3, PARAM_COUNTS,  # noqa: F401
                                       noisy_inputs, synth_weights)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max())

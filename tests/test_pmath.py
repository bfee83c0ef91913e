"""Portable math (csrc/uam_pmath.h, via the oracle build): accuracy against libm.  The bit-identity of
host and device results is what the GPU parity tests rely on (float state compared to the bit)."""
import math

import numpy as np


def ulp_err(got, want):
    return abs(got - want) / np.spacing(abs(want)) if want != 0 else abs(got)


def test_sincos_within_one_ulp(oracle):
    rs = np.random.RandomState(0)
    xs = np.concatenate([rs.uniform(-8, 8, 4000), rs.uniform(-1e3, 1e3, 3000), rs.uniform(-3e5, 3e5, 2000),
                         [0.0, math.pi / 4, math.pi / 2, math.pi, 2 * math.pi, -math.pi / 4, 1e-9, -1e-300]])
    worst = 0.0
    for x in xs:
        s, c = oracle.pmath_sincos(float(x))
        worst = max(worst, ulp_err(s, math.sin(x)), ulp_err(c, math.cos(x)))
    assert worst <= 1.0, worst


def test_log_within_one_ulp(oracle):
    rs = np.random.RandomState(1)
    xs = np.concatenate([rs.uniform(0, 1, 5000), 2.0 ** rs.uniform(-52, 0, 2000), [1.0, 0.5, 2.0 ** -53, 0.999999999]])
    xs = xs[xs > 0]
    worst = max(ulp_err(oracle.pmath_log(float(x)), math.log(x)) for x in xs)
    assert worst <= 1.0, worst


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(v) for v in oracle.philox4x32_10(ctr, key)) == want

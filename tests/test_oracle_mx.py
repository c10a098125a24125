"""The block-scaled e4m3 quantiser of oracle/fp8.py (the definition the HIP producers are tested against bit for bit, tests/test_gpu_mx8.py) against
an INDEPENDENT restatement: the e4m3 value grid built from the OCP format's definition (1 sign, 4 exponent bits with bias 7, 3 mantissa bits,
subnormals, no infinities, S.1111.111 = NaN, largest value 448), nearest value with ties to the even code, in float64; the block exponent from
math.frexp on exact rationals.  The reference has no fp8 path (example/sd1.py:33: fp32 throughout), so there is nothing of its own to pin this to."""
import math
from fractions import Fraction

import numpy as np

from oracle import fp8 as O8


def e4m3_grid():
    vals = []
    for code in range(128):                       # non-negative codes
        e, m = code >> 3, code & 7
        if e == 15 and m == 7:
            continue                              # NaN
        vals.append((2.0 ** -6) * (m / 8.0) if e == 0 else (2.0 ** (e - 7)) * (1 + m / 8.0))
    return np.array(vals, dtype=np.float64)       # ascending; index == code


def quant_e4m3_rne(x):
    """nearest e4m3 value, ties to the even code, saturating at 448 (callers never exceed it)."""
    g = e4m3_grid()
    a = np.abs(x.astype(np.float64))
    hi = np.searchsorted(g, a, side="left").clip(1, len(g) - 1)
    lo = hi - 1
    dl, dh = a - g[lo], g[hi] - a
    pick = np.where(dl < dh, lo, np.where(dh < dl, hi, np.where(lo % 2 == 0, lo, hi)))
    pick = np.where(a >= g[-1], len(g) - 1, pick)
    return np.sign(x) * g[pick], pick.astype(np.uint8) | ((x < 0) | ((x == 0) & np.signbit(x))).astype(np.uint8) << 7


def block_exponent(amax):
    """e = ceil(log2(amax / 448)) in exact arithmetic, on the fp32 quotient the device forms; -127 for a zero / subnormal quotient."""
    q = np.float32(amax) / np.float32(448.0)
    if q < np.float32(2.0 ** -126):
        return -127
    f = Fraction(float(q))
    m, e = math.frexp(float(q))                   # q = m 2^e, m in [0.5, 1)
    return e - 1 if f == Fraction(2) ** (e - 1) else e


def test_grid_is_the_ocp_e4m3_format():
    g = e4m3_grid()
    assert len(g) == 127 and g[0] == 0.0 and g[1] == 2.0 ** -9 and g[8] == 2.0 ** -6 and g[-1] == 448.0 and (np.diff(g) > 0).all()
    codes = np.arange(127, dtype=np.uint8)
    np.testing.assert_array_equal(O8.decode_e4m3(codes), g.astype(np.float32))          # torch's float8_e4m3fn decodes to the same grid


def test_quant_act_mx_matches_the_independent_restatement():
    rng = np.random.default_rng(5)
    rows, c = 64, 160
    x = rng.standard_normal((rows, c)).astype(np.float32)
    x *= np.exp2(rng.integers(-20, 12, size=(rows, c // 32))).repeat(32, axis=1).astype(np.float32)
    x[0, :32] = 0.0
    x[1, :32] = 0.0; x[1, 3] = 448.0
    x[2, :32] = 0.0; x[2, 4] = -896.0
    x[3, :32] = 0.0; x[3, 5] = 449.0
    x[4, :32] = np.float32(1e-38)                 # amax / 448 subnormal: scale byte 0
    x[5, :32] = np.linspace(-30000, 30000, 32, dtype=np.float32)
    x[6, :32] = 0.5 * (e4m3_grid()[40:72].astype(np.float32) + e4m3_grid()[41:73].astype(np.float32))     # exact ties between neighbouring codes (scale 2^0 below)
    x[6, 0] = 448.0
    x = x.astype(np.float16).astype(np.float32) if False else x
    got_codes, got_sc = O8.quant_act_mx_codes(x)
    got = O8.quant_act_mx(x).numpy()
    for r in range(rows):
        for b in range(c // 32):
            blk = x[r, 32 * b:32 * b + 32]
            e = block_exponent(np.abs(blk).max())
            assert got_sc[r, b] == e + 127, (r, b, got_sc[r, b], e)
            want, codes = quant_e4m3_rne(np.ldexp(blk.astype(np.float64), -e))
            np.testing.assert_array_equal(got[r, 32 * b:32 * b + 32].astype(np.float64), np.ldexp(want, e))
            live = blk != 0
            np.testing.assert_array_equal(got_codes[r, 32 * b:32 * b + 32][live], codes[live])
    assert got_sc[0, 0] == 0 and got_sc[1, 0] == 127 and got_sc[2, 0] == 128 and got_sc[3, 0] == 128 and got_sc[4, 0] == 0


def test_mx_gemm_supported_is_the_documented_rule():
    # one of the 192- / 256-row tiles must give the launch >= 128 blocks; GEGLU and the block-scaled output need the 128-wide tile;
    # a time-embedding bias needs HoWo >= the tile's rows; 256 x 160 only on the 128 grid
    assert O8.mx_gemm_supported(73728, 320, 2880, howo=9216, c_parts=(320,))
    assert O8.mx_gemm_supported(18432, 2560, 640, act=1, out_mx=True) and not O8.mx_gemm_supported(512, 2560, 640, act=1, out_mx=True)
    assert not O8.mx_gemm_supported(256, 1280, 1280) and not O8.mx_gemm_supported(4608, 1280, 1290) and not O8.mx_gemm_supported(4608, 1284, 1280)
    assert O8.mx_gemm_supported(24576, 160, 1280, c_parts=(1280,)) and not O8.mx_gemm_supported(24576, 160, 1280, howo=144)

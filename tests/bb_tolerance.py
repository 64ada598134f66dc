"""How EK80 broadband (pulse-compressed) outputs are judged against the oracle.

The reference's pulse-compressed samples are complex64 (ek80_complex.py:304) and its own tests compare dB values with
absolute tolerances of 2e-3 ... 5.5e-3 dB (test_calibrate_ek80_CW / _BB).  Here:
  float64 output (complex128 transform): 2e-4 dB for every sample within 60 dB of its ping's strongest echo;
  float32 output (complex64 transform, whose error is relative to the strongest echo of the 2048-sample tile, not to
      the sample): 2e-3 dB within 40 dB of the ping's peak, and the north-star's fp32 tolerance -- 1e-3 relative on the
      dB value -- within 60 dB;
  everything finite (float64) / within 90 dB of the ping's peak (float32: a complex64 transform has a noise floor
      about 120 dB below the tile's strongest echo): 0.5 dB (samples at the reference's own float32 noise floor)."""
import numpy as np


def assert_bb_close(got, exp, dtype):
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    fin = np.isfinite(exp)
    peak = np.nanmax(np.where(fin, exp, -np.inf), axis=2, keepdims=True)
    err = np.abs(got - exp)
    within = lambda db: fin & (exp > peak - db)  # noqa: E731
    if str(dtype) == "float64":
        assert err[within(60)].max() < 2e-4
    else:
        assert err[within(40)].max() < 2e-3
        s = within(60)
        assert (err[s] / np.maximum(np.abs(exp[s]), 1.0)).max() < 1e-3
    assert err[fin if str(dtype) == "float64" else within(90)].max() < 0.5

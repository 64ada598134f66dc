"""How EK80 broadband (pulse-compressed) outputs are judged against the oracle: in LINEAR power, with a bound that
follows the error model of either side.

  * The reference rounds every sector's pulse-compressed samples to complex64 (ek80_complex.py:304) before the sector
    mean: an amplitude error of up to ~6e-8 of the sector's |y_b| <= the strongest echo its transform saw.  The oracle
    does the same.
  * A transform (complex128 or complex64 here) returns every sample with an amplitude error delta = kappa * A_peak,
    A_peak = the strongest echo of the 2048-sample tile (every ping is tiled on its own: a ping's values never depend
    on its neighbours); the peak is taken per (channel, ping).
  => |prx_got / prx_exp - 1| <= c + 2 kappa sqrt(prx_peak / prx_exp): a constant for the arithmetic around the transform
     (logarithms, float32 rounding of the dB value) plus the peak-relative term.
       float64 output, complex128 transform: c = 2e-6 (5e-6 dB), kappa = 2e-7 (the REFERENCE's complex64 storage)
       float32 output, complex64 transform:  c = 1e-3 (the north-star's fp32 tolerance), kappa = 3e-5 (measured 6e-6:
                                             radix-8 passes whose twiddles come from product trees in float32)
     Samples whose bound exceeds 0.1 (0.4 dB) lie at the noise floor of one side or the other (100 dB under the peak for
     float64 against the reference's storage, 65 dB for a complex64 transform) and are held to the NaN pattern only.
``prx`` = the oracle's received power (same shape as the dB values): the ratio is taken there, the time-varied gain in
the dB values cancels in got - exp.  Without it the dB values themselves stand in (conservative where the gain grows with
range)."""
import numpy as np

PARAMS = {"float64": (2e-6, 2e-7), "float32": (1e-3, 3e-5)}


def assert_bb_close(got, exp, dtype, prx=None):
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    fin = np.isfinite(exp)
    lin = np.asarray(prx, np.float64) if prx is not None else 10.0 ** (exp / 10.0)
    lin = np.where(fin & (lin > 0), lin, np.nan)
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            peak = np.nanmax(lin, axis=-1, keepdims=True)
    c, kappa = PARAMS[str(dtype)]
    with np.errstate(invalid="ignore", over="ignore"):
        bound = c + 2.0 * kappa * np.sqrt(peak / lin)
        rel = np.abs(10.0 ** ((got - exp) / 10.0) - 1.0)
    judged = fin & (bound < 0.1)
    assert judged.sum() > 0.5 * fin.sum() or str(dtype) == "float32", "most samples must be judged"
    worst = np.nanmax(np.where(judged, rel / bound, 0.0))
    assert worst < 1.0, (str(dtype), float(worst), np.unravel_index(np.nanargmax(np.where(judged, rel / bound, 0.0)), rel.shape))
    return float(worst)

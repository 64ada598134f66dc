"""Oracle: seawater sound speed and absorption closed forms (test infrastructure).

Restates /root/reference/echopype/utils/uwa.py:8-53 (sound speed) and :56-189
(absorption).  Pinned by tests/golden/ref_leaf_goldens.npz (values produced by the
reference's own functions) and by the cross-formula tolerances
of /root/reference/echopype/tests/utils/test_utils_uwa.py:12-67.
"""
import numpy as np

__all__ = ["sound_speed", "absorption"]


def _poly(x, coeffs):
    """Horner evaluation, coeffs lowest order first."""
    acc = 0.0
    for c in reversed(coeffs):
        acc = acc * x + c
    return acc


def sound_speed(T=27, S=35, P=10, formula="Mackenzie"):
    """Sound speed [m/s]; T degC, S psu, P dbar.  (uwa.py:38-52)"""
    T = np.asarray(T, dtype=np.float64) if not np.isscalar(T) else T
    if formula == "Mackenzie":
        # nine-term Mackenzie (1981); reference groups it as three partial sums (:39-41)
        dS = S - 35
        c = 1448.96 + 4.591 * T - 5.304e-2 * T**2 + 2.374e-4 * T**3
        c = c + (1.340 * dS + 1.630e-2 * P + 1.675e-7 * P**2)
        c = c + (-1.025e-2 * T * dS - 7.139e-13 * T * P**3)
        return c
    if formula == "AZFP":
        # uwa.py:43-49 -- nested polynomial in z = T/10 and p = P/1000
        z = T / 10
        p = P / 1000
        return (
            1449.05
            + z * (45.7 + z * (-5.21 + 0.23 * z))
            + (1.333 + z * (-0.126 + z * 0.009)) * (S - 35.0)
            + p * (16.3 + 0.18 * p)
        )
    raise ValueError(f"unknown sound-speed formula {formula!r}")


def absorption(f_hz, T=27, S=35, P=10, pH=8.1, c=None, formula="AM"):
    """Absorption [dB/m] at frequency f_hz [Hz].  (uwa.py:110-189)"""
    f_hz = np.asarray(f_hz, dtype=np.float64) if not np.isscalar(f_hz) else f_hz
    if formula == "FG":  # Francois & Garrison 1982, uwa.py:110-144
        fk = f_hz / 1000.0
        cw = (1412.0 + 3.21 * T + 1.19 * S + 0.0167 * P) if c is None else c
        TK = T + 273
        # boric acid
        A1 = 8.86 / cw * 10 ** (0.78 * pH - 5)
        f1 = 2.8 * np.sqrt(S / 35) * 10 ** (4 - 1245 / TK)
        # magnesium sulphate
        A2 = 21.44 * S / cw * (1 + 0.025 * T)
        P2 = 1.0 - 1.37e-4 * P + 6.2e-9 * P**2
        f2 = 8.17 * 10 ** (8 - 1990 / TK) / (1 + 0.0018 * (S - 35))
        # pure water, two temperature branches (:124-137)
        P3 = 1.0 - 3.83e-5 * P + 4.9e-10 * P**2
        if np.all(T < 20):
            A3 = 4.937e-4 - 2.59e-5 * T + 9.11e-7 * T**2 - 1.5e-8 * T**3
        else:
            A3 = 3.964e-4 - 1.146e-5 * T + 1.45e-7 * T**2 - 6.5e-10 * T**3
        f_sq = fk**2
        per_km = (
            A1 * 1.0 * f1 * f_sq / (f_sq + f1**2)
            + A2 * P2 * f2 * f_sq / (f_sq + f2**2)
            + A3 * P3 * f_sq
        )
        return per_km / 1000
    if formula == "AM":  # Ainslie & McColm 1998, uwa.py:146-161
        fk = f_hz / 1000
        D = P / 1000
        f1 = 0.78 * np.sqrt(S / 35) * np.exp(T / 26)
        f2 = 42 * np.exp(T / 17)
        boric = 0.106 * (f1 * (fk**2)) / ((f1**2) + (fk**2)) * np.exp((pH - 8) / 0.56)
        mgso4 = (
            0.52 * (1 + T / 43) * (S / 35) * (f2 * (fk**2)) / ((f2**2) + (fk**2)) * np.exp(-D / 6)
        )
        water = 0.00049 * fk**2 * np.exp(-(T / 27 + D))
        return (boric + mgso4 + water) / 1000
    if formula == "AZFP":  # AZFP Matlab formula, uwa.py:163-187
        TK = T + 273.0
        f1 = 1320.0 * TK * np.exp(-1700 / TK)
        f2 = 1.55e7 * TK * np.exp(-3052 / TK)
        k = 1 + P / 10.0
        a = 8.95e-8 * (1 + T * (2.29e-2 - 5.08e-4 * T))
        b = (S / 35.0) * 4.88e-7 * (1 + 0.0134 * T) * (1 - 0.00103 * k + 3.7e-7 * k**2)
        cc = (
            4.86e-13
            * (1 + T * (-0.042 + T * (8.53e-4 - T * 6.23e-6)))
            * (1 + k * (-3.84e-4 + k * 7.57e-8))
        )
        fsq = f_hz**2
        if np.all(np.asarray(S) == 0):
            return cc * fsq
        return (a * f1 * fsq) / (f1**2 + fsq) + (b * f2 * fsq) / (f2**2 + fsq) + cc * fsq
    raise ValueError(f"unknown absorption formula {formula!r}")

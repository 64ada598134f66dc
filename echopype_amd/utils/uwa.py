"""Host-side seawater acoustics (per-channel / per-ping parameter preparation, O(C*P)).

Same call signatures as the reference's ``echopype.utils.uwa`` (/root/reference/echopype/utils/
uwa.py:8-53 calc_sound_speed, :56-189 calc_absorption) so user code and the calibrators can
call them unchanged.  These are scalar/per-ping closed forms evaluated on the host while the
sample-sized work runs on the GPU; tests check them against goldens produced by the reference.
"""
import numpy as np


def calc_sound_speed(temperature=27, salinity=35, pressure=10, formula_source="Mackenzie"):
    """Sound speed [m/s] from temperature [degC], salinity [PSU], pressure [dbar]."""
    t, s, p = temperature, salinity, pressure
    if formula_source == "Mackenzie":
        # Mackenzie (1981) nine-term equation; summed in the reference's order (uwa.py:39-41)
        first = 1448.96 + 4.591 * t - 5.304e-2 * t**2 + 2.374e-4 * t**3
        second = 1.340 * (s - 35) + 1.630e-2 * p + 1.675e-7 * p**2
        third = -1.025e-2 * t * (s - 35) - 7.139e-13 * t * p**3
        return (first + second) + third
    if formula_source == "AZFP":
        z, pk = t / 10, p / 1000
        return (
            1449.05
            + z * (45.7 + z * (-5.21 + 0.23 * z))
            + (1.333 + z * (-0.126 + z * 0.009)) * (s - 35.0)
            + pk * (16.3 + 0.18 * pk)
        )
    raise ValueError("Unknown formula source")


def _relax(amp, f_relax, f_sq):
    """One relaxation term A * f_r * f^2 / (f^2 + f_r^2), grouped as the reference writes it."""
    return amp * f_relax * f_sq / (f_sq + f_relax**2)


def calc_absorption(frequency, temperature=27, salinity=35, pressure=10, pH=8.1, sound_speed=None,
                    formula_source="AM"):
    """Sea-water absorption [dB/m] at ``frequency`` [Hz] (scalar or array)."""
    t, s, p = temperature, salinity, pressure
    if formula_source == "FG":
        f_sq = (frequency / 1000.0) ** 2
        c = 1412.0 + 3.21 * t + 1.19 * s + 0.0167 * p if sound_speed is None else sound_speed
        kelvin = t + 273
        boric = _relax(8.86 / c * 10 ** (0.78 * pH - 5) * 1.0,
                       2.8 * np.sqrt(s / 35) * 10 ** (4 - 1245 / kelvin), f_sq)
        mg = _relax(21.44 * s / c * (1 + 0.025 * t) * (1.0 - 1.37e-4 * p + 6.2e-9 * p**2),
                    8.17 * 10 ** (8 - 1990 / kelvin) / (1 + 0.0018 * (s - 35)), f_sq)
        if np.all(t < 20):
            a3 = 4.937e-4 - 2.59e-5 * t + 9.11e-7 * t**2 - 1.5e-8 * t**3
        else:
            a3 = 3.964e-4 - 1.146e-5 * t + 1.45e-7 * t**2 - 6.5e-10 * t**3
        water = a3 * (1.0 - 3.83e-5 * p + 4.9e-10 * p**2) * f_sq
        return (boric + mg + water) / 1000
    if formula_source == "AM":
        fk = frequency / 1000
        depth_km = p / 1000
        f1 = 0.78 * np.sqrt(s / 35) * np.exp(t / 26)
        f2 = 42 * np.exp(t / 17)
        a1 = 0.106 * (f1 * (fk**2)) / ((f1**2) + (fk**2)) * np.exp((pH - 8) / 0.56)
        a2 = 0.52 * (1 + t / 43) * (s / 35) * (f2 * (fk**2)) / ((f2**2) + (fk**2)) * np.exp(-depth_km / 6)
        a3 = 0.00049 * fk**2 * np.exp(-(t / 27 + depth_km))
        return (a1 + a2 + a3) / 1000
    if formula_source == "AZFP":
        kelvin = t + 273.0
        f1 = 1320.0 * kelvin * np.exp(-1700 / kelvin)
        f2 = 1.55e7 * kelvin * np.exp(-3052 / kelvin)
        k = 1 + p / 10.0
        a = 8.95e-8 * (1 + t * (2.29e-2 - 5.08e-4 * t))
        b = (s / 35.0) * 4.88e-7 * (1 + 0.0134 * t) * (1 - 0.00103 * k + 3.7e-7 * k**2)
        c = 4.86e-13 * (1 + t * (-0.042 + t * (8.53e-4 - t * 6.23e-6))) * (1 + k * (-3.84e-4 + k * 7.57e-8))
        fsq = frequency**2
        if np.all(np.asarray(s) == 0):
            return c * fsq
        return (a * f1 * fsq) / (f1**2 + fsq) + (b * f2 * fsq) / (f2**2 + fsq) + c * fsq
    raise ValueError("Unknown formula source")

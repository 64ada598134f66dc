"""AZFP calibrator (mirrors calibrate/calibrate_azfp.py:10-117 and range.py:11-95).

Host: env/cal parameters and one coefficient row per (channel, ping).  GPU: one epa_sv_power pass.
  R(s) = c*L/(2f) + (c/4) * (((2(s+1)-1)*N - 1)/f + tau) - offset        (range.py:81-89)
       = s * [c*N/(2f)]  +  [c*L/(2f) + (c/4)*((N-1)/f + tau) - offset]   ->  rb*s + r0
  EL = EL0 - 2.5/DS + counts/(26214*DS);  SL = TVR + 20 log10 VTX0       (calibrate_azfp.py:66-74)
  Sv = EL - SL + 20log10 R + 2 alpha R - 10log10(0.5 c tau psi) + Sv_offset ; TS = EL - SL + 40log10 R + 2 alpha R
"""
import numpy as np
import torch

from .. import _lib, ops
from ..echodata import BEAM1
from ..xr_lite import Dataset
from .cal_params import get_cal_params_AZFP
from .calibrate_base import ECHO_DIMS, CalibrateBase, cp_array
from .env_params import get_env_params_AZFP


class CalibrateAZFP(CalibrateBase):
    def __init__(self, echodata, env_params, cal_params, ecs_file=None, **kwargs):
        super().__init__(echodata, env_params, cal_params, ecs_file, **kwargs)
        self.sonar_type = "AZFP"
        self.ed_beam_group = BEAM1
        self.beam = self.echodata[BEAM1]
        self.vend = self.echodata["Vendor_specific"]
        self.env_params = get_env_params_AZFP(self.echodata, self.env_params)
        self.cal_params = get_cal_params_AZFP(self.beam, self.vend, self.cal_params)
        # range differs between Sv and TS, so it is computed inside _cal_power_samples (:31-32)

    def compute_echo_range(self, cal_type=None):
        if cal_type is None:
            raise ValueError('cal_type must be "Sv" or "TS"')
        if "sound_speed" not in self.env_params:
            raise RuntimeError(
                "sounds_speed not included in env_params, "
                "use echopype.calibrate.env_params.get_env_params_AZFP() to compute env_params "
                "by supplying temperature, salinity, and pressure.")

    def _rows(self, cal_type):
        C, P, S = self.beam["backscatter_r"].shape
        cp = lambda v, n: cp_array(v, C, P, n)  # noqa: E731
        cw = cp(self.env_params["sound_speed"], "sound_speed")
        alpha = cp(self.env_params["sound_absorption"], "sound_absorption")
        tau = cp(self.beam["transmit_duration_nominal"], "transmit_duration_nominal")
        N = cp(self.vend["number_of_samples_per_average_bin"], "number_of_samples_per_average_bin")
        f = cp(self.vend["digitization_rate"], "digitization_rate")
        L = cp(self.vend["lock_out_index"], "lock_out_index")
        EL, DS = cp(self.cal_params["EL"], "EL"), cp(self.cal_params["DS"], "DS")
        TVR, VTX0 = cp(self.cal_params["TVR"], "TVR"), cp(self.cal_params["VTX0"], "VTX0")
        psi = cp(self.cal_params["equivalent_beam_angle"], "equivalent_beam_angle")
        svoff = cp(self.cal_params["Sv_offset"], "Sv_offset")
        offset = 0.0 if cal_type == "Sv" else cw * tau / 4
        k = cw * N / (2 * f)
        r0 = cw * L / (2 * f) + (cw / 4) * ((N - 1) / f + tau) - offset
        with np.errstate(invalid="ignore", divide="ignore"):
            SL = TVR + 20 * np.log10(VTX0)
            A = EL - 2.5 / DS - SL
            if cal_type == "Sv":
                A = A - 10 * np.log10(0.5 * cw * tau * psi) + svoff
            n = 20.0 if cal_type == "Sv" else 40.0
            rows = np.zeros((C, P, _lib.NCOEF))
            rows[..., _lib.CF_RA] = 1.0
            rows[..., _lib.CF_RB] = k
            rows[..., _lib.CF_R0] = r0
            rows[..., _lib.CF_SHIFT] = 0.0
            rows[..., _lib.CF_ALPHA2] = 2 * alpha
            rows[..., _lib.CF_A0] = A + n * np.log10(k)
            rows[..., _lib.CF_G] = 1.0 / (26214 * DS)
            rows[..., _lib.CF_D] = -r0 / k
        self._reach_terms = (k, r0)
        return rows

    def _host_reach_bound(self, S):
        """Upper bound of every echo_range of the rows ``_rows`` made last: max((S - 1) k + r0) over (channel, ping) --
        the AZFP range starts at r0 = c L / (2 f) + (c / 4) ((N - 1) / f + tau) > 0 (range.py:81-89), which the EK
        bound of the base class does not know.  The rows are host arrays: no device reduction, no wait."""
        terms = getattr(self, "_reach_terms", None)
        if terms is None:
            return None
        k, r0 = terms
        with np.errstate(invalid="ignore"):
            reach = (S - 1) * k + r0
            m = float(np.fmax.reduce(reach, axis=None)) if reach.size else float("nan")
        if not (np.isfinite(m) and m > 0):
            return None
        return m * (1 + (1e-12 if str(self.dtype).endswith("64") else 1e-6))

    def _power_inputs(self, cal_type):
        if cal_type not in ("Sv", "TS"):
            raise ValueError("cal_type not recognized!")
        self.compute_echo_range(cal_type=cal_type)
        coef = self._dev(self._rows(cal_type), torch.float64)
        raw = self._dev(self.beam["backscatter_r"].data, torch.float32)
        return raw, coef, 0, None  # no R' <= 0 guard, echo_range not masked (calibrate_azfp.py)

    def _finish(self, cal_type, out_t, range_t, tau_eff=None, range_stats=None):
        ds = Dataset(coords={k: self.beam.coords[k] for k in ECHO_DIMS})
        ds[cal_type] = self._wrap(out_t, ECHO_DIMS)
        ds["echo_range"] = self._wrap(range_t, ECHO_DIMS, stats=range_stats)
        self.range_meter = ds["echo_range"]
        ds["frequency_nominal"] = self.beam["frequency_nominal"]
        return self._add_params_to_output(ds)

    def _cal_power_samples(self, cal_type, **kwargs):
        raw, coef, flags, _ = self._power_inputs(cal_type)
        out_t, range_t, stats = self._sv_power_lazy_range(raw, coef, cal_type, flags)
        return self._finish(cal_type, out_t, range_t, range_stats=stats)

    def compute_Sv(self, **kwargs):
        return self._cal_power_samples("Sv")

    def compute_TS(self, **kwargs):
        return self._cal_power_samples("TS")

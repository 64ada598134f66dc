"""Expected outputs for the synthetic echograms, computed with the CPU oracle only (no product
code): the parameter selection of the reference's calibrators restated for the synth dicts."""
import numpy as np

from oracle import calibrate as ocal
from oracle import ek80 as oek
from oracle import uwa as ouwa


def ek60(d, cal_type="Sv", env=None, gain=None):
    """compute_Sv/compute_TS on an EK60 file: env from the Environment group unless T,S,P,pH are
    all user-supplied (env_params.py:270-340); gain/sa by pulse-length lookup unless given."""
    C, P, S = d["backscatter_r"].shape
    if env and all(k in env for k in ("temperature", "salinity", "pressure", "pH")):
        ss = ouwa.sound_speed(env["temperature"], env["salinity"], env["pressure"], "Mackenzie")
        ab = ouwa.absorption(d["frequency_nominal"], env["temperature"], env["salinity"], env["pressure"],
                             env["pH"], c=ss, formula=env.get("formula_absorption", "FG"))
    else:
        ss, ab = d["sound_speed_indicative"], d["absorption_indicative"]
    g = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["gain_correction"]) \
        if gain is None else np.asarray(gain, float)
    sa = ocal.vend_cal_params_power(d["transmit_duration_nominal"], d["pulse_length"], d["sa_correction"])
    return ocal.cal_power_ek(
        d["backscatter_r"], sonar="EK60", cal_type=cal_type, sample_interval=d["sample_interval"],
        sound_speed=ss, absorption=ab, transmit_power=d["transmit_power"],
        tau_nominal=d["transmit_duration_nominal"], gain=g, sa_correction=sa, psi=d["equivalent_beam_angle"],
        f_nominal=d["frequency_nominal"], tau_eff=d["transmit_duration_nominal"][:, 0])


def ek80_replicas(d, filters, waveform):
    reps, teff = [], []
    for c in range(len(d["frequency_nominal"])):
        f0 = d["f_start"][c] if waveform == "BB" else d["frequency_nominal"][c]
        f1 = d["f_stop"][c] if waveform == "BB" else d["frequency_nominal"][c]
        y, t = oek.transmit_replica(d["fs"][c], d["tau"][c], d["slope"][c], f0, f1, filters)
        reps.append(y)
        teff.append(float(np.ravel(oek.tau_effective(y, 1 / np.diff(t[:2]), waveform))[0]))
    return reps, np.array(teff)


def ek80_complex(d, filters, cal_type="Sv"):
    """EK80 complex (BB or CW) as CalibrateEK80 would run it on the synth file: env T/S/depth/pH from
    the Environment group -> FG absorption at the centre frequency with the file sound speed
    (env_params.py:284-340); gain from the narrowband table minus B_theta_phi_m for BB."""
    wf = d.get("waveform", "BB")
    C, P, S, B = d["backscatter_r"].shape
    reps, teff = ek80_replicas(d, filters, wf)
    fnom = d["frequency_nominal"]
    fc = (d["f_start"] + d["f_stop"]) / 2 if wf == "BB" else fnom
    ss = float(np.asarray(d["sound_speed"]).flat[0])  # Environment.sound_speed_indicative, one time1
    ab = ouwa.absorption(fc, 10.0, 35.0, 10.0, 8.0, c=ss, formula="FG")
    tau = np.tile(d["tau"][:, None], (1, P))
    pl = np.tile(np.array([256e-6, 512e-6, 1024e-6, 2048e-6, 4096e-6]), (C, 1))
    g0 = d["gain"]
    gtab = np.stack([g0 - 1, g0 - .5, g0, g0 + .2, g0 + .3], axis=1)
    gain = ocal.vend_cal_params_power(tau, pl, gtab)
    sa = np.tile(d["sa"][:, None], (1, P))
    if wf == "BB":
        bw_a = d["beamwidth_alongship"] * fnom / fc
        bw_t = d["beamwidth_athwartship"] * fnom / fc
        gain = gain - ocal.b_theta_phi_m(d["angle_offset_alongship"], d["angle_offset_athwartship"], bw_a, bw_t)[:, None]
        psi = d["psi"] + 20 * np.log10(fnom / fc)
    else:
        psi = d["psi"]
    return ocal.cal_complex_ek80(
        d["backscatter_r"], d["backscatter_i"], waveform_mode=wf, cal_type=cal_type,
        sample_interval=d["sample_interval"], sound_speed=ss, absorption=ab,
        transmit_power=d["transmit_power"], tau_nominal=tau, gain=gain, sa_correction=sa, psi_fc=psi,
        f_center=fc, tau_eff=teff, z_er=d["z_er"], z_et=d["z_et"], replicas=reps), teff


def azfp(d, cal_type="Sv"):
    T, S_, P_ = d["temperature"], d["salinity"], d["pressure"]
    ss = ouwa.sound_speed(T, S_, P_, "AZFP")  # (P,)
    ab = ouwa.absorption(d["frequency_nominal"][:, None], T[None, :], S_, P_, formula="AZFP")  # (C,P)
    C, P, S = d["backscatter_r"].shape
    return ocal.cal_azfp(
        d["backscatter_r"], cal_type=cal_type, sound_speed=np.tile(ss, (C, 1)), absorption=ab,
        tau=d["transmit_duration_nominal"], n_avg=d["number_of_samples_per_average_bin"],
        dig_rate=d["digitization_rate"], lockout=d["lock_out_index"], EL=d["EL"], DS=d["DS"], TVR=d["TVR"],
        VTX0=d["VTX0"], psi_lin=d["equivalent_beam_angle"], Sv_offset=d["Sv_offset"])

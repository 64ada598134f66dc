"""estimate_background_noise / remove_background_noise with the reference's signatures
(/root/reference/echopype/clean/api.py:362-433, :436-511; De Robertis & Higginbottom 2007).
The per-sample work is two HIP kernel launches (epa_noise_estimate, epa_noise_apply).
``remove_background_noise`` adds ``Sv_noise`` and ``Sv_corrected`` to the CALLER's dataset, as the
reference does (api.py:490-502).
"""
import numpy as np
import torch

from .. import ops
from ..commongrid.api import _dev, _full
from ..utils.prov import echopype_prov_attrs, insert_processing_level
from ..xr_lite import DataArray, DeviceArray, from_xarray
from .utils import add_remove_background_noise_attrs, extract_dB


def _inputs(ds_Sv):
    sv_da = ds_Sv["Sv"]
    order = tuple(sv_da.dims)
    sv_t = _dev(sv_da)
    if sv_t.dtype not in (torch.float32, torch.float64):
        sv_t = sv_t.double()
    rg_t = _dev(_full(ds_Sv["echo_range"], ds_Sv, order), sv_t.dtype)
    C, P, S = sv_t.shape
    a = ds_Sv["sound_absorption"]
    av = np.asarray(a.values, dtype=np.float64)
    if av.ndim == 0:
        a2 = np.full((C, P), 2 * float(av))
    elif av.ndim == 1:
        a2 = np.broadcast_to((2 * av)[:, None] if a.dims == (order[0],) or av.shape[0] == C else (2 * av)[None, :], (C, P))
    else:
        a2 = 2 * (av if a.dims[0] == order[0] else av.T)
    return order, sv_t, rg_t, ops.to_device(np.ascontiguousarray(a2, dtype=np.float64))


def _estimate(ds_Sv, ping_num, range_sample_num, background_noise_max):
    if background_noise_max is not None:
        background_noise_max = extract_dB(background_noise_max)
    order, sv_t, rg_t, a2 = _inputs(ds_Sv)
    nmax = float("nan") if background_noise_max is None else float(background_noise_max)
    noise = ops.noise_estimate(sv_t, a2, ping_num, range_sample_num, range=rg_t, noise_max=nmax)
    return order, sv_t, rg_t, a2, noise, background_noise_max


def estimate_background_noise(ds_Sv, ping_num, range_sample_num, background_noise_max=None):
    """Sv_noise (same shape as Sv): block-minimum noise + transmission loss (api.py:392-431)."""
    ds_Sv = from_xarray(ds_Sv)
    order, sv_t, rg_t, a2, noise, _ = _estimate(ds_Sv, ping_num, range_sample_num, background_noise_max)
    sn, _ = ops.noise_apply(sv_t, a2, noise, ping_num, 0.0, range=rg_t, want_corrected=False)
    return DataArray(DeviceArray(sn), order, {d: ds_Sv[d].values for d in order if d in ds_Sv.coords},
                     name="Sv_noise")


def remove_background_noise(ds_Sv, ping_num, range_sample_num, background_noise_max=None,
                            SNR_threshold="3.0dB"):
    """Adds Sv_noise and Sv_corrected to ``ds_Sv`` and returns it (api.py:472-511)."""
    ds_Sv = from_xarray(ds_Sv)
    if SNR_threshold is not None:
        SNR_threshold = extract_dB(SNR_threshold)
    order, sv_t, rg_t, a2, noise, nmax = _estimate(ds_Sv, ping_num, range_sample_num, background_noise_max)
    sn, sc = ops.noise_apply(sv_t, a2, noise, ping_num, float(SNR_threshold), range=rg_t)
    for name, t, kind in (("Sv_noise", sn, "noise"), ("Sv_corrected", sc, "corrected")):
        da = DataArray(DeviceArray(t), order)
        ds_Sv[name] = add_remove_background_noise_attrs(da, kind, ping_num, range_sample_num, SNR_threshold,
                                                        nmax, ops.nanminmax(t))
    prov = echopype_prov_attrs(process_type="processing")
    prov["processing_function"] = "clean.remove_background_noise"
    ds_Sv.attrs.update(prov)
    return insert_processing_level(ds_Sv, "L*B", input_ds=ds_Sv)


# names used by older echopype releases (docs/source/whats-new.md:366)
estimate_noise = estimate_background_noise
remove_noise = remove_background_noise

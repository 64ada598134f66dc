"""Test files cut into ping shards (shared by the CPU gloo tests and the multi-rank GPU tests): an EK80 broadband file
whose second channel starts recording late (NaN pulse length on its first pings -- the first valid ping, the filter
interval starts and the replica parameters are then facts of the WHOLE file that most shards cannot see), with one
filter set or several filter_time stamps; an AZFP file."""
import numpy as np

import echopype_amd as ep
from echopype_amd.echodata import BEAM1, EchoData

BEAM_PING_VARS = ("backscatter_r", "backscatter_i", "sample_interval", "transmit_duration_nominal", "transmit_power",
                  "slope", "transmit_frequency_start", "transmit_frequency_stop", "transmit_type")


def ek80_bb_file(P=40, S=256, B=4, late=12, filter_pings=None, seed=5):
    """Whole-file EK80 BB EchoData (host arrays).  Channel 1 has NaN transmit parameters and NaN samples on its first
    ``late`` pings.  ``filter_pings``: ping indices of the filter_time stamps (None: one filter set, no filter_time)."""
    from echopype_amd.calibrate.ek80_complex import filter_decimate_chirp, tapered_chirp

    filt = ep.synth.ek80_filters()
    reps = []
    for c in range(2):
        y, _ = tapered_chirp(1.5e6, ep.synth.EK80_BB["tau"][c], 0.05, ep.synth.EK80_BB["f_start"][c], ep.synth.EK80_BB["f_stop"][c])
        reps.append(filter_decimate_chirp(filt, y, 1.5e6)[0])
    d = ep.synth.ek80_numpy(2, P, S, B, seed=seed, replicas=reps)
    ed = ep.echodata.from_ek80_arrays(d, filt, filter_time_idx=filter_pings)
    beam = ed[BEAM1]
    for name in ("transmit_duration_nominal", "slope", "transmit_frequency_start", "transmit_frequency_stop"):
        a = np.array(beam[name].values, dtype=np.float64)
        a[1, :late] = np.nan
        beam[name] = (beam[name].dims, a)
    for name in ("backscatter_r", "backscatter_i"):
        a = np.array(beam[name].values)
        a[1, :late] = np.nan
        beam[name] = (beam[name].dims, a)
    return ed


def azfp_file(P=70, S=300, seed=3):
    d = ep.synth.azfp_numpy(4, P, S, seed=seed)
    d["temperature"] = 8.0 + 0.01 * np.arange(P)  # (sound speed / absorption then differ from ping to ping)
    return ep.echodata.from_azfp_arrays(d), dict(salinity=d["salinity"], pressure=d["pressure"])


def shard_of(ed, p0, p1):
    """The EchoData a converter run on pings [p0, p1) of the file would give: the beam group cut along ping_time, the
    small groups (Vendor_specific with ALL its filter_time stamps, Environment, Sonar) whole."""
    groups = {k: ed[k] for k in ed.group_paths}
    groups[BEAM1] = ed[BEAM1].isel(ping_time=slice(p0, p1))
    return EchoData(ed.sonar_model, groups, source_file=ed.source_file)

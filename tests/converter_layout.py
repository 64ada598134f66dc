"""EchoData objects shaped like what echopype's converter writes (TEST INFRASTRUCTURE): a DataTree-like container
(tests/fake_xarray.py::TreeEchoData) whose groups are "xarray" Datasets with the converter's variables, dimensions and
dtypes -- not only the handful the calibrators read:

  EK60 (convert/set_groups_ek60.py:80-152, 578-726, 728-787): Beam_group1 with float32 backscatter_r AND the angle planes,
        byte-typed data_type / channel_mode, transmit_bandwidth, sample_time_offset, beam_type ...; Environment on
        (channel, time1) with its own frequency_nominal; Vendor_specific pulse-length tables + frequency_nominal; Sonar,
        Platform (time1 / time2 axes), Provenance and Top-level groups present.
  EK80 (set_groups_ek80.py:796-840, 1234-1518): complex samples as FLOAT64 r / i planes (parse_base.py:306-309) with a
        string-labelled ``beam`` dimension, int64 range_sample, transmit_type strings, the filter coefficients NaN-padded
        on (channel, filter_time, n), Environment on a single time1, Sonar with beam_group_descr / waveform_encode_descr.
  AZFP (set_groups_azfp.py:417-466, 583-608, 736-770): float64 counts, temperature on time1, tilt variables.
Built from echopype_amd's own synthetic EchoData (same numbers), so results can be compared bit for bit.

``use(module)`` switches the container library: tests/fake_xarray.py (default; xarray cannot be installed in the build
image) or the REAL ``xarray`` -- the groups are then ``xarray.Dataset`` objects in an ``xarray.DataTree``, read through
the same thin EchoData wrapper echopype puts around its tree (echodata/echodata.py:43-346)."""
import numpy as np

import fake_xarray as fx

_FAKE = fx


def use(module):
    """Select the container library for the builders below; returns the previous one."""
    global fx
    prev, fx = fx, module
    return prev


class RealTreeEchoData:
    """The read API of echopype's EchoData over a real ``xarray.DataTree`` (echodata.py:327-335: ``ed[path]`` gives the
    group's Dataset, None for a group the file does not have)."""

    def __init__(self, sonar_model, tree, source_file=None):
        self.sonar_model, self.source_file, self.converted_raw_path = sonar_model, source_file, None
        self._tree = tree

    @property
    def group_paths(self):
        return ["Top-level"] + [g.lstrip("/") for g in self._tree.groups if g != "/"]

    def __getitem__(self, key):
        if key in (None, "Top-level"):
            return self._tree.to_dataset()
        try:
            return self._tree[key].to_dataset()
        except KeyError:
            return None


def _fake(ds, extra=None, coord_map=None):
    """lite Dataset -> fake-xarray Dataset (+ extra variables {name: (dims, values[, attrs])})."""
    coords = {}
    for k, c in ds.coords.items():
        v = np.asarray(c.values)
        if coord_map and k in coord_map:
            v = coord_map[k](v)
        coords[k] = (c.dims, v, dict(c.attrs))
    out = fx.Dataset({k: (v.dims, np.asarray(v.values), dict(v.attrs)) for k, v in ds.data_vars.items()}, coords=coords,
                     attrs=dict(ds.attrs))
    for k, v in (extra or {}).items():
        out[k] = v
    return out


def _common_groups(sonar_model, ping_time, extra_sonar=None):
    P = len(ping_time)
    plat = fx.Dataset({"latitude": (("time1",), np.linspace(47.0, 47.1, P)),
                       "longitude": (("time1",), np.linspace(-125.0, -124.9, P)),
                       "pitch": (("time2",), np.zeros(P)), "roll": (("time2",), np.zeros(P)),
                       "vertical_offset": (("time2",), np.zeros(P)),
                       "water_level": ((), np.float64(0.0))},
                      coords={"time1": (("time1",), ping_time), "time2": (("time2",), ping_time)})
    prov = fx.Dataset(attrs={"conversion_software_name": "echopype", "conversion_software_version": "0.10.x"})
    top = fx.Dataset(attrs={"conventions": "CF-1.7, SONAR-netCDF4-1.0, ACDD-1.3", "keywords": sonar_model})
    return plat, prov, top


def _tree(sonar_model, top, groups, source_file):
    if fx is not _FAKE:  # real xarray: a DataTree of the groups
        tree = fx.DataTree.from_dict({"/": top, **{"/" + path: ds for path, ds in groups.items()}})
        return RealTreeEchoData(sonar_model, tree, source_file=source_file)
    root = fx.DataTreeNode(top)
    for path, ds in groups.items():
        node = root
        parts = path.split("/")
        for part in parts[:-1]:
            node = node.children.setdefault(part, fx.DataTreeNode())
        if parts[-1] in node.children:
            node.children[parts[-1]]._ds = ds
        else:
            node.children[parts[-1]] = fx.DataTreeNode(ds)
    return fx.TreeEchoData(sonar_model, root, source_file=source_file)


def ek60(lite_ed, rng=None):
    rng = rng or np.random.default_rng(0)
    beam, vend, env = lite_ed["Sonar/Beam_group1"], lite_ed["Vendor_specific"], lite_ed["Environment"]
    C, P, S = beam["backscatter_r"].shape
    cp, cps, ch = ("channel", "ping_time"), ("channel", "ping_time", "range_sample"), ("channel",)
    assert np.asarray(beam["backscatter_r"].values).dtype == np.float32
    extra = {
        "angle_athwartship": (cps, rng.integers(-128, 127, (C, P, S)).astype(np.float32)),
        "angle_alongship": (cps, rng.integers(-128, 127, (C, P, S)).astype(np.float32)),
        "transmit_bandwidth": (cp, np.full((C, P), 2425.15)),
        "data_type": (cp, np.full((C, P), 3, dtype=np.byte)),
        "channel_mode": (cp, np.zeros((C, P), dtype=np.byte)),
        "sample_time_offset": (cp, np.zeros((C, P))),
        "beam_type": (ch, np.ones(C, dtype=np.int64)),
        "beamwidth_twoway_alongship": (ch, np.full(C, 7.0)), "beamwidth_twoway_athwartship": (ch, np.full(C, 7.0)),
        "angle_offset_alongship": (ch, np.zeros(C)), "angle_offset_athwartship": (ch, np.zeros(C)),
        "angle_sensitivity_alongship": (ch, np.full(C, 21.9)), "angle_sensitivity_athwartship": (ch, np.full(C, 21.9)),
        "gain_correction": (ch, np.full(C, 26.0)),
        "transceiver_software_version": (ch, np.array(["070413"] * C)),
    }
    fb = _fake(beam, extra, {"range_sample": lambda v: v.astype(np.int64)})
    fb.attrs.update(beam_mode="vertical", conversion_equation_t="type_3")
    fnom = np.asarray(beam["frequency_nominal"].values)
    fe = _fake(env, {"frequency_nominal": (ch, fnom)})
    fv = _fake(vend, {"frequency_nominal": (ch, fnom)})
    pt = np.asarray(beam["ping_time"].values)
    plat, prov, top = _common_groups("EK60", pt)
    sonar = fx.Dataset({"beam_group_descr": (("beam_group",), np.array(["contains backscatter power (uncalibrated) ..."]))},
                       coords={"beam_group": (("beam_group",), np.array(["Beam_group1"]))},
                       attrs={"sonar_manufacturer": "Simrad", "sonar_model": "EK60", "sonar_type": "echosounder"})
    return _tree("EK60", top, {"Environment": fe, "Platform": plat, "Provenance": prov, "Sonar": sonar,
                               "Sonar/Beam_group1": fb, "Vendor_specific": fv}, lite_ed.source_file)


def ek80(lite_ed):
    beam, vend, env, sonar0 = (lite_ed[g] for g in ("Sonar/Beam_group1", "Vendor_specific", "Environment", "Sonar"))
    assert "filter_time" in vend.sizes, "build the lite EchoData with filter_time_idx=[0]: the converter always writes it"
    r = np.asarray(beam["backscatter_r"].values)
    assert r.dtype == np.float64 and r.ndim == 4  # parse_base.py:306-309
    C, P = r.shape[:2]
    cp = ("channel", "ping_time")
    extra = {"transmit_bandwidth": (cp, np.full((C, P), np.nan)), "data_type": (cp, np.full((C, P), 8, dtype=np.byte)),
             "channel_mode": (cp, np.zeros((C, P), dtype=np.byte)), "sample_time_offset": (cp, np.zeros((C, P))),
             "beam_type": (("channel",), np.ones(C, dtype=np.int64))}
    fb = _fake(beam, extra, {"range_sample": lambda v: v.astype(np.int64),
                             "beam": lambda v: (np.arange(len(v)) + 1).astype(str)})  # set_groups_ek80.py:829-833
    fe = _fake(env)
    fv = _fake(vend, {"frequency_nominal": (("channel",), np.asarray(beam["frequency_nominal"].values))})
    pt = np.asarray(beam["ping_time"].values)
    plat, prov, top = _common_groups("EK80", pt)
    descr = np.asarray(sonar0["waveform_encode_descr"].values)
    sonar = fx.Dataset({"beam_group_descr": (("beam_group",), np.array(["contains complex backscatter data ..."])),
                        "waveform_encode_descr": (("beam_group",), descr),
                        "transducer_serial_number": (("channel",), np.array(["123"] * C))},
                       coords={"beam_group": (("beam_group",), np.array(["Beam_group1"]))},
                       attrs={"sonar_manufacturer": "Simrad", "sonar_model": "EK80", "sonar_type": "echosounder"})
    return _tree("EK80", top, {"Environment": fe, "Platform": plat, "Provenance": prov, "Sonar": sonar,
                               "Sonar/Beam_group1": fb, "Vendor_specific": fv}, lite_ed.source_file)


def azfp(lite_ed):
    beam, vend, env = lite_ed["Sonar/Beam_group1"], lite_ed["Vendor_specific"], lite_ed["Environment"]
    C, P, S = beam["backscatter_r"].shape
    cp = ("channel", "ping_time")
    fb = _fake(beam, {"sample_interval": (cp, np.full((C, P), 1.5625e-5)), "transmit_bandwidth": (cp, np.full((C, P), np.nan)),
                      "tilt_x": (("ping_time",), np.zeros(P)), "tilt_y": (("ping_time",), np.zeros(P)),
                      "beam_type": (("channel",), np.zeros(C, dtype=np.int64))},
               {"range_sample": lambda v: v.astype(np.int64)})
    pt = np.asarray(beam["ping_time"].values)
    plat, prov, top = _common_groups("AZFP", pt)
    sonar = fx.Dataset(attrs={"sonar_manufacturer": "ASL Environmental Sciences", "sonar_model": "AZFP"})
    return _tree("AZFP", top, {"Environment": _fake(env), "Platform": plat, "Provenance": prov, "Sonar": sonar,
                               "Sonar/Beam_group1": fb, "Vendor_specific": _fake(vend)}, lite_ed.source_file)

"""CPU oracle for the echopype hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a NumPy fp64 restatement of the reference algorithm
(OSOceanAcoustics/echopype @ 2026-05-01, /root/reference) for the path

    calibrate.compute_Sv / compute_TS  (EK60 CW, EK80 CW power/complex, EK80 BB, AZFP)
    -> clean.estimate_background_noise / remove_background_noise
    -> commongrid.compute_MVBS / compute_MVBS_index_binning

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``echopype_amd/`` imports it, and the product
path fails loudly when the HIP library is missing (see echopype_amd/_lib.py).

Pinning status (see DESIGN.md "Oracle"):
  * leaf functions (uwa formulas, transmit replica, filter/decimate, tau_effective,
    norm factor, matched-filter convolution) are pinned against outputs of the
    reference's OWN code executed in the authoring container through a stub-module
    loader (oracle/gen_goldens.py -> tests/golden/ref_leaf_goldens.npz);
  * the Sv/TS chains (EK60, EK80 power incl. GPT, AZFP, EK80 complex CW / BB) are pinned
    against outputs of the reference's OWN calibrator methods, executed over a strict
    named-dimension array shim (oracle/xr_shim.py, oracle/gen_chain_goldens.py ->
    tests/golden/ref_chain_goldens.npz); the noise-mask leaf functions likewise
    (oracle/gen_mask_goldens.py -> tests/golden/ref_mask_goldens.npz) and the three mask API
    functions end to end over the shim (oracle/gen_maskapi_goldens.py ->
    tests/golden/ref_maskapi_goldens.npz);
  * noise removal likewise: the reference's own estimate / remove_background_noise run over
    the shim (oracle/gen_noise_goldens.py -> tests/golden/ref_noise_goldens.npz);
  * compute_MVBS_index_binning (coarsen) and the bin-string parsers run from the reference too
    (oracle/gen_mvbs_index_goldens.py -> tests/golden/ref_mvbs_index_goldens.npz);
  * the EchoData-driven inputs of add_depth likewise (oracle/gen_depth_goldens.py ->
    tests/golden/ref_depth_goldens.npz);
  * MVBS / NASC (flox group-bys) cannot be executed from
    the reference here, so they are pinned against the reference's synthetic
    known-answer tests restated in tests/test_oracle_kat.py, test_oracle_masks.py,
    test_oracle_nasc.py (noise seed-1 => 6 NaNs, pulse-length lookup tables, MVBS
    brute-force values/NaN masks, index-binning coarsen formula, Echoview NASC value, ...).

Every function cites the reference file:line it follows.  The pass structure of the
reference (whole-array temporaries, np.log10 / 10**x, scipy.signal.convolve per
(ping, beam) slab, group-by with bincount) is kept on purpose so that timing this
code is a fair stand-in for the reference CPU path (bench.py cpu_baseline, kind="port").
"""

from . import uwa, ek80, calibrate, clean, commongrid, masks, nasc  # noqa: F401

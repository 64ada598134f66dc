#!/usr/bin/env python3
"""Read the signatures of the reference's public functions on the hot path with ``ast`` (no import: the reference
needs xarray / dask / flox and Python >= 3.11) and write them to tests/golden/ref_signatures.json -- data only: for each
function the ordered list of [name, kind, default-as-source-text].  ``compute_Sv`` / ``compute_TS`` are
``(echodata, **kwargs)`` forwarding to ``_compute_cal`` (calibrate/api.py:23-33, 345, 449): their entry lists the
parameters of ``_compute_cal`` after ``cal_type``, ``echodata`` as keyword-only ones -- what a caller can actually pass.
Authoring container only (needs /root/reference); tests/test_signatures.py compares ``inspect.signature`` of every
drop-in function with this file."""
import ast
import json
import os

REF = "/root/reference/echopype"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_signatures.json")
WANTED = {
    "calibrate/api.py": ["_compute_cal", "compute_Sv", "compute_TS"],
    "commongrid/api.py": ["compute_MVBS", "compute_MVBS_index_binning", "compute_NASC"],
    "clean/api.py": ["estimate_background_noise", "remove_background_noise", "mask_transient_noise", "mask_impulse_noise",
                     "mask_attenuated_signal"],
    "consolidate/api.py": ["add_depth", "swap_dims_channel_frequency"],
    "mask/api.py": ["apply_mask"],
}


def params(fn):
    a = fn.args
    out = []
    pos = a.posonlyargs + a.args
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    for arg, d in zip(pos, defaults):
        out.append([arg.arg, "positional_or_keyword", None if d is None else ast.unparse(d)])
    if a.vararg:
        out.append([a.vararg.arg, "var_positional", None])
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append([arg.arg, "keyword_only", None if d is None else ast.unparse(d)])
    if a.kwarg:
        out.append([a.kwarg.arg, "var_keyword", None])
    return out


def main():
    sigs = {}
    for rel, names in WANTED.items():
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
        mod = rel.replace("/api.py", "")
        for name in names:
            sigs[f"{mod}.{name}"] = {"params": params(fns[name]), "line": fns[name].lineno, "file": rel}
    cal = sigs.pop("calibrate._compute_cal")
    for name in ("compute_Sv", "compute_TS"):
        own = sigs[f"calibrate.{name}"]
        assert [p[:2] for p in own["params"]] == [["echodata", "positional_or_keyword"], ["kwargs", "var_keyword"]]
        own["params"] = [own["params"][0]] + [[n, "keyword_only", d] for n, _, d in cal["params"][2:]]
        own["forwards_to"] = f"_compute_cal (calibrate/api.py:{cal['line']})"
    json.dump(sigs, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", os.path.normpath(OUT), len(sigs), "signatures")
    for k, v in sigs.items():
        print(f"  {k}({', '.join(p[0] + ('=' + p[2] if p[2] is not None else '') for p in v['params'])})")


if __name__ == "__main__":
    main()

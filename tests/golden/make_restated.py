"""Refreshes the "restated" fields of tests/golden/reference_kats.json from oracle/cloudy_oracle.c.

The "expected" fields are DATA transcribed from the reference's unit tests (file:line cited per entry)
and are never touched by this script; "restated" is what the C oracle produced in the build container,
kept so that a change in the oracle's numerics shows up as a test failure.
Run:  python tests/golden/make_restated.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cloudy_oracle as O  # noqa: E402

TYPES = {"exponential": 0, "gamma": 1, "monodisperse": 2, "lognormal": 3}


def mk(spec):
    return O.make_dist(TYPES[spec[0]], *spec[1:])


def main():
    path = os.path.join(ROOT, "tests", "golden", "reference_kats.json")
    d = json.load(open(path))
    for e in d["moment_source_helper"]:
        if "restated" in e:
            e["restated"] = O.moment_source_helper(mk(e["dist"]), e["p1"], e["p2"], e["x_threshold"],
                                                   e["n_bins_per_log_unit"])
    for e in d["compute_thresholds"]:
        pd = [mk(s) for s in e["pdists"]]
        if "thresholds_default" in e:
            e["restated"] = float(O.compute_thresholds(pd)[0])
        if "percentiles" in e:
            e["restated"] = float(O.compute_thresholds(pd, e["percentiles"])[0])
    e = d["gamma_exp_coal_ints"]
    thr = [np.inf if t == "inf" else t for t in e["thresholds"]]
    cd = O.coalescence_data(np.array(e["kernel_c"]), e["NProgMoms"], thr)
    e["restated"] = [float(v) for v in O.get_coal_ints([mk(s) for s in e["pdists"]], cd)]
    json.dump(d, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()

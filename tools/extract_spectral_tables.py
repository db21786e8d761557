#!/usr/bin/env python3
"""Extract the spectral coefficient TABLES (data, not code) the GORT path needs.

Run in the build container only (reads /root/reference; the GPU box never does):

    python tools/extract_spectral_tables.py

Outputs (committed, small):
  gort_amd/data/prospect_d_coeffs.f32   7 x 2101 float32, row order
        refractive, k_Cab, k_Car, k_Anth, k_Brown, k_Cw, k_Cm  (400..2500 nm @ 1 nm)
        Source: PROSPECT-D v6.0 (Feret, Gitelson, Noble & Jacquemoud 2017, RSE 193:204-215),
        DATA statements of /root/reference/PROSPECT-D/dataSpec_PDB.f90:272-1179.
        The Fortran literals are default-REAL, i.e. they reach the model rounded to
        IEEE binary32 (SURVEY.md section 8a 'precision trap'), so float32 storage is the
        exact representation of what the reference computes with.
  gort_amd/data/price_soil_eofs.f64     4 x 421 float64, Price (1990) soil EOFs,
        400..2500 nm @ 5 nm; source /root/reference/include/soil_rho.h:4-7.
"""
import os
import re
import sys

import numpy as np

REF = os.environ.get("GORT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "gort_amd", "data")

NW = 2101
ORDER = ["refractive", "k_Cab", "k_Car", "k_Anth", "k_Brown", "k_Cw", "k_Cm"]


def parse_fortran_data(path):
    txt = open(path, encoding="latin-1").read()
    # join continuation lines
    txt = re.sub(r"&\s*\n", "", txt)
    arrays = {name: np.full(NW, np.nan, dtype=np.float32) for name in ORDER}
    pat = re.compile(r"data\s*\(\s*(\w+)\(i\)\s*,\s*i\s*=\s*(\d+)\s*,\s*(\d+)\s*\)\s*/([^/]*)/", re.I)
    for m in pat.finditer(txt):
        name, lo, hi, body = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
        if name not in arrays:
            continue
        vals = []
        for tok in body.split(","):
            tok = tok.strip()
            if not tok:
                continue
            if "*" in tok:
                rep, v = tok.split("*")
                vals.extend([v] * int(rep))
            else:
                vals.append(tok)
        assert len(vals) == hi - lo + 1, (name, lo, hi, len(vals))
        # default-REAL literal -> binary32 (a 'd' exponent would be a genuine double; none exist)
        assert not any("d" in v.lower() for v in vals), name
        arrays[name][lo - 1:hi] = np.array([np.float32(v) for v in vals], dtype=np.float32)
    for name, a in arrays.items():
        assert not np.isnan(a).any(), name
    return np.stack([arrays[n] for n in ORDER])


def parse_soil(path):
    txt = open(path).read()
    rows = []
    for i in range(1, 5):
        m = re.search(r"default_soil_vector_%d\s*\[\s*\]\s*=\s*\{([^}]*)\}" % i, txt)
        vals = [float(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
        rows.append(vals)
    a = np.array(rows, dtype=np.float64)
    assert a.shape == (4, 421), a.shape
    return a


def main():
    os.makedirs(OUT, exist_ok=True)
    p = parse_fortran_data(os.path.join(REF, "PROSPECT-D", "dataSpec_PDB.f90"))
    assert p.shape == (7, NW)
    p.astype("<f4").tofile(os.path.join(OUT, "prospect_d_coeffs.f32"))
    s = parse_soil(os.path.join(REF, "include", "soil_rho.h"))
    s.astype("<f8").tofile(os.path.join(OUT, "price_soil_eofs.f64"))
    print("prospect_d_coeffs.f32", p.shape, "refractive[0]=%r k_Cab[0]=%r" % (p[0, 0], p[1, 0]))
    print("price_soil_eofs.f64", s.shape, "v1[0]=%r v4[420]=%r" % (s[0, 0], s[3, 420]))


if __name__ == "__main__":
    sys.exit(main())

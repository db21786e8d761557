#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (build container only).

    make -C oracle ref            # compiles /root/reference in place -> oracle/_ref/
    python tools/make_golden.py   # ~3-5 min on 8 cores

Sources of truth:
  oracle/_ref/gortt      the reference CLI, unmodified                    (text, %f)
  oracle/_ref/gortt_fp   same program, printf -> "%.17g" (oracle/ref_capture.c)
  oracle/_ref/libgortt_ref.so  reference objects + oracle/ref_shim.c      (function level)

Only inputs and expected outputs are stored (data); no reference source text.
The GPU box never runs this script and never sees /root/reference.
"""
import ctypes as C
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
GORTT = os.path.join(REFDIR, "gortt")
GORTT_FP = os.path.join(REFDIR, "gortt_fp")
D = C.c_double


def run(binary, args, stdin_text, timeout=600):
    p = subprocess.run([binary] + list(args), input=stdin_text.encode(), capture_output=True,
                       timeout=timeout)
    return p.returncode, p.stdout.decode("latin-1"), p.stderr.decode("latin-1")


def stream_text(angles, wl):
    head = "%d %d %s\n" % (len(angles), len(wl), " ".join(repr(float(w)) if float(w) != int(w) else str(int(w)) for w in wl))
    assert len(head) < 999, len(head)
    body = "".join("%r %r %r %r\n" % tuple(float(x) for x in a) for a in angles)
    return head + body


def parse_rows(stdout, nw, prnspec=False, prnprop=False, energy=False):
    """Rows of a gortt_fp run -> dict of arrays (header line skipped)."""
    lines = stdout.strip("\n").split("\n")[1:]
    rs, sc, K, en = [], [], [], []
    for ln in lines:
        tok = ln.replace("{", " ").replace("}", " ").replace("[", " ").replace("]", " ").split()
        v = [float(t) for t in tok]
        i = 4
        r, s = [], []
        for _ in range(nw):
            r.append(v[i]); i += 1
            if prnspec:
                s.append(v[i:i + 4]); i += 4
        rs.append(r)
        if prnspec: sc.append(s)
        if prnprop:
            K.append(v[i:i + 4]); i += 4
        if energy:
            en.append(np.array(v[i:i + 3 * nw]).reshape(nw, 3)); i += 3 * nw
        assert i == len(v), (i, len(v))
    out = {"rsurf": np.array(rs)}
    if prnspec: out["scomp"] = np.array(sc)
    if prnprop: out["K"] = np.array(K)
    if energy: out["energy"] = np.array(en)
    return out


def fp_stream(args, angles, wl, **kw):
    flags = list(args)
    if kw.get("prnspec"): flags.append("-prnspec")
    if kw.get("prnprop"): flags.append("-prnprop")
    if kw.get("energy"): flags.append("-energy")
    rc, out, err = run(GORTT_FP, flags, stream_text(angles, wl))
    assert rc == 0, (rc, err)
    return parse_rows(out, len(wl), **kw)


def fp_lut(args):
    rc, out, err = run(GORTT_FP, list(args) + ["-W"], "")
    assert rc == 0, (rc, err)
    rows = [ln.split() for ln in out.strip().split("\n")]
    pn0 = np.array([float(r[1]) for r in rows[:90]])
    ep = np.array([float(r[2]) for r in rows[:90]])
    assert rows[90][0] == "-1"
    return pn0, ep, float(rows[90][1]), float(rows[90][2])


# ------------------------------------------------------------------ CLI cases
def cli_cases():
    readme = "1 4 450 600 800 1000\n10 0 30 20\n"
    two = "2 3 550 670 865\n0 0 45 0\n-30 0 45 0\n"
    cases = [
        ("readme", ["-LAI", "4.0"], readme),
        ("readme_prnprop", ["-LAI", "4.0", "-prnprop"], readme),
        ("readme_prnspec", ["-LAI", "4.0", "-prnspec"], readme),
        ("readme_energy", ["-LAI", "4.0", "-energy"], readme),
        ("readme_all", ["-LAI", "4.0", "-prnspec", "-prnprop", "-energy"], readme),
        ("readme_q08", ["-LAI", "4.0", "-q08_pn_kopen"], readme),
        ("lut_default", ["-LAI", "4.0", "-W"], ""),
        ("lut_q08", ["-LAI", "4.0", "-q08_pn_kopen", "-W"], ""),
        ("lut_newstyle", ["-HB", "2.0", "-BR", "2.0", "-PCC", "0.6", "-LAI", "3.3", "-W"], ""),
        ("newstyle", ["-HB", "2.0", "-BR", "2.0", "-PCC", "0.6", "-LAI", "3.3", "-prnprop"], two),
        ("overrides", ["-LAI", "2", "-alb_leaf", "0.9", "-alb_soil", "0.2", "-diffuse", "0.3", "-beta", "0.5"],
         "1 2 550 865\n25 40 35 170\n"),
        ("oldstyle", ["-favd", "0.6", "-h1", "2.5", "-h2", "9", "-lambda", "0.3", "-r", "1.1", "-b", "2.0", "-prnprop"],
         "3 2 500.5 1650.25\n20 10 40 250\n-20 350 -40 10\n60 400 10 -30\n"),
        ("prospect_flags", ["-LAI", "3", "-N", "1.8", "-cab", "45", "-car", "8", "-canth", "2", "-cbrown", "0.1",
                            "-cw", "0.01", "-cm", "0.005", "-rsl1", "0.3", "-rsl2", "0.05", "-rsl3", "0.01",
                            "-rsl4", "0.001", "-prnspec"],
         "2 5 400 555.5 1200 2100 2500\n15 20 25 200\n0 0 0 0\n"),
        ("case_insensitive", ["-FAVD", "0.7", "-H1", "3.5", "-Cab", "20", "-n", "1.5"], readme),
        ("prefix_flags", ["-LAI", "4.0", "-prnspecXYZ", "-prnpropABC", "-diffusion", "0.2", "-betaX", "0.3"], readme),
        ("horizon", ["-LAI", "4.0", "-prnprop"], "5 1 800\n89 0 30 0\n89.5 0 30 180\n90 0 30 0\n30 0 90 0\n-90 0 0 0\n"),
        ("azimuth_wrap", ["-LAI", "4.0", "-prnprop"],
         "6 1 650\n30 200 40 0\n30 0 40 200\n30 -45 40 45\n30 725 40 -370\n-30 10 -40 20\n30 180 40 0\n"),
        ("extra_columns", ["-LAI", "4.0"], "1 1 700\n10 0 30 20 99 98 junk\n"),
        ("header_spacing", ["-LAI", "4.0"], "  1   2\t450   600  \n10 0 30 20\n"),
        ("count_mismatch", ["-LAI", "4.0"], "3 1 800\n10 0 30 20\n"),
        ("usage", ["-u"], ""),
        ("unknown_option", ["-Anth", "1"], readme),
        ("unknown_argument", ["-LAI", "4.0", "foo"], readme),
        ("empty_stdin", ["-LAI", "4.0"], ""),
        ("bad_header_nw", ["-LAI", "4.0"], "1 3 450 600\n10 0 30 20\n"),
        ("bad_angle_line", ["-LAI", "4.0"], "2 1 800\n10 0 30 20\n10 0 thirty 20\n"),
        ("wl_out_of_range", ["-LAI", "4.0"], "1 1 399\n10 0 30 20\n"),
        ("missing_lut_file", ["-LAI", "4.0", "-P", "/nonexistent/lut.dat"], readme),
        ("header_only_na", ["-LAI", "4.0"], "1\n"),
        # round 2: empty and ragged inputs
        ("zero_lines", ["-LAI", "4.0"], "0 2 500 600\n"),
        ("zero_bands", ["-LAI", "4.0", "-prnprop"], "2 0\n10 0 30 20\n-20 5 40 100\n"),
        ("more_lines_than_announced", ["-LAI", "4.0"], "1 1 800\n10 0 30 20\n20 0 30 20\n30 0 30 20\n"),
        ("no_final_newline", ["-LAI", "4.0"], "2 2 650 865\n10 0 30 20\n20 10 35 200"),
        ("crlf_lines", ["-LAI", "4.0"], "2 2 650 865\r\n10 0 30 20\r\n20 10 35 200\r\n"),
        ("blank_line_in_stream", ["-LAI", "4.0"], "3 1 800\n10 0 30 20\n\n20 0 30 20\n"),
        ("three_numbers_on_a_line", ["-LAI", "4.0"], "2 1 800\n10 0 30 20\n10 0 30\n"),
        ("energy_prnprop", ["-LAI", "4.0", "-energy", "-prnprop"], "2 2 650 865\n0 0 0 0\n-45 90 60 300\n"),
        ("scientific_notation", ["-LAI", "4.0"], "1 2 6.5e2 8.65E+02\n1e1 0.0e0 3.0e+1 2e1\n"),
        ("negative_count", ["-LAI", "4.0"], "-1 1 800\n10 0 30 20\n"),
        ("wl_at_the_limits", ["-LAI", "4.0"], "1 2 400 2500\n10 0 30 20\n"),
        ("lai_zero", ["-LAI", "0"], "1 2 650 865\n10 0 30 20\n"),
        ("diffuse_one", ["-LAI", "4.0", "-diffuse", "1.0", "-prnprop"], "1 2 650 865\n10 0 30 20\n"),
    ]
    out = []
    for name, args, stdin in cases:
        rc, so, se = run(GORTT, args, stdin)
        rc2, so2, se2 = run(GORTT_FP, args, stdin)
        se = se.replace(GORTT, "gortt"); se2 = se2.replace(GORTT_FP, "gortt")
        assert rc == rc2
        out.append({"name": name, "args": args, "stdin": stdin, "rc": rc, "stdout": so,
                    "stdout_fp": so2, "stderr": se})
        print("cli", name, "rc", rc, "| out", len(so), "| err", se.strip().split("\n")[0][:70])
    # -P round trip: BRDF through a LUT file written by -W
    rc, lut, _ = run(GORTT, ["-LAI", "4.0", "-W"], "")
    lut_path = "/tmp/gort_golden_lut.dat"
    open(lut_path, "w").write(lut)
    pp = "4 3 450 800 1600\n10 0 30 20\n60 0 30 180\n88.5 0 30 0\n30 0 89 0\n"
    rc, so, se = run(GORTT, ["-LAI", "4.0", "-P", lut_path, "-prnprop"], pp)
    rc2, so2, _ = run(GORTT_FP, ["-LAI", "4.0", "-P", lut_path, "-prnprop"], pp)
    out.append({"name": "lut_roundtrip", "args": ["-LAI", "4.0", "-P", "@LUT@", "-prnprop"], "stdin": pp,
                "rc": rc, "stdout": so, "stdout_fp": so2, "stderr": se, "lut_text": lut})
    # round 2: damaged -P files (fscanf("%d %lf %lf") takes what it can: gortt.c:131-146)
    rows = lut.strip().split("\n")
    damaged = {
        "lut_truncated": "\n".join(rows[:45]) + "\n",
        "lut_garbage_line": "\n".join(rows[:30] + ["garbage here"] + rows[30:]) + "\n",
        "lut_empty": "",
        "lut_without_kopen": "\n".join(r for r in rows if not r.startswith("-1")) + "\n",
        "lut_two_kopen_lines": lut + "-1 0.5 0.25\n",
        "lut_rows_reversed": "\n".join(reversed(rows)) + "\n",
        "lut_crlf": lut.replace("\n", "\r\n"),
    }
    pp2 = "2 2 650 865\n10 0 30 20\n60 0 50 180\n"
    for name, text in damaged.items():
        open(lut_path, "w").write(text)
        rc, so, se = run(GORTT, ["-LAI", "4.0", "-P", lut_path, "-prnprop"], pp2)
        rc2, so2, _ = run(GORTT_FP, ["-LAI", "4.0", "-P", lut_path, "-prnprop"], pp2)
        assert rc in (0, 1)
        out.append({"name": name, "args": ["-LAI", "4.0", "-P", "@LUT@", "-prnprop"], "stdin": pp2,
                    "rc": rc, "stdout": so, "stdout_fp": so2, "stderr": se.replace(GORTT, "gortt"), "lut_text": text})
        print("cli", name, "rc", rc, "| out", len(so))
    json.dump(out, open(os.path.join(GOLD, "cli_cases.json"), "w"), indent=1)


# ------------------------------------------------------- function-level dumps
def shim():
    L = C.CDLL(os.path.join(REFDIR, "libgortt_ref.so"))
    return L


def p(a):
    return a.ctypes.data_as(C.POINTER(D))


def shim_canopy(L, flags):
    argv = [b"gortt"] + [f.encode() for f in flags]
    arr = (C.c_char_p * len(argv))(*argv)
    L.refshim_canopy(len(argv), arr)
    sc = np.zeros(32)
    L.refshim_canopy_scalars(p(sc))
    pn0 = np.zeros(15 * 91); ps0 = np.zeros(15 * 91); vg = np.zeros(15 * 91)
    ep = np.zeros(91); th = np.zeros(91); thp = np.zeros(91); hh = np.zeros(15); hp = np.zeros(15)
    kk = np.zeros(2)
    L.refshim_gap_tables(p(pn0), p(ps0), p(vg), p(ep), p(th), p(thp), p(hh), p(hp), p(kk))
    return dict(scalars=sc, p_n0=pn0.reshape(15, 91), p_s0=ps0.reshape(15, 91), v_g=vg.reshape(15, 91),
                epgap0=ep, theta=th, theta_p=thp, height=hh, height_p=hp, kk=kk)


CANOPIES = {
    "default_lai4": ["-LAI", "4.0"],
    "newstyle": ["-HB", "2.0", "-BR", "2.0", "-PCC", "0.6", "-LAI", "3.3"],
    "q08_lai4": ["-LAI", "4.0", "-q08_pn_kopen"],
    "sparse": ["-HB", "1.2", "-BR", "3.4", "-PCC", "0.25", "-LAI", "0.7"],
    "dense_flat": ["-HB", "2.9", "-BR", "1.0", "-PCC", "0.78", "-LAI", "5.8"],
    "oldstyle": ["-favd", "0.6", "-h1", "2.5", "-h2", "9", "-lambda", "0.3", "-r", "1.1", "-b", "2.0"],
}


def c5_members(n):
    """SURVEY.md 8(d) C5 draw: numpy default_rng(12345); HB,BR,PCC,LAI through float32."""
    rng = np.random.default_rng(12345)
    m = []
    for _ in range(n):
        hb, br, pcc, lai = rng.uniform(1, 3), rng.uniform(1, 3.5), rng.uniform(0.2, 0.8), rng.uniform(0.5, 6)
        cab, cw, cm = rng.uniform(10, 60), rng.uniform(0.005, 0.03), rng.uniform(0.002, 0.015)
        N, rsl1 = rng.uniform(1, 2.5), rng.uniform(0.05, 0.4)
        m.append(dict(hb=float(np.float32(hb)), br=float(np.float32(br)), pcc=float(np.float32(pcc)),
                      lai=float(np.float32(lai)), cab=cab, cw=cw, cm=cm, N=N, rsl1=rsl1))
    return m


def member_flags(m):
    return ["-HB", repr(m["hb"]), "-BR", repr(m["br"]), "-PCC", repr(m["pcc"]), "-LAI", repr(m["lai"]),
            "-cab", repr(m["cab"]), "-cw", repr(m["cw"]), "-cm", repr(m["cm"]), "-N", repr(m["N"]),
            "-rsl1", repr(m["rsl1"])]


def canopies_and_spectra():
    L = shim()
    store = {}
    for tag, flags in CANOPIES.items():
        d = shim_canopy(L, flags)
        for k, v in d.items():
            store["%s/%s" % (tag, k)] = v
        print("canopy", tag, "kopen", d["kk"])
    np.savez_compressed(os.path.join(GOLD, "canopies.npz"), **store)
    json.dump(CANOPIES, open(os.path.join(GOLD, "canopies_flags.json"), "w"), indent=1)

    # spectra: raw PROSPECT-D for several parameter sets, all 2101 bands
    sp = {}
    psets = {
        "default": (1.2, 30., 10., 1.0, 0.0, 0.015, 0.009),
        "dry": (2.5, 5., 2., 0.0, 0.6, 0.0005, 0.012),
        "dense": (1.0, 80., 20., 5.0, 0.0, 0.04, 0.002),
        "zero_abs": (1.5, 0., 0., 0., 0., 0., 0.),
        "opaque": (1.3, 100., 30., 10., 2.0, 0.05, 0.02),
    }
    for tag, ps in psets.items():
        out = np.zeros(4202)
        L.refshim_prospect_raw(*[D(x) for x in ps], p(out))
        sp["prospect/%s/params" % tag] = np.array(ps)
        sp["prospect/%s/RT" % tag] = out
    # interpolated spectra through the reference's C glue (float fraction!), incl. non-integer nm
    shim_canopy(L, ["-LAI", "4.0", "-q08_pn_kopen"])
    wl = np.concatenate([np.arange(400., 2501., 1.0), np.array([400.25, 555.5, 1650.75, 2499.99, 703.1, 1000.000001])])
    rs = np.zeros(wl.size); rl = np.zeros(wl.size); tl = np.zeros(wl.size)
    L.refshim_spectra(p(wl), wl.size, p(rs), p(rl), p(tl))
    sp["interp/default/wl"] = wl; sp["interp/default/rsoil"] = rs
    sp["interp/default/rleaf"] = rl; sp["interp/default/tleaf"] = tl
    shim_canopy(L, ["-LAI", "4.0", "-q08_pn_kopen", "-rsl1", "0.31", "-rsl2", "-0.02", "-rsl3", "0.05", "-rsl4", "0.004",
                    "-N", "1.9", "-cab", "12.5"])
    rs2 = np.zeros(wl.size); rl2 = np.zeros(wl.size); tl2 = np.zeros(wl.size)
    L.refshim_spectra(p(wl), wl.size, p(rs2), p(rl2), p(tl2))
    sp["interp/alt/wl"] = wl; sp["interp/alt/rsoil"] = rs2; sp["interp/alt/rleaf"] = rl2; sp["interp/alt/tleaf"] = tl2
    sp["interp/alt/rsl"] = np.array([0.31, -0.02, 0.05, 0.004]); sp["interp/alt/prospect"] = np.array([1.9, 12.5, 10., 1.0, 0.0, 0.015, 0.009])
    x = np.zeros(32); w = np.zeros(32)
    L.refshim_gauleg(p(x), p(w), 32)
    sp["gauleg32/x"] = x; sp["gauleg32/w"] = w
    np.savez_compressed(os.path.join(GOLD, "spectra.npz"), **sp)
    print("spectra done")


# ----------------------------------------------------------- config goldens
def chunks(seq, n):
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def config_goldens():
    pool = ThreadPoolExecutor(8)
    lai4 = ["-LAI", "4.0"]

    # C2: principal plane, 181 view zeniths, sza=30, 800 nm (and with K's)
    ang = [(float(v), 0.0, 30.0, 0.0) for v in range(-90, 91)]
    r = fp_stream(lai4, ang, [800], prnprop=True, prnspec=True)
    np.savez_compressed(os.path.join(GOLD, "c2_principal_plane.npz"), angles=np.array(ang), wl=np.array([800.]),
                        rsurf=r["rsurf"], K=r["K"], scomp=r["scomp"])
    print("C2 done; nan rows:", int(np.isnan(r["rsurf"]).any(axis=1).sum()))

    # C3 subgrid: stream order "vza phi sza 0" (SURVEY 8d), 800 nm
    zen = [0, 1, 7, 15, 30, 45, 60, 75, 85, 88, 89, 90]
    phi = list(range(0, 361, 15)) + [1, 89, 91, 179, 181, 269, 271, 359]
    ang = [(float(v), float(f), float(s), 0.0) for s in zen for v in zen for f in phi]
    parts = list(pool.map(lambda a: fp_stream(lai4, a, [800], prnprop=True), chunks(ang, 600)))
    np.savez_compressed(os.path.join(GOLD, "c3_subgrid.npz"), angles=np.array(ang), wl=np.array([800.]),
                        rsurf=np.concatenate([q["rsurf"] for q in parts]), K=np.concatenate([q["K"] for q in parts]))
    print("C3 subgrid done", len(ang))

    # second canopy, random off-grid angles x 24 wavelengths (incl. non-integer nm), all outputs
    rng = np.random.default_rng(7)
    ang = np.stack([rng.uniform(-89, 89, 400), rng.uniform(-400, 400, 400), rng.uniform(-89, 89, 400),
                    rng.uniform(-400, 400, 400)], axis=1)
    wl = sorted(set(np.round(rng.uniform(400, 2500, 20), 2).tolist() + [400.0, 2500.0, 700.5, 1400.0]))
    ns = CANOPIES["newstyle"]
    parts = list(pool.map(lambda a: fp_stream(ns, a, wl, prnprop=True, prnspec=True), chunks(ang.tolist(), 50)))
    np.savez_compressed(os.path.join(GOLD, "random_stream_newstyle.npz"), angles=ang, wl=np.array(wl),
                        rsurf=np.concatenate([q["rsurf"] for q in parts]), K=np.concatenate([q["K"] for q in parts]),
                        scomp=np.concatenate([q["scomp"] for q in parts]))
    print("random stream done")

    # C4: albedo / favegt / fasoil. (a) all 2101 bands at 5 sun zeniths, (b) 21 bands at all 91 sun zeniths
    def energy_run(args):
        szas, wl = args
        a = [(0.0, 0.0, float(s), 0.0) for s in szas]
        return fp_stream(lai4, a, wl, energy=True)["energy"]
    wl_all = list(range(400, 2501))
    sz_a = [0, 30, 60, 80, 89]
    parts = list(pool.map(energy_run, [(sz_a, c) for c in chunks(wl_all, 32)]))
    en_a = np.concatenate(parts, axis=1)            # [5][2101][3]
    wl_b = list(range(400, 2501, 105))
    sz_b = list(range(0, 91))
    parts = list(pool.map(energy_run, [(c, wl_b) for c in chunks(sz_b, 12)]))
    en_b = np.concatenate(parts, axis=0)            # [91][21][3]
    np.savez_compressed(os.path.join(GOLD, "c4_albedo.npz"), sza_a=np.array(sz_a, float), wl_a=np.array(wl_all, float),
                        energy_a=en_a, sza_b=np.array(sz_b, float), wl_b=np.array(wl_b, float), energy_b=en_b)
    print("C4 done", en_a.shape, en_b.shape)



def c5_goldens():
    pool = ThreadPoolExecutor(8)
    wl_all = list(range(400, 2501))
    # C5: first 8 ensemble members; LUT + BRDF at sza=30 on a (vza,phi) subgrid x 2101 bands (12 CLI chunks)
    mem = c5_members(8)
    vz = [0, 45, 89, 90]
    ph = [0, 90, 180, 300]
    ang = [(float(v), float(f), 30.0, 0.0) for v in vz for f in ph]
    store = {"angles": np.array(ang), "wl": np.array(wl_all, float)}
    for i, m in enumerate(mem):
        fl = member_flags(m)
        pn0, ep, ko, kep = fp_lut(fl)
        parts = list(pool.map(lambda c: fp_stream(fl, ang, c)["rsurf"], chunks(wl_all, 180)))
        store["m%d/rsurf" % i] = np.concatenate(parts, axis=1)
        store["m%d/p_n0" % i] = pn0; store["m%d/epgap0" % i] = ep; store["m%d/kk" % i] = np.array([ko, kep])
        store["m%d/params" % i] = np.array([m[k] for k in ("hb", "br", "pcc", "lai", "cab", "cw", "cm", "N", "rsl1")])
        print("C5 member", i, "kopen", ko, kep)
    np.savez_compressed(os.path.join(GOLD, "c5_members.npz"), **store)


# ------------------------------------------------- fuzz canopies (task 2)
def fuzz_canopy_specs():
    """The canopies the GPU fuzz tests draw (tests/test_gpu_parity.py), so that those tests compare with the
    REFERENCE and not only with our restatement: the exact-tie geometries (-BR 1/2/3: s/ds + 0.5 an exact integer
    in the path-length histogram, gortt_pn_kopen.c:134-139), oblate crowns, very sparse / very dense stands, the
    first 150 draws of test_gap_probabilities_fuzz_incl_tie_hazards (seed 4242), the first 40 C5 members."""
    specs = []

    def ns(hb, br, pcc, lai, tag):
        hb, br, pcc, lai = (float(np.float32(x)) for x in (hb, br, pcc, lai))
        specs.append({"tag": tag, "flags": ["-HB", repr(hb), "-BR", repr(br), "-PCC", repr(pcc), "-LAI", repr(lai)],
                      "kw": {"newstyle": [hb, br, pcc], "lai": lai}})
    for br in (1.0, 2.0, 3.0):
        for hb in (1.0, 2.0, 2.5):
            for pcc in (0.3, 0.6):
                ns(hb, br, pcc, 3.0, "tie")
    rng = np.random.default_rng(4242)
    for _ in range(150):
        ns(rng.uniform(0.5, 4), rng.uniform(0.4, 4), rng.uniform(0.05, 0.95), rng.uniform(0.1, 9), "fuzz4242")
    for br in (0.4, 0.5, 0.65, 0.8, 0.95):                        # oblate crowns, b/r < 1
        for pcc in (0.2, 0.7):
            ns(1.5, br, pcc, 2.5, "oblate")
    for lai in (0.1, 9.0):
        for br in (1.0, 2.5):
            for pcc in (0.1, 0.9):
                ns(2.0, br, pcc, lai, "lai_extreme")
    for m in c5_members(40):
        ns(m["hb"], m["br"], m["pcc"], m["lai"], "c5")
    old = [dict(favd=0.3, h1=1.0, h2=4.0, lam=0.9, r=0.5, b=0.5), dict(favd=1.2, h1=6.0, h2=20.0, lam=0.05, r=2.5, b=7.0),
           dict(favd=0.8, h1=2.0, h2=2.5, lam=0.2, r=1.0, b=3.0), dict(favd=0.05, h1=4.0, h2=12.0, lam=0.4, r=0.9, b=0.6)]
    for kw in old:
        specs.append({"tag": "oldstyle", "kw": kw,
                      "flags": ["-favd", repr(kw["favd"]), "-h1", repr(kw["h1"]), "-h2", repr(kw["h2"]), "-lambda", repr(kw["lam"]),
                                "-r", repr(kw["r"]), "-b", repr(kw["b"])]})
    return specs


def fuzz_goldens():
    pool = ThreadPoolExecutor(8)
    specs = fuzz_canopy_specs()

    def lut(spec):
        rc, out, err = run(GORTT_FP, spec["flags"] + ["-W"], "")
        if rc != 0:
            return None
        rows = [ln.split() for ln in out.strip().split("\n")]
        if len(rows) != 91 or rows[90][0] != "-1":
            return None
        # NaN rows are kept: HB = BR = 1 gives `-nan` in epgap(30 deg) and KOpenEP in the reference itself, and
        # the NaN pattern is part of what the restatement and the kernel have to reproduce
        return np.array([[float(r[1]), float(r[2])] for r in rows])
    tabs = list(pool.map(lut, specs))
    keep = [i for i, t in enumerate(tabs) if t is not None]
    print("fuzz canopies: %d of %d gave a 91-row -W table, %d of them with NaN entries"
          % (len(keep), len(specs), sum(int(np.isnan(tabs[i]).any()) for i in keep)))
    specs = [specs[i] for i in keep]
    tab = np.array([tabs[i] for i in keep])                         # [n][91][2]: rows 0..89 p_n0, epgap; row 90 k_open, k_openep
    json.dump(specs, open(os.path.join(GOLD, "fuzz_canopies.json"), "w"), indent=0)

    # BRDF rows of 24 of them (every tag represented): random lines, exact hot-spot lines, table nodes, near-horizon
    rng = np.random.default_rng(2024)
    pick = sorted(set([0, 5, 11, 17] + list(range(18, len(specs), max(1, (len(specs) - 18) // 20)))))[:24]
    wl = [450.0, 555.5, 670.0, 865.0, 1240.25, 1650.0, 2130.0]
    lines = np.stack([rng.uniform(-89.9, 89.9, 48), rng.uniform(-400, 400, 48), rng.uniform(-89.9, 89.9, 48),
                      rng.uniform(-400, 400, 48)], 1)
    lines[:8, 0] = lines[:8, 2]; lines[:8, 1] = lines[:8, 3]           # exact hot spot
    lines[8:14, [0, 2]] = np.round(lines[8:14, [0, 2]])                # integer zeniths (table nodes)
    lines[14:20, 0] = rng.uniform(88.5, 89.99, 6)                     # view near the horizon
    lines[20:24, 2] = rng.uniform(88.0, 89.9, 4)                      # sun near the horizon
    lines[24] = [89.0, 10.0, 89.0, 10.0]                              # hot spot at the horizon (ill-conditioned in the reference)
    lines[25] = [0.0, 0.0, 0.0, 0.0]

    def rows(i):
        return fp_stream(specs[i]["flags"], lines.tolist(), wl, prnprop=True, prnspec=True)
    res = list(pool.map(rows, pick))
    np.savez_compressed(os.path.join(GOLD, "fuzz_canopies.npz"), lut=tab, brdf_pick=np.array(pick), brdf_lines=lines,
                        brdf_wl=np.array(wl), brdf_rsurf=np.array([r["rsurf"] for r in res]),
                        brdf_K=np.array([r["K"] for r in res]), brdf_scomp=np.array([r["scomp"] for r in res]))
    print("fuzz goldens done: %d canopies, %d with BRDF rows" % (len(specs), len(pick)))


def cli_fuzz_goldens(n_cases=160, seed=777):
    """Random command lines and inputs through the REAL reference CLI: flag combinations (old / new style crown geometry,
    -LAI / -favd, -beta, -diffuse, leaf / soil overrides, PROSPECT and Price flags in mixed case, -q08_pn_kopen,
    output flags), 1..6 wavelengths incl. non-integers and the limits, 1..8 angle lines incl. negative zeniths,
    azimuths beyond +-360, near-horizon and horizon geometries.  stdout / stderr / rc kept; the test compares bytes."""
    rng = np.random.default_rng(seed)
    out, tried = [], 0

    def num(x, nd=None):
        if nd is None:
            nd = int(rng.integers(0, 7))
        t = ("%." + str(nd) + "f") % x
        return t

    while len(out) < n_cases and tried < 4 * n_cases:
        tried += 1
        args = []
        style = rng.integers(0, 3)
        if style == 0:
            args += ["-HB", num(rng.uniform(0.5, 4), 3), "-BR", num(rng.uniform(0.4, 4), 3), "-PCC", num(rng.uniform(0.05, 0.95), 3)]
        elif style == 1:
            h1 = rng.uniform(1, 6)
            args += ["-h1", num(h1, 2), "-h2", num(h1 + rng.uniform(0.5, 12), 2), "-b", num(rng.uniform(0.5, 6), 2),
                     "-r", num(rng.uniform(0.4, 2.5), 2), "-lambda", num(rng.uniform(0.02, 0.9), 3)]
        args += (["-LAI", num(rng.uniform(0.1, 8), 2)] if rng.random() < 0.7 else ["-favd", num(rng.uniform(0.05, 1.5), 3)])
        if rng.random() < 0.25: args += ["-beta", num(rng.uniform(0, 1), 2)]
        if rng.random() < 0.25: args += [str(rng.choice(["-diffuse", "-DIFF", "-diffusivity"])), num(rng.uniform(0, 1), 2)]
        if rng.random() < 0.15: args += ["-alb_leaf", num(rng.uniform(0.05, 0.98), 2)]
        if rng.random() < 0.15: args += ["-alb_soil", num(rng.uniform(0.01, 0.6), 2)]
        for flag, lo, hi in (("-N", 1.0, 3.0), ("-Cab", 5, 80), ("-car", 1, 20), ("-cw", 0.002, 0.04), ("-CM", 0.001, 0.02),
                             ("-canth", 0, 5), ("-cbrown", 0, 1), ("-rsl1", 0.05, 0.5), ("-rsl2", -0.1, 0.2), ("-RSL3", -0.05, 0.05),
                             ("-rsl4", -0.01, 0.01)):
            if rng.random() < 0.2: args += [flag, num(rng.uniform(lo, hi), 4)]
        if rng.random() < 0.12: args += ["-q08_pn_kopen"]
        nw = int(rng.integers(1, 7))
        if rng.random() < 0.35: args += ["-prnprop"]
        if rng.random() < 0.25: args += ["-prnspec"]
        if rng.random() < 0.15: args += ["-energy"]
        order = rng.permutation(len(args)) if False else None   # flag order matters (new style switches old style off): keep
        wl = [float(rng.choice([400.0, 2500.0, rng.uniform(400, 2500), float(rng.integers(400, 2501))])) for _ in range(nw)]
        na = int(rng.integers(1, 9))
        lines = []
        for _ in range(na):
            kind = rng.integers(0, 6)
            vza, sza = rng.uniform(-85, 85), rng.uniform(0, 80)
            if kind == 0: vza, sza = rng.uniform(86, 89.9), rng.uniform(0, 89.9)
            if kind == 1: sza = rng.uniform(-80, -1)
            if kind == 2: vza = sza                                   # hot spot direction when the azimuths agree
            vaa, saa = rng.uniform(-400, 760), rng.uniform(-400, 760)
            if kind == 2: vaa = saa
            if kind == 3: vza, sza = float(rng.integers(-89, 90)), float(rng.integers(0, 90))     # table nodes
            # exactly the horizon (the reference prints -nan); zeniths BEYOND 90 degrees are left out: there the reference
            # reads past the end of its 91-entry tables and prints whatever lies behind them (deviations)
            if kind == 4 and rng.random() < 0.3: vza = 90.0
            lines.append(" ".join(num(v) for v in (vza, vaa, sza, saa)))
        stdin = "%d %d %s\n" % (na, nw, " ".join(num(w, int(rng.integers(0, 4))) for w in wl)) + "\n".join(lines) + "\n"
        rc, so, se = run(GORTT, args, stdin)
        if rc not in (0, 1) or len(so) > 20000:
            continue                                                  # a crash of the reference is not a fixture
        rc2, so2, _ = run(GORTT_FP, args, stdin)                      # the same at %.17g: compared with --binary-out
        out.append({"name": "fuzz%03d" % len(out), "args": args, "stdin": stdin, "rc": rc, "stdout": so, "stdout_fp": so2,
                    "stderr": se.replace(GORTT, "gortt")})
    print("cli fuzz: %d cases kept of %d tried; %d with rc 1, %d with -nan" %
          (len(out), tried, sum(c["rc"] for c in out), sum("-nan" in c["stdout"] for c in out)))
    json.dump(out, open(os.path.join(GOLD, "cli_fuzz_cases.json"), "w"), indent=0)


def prospect_fuzz_golden(n=96, seed=5150):
    """PROSPECT-D through the reference's own Fortran for random parameter vectors: the usual ranges, the edges (N = 1:
    one layer, the Stokes system degenerates; no absorbers at all: tau = 1; absorbers so strong that k > 85: tau = 0,
    and k in (4, 85]: the second Chebyshev fit), and values an ensemble filter may wander into (negative contents).
    Every 7th band of R and T is kept (301 + 301 values per vector)."""
    L = shim()
    rng = np.random.default_rng(seed)
    ps = []
    for i in range(n):
        kind = i % 8
        N, cab, car, anth, cbrown, cw, cm = (rng.uniform(1.0, 3.5), rng.uniform(0, 100), rng.uniform(0, 30), rng.uniform(0, 10),
                                             rng.uniform(0, 1.5), rng.uniform(0.0002, 0.05), rng.uniform(0.0005, 0.03))
        if kind == 1: N = 1.0
        if kind == 2: cab = car = anth = cbrown = 0.0
        if kind == 3: cw, cm = rng.uniform(0.5, 5.0), rng.uniform(0.1, 2.0)          # k beyond 4, up to beyond 85
        if kind == 4: cab, cbrown = rng.uniform(500, 5000), rng.uniform(5, 50)
        if kind == 5: cw = cm = 0.0
        if kind == 6: N = rng.uniform(1.0, 1.05)
        if kind == 7: cab, cw = -rng.uniform(0, 5), -rng.uniform(0, 0.001)               # k may go negative: tau = 1 branch
        ps.append((N, cab, car, anth, cbrown, cw, cm))
    ps = np.array(ps)
    keep = np.arange(0, 2101, 7)
    RT = np.zeros((n, 2, keep.size))
    for i, p_ in enumerate(ps):
        out = np.zeros(4202)
        L.refshim_prospect_raw(*[D(x) for x in p_], p(out))
        RT[i, 0], RT[i, 1] = out[:2101][keep], out[2101:][keep]
    np.savez_compressed(os.path.join(GOLD, "prospect_fuzz.npz"), params=ps, bands=keep, RT=RT)
    print("prospect fuzz: %d vectors, NaN in %d of them, R range %.3g..%.3g" %
          (n, int(np.isnan(RT).any(axis=(1, 2)).sum()), np.nanmin(RT[:, 0]), np.nanmax(RT[:, 0])))


def ensemble_states_golden(n=24, seed=6060):
    """The ensemble layer end to end against forward runs of the real reference: 24 random state vectors (HB, BR, PCC,
    LAI, N, Cab, Car, Cw, Cm, rsl1 - gort_amd/ensemble.py STATE), each one run of gortt_fp with that member's flags on
    the same 12 angle lines x 7 MODIS land bands (+ -energy for the first 6 members, 3 lines)."""
    rng = np.random.default_rng(seed)
    wl = np.array([450.0, 555.0, 645.0, 858.5, 1240.0, 1640.0, 2130.0])
    ang = np.round(np.stack([rng.uniform(-70, 70, 12), rng.uniform(0, 360, 12), rng.uniform(0, 75, 12), rng.uniform(0, 360, 12)], 1), 4)
    names = ("HB", "BR", "PCC", "LAI", "N", "Cab", "Car", "Cw", "Cm", "rsl1")
    states, rows, energy = [], [], []
    for i in range(n):
        st = [float(np.float32(rng.uniform(1, 3))), float(np.float32(rng.uniform(1, 3.5))), float(np.float32(rng.uniform(0.2, 0.8))),
              float(np.float32(rng.uniform(0.5, 6))), rng.uniform(1, 2.5), rng.uniform(10, 60), rng.uniform(2, 15),
              rng.uniform(0.005, 0.03), rng.uniform(0.002, 0.015), rng.uniform(0.05, 0.4)]
        flags = ["-HB", repr(st[0]), "-BR", repr(st[1]), "-PCC", repr(st[2]), "-LAI", repr(st[3]), "-N", repr(st[4]),
                 "-cab", repr(st[5]), "-car", repr(st[6]), "-cw", repr(st[7]), "-cm", repr(st[8]), "-rsl1", repr(st[9])]
        rc, so, se = run(GORTT_FP, flags, stream_text(ang, wl))
        assert rc == 0, se
        rows.append([[float(t) for t in ln.split()[4:]] for ln in so.strip().split("\n")[1:]])
        if i < 6:
            rc, so, se = run(GORTT_FP, flags + ["-energy"], stream_text(ang[:3], wl))
            assert rc == 0, se
            energy.append([[float(t) for t in ln.split()[4 + len(wl):]] for ln in so.strip().split("\n")[1:]])
        states.append(st)
    np.savez_compressed(os.path.join(GOLD, "ensemble_states.npz"), names=np.array(names), states=np.array(states), wl=wl,
                        angles=ang, rsurf=np.array(rows), energy=np.array(energy).reshape(6, 3, len(wl), 3))
    print("ensemble states: %d members x %d lines x %d bands from the reference" % (n, len(ang), len(wl)))


def cli_hostile_goldens(n_cases=100, seed=999):
    """Command lines a careless or hostile user types, through the real reference: non-numeric and negative values,
    repeated and contradicting flags, unknown options in every position, prefixes that fall through to the catch-alls
    (-b, -r, -n), headers with signs, fractions and garbage, angle lines with too few / non-numeric fields.  Cases on
    which the reference itself crashes (a value flag as the last argument: SIGSEGV) or hangs are not fixtures."""
    rng = np.random.default_rng(seed)
    values = ["abc", "-1", "0", "1e3", "nan", "inf", "3.3.3", "", "  2", "0x10", "1,5", "-0.0", "1e-320", "9" * 30]
    flags = ["-LAI", "-favd", "-h1", "-h2", "-b", "-r", "-lambda", "-HB", "-BR", "-PCC", "-beta", "-diffuse", "-alb_leaf",
             "-alb_soil", "-N", "-cab", "-car", "-cw", "-cm", "-canth", "-cbrown", "-rsl1", "-rsl4", "-bogus", "-Bx", "-Rz",
             "-nx", "-lam", "-Lambda", "-hb", "-pcc", "-PCCx", "-LAIx", "-Wx", "-prn", "-prnsp", "-energyzz", "-q08_pn", "-lidar",
             "-soil", "-U", "--", "-", "-P"]
    headers = ["1 2 650 865", "1 2 650 865 999", "1 2 650", "+1 +2 650 865", "1.9 2.9 650 865", "1 2 65e1 865.", "01 02 0650 0865",
               "1 2 abc 865", "1 2 2500.0001 865", "x 2 650 865", "1 x 650 865", "1 -1", "", " ", "1 2 650 865 # comment"]
    lines = ["10 0 30 20", "10 0 30", "10 0 30 x", "1e1 0e0 3e1 2e1", "10,0,30,20", "  10\t0  30   20  ", "+10 -0 +30 +20",
             "10 0 30 20 extra", "nan 0 30 20", "inf 0 30 20", "10 0 1e400 20", "0x10 0 30 20", "10. .0 30. 20."]
    out, tried = [], 0
    while len(out) < n_cases and tried < 6 * n_cases:
        tried += 1
        args = []
        for _ in range(int(rng.integers(1, 5))):
            f = str(rng.choice(flags))
            args.append(f)
            if rng.random() < 0.8:
                args.append(str(rng.choice(values)) if rng.random() < 0.6 else "%.3f" % rng.uniform(0.1, 5))
        if rng.random() < 0.5:
            args = ["-LAI", "4.0"] + args
        if args and args[-1] in flags and rng.random() < 0.9:
            args.append("1.5")                                     # keep most cases away from the trailing-flag SIGSEGV
        stdin = str(rng.choice(headers)) + "\n" + "".join(str(rng.choice(lines)) + "\n" for _ in range(int(rng.integers(0, 4))))
        try:
            rc, so, se = run(GORTT, args, stdin, timeout=20)
        except subprocess.TimeoutExpired:
            continue
        if rc not in (0, 1) or len(so) > 20000:
            continue
        out.append({"name": "hostile%03d" % len(out), "args": args, "stdin": stdin, "rc": rc, "stdout": so,
                    "stderr": se.replace(GORTT, "gortt")})
    print("cli hostile: %d cases kept of %d tried; %d with rc 1" % (len(out), tried, sum(c["rc"] for c in out)))
    json.dump(out, open(os.path.join(GOLD, "cli_hostile_cases.json"), "w"), indent=0)


def cli_number_formats_golden(seed=1212):
    """How the reference READS numbers (sscanf "%lf" for angle fields, atoi / atof for the header, gortt.c:153-237): 140
    single-line inputs with every spelling C's number parsers know - signs, hex floats, inf / nan / nan(chars), exponents
    without digits, leading zeros, tabs, trailing junk glued to a number, numbers too large or too small for a double."""
    rng = np.random.default_rng(seed)
    toks = ["10", "+10", "-10", "010", "1e1", "1E1", "1e+1", "1.e1", ".1e2", "10.", "0x1p3", "0xA", "0XA.8", "1e", "1e+", "1.5e400",
            "-1e400", "1e-400", "4.9e-324", "inf", "-inf", "INF", "infinity", "nan", "-nan", "NAN", "nan(abc)", "nan(", "10abc", "10e", "10x",
            "1_0", "1'0", "١٠", "10\t", "\t10", "1 0", "--10", "+-10", "1.2.3", "0.0.0", "0", "-0", "+0", "00", ".", "-", "+", "e5", ".e5",
            "1d1", "1f", "1L", "0b11", "1e1e1", "0x", "0x.p1", "0x1p", "1,0"]
    out = []
    for i in range(140):
        if i % 2 == 0:                                              # one odd token in an angle line
            f = ["10", "0", "30", "20"]
            f[int(rng.integers(0, 4))] = str(rng.choice(toks))
            stdin = "1 2 650 865\n" + " ".join(f) + "\n"
        else:                                                       # one odd token in the header
            h = ["1", "2", "650", "865"]
            h[int(rng.integers(0, 4))] = str(rng.choice(toks))
            stdin = " ".join(h) + "\n10 0 30 20\n"
        try:
            rc, so, se = run(GORTT, ["-LAI", "4.0"], stdin, timeout=20)
        except subprocess.TimeoutExpired:
            continue
        if rc not in (0, 1) or len(so) > 20000:
            continue
        out.append({"name": "numfmt%03d" % len(out), "args": ["-LAI", "4.0"], "stdin": stdin, "rc": rc, "stdout": so,
                    "stderr": se.replace(GORTT, "gortt")})
    print("cli number formats: %d cases, %d with rc 1" % (len(out), sum(c["rc"] for c in out)))
    json.dump(out, open(os.path.join(GOLD, "cli_number_format_cases.json"), "w"), indent=0, ensure_ascii=False)


def cli_scanf_corner_golden():
    """Where scanf("%lf")'s greedy matching and strtod's longest valid prefix part ways: "0x." is 0 for scanf,
    "0xp1", "nan()", "nan(1)" and "infinit" are matching failures, "1e" / "0x1p" swallow the marker.  Every such token in
    every one of the four fields of an angle line (the other three plain), deterministic: the reference's exit code,
    stderr and stdout say what its sscanf did with it."""
    toks = ["0x.", "0xp1", "0x", "0x.p1", "0x.8p1", "0x1.8", "0x1p", "0x1p+", "0X1P-", "nan()", "nan(1)", "nan(abc", "nan(ab c)",
            "infinit", "infinityx", "infinity", "in", "i", "na", "+inf", "-infinity", "1e", "1e+", "1e-", "0e", ".e1", "+.5", "-.5e1",
            "5.", "1e5x", "12abc", "1..2", "1e0005", "00x1", "1.e", "+", "-.", "1e+5e"]
    out = []
    for tok in toks:
        for pos in range(4):
            f = ["10", "0", "30", "20"]
            f[pos] = tok
            stdin = "1 2 650 865\n" + " ".join(f) + "\n"
            try:
                rc, so, se = run(GORTT, ["-LAI", "4.0"], stdin, timeout=20)
            except subprocess.TimeoutExpired:
                continue
            if rc not in (0, 1) or len(so) > 20000:
                continue
            out.append({"name": "scanf_%s_field%d" % (tok, pos), "args": ["-LAI", "4.0"], "stdin": stdin, "rc": rc, "stdout": so,
                        "stderr": se.replace(GORTT, "gortt")})
    print("cli scanf corners: %d cases, %d with rc 1" % (len(out), sum(c["rc"] for c in out)))
    json.dump(out, open(os.path.join(GOLD, "cli_scanf_corner_cases.json"), "w"), indent=0, ensure_ascii=False)


def cli_bulk_golden(n=4000, seed=4040):
    """One LONG stream through the real reference (4000 random lines x 3 bands, -prnspec -prnprop): what the drop-in's
    chunked, multi-threaded text path has to reproduce row for row.  Kept gzipped (stdin + stdout)."""
    import gzip
    rng = np.random.default_rng(seed)
    ang = np.stack([rng.uniform(-89, 89, n), rng.uniform(-360, 720, n), rng.uniform(0, 89, n), rng.uniform(-360, 720, n)], 1)
    ang[::97, 2] = np.round(ang[::97, 2])                              # some table nodes
    ang[::101, 0] = ang[::101, 2]; ang[::101, 1] = ang[::101, 3]       # some hot-spot directions
    stdin = "%d 3 550.5 865 1650.25\n" % n + "".join("%.5f %.4f %.6f %.3f\n" % tuple(r) for r in ang)
    args = ["-LAI", "3.5", "-prnspec", "-prnprop"]
    rc, so, se = run(GORTT, args, stdin)
    assert rc == 0 and so.count("\n") == n + 1
    with gzip.open(os.path.join(GOLD, "cli_bulk.json.gz"), "wt") as f:
        json.dump({"args": args, "stdin": stdin, "stdout": so, "stderr": se}, f)
    print("cli bulk: %d lines, %d bytes of reference output" % (n, len(so)))


def wide_stream_lines(n=60000, seed=9090):
    """The angle lines of the wide-stream fixture: shared by the generator and tests/test_gpu_parity.py."""
    rng = np.random.default_rng(seed)
    ang = np.stack([rng.uniform(-89, 89, n), rng.uniform(-360, 720, n), rng.integers(0, 90, n).astype(float), rng.uniform(-360, 720, n)], 1)
    k = n // 2
    ang[k:, 2] = rng.uniform(0, 89, n - k)                              # half of them with a sun zenith of their own
    ang[::211, 0] = ang[::211, 2]; ang[::211, 1] = ang[::211, 3]        # hot spot
    ang[::307, 0] = 89.5
    return np.round(ang, 6)


def wide_stream_golden():
    """The WIDE stream kernels (>= 128 bands, >= 4.2e6 samples per call: the per-line flat kernel and the grouped form)
    against the real reference and not only against our restatement: 180 bands (what fits the reference's 999-character
    header) x 60 000 lines run on the GPU, of which the reference computes a sample of ~100 at %.17g here."""
    wl = np.round(np.linspace(400, 2500, 180))
    ang = wide_stream_lines()
    pick = np.unique(np.concatenate([np.arange(0, len(ang), 641), [0, 211, 307, 30000, 30001, len(ang) - 1]]))
    args = ["-HB", "2.0", "-BR", "2.0", "-PCC", "0.6", "-LAI", "3.3"]
    rc, so, se = run(GORTT_FP, args, stream_text(ang[pick], wl))
    assert rc == 0, se
    rows = [ln.split() for ln in so.strip().split("\n")[1:]]
    vals = np.array([[float(t) for t in r[4:]] for r in rows])
    assert vals.shape == (len(pick), len(wl))
    np.savez_compressed(os.path.join(GOLD, "wide_stream.npz"), wl=wl, pick=pick, rsurf=vals,
                        canopy=np.array([2.0, 2.0, 0.6, 3.3]), n_lines=len(ang), seed=9090)
    print("wide stream: %d sample lines x %d bands from the reference, NaN rows %d" % (len(pick), len(wl), int(np.isnan(vals).all(1).sum())))


def lut_nodes_golden(n=160, seed=3131):
    """The LUT kernel (expand_flat_kernel, the headline kernel) against the real reference directly: 160 random nodes
    (sun zenith, view zenith, azimuth) of the integer-degree metric grid x 180 bands at %.17g, plus the grid's corners
    and the horizon rows.  The reference sees a node as the line `vza phi sza 0`."""
    rng = np.random.default_rng(seed)
    wl = np.round(np.linspace(400, 2500, 180))
    nodes = np.stack([rng.integers(0, 91, n), rng.integers(0, 91, n), rng.integers(0, 361, n)], 1)
    nodes = np.concatenate([nodes, [[0, 0, 0], [0, 0, 360], [89, 89, 0], [89, 89, 180], [30, 30, 0], [30, 89, 90], [89, 30, 270],
                                    [90, 45, 10], [45, 90, 10]]])       # the last two: horizon rows, -nan in the reference
    ang = np.stack([nodes[:, 1], nodes[:, 2], nodes[:, 0], np.zeros(len(nodes))], 1).astype(float)
    args = ["-LAI", "4.0"]
    rc, so, se = run(GORTT_FP, args, stream_text(ang, wl))
    assert rc == 0, se
    vals = np.array([[float(t) for t in ln.split()[4:]] for ln in so.strip().split("\n")[1:]])
    assert vals.shape == (len(nodes), len(wl))
    np.savez_compressed(os.path.join(GOLD, "lut_nodes.npz"), wl=wl, nodes=nodes, rsurf=vals)
    print("lut nodes: %d nodes x %d bands from the reference, NaN rows %d" % (len(nodes), len(wl), int(np.isnan(vals).all(1).sum())))


def main():
    for b in (GORTT, GORTT_FP):
        if not os.path.exists(b):
            sys.exit("missing %s: run `make -C oracle ref` first" % b)
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["cli", "func", "config", "c5", "fuzz"]
    if "cli" in what: cli_cases()
    if "func" in what: canopies_and_spectra()
    if "config" in what: config_goldens()
    if "c5" in what: c5_goldens()
    if "fuzz" in what: fuzz_goldens()
    if "clifuzz" in what: cli_fuzz_goldens()
    if "clibulk" in what: cli_bulk_golden()
    if "clihostile" in what: cli_hostile_goldens()
    if "clinumfmt" in what: cli_number_formats_golden()
    if "cliscanf" in what: cli_scanf_corner_golden()
    if "prospect" in what: prospect_fuzz_golden()
    if "ensemble" in what: ensemble_states_golden()
    if "wide" in what: wide_stream_golden()
    if "lutnodes" in what: lut_nodes_golden()


if __name__ == "__main__":
    main()

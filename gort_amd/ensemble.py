"""Ensemble-facing host layer over the member-batched kernels.

The reference README names the use - inverting canopy parameters from multi-angle reflectance with an ensemble
filter - but holds no code for it: each ensemble member is one forward run of `gortt` with that member's flags
(-HB -BR -PCC -LAI for the crown geometry, the PROSPECT-D / Price parameters for the spectra).  What is built
here is therefore exactly that and nothing beyond it: a state vector -> the `gortt` inputs of a member, all
members' observation vectors in one launch pair (gort_rsurf_members_stream), and finite-difference Jacobians as
an ensemble of perturbed members.  Parity is parity of the forward runs (tests/test_gpu_parity.py).
"""
import numpy as np

from . import api

#: state vector layout; the first four are the `gortt` new-style crown flags (gortt.c:1086-1131), the rest
#: PROSPECT-D leaf parameters (prospect_DB.f90) and the first Price soil coefficient (price_soil.c)
STATE = ("HB", "BR", "PCC", "LAI", "N", "Cab", "Car", "Cw", "Cm", "rsl1")
DEFAULT = dict(HB=2.0, BR=2.0, PCC=0.6, LAI=4.0, N=1.2, Cab=30.0, Car=10.0, Cw=0.015, Cm=0.009, rsl1=0.2)


def f32(x):
    """The reference parses flag values with atof into float variables: a member is what gortt would see."""
    return float(np.float32(x))


def member_inputs(state):
    """state (dict or sequence in STATE order) -> (Canopy, LeafSoil) as `gortt -HB .. -BR .. -PCC .. -LAI ..`
    with PROSPECT-D / Price parameters would build them."""
    s = dict(DEFAULT)
    s.update(state if isinstance(state, dict) else dict(zip(STATE, state)))
    canopy = api.make_canopy(newstyle=(f32(s["HB"]), f32(s["BR"]), f32(s["PCC"])), lai=f32(s["LAI"]))
    leaf = api.leaf_soil(prospect=dict(N=s["N"], Cab=s["Cab"], Car=s["Car"], Cw=s["Cw"], Cm=s["Cm"]),
                         rsl=(s["rsl1"], 0.1, 0.03726, -0.002426))
    return canopy, leaf


class Ensemble:
    """N members resident on one GPU: gap probabilities, spectra and band tables of all of them are (re)computed
    on the device by `set_states`, `observe` evaluates the same sun/view geometries for every member."""

    def __init__(self, wavelengths, engine=None):
        self.wl = np.ascontiguousarray(wavelengths, dtype=np.float64)
        self.eng = engine or api.Engine()
        self.n = 0

    def set_states(self, states):
        states = [dict(zip(STATE, s)) if not isinstance(s, dict) else s for s in states]
        pairs = [member_inputs(s) for s in states]
        self.eng.set_members_leaf([p[0] for p in pairs], [p[1] for p in pairs], self.wl, compute_gaps=True)
        self.n = len(pairs)
        return self

    def observe(self, angles_deg, member_begin=0, member_end=None):
        """rsurf[member][line][band] for lines of (vza, vaa, sza, saa) in degrees."""
        return self.eng.rsurf_members_stream(angles_deg, member_begin, self.n if member_end is None else member_end)

    def albedo(self, angles_deg, member_begin=0, member_end=None):
        """energy[member][line][band][3] = albedo, vegetation and soil absorption (gortt -energy per member)."""
        import torch
        ang = torch.as_tensor(np.ascontiguousarray(angles_deg, dtype=np.float64).reshape(-1, 4), device="cuda")
        m1 = self.n if member_end is None else member_end
        out = torch.empty((m1 - member_begin, ang.shape[0], self.wl.size, 3), dtype=torch.float64, device="cuda")
        self.eng.energy_members_dev(ang, member_begin, m1, out)
        self.eng.synchronize()
        return out.cpu().numpy()

    def close(self):
        self.eng.close()


def jacobian(wavelengths, state, angles_deg, rel_step=1e-3, params=STATE, engine=None):
    """Central finite-difference Jacobian d rsurf[line][band] / d state[p] as ONE ensemble of 2 P + 1 members.
    Returns (rsurf0[line][band], J[p][line][band], steps[p]).  Steps are relative to the parameter's value and
    are applied before the float32 rounding of the crown flags, so they must stay well above 1e-7 relative."""
    base = dict(DEFAULT)
    base.update(state if isinstance(state, dict) else dict(zip(STATE, state)))
    members, steps = [base], []
    for p in params:
        h = abs(base[p]) * rel_step
        lo, hi = dict(base), dict(base)
        lo[p], hi[p] = base[p] - h, base[p] + h
        if p in ("HB", "BR", "PCC", "LAI"):              # what the member really gets after the float32 flag parse
            h = 0.5 * (f32(hi[p]) - f32(lo[p]))
        members += [lo, hi]
        steps.append(h)
    ens = Ensemble(wavelengths, engine).set_states(members)
    r = ens.observe(angles_deg)
    if engine is None:
        ens.close()
    J = np.stack([(r[2 + 2 * k] - r[1 + 2 * k]) / (2.0 * steps[k]) for k in range(len(params))])
    return r[0], J, np.array(steps)

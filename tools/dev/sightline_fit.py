"""Feasibility: piecewise degree-P fits in s = 1/d^2 (K uniform intervals per octave of s) of the three
density components along a sightline; max relative error of the mixture and per component."""
import sys
import numpy as np
sys.path.insert(0, ".")
from brutus_amd import galprior as gp

P = int(sys.argv[1]) if len(sys.argv) > 1 else 7
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def comps(d, coord, sgn=None):
    R, Z = gp.galactic_to_RZ(d, coord)
    aZ = np.abs(Z) if sgn is None else sgn * Z
    def disk(Rsc, Zsc, Rs):
        return np.exp(-((np.sqrt(R * R + Rs ** 2) - 8.2) / Rsc + (aZ - 0.025) / Zsc))
    t0 = disk(2.6, 0.3, 2.0)
    t1 = 0.04 * disk(2.0, 0.9, 2.0)
    t2 = 0.005 * np.exp(gp.logn_halo(R, Z))
    return np.array([t0, t1, t2]), Z


nodes = np.cos((2 * np.arange(P + 1) + 1) * np.pi / (2 * (P + 1)))
V = np.vander(nodes, P + 1, increasing=True)
Vinv = np.linalg.inv(V)
worst = 0.
rng = np.random.default_rng(1)
coords = [(0., 0.), (0., 90.), (180., 0.), (0., -90.), (90., 30.), (0., 5.), (0.0, 0.17), (359.9, -0.2)]
coords += [(rng.uniform(0, 360), np.degrees(np.arcsin(rng.uniform(-1, 1)))) for _ in range(24)]
for coord in coords:
    w_mix = w_c = np.zeros(3)
    worst_mix = 0.
    worst_c = np.zeros(3)
    for e in range(-16, 14):          # s octaves: d from 2^8 down to 2^-7
        for k in range(K):
            s_lo = 2. ** e * (1 + k / K)
            w = 2. ** e / K
            sn = s_lo + w * (nodes + 1) / 2
            dn = 1 / np.sqrt(sn)
            _, Zn = comps(dn, coord)
            Zl = gp.galactic_to_RZ(np.array([1 / np.sqrt(s_lo), 1 / np.sqrt(s_lo + w)]), coord)[1]
            sides = [np.sign(Zl[0])] if Zl[0] * Zl[1] > 0 else [1., -1.]
            for sg in sides:
                f, _ = comps(dn, coord, sg)
                c = f @ Vinv.T                      # (3, P+1) monomial coefficients
                t = np.linspace(-1, 1, 400, endpoint=False)
                st = s_lo + w * (t + 1) / 2
                dt = 1 / np.sqrt(st)
                ft, Zt = comps(dt, coord)
                m = (np.sign(Zt) == sg) | (Zt == 0)
                if not m.any():
                    continue
                pt = np.array([np.polyval(c[i][::-1], t) for i in range(3)])
                tot = ft.sum(0)
                err_mix = np.abs(pt - ft).sum(0) / tot
                worst_mix = max(worst_mix, err_mix[m].max())
                worst_c = np.maximum(worst_c, (np.abs(pt - ft) / ft)[:, m].max(1))
    print("l %7.2f b %7.2f  mix %.2e  comps %s" % (coord[0], coord[1], worst_mix, worst_c))
    worst = max(worst, worst_mix)
print("P", P, "K", K, "worst mixture error", worst)

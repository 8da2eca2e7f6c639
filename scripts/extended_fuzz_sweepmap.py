"""CPU fuzz of CircuitSweep._batch (the learned swept-name -> table-entry map, api.py): random builders whose resistor / capacitor
values are random expressions of the sweep variables (identity, proportional, affine, product, sum, reciprocal, square, conditional,
clipped), product and tandem sweeps; the table the map assembles must equal one build per point.  No GPU.
usage: python scripts/extended_fuzz_sweepmap.py [first_seed] [n_seeds]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cedarsim_jl_amd import Circuit, CircuitSweep, ProductSweep, TandemSweep  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fails, hows = [], {}
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    nvar = int(rng.integers(1, 4))
    vnames = ["a", "b", "m"][:nvar]

    def expr():
        kind = rng.integers(0, 10)
        v = vnames[rng.integers(nvar)]
        w = vnames[rng.integers(nvar)]
        k, a0 = float(rng.uniform(0.5, 3.0)), float(rng.uniform(10.0, 1e3))
        if kind == 0: return lambda p: a0
        if kind == 1: return lambda p: p[v]
        if kind == 2: return lambda p: k * p[v]
        if kind == 3: return lambda p: a0 + k * p[v]
        if kind == 4: return lambda p: (p[v] * p[w]) if v != w else p[v] * p[v]
        if kind == 5: return lambda p: a0 + p[v] + p[w]
        if kind == 6: return lambda p: 1e4 / p[v]
        if kind == 7: return lambda p: a0 + p[v] ** 2
        if kind == 8: return lambda p: (p[v] if p[w] > 2.0 else a0)
        return lambda p: max(p[v], 2.0) * k
    exprs = [expr() for _ in range(3)]
    defaults = {n: float(rng.uniform(1.0, 4.0)) for n in vnames}

    def build(**kw):
        p = dict(defaults); p.update(kw)
        c = Circuit()
        c.V("V", "vcc", 0, dc=1.0)
        c.R("R1", "vcc", "mid", 1.0 + abs(exprs[0](p)))
        c.R("R2", "mid", 0, 1.0 + abs(exprs[1](p)))
        c.C("C1", "mid", 0, 1e-12 * (1.0 + abs(exprs[2](p))))
        return c
    if rng.random() < 0.5:
        sweep = ProductSweep(**{n: [float(x) for x in np.round(rng.uniform(0.5, 4.0, int(rng.integers(2, 7))), 3)] for n in vnames})
    else:
        npts = int(rng.integers(5, 60))
        sweep = TandemSweep(**{n: [float(x) for x in rng.uniform(0.5, 4.0, npts)] for n in vnames})
    try:
        cs = CircuitSweep(build, sweep)
        n = len(cs.points)
        base, ids, vals = cs._batch(0, n)
        hows[cs.setup["how"].split(" (")[0]] = hows.get(cs.setup["how"].split(" (")[0], 0) + 1
        for r in range(n):
            c = build(**cs.points[r])
            for i, sl in enumerate(base.slots):
                want = c.dev_par[sl[1]][sl[2]]
                got = vals[i][r]
                if not (got == want or abs(got - want) <= 1e-12 * abs(want)):
                    fails.append((seed, r, sl, got, want, cs.setup["how"]))
                    break
            else:
                # entries that are NOT slots must equal the base build's
                for d in range(len(c.dev_par)):
                    for j in range(len(c.dev_par[d])):
                        if (1, d, j) not in [tuple(s) for s in base.slots] and not (c.dev_par[d][j] == base.dev_par[d][j] or (c.dev_par[d][j] != c.dev_par[d][j] and base.dev_par[d][j] != base.dev_par[d][j])):
                            fails.append((seed, r, "unslotted entry differs", d, j, c.dev_par[d][j], base.dev_par[d][j], cs.setup["how"]))
                continue
            break
    except Exception as ex:  # noqa: BLE001
        fails.append((seed, "raised", type(ex).__name__, str(ex)[:200]))
for f in fails[:30]:
    print("FAIL", f)
print("%d seeds, %d failures, how: %s" % (n_seeds, len(fails), hows))
sys.exit(1 if fails else 0)

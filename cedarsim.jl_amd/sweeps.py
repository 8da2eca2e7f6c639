"""Sweep iterators — host-side mirror of src/sweeps.jl:150-354 (same names, same orderings).

`Sweep`, `ProductSweep`, `TandemSweep`, `SerialSweep`, `sweepify`, `sweepvars`, `split_axes`.
Every sweep iterates tuples of `(name, value)` pairs, exactly like the Julia iterators yield
`((selector, value), ...)`; `SerialSweep` fills the variables it does not set with `None`
(`nothing`, src/sweeps.jl:327-331).  Iteration order of `ProductSweep` is column-major (first axis
fastest), matching `Iterators.product`.
"""
import itertools

import numpy as np


def frange(start, step, stop):
    """Julia `start:step:stop` for floats (inclusive)."""
    n = int(np.floor((stop - start) / step + 1e-9)) + 1
    return [start + i * step for i in range(max(n, 0))]


class Sweep:
    """1-D sweep: Sweep("R1", values) / Sweep(R1=values) (src/sweeps.jl:175-249)."""

    def __init__(self, selector=None, values=None, **kw):
        if selector is None:
            if len(kw) != 1:
                raise ValueError("`Sweep` takes a single variable at a time!")  # src/sweeps.jl:192
            (selector, values), = kw.items()
        elif isinstance(selector, tuple) and values is None:
            selector, values = selector
        elif isinstance(selector, dict) and values is None:
            if len(selector) != 1:
                raise ValueError("`Sweep` takes a single variable at a time!")
            (selector, values), = selector.items()
        self.selector = str(selector)
        self.values = list(np.atleast_1d(values)) if not isinstance(values, (list, tuple, range)) else list(values)

    def __iter__(self):
        for v in self.values:
            yield ((self.selector, v),)

    def __len__(self):
        return len(self.values)

    @property
    def shape(self):
        return (len(self.values),)

    def vars(self):
        return {self.selector}

    def __eq__(self, o):
        return isinstance(o, Sweep) and self.selector == o.selector and list(self.values) == list(o.values)

    def __repr__(self):
        if len(self.values) > 1:
            return "Sweep of %s with %d values over [%s .. %s]" % (self.selector, len(self.values), min(self.values), max(self.values))
        return "Sweep of %s set to %s" % (self.selector, self.values[0])


def _as_sweeps(args, kw):
    out = []
    for a in args:
        if isinstance(a, (Sweep, _Composite)):
            out.append(a)
        elif isinstance(a, dict):
            out.extend(Sweep(k, v) for k, v in a.items())
        elif isinstance(a, tuple) and len(a) == 2:
            out.append(Sweep(a[0], a[1]))
        else:
            raise TypeError("cannot make a Sweep from %r" % (a,))
    out.extend(Sweep(k, v) for k, v in kw.items())
    return out


class _Composite:
    def vars(self):
        s = set()
        for it in self.iterators:
            s |= it.vars()
        return s

    def __len__(self):
        return int(np.prod(self.shape)) if self.shape else 0


class _Product(_Composite):
    def __init__(self, its):
        self.iterators = its

    @property
    def shape(self):
        sh = ()
        for it in self.iterators:
            sh += it.shape
        return sh

    def __iter__(self):
        # column-major: the FIRST iterator varies fastest (Iterators.product)
        lists = [list(it) for it in self.iterators]
        for combo in itertools.product(*reversed(lists)):
            out = ()
            for part in reversed(combo):
                out += part
            yield out


class _Tandem(_Composite):
    def __init__(self, its):
        lens = [len(i) for i in its]
        if any(n != lens[0] for n in lens):
            raise ValueError("TandemSweep requires all sweeps be of the same length!")  # src/sweeps.jl:286
        self.iterators = its

    @property
    def shape(self):
        return (len(self.iterators[0]),)

    def __iter__(self):
        for combo in zip(*self.iterators):
            out = ()
            for part in combo:
                out += part
            yield out


class _Serial(_Composite):
    def __init__(self, its):
        self.iterators = its

    @property
    def shape(self):
        return (sum(len(i) for i in self.iterators),)

    def __iter__(self):
        allv = sorted(self.vars())
        for it in self.iterators:
            for point in it:
                m = {v: None for v in allv}
                m.update(dict(point))
                yield tuple((v, m[v]) for v in allv)


def ProductSweep(*args, **kw):
    """Cartesian product (src/sweeps.jl:261-268); a single argument degenerates to a `Sweep`."""
    its = _as_sweeps(args, kw)
    return its[0] if len(its) == 1 else _Product(its)


def TandemSweep(*args, **kw):
    """Zip (src/sweeps.jl:278-290); all inputs must have the same length."""
    its = _as_sweeps(args, kw)
    return its[0] if len(its) == 1 else _Tandem(its)


def SerialSweep(*args, **kw):
    """Concatenation (src/sweeps.jl:300-338); unset variables are `None`."""
    its = _as_sweeps(args, kw)
    return its[0] if len(its) == 1 else _Serial(its)


def sweepify(x):
    """src/sweeps.jl:349-354: dict → ProductSweep, list → SerialSweep of sweepified items."""
    if isinstance(x, (Sweep, _Composite)):
        return x
    if isinstance(x, dict):
        return ProductSweep(**x)
    if isinstance(x, (list, tuple)):
        return SerialSweep(*[sweepify(i) for i in x])
    return Sweep(x)


def sweepvars(*sweeps):
    s = set()
    for sw in sweeps:
        s |= sw.vars()
    return s


def find_param_ranges(sweep):
    """Per-variable (min, max, count) over a sweep — src/sweeps.jl:507-546."""
    acc = {}
    for point in sweep:
        for k, v in point:
            if v is None:
                continue
            lo, hi, vals = acc.get(k, (v, v, set()))
            vals.add(v)
            acc[k] = (min(lo, v), max(hi, v), vals)
    return {k: (lo, hi, len(vals)) for k, (lo, hi, vals) in acc.items()}


def shard_range(n, rank, world):
    """Contiguous block of sweep points owned by `rank` (S/G samples per GPU)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)

"""The handful of polars calls the reference's example scripts make on small constant tables, over pandas.

examples/rocket/main.py builds its aerodynamic coefficient grid with polars (`pl.from_dict`, nested
`df.group_by([col], maintain_order=True)` iteration, `.agg(pl.col(names).min())`, `.select(pl.col(names))`, `.to_numpy()`;
`df["Mach"].min() / .max() / .unique()`, main.py:150-262) — host-side table preparation, nothing traced.  polars is not in
this image; pandas is.  elodin_amd.compat.install() registers this module as `polars` only when the real one is absent, so
such a script imports unmodified.  Anything beyond this subset raises AttributeError naming what is missing."""
from __future__ import annotations

import types

import numpy as _np
import pandas as _pd


class _Col:
    def __init__(self, names, agg=None):
        self.names, self.agg = ([names] if isinstance(names, str) else list(names)), agg

    def min(self): return _Col(self.names, "min")
    def max(self): return _Col(self.names, "max")
    def mean(self): return _Col(self.names, "mean")
    def first(self): return _Col(self.names, "first")


def col(names, *more):
    return _Col([names, *more] if more else names)


class Series:
    def __init__(self, s: "_pd.Series"): self._s = s
    def min(self): return self._s.min()
    def max(self): return self._s.max()
    def unique(self): return Series(_pd.Series(self._s.unique()))
    def to_numpy(self): return self._s.to_numpy()
    def to_list(self): return self._s.tolist()
    def __len__(self): return len(self._s)
    def __iter__(self): return iter(self._s)


class _GroupBy:
    def __init__(self, df, keys, maintain_order):
        self._df, self._keys = df, ([keys] if isinstance(keys, str) else list(keys))

    def _groups(self):
        return self._df._d.groupby(self._keys, sort=False)          # first-appearance order = maintain_order=True

    def __iter__(self):
        for key, sub in self._groups():
            yield (key if isinstance(key, tuple) else (key,)), DataFrame(sub.reset_index(drop=True))

    def agg(self, *exprs):
        parts = {}
        for e in exprs:
            if not isinstance(e, _Col) or e.agg is None:
                raise NotImplementedError("group_by(...).agg takes pl.col(names).min() / .max() / .mean() / .first() here")
            for n in e.names:
                parts[n] = getattr(self._groups()[n], e.agg)()
        out = _pd.DataFrame(parts).reset_index()
        return DataFrame(out)


class DataFrame:
    def __init__(self, data):
        self._d = data if isinstance(data, _pd.DataFrame) else _pd.DataFrame(data)

    def __getitem__(self, name): return Series(self._d[name])
    def __len__(self): return len(self._d)
    @property
    def columns(self): return list(self._d.columns)
    def group_by(self, by, *more, maintain_order: bool = False):
        keys = ([by] if isinstance(by, str) else list(by)) + list(more)
        return _GroupBy(self, keys, maintain_order)

    def select(self, *exprs):
        names = []
        for e in exprs:
            names += e.names if isinstance(e, _Col) else ([e] if isinstance(e, str) else list(e))
        return DataFrame(self._d[names])

    def to_numpy(self): return self._d.to_numpy(dtype=_np.float64)
    def to_dict(self, as_series: bool = True): return {c: self._d[c].tolist() for c in self._d.columns}


def from_dict(data): return DataFrame(data)


def module() -> types.ModuleType:
    m = types.ModuleType("polars")
    m.__doc__ = __doc__
    m.DataFrame, m.Series, m.col, m.from_dict = DataFrame, Series, col, from_dict

    def missing(name):
        raise AttributeError(f"polars.{name} is not provided by elodin_amd.compat_polars (a pandas-backed subset for example scripts)")
    m.__getattr__ = missing
    return m

"""Dict-of-arrays <-> flat float vector, same contract as pymc/blocking.py:40-103."""

from __future__ import annotations

from typing import Dict, NamedTuple, Optional, Tuple

import numpy as np

PointType = Dict[str, np.ndarray]


class RaveledVars(NamedTuple):
    """pymc/blocking.py:44-46."""

    data: np.ndarray
    point_map_info: Tuple[Tuple[str, Tuple[int, ...], int, np.dtype], ...]


class DictToArrayBijection:
    """pymc/blocking.py:60-103: variables raveled in C order and concatenated in dict order."""

    @staticmethod
    def map(var_dict: PointType) -> RaveledVars:
        items = [(k, np.asarray(v)) for k, v in var_dict.items()]
        if items:
            flat = np.concatenate([v.ravel() for _, v in items])
        else:
            flat = np.array([])
        return RaveledVars(flat, tuple((k, v.shape, v.size, v.dtype) for k, v in items))

    @staticmethod
    def rmap(array: RaveledVars, start_point: Optional[PointType] = None) -> PointType:
        out = dict(start_point) if start_point else {}
        pos = 0
        for name, shape, size, dtype in array.point_map_info:
            out[name] = array.data[pos : pos + size].reshape(shape).astype(dtype)
            pos += size
        return out

"""Tunables of the host side, in ONE place: module constants with built-in defaults that
`F2G_OPTS="name=value,..."` may override at import (the same variable carries the library's own dispatch
options, csrc/common.h: names neither side knows are ignored by that side).  Tests and tools change a
tunable by assigning the module attribute (`ops.X6F_MIN_K = 32`) or, for the library's,
`flow2gan_amd._lib.set_option("x6p", 2)`.

Environment switches that remain on their own (user-facing, or needed before anything is imported):
F2G_GEMM (arithmetic of the GEMMs), F2G_STREAMS (launch lanes), F2G_DETERMINISTIC (no splits on the library's
own initiative), F2G_WEIGHT_CACHE, F2G_LIB_PATH, F2G_DRYRUN, F2G_DIST_TIMEOUT_S."""
from __future__ import annotations

import os


def _parse(text: str) -> dict:
    out = {}
    for item in text.split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip().lower()] = v.strip()
    return out


_OPTS = _parse(os.environ.get("F2G_OPTS", ""))


def opt(name: str, default):
    """Value of tunable `name` (lower case): F2G_OPTS's, else `default`; typed like the default."""
    v = _OPTS.get(name)
    if v is None:
        return default
    if isinstance(default, bool):
        return v not in ("0", "false", "off", "")
    return type(default)(v)

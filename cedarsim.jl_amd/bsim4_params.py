"""Index table of BSIM4 model-card parameters, parsed from include/cedarhip_bsim4_params.def.

The .def file is the single source of truth for the layout of `ch_desc.model_par`
(include/cedarhip.h).  Model cards reach the reference as keyword arguments of the VA-generated
functor, case-insensitively (src/spectre.jl:1113-1149); this table plays that role here.
"""
import os
import re

_DEF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "cedarhip_bsim4_params.def")


def _parse():
    names, ignored = [], set()
    with open(_DEF) as f:
        text = f.read()
    # drop comments and the preprocessor scaffolding
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("#"):
            continue
        for m in re.finditer(r"\b([PBI])\(\s*([A-Za-z0-9_]+)\s*(?:,[^)]*)?\)", line):
            kind, n = m.group(1), m.group(2)
            if kind == "P":
                names.append(n)
            elif kind == "B":
                names.extend([n, "l" + n, "w" + n, "p" + n])
            else:
                ignored.add(n)
    return names, ignored


PARAM_NAMES, IGNORED = _parse()
PARAM_INDEX = {n: i for i, n in enumerate(PARAM_NAMES)}
NPAR = len(PARAM_NAMES)
# binned variants of ignored parameters are ignored too
IGNORED |= {p + n for n in list(IGNORED) for p in "lwp"}

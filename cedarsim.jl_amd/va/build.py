"""Ahead-of-time build of the Verilog-A model library.

    python -m cedarsim_jl_amd.va.build            (also run by the csrc Makefile and __graft_entry__.build())

Parses every model source, generates `csrc/_generated/va_models.hpp` (compiled into libcedarhip.so and into
the test oracle) and writes `lib/va_modules.json`, the parsed modules the host side needs at run time
(parameter names, defaults and ranges, node lists, the analog block for the structure probe).

Sources: `cedarsim.jl_amd/va/library/*.va`, the files named in `CEDARHIP_VA_SOURCES` (os.pathsep separated),
and — when the reference checkout is present on this machine — the CMC BSIM-CMG 107 model the reference
ships and tests with (VerilogAParser.jl/cmc_models/bsimcmg107/bsimcmg.va).  Its text is read where it lies;
only build outputs (git-ignored) derive from it.  The counterpart in the reference is ModelLoader.jl /
the precompiled model packages (src/ModelLoader.jl).
"""
import glob
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
GEN_DIR = os.path.join(PKG, "csrc", "_generated")
GEN_HDR = os.path.join(GEN_DIR, "va_models.hpp")
MOD_JSON = os.path.join(PKG, "lib", "va_modules.json")
REFERENCE_MODELS = ["/root/reference/VerilogAParser.jl/cmc_models/bsimcmg107/bsimcmg.va"]


def sources():
    srcs = sorted(glob.glob(os.path.join(HERE, "library", "*.va")))
    for p in os.environ.get("CEDARHIP_VA_SOURCES", "").split(os.pathsep):
        if p:
            srcs.append(p)
    if os.environ.get("CEDARHIP_VA_NO_REFERENCE_MODELS") != "1":
        srcs += [p for p in REFERENCE_MODELS if os.path.isfile(p)]
    return srcs


def _jsonable(x):
    if isinstance(x, tuple):
        return {"t": [_jsonable(c) for c in x]}
    if isinstance(x, list):
        return [_jsonable(c) for c in x]
    if isinstance(x, dict):
        return {"d": {k: _jsonable(v) for k, v in x.items()}}
    if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
        return {"f": repr(x)}
    return x


def _unjson(x):
    if isinstance(x, dict):
        if "t" in x:
            return tuple(_unjson(c) for c in x["t"])
        if "d" in x:
            return {k: _unjson(v) for k, v in x["d"].items()}
        if "f" in x:
            return float(x["f"])
    if isinstance(x, list):
        return [_unjson(c) for c in x]
    return x


def module_to_json(m):
    return {"name": m.name, "ports": m.ports, "internal": m.internal, "params": _jsonable(m.params), "aliases": m.aliases,
            "vars": m.vars, "var_desc": m.var_desc, "arrays": {k: list(v) for k, v in m.arrays.items()}, "branches": _jsonable(m.branches), "analog": _jsonable(m.analog),
            "functions": {k: {"rtype": f.rtype, "args": _jsonable(f.args), "vars": f.vars, "body": _jsonable(f.body)} for k, f in m.functions.items()}}


def module_from_json(j):
    from .frontend import Function, Module
    m = Module(j["name"])
    m.ports, m.internal = list(j["ports"]), list(j["internal"])
    m.params = [tuple(p) for p in _unjson(j["params"])]
    m.aliases, m.vars = dict(j["aliases"]), dict(j["vars"])
    m.var_desc = dict(j.get("var_desc", {}))
    m.arrays = {k: tuple(v) for k, v in j.get("arrays", {}).items()}
    m.branches = {k: tuple(v) for k, v in _unjson(j["branches"]).items()}
    m.analog = _unjson(j["analog"])
    for k, fj in j["functions"].items():
        f = Function(k, fj["rtype"])
        f.args = [tuple(a) for a in _unjson(fj["args"])]
        f.vars = dict(fj["vars"])
        f.body = _unjson(fj["body"])
        m.functions[k] = f
    return m


def build(verbose=True):
    from .codegen import generate_header
    from .frontend import parse_va_file
    mods, tags = [], []
    for src in sources():
        with open(src, "rb") as f:
            tags.append("%s:%s" % (os.path.basename(src), hashlib.sha256(f.read()).hexdigest()[:12]))
        for m in parse_va_file(src):
            if any(x.name == m.name for x in mods):
                raise SystemExit("duplicate Verilog-A module '%s' (%s)" % (m.name, src))
            mods.append(m)
    hdr = generate_header(mods, "sources: " + ", ".join(tags))
    os.makedirs(GEN_DIR, exist_ok=True)
    os.makedirs(os.path.dirname(MOD_JSON), exist_ok=True)
    old = open(GEN_HDR).read() if os.path.exists(GEN_HDR) else None
    if old != hdr:
        with open(GEN_HDR, "w") as f:
            f.write(hdr)
    with open(MOD_JSON, "w") as f:
        json.dump({"sources": tags, "modules": [module_to_json(m) for m in mods]}, f)
    if verbose:
        print("va build: %d modules (%s) -> %s" % (len(mods), ", ".join(m.name for m in mods), os.path.relpath(GEN_HDR)))
    return mods


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(PKG))
    build()

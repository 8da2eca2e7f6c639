"""Extracts the two BSIM-CMG model cards the reference's inverter deck uses (nmos_lvt, pmos_lvt) from the ASAP7 TT
card file the reference's parser tests hold (SpectreNetlistParser.jl/test/examples/7nm_TT.scs, BSD 3-Clause,
Copyright 2020 Lawrence T. Clark, Vinay Vashishtha, Arizona State University) into a parameter table.
Run in the container that has /root/reference:  python tests/golden/make_asap7_cards.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cedarsim_jl_amd import parse_spectre_models  # noqa: E402

SRC = "/root/reference/SpectreNetlistParser.jl/test/examples/7nm_TT.scs"
models = parse_spectre_models(open(SRC).read())
out = {"source": "SpectreNetlistParser.jl/test/examples/7nm_TT.scs (ASAP7 TT models v1.0 8/3/16; BSD 3-Clause, (c) 2020 L. T. Clark, "
                 "V. Vashishtha, Arizona State University)",
       "cards": {name: {"master": models[name][0], "params": models[name][1]} for name in ("nmos_lvt", "pmos_lvt")}}
json.dump(out, open(os.path.join(HERE, "asap7_tt_lvt_cards.json"), "w"), indent=0, sort_keys=True)
print({k: len(v["params"]) for k, v in out["cards"].items()})

"""cedarsim.jl_amd — MI355X-native DC/transient Newton engine behind CedarSim's problem surface.

Host-side mirror (Python) of the reference's user-facing interface for the hot path:
`Circuit` (flat device table), `dc`/`tran` (dc!/tran!, src/sweeps.jl:437-465), `CircuitSweep` and the
sweep iterators (src/sweeps.jl:150-435).  All numerics run in libcedarhip.so (HIP, gfx950) through
the C-ABI of include/cedarhip.h; there is no CPU fallback.
"""
from .circuit import (Circuit, CedarError, DC, PWL, PULSE, SIN, dc_opts, tran_opts, RETCODES)  # noqa: F401
from .sweeps import (Sweep, ProductSweep, TandemSweep, SerialSweep, sweepify, sweepvars, find_param_ranges, frange, shard_range)  # noqa: F401
from .netlist import parse_spice, parse_spice_file, parse_spectre_models, parse_number, NoBinException  # noqa: F401
from .api import dc, tran, ac, noise, acdec, ACSolution, NoiseSolution, CircuitSweep, Solution, gather_sharded, gather_sharded_device  # noqa: F401

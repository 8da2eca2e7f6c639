# CedarHIP.jl — the reference-side binding of libcedarhip.so.
#
# STATUS: written against the reference's sources (file:line below), NEVER EXECUTED — there is no Julia in this pipeline and
# the reference needs a custom Julia build plus un-vendored packages (DESIGN.md 3).  Every ccall below has the argument shapes
# that examples/c_abi_demo.c and the ctypes binding (cedarsim.jl_amd/engine.py, circuit.py) exercise on the GPU; the
# struct mirrors follow include/cedarhip.h field by field.  A maintainer should expect to fix Julia-level details on first run.
#
# What it does: keeps CedarSim's netlist-compiled circuit closure and the SciMLBase problem surface, and replaces what
# happens inside `solve` (src/sweeps.jl:456) / `dc!` / `tran!` (src/sweeps.jl:437-465):
#   1. StampPass — a binding overlay in the style of AliasInterp (src/aliasextract.jl:10-39): the circuit closure is run ONCE
#      with fake nets; every device functor of src/simpledevices.jl:49-373 and every VA-generated functor
#      (src/vasim.jl:853-867) records (kind, nets, fields, multiplier, scope) instead of emitting equations; the waveform
#      functions of src/spectre_env.jl:144-176 return a tagged reference instead of a value.
#   2. make_desc — the records become the flat ch_desc of include/cedarhip.h.
#   3. ch_circuit_build / ch_set_params / ch_dc / ch_tran — the engine; results come back as a CedarHIPSolution whose
#      `sol[sys.x]`, `sol.t`, `sol.retcode`, `sol(t; idxs)` cover what the reference's tests use
#      (test/gf180_dff.jl:28-33, test/basic.jl:37-42).
module CedarHIP

using CedarSim, SciMLBase
using CedarSim: AbstractNet, AbstractSim, DefaultSim, ParamSim, SimSpec, ParallelInstances, DScope, debug_scope, spec, sim_mode,
                SimpleResistor, SimpleCapacitor, SimpleInductor, VoltageSource, CurrentSource, vcvs, vccs, Gnd, Net,
                undefault, isdefault, CedarDCOp, CedarTranOp
using CedarSim.VerilogAEnvironment: VAModel
import CedarSim.SpectreEnvironment
using CassetteOverlay
using Base.ScopedValues: with

const lib = joinpath(@__DIR__, "..", "lib", "libcedarhip.so")
const CH_DEV = (R = Int32(1), C = Int32(2), L = Int32(3), V = Int32(4), I = Int32(5), VCVS = Int32(6), VCCS = Int32(7), MOS = Int32(8), VA = Int32(9))
const CH_SRC = (DC = Int32(0), PWL = Int32(1), PULSE = Int32(2), SIN = Int32(3))
const NNODE, NPAR, NIPAR, SRC_NPAR = 8, 8, 2, 8   # CH_DEV_NNODE, CH_DEV_NPAR, CH_DEV_NIPAR, CH_SRC_NPAR

# ---- struct mirrors (field order == include/cedarhip.h) ---------------------------------------------------------------
struct ChDesc
    n_nodes::Int32; n_dev::Int32
    dev_kind::Ptr{Int32}; dev_node::Ptr{Int32}; dev_ipar::Ptr{Int32}; dev_par::Ptr{Float64}; dev_mult::Ptr{Float64}
    n_src::Int32
    src_kind::Ptr{Int32}; src_dc::Ptr{Float64}; src_par::Ptr{Float64}; src_pwl_ofs::Ptr{Int32}; pwl_t::Ptr{Float64}; pwl_y::Ptr{Float64}
    n_model::Int32; model_par::Ptr{Float64}
    temp::Float64; gmin::Float64; scale::Float64
    n_slot::Int32; slot_kind::Ptr{Int32}; slot_a::Ptr{Int32}; slot_b::Ptr{Int32}
    n_obs::Int32; obs_kind::Ptr{Int32}; obs_index::Ptr{Int32}
    src_ac::Ptr{Float64}
    n_va_par::Int64; va_par::Ptr{Float64}
end
struct ChDcOpts
    abstol::Float64; maxiters::Int32; n_restarts::Int32; seed::UInt64; tran_mode::Int32; dv_max::Float64; x0::Ptr{Float64}
end
struct ChTranOpts
    abstol::Float64; reltol::Float64; max_order::Int32; dtmin::Float64; dtmax::Float64; dt0::Float64
    max_steps::Int32; newton_maxiters::Int32; n_saveat::Int32; saveat::Ptr{Float64}; dc::ChDcOpts; skip_dc::Int32; stepper::Int32
end
struct ChStats
    nf::Int64; njacs::Int64; nfactors::Int64; nsolve::Int64; nnonliniter::Int64; nnonlinconvfail::Int64
    naccept::Int64; nreject::Int64; nrestarts::Int64
    wall_seconds::Float64; dc_seconds::Float64; device_seconds::Float64
    n_kernel_launches::Int64; n_block_iters::Int64; n_step_attempts::Int64
    barrier_seconds::Float64; stepper::Int32; stepper_mode::Int32
    step_kernel_seconds::Float64; step_kernel_launches::Int64; step_block_iters::Int64
end
ChStats() = ChStats(ntuple(_ -> 0, 9)..., 0.0, 0.0, 0.0, 0, 0, 0, 0.0, Int32(0), Int32(0), 0.0, 0, 0)
struct ChInfo
    n_nodes::Int32; n_branches::Int32; n_mna::Int32; n_unknowns::Int32; n_known::Int32; n_alias::Int32; n_components::Int32
    max_component::Int32; n_classes::Int32; n_mos::Int32; n_mos_classes::Int32; path::Int32
    nnz_jac::Int64; nnz_lu::Int64; n_samples::Int32
end

# ---- what the overlay records ------------------------------------------------------------------------------------------
struct FakeNet <: AbstractNet     # stands in for Net (src/simulate_ir.jl:28-54): an id instead of a DAE variable
    id::Int32
    name::Any
    multiplier::Float64
end
struct WaveRef <: Real            # a source waveform captured instead of evaluated (spectre_env.jl:144-176)
    id::Int32
end
Base.promote_rule(::Type{WaveRef}, ::Type{<:Real}) = Any   # VoltageSource's promote(dc, tran, ...) keeps the reference as is
struct Wave
    kind::Int32; par::NTuple{8,Float64}; ts::Vector{Float64}; ys::Vector{Float64}
end
mutable struct StampTable
    net_ids::Dict{Any,Int32}; net_names::Vector{Any}
    kind::Vector{Int32}; node::Vector{NTuple{NNODE,Int32}}; ipar::Vector{NTuple{NIPAR,Int32}}; par::Vector{NTuple{NPAR,Float64}}
    mult::Vector{Float64}; scope::Vector{Any}
    waves::Vector{Wave}                       # waveform table (WaveRef.id indexes it)
    src_dc::Vector{Float64}; src_wave::Vector{Int32}; src_ac::Vector{Float64}   # one entry per V / I device
    models::Vector{Vector{Float64}}; model_keys::Vector{Any}                     # BSIM4 cards [CH_B4_NPAR], NaN = not given
    va_par::Vector{Float64}
end
StampTable() = StampTable(Dict{Any,Int32}(), Any[], Int32[], NTuple{NNODE,Int32}[], NTuple{NIPAR,Int32}[], NTuple{NPAR,Float64}[],
                          Float64[], Any[], Wave[], Float64[], Int32[], Float64[], Vector{Float64}[], Any[], Float64[])

function intern!(tbl::StampTable, name)
    name === nothing && (name = gensym(:net))
    get!(tbl.net_ids, name) do
        push!(tbl.net_names, name)
        Int32(length(tbl.net_names))          # node ids start at 1; ground is whatever net Gnd() ties to 0 (resolved in make_desc)
    end
end
nodes8(nets) = ntuple(k -> k <= length(nets) ? nets[k].id : Int32(0), NNODE)
par8(vals) = ntuple(k -> k <= length(vals) ? Float64(vals[k]) : NaN, NPAR)
function record!(tbl::StampTable, kind, nets, pars, dscope; ipar = (Int32(0), Int32(0)))
    push!(tbl.kind, kind); push!(tbl.node, nodes8(nets)); push!(tbl.ipar, ipar); push!(tbl.par, par8(pars))
    # ParallelInstances multiplies the nets' multipliers (simulate_ir.jl:56-75): a device's own m is that of its first net
    push!(tbl.mult, isempty(nets) ? 1.0 : nets[1].multiplier); push!(tbl.scope, dscope)
    length(tbl.kind)
end
function source!(tbl::StampTable, dc, tran, ac)
    push!(tbl.src_dc, Float64(dc isa WaveRef ? 0.0 : dc))
    if tran isa WaveRef
        push!(tbl.src_wave, tran.id)
    else                                      # a plain number: constant waveform
        push!(tbl.waves, Wave(CH_SRC.DC, par8((Float64(tran),)), Float64[], Float64[])); push!(tbl.src_wave, Int32(length(tbl.waves)))
    end
    push!(tbl.src_ac, abs(ac))
    Int32(length(tbl.src_dc) - 1)             # 0-based source index for dev_ipar[0]
end

# ---- the overlay pass (same shape as AliasInterp, src/aliasextract.jl:10-39; call site src/simulate_ir.jl:85-87) ---------
struct StampPass <: CassetteOverlay.AbstractBindingOverlay{nothing,nothing}
    tbl::StampTable
    ground::Vector{Int32}
end
StampPass() = StampPass(StampTable(), Int32[])

(self::StampPass)(::Type{Net}, name = nothing, multiplier::Float64 = 1.0) = FakeNet(intern!(self.tbl, name), name, multiplier)
(self::StampPass)(::Type{Net}, net::FakeNet, multiplier::Float64) = FakeNet(net.id, net.name, net.multiplier * multiplier)
(self::StampPass)(::typeof(with), f, pairs...) = with(pairs...) do; self(f); end
# kcl!/branch!/equation! never run: every functor below returns before reaching them.

function (self::StampPass)(R::SimpleResistor, A, B; dscope = CedarSim.defaultscope(R))      # simpledevices.jl:65-77
    res = isdefault(R.r) ? R.rsh * (R.l - R.short) / (R.w - R.narrow) : undefault(R.r)
    record!(self.tbl, CH_DEV.R, (A, B), (res,), dscope); nothing
end
(self::StampPass)(C::SimpleCapacitor, A, B; dscope = CedarSim.defaultscope(C)) = (record!(self.tbl, CH_DEV.C, (A, B), (C.capacitance,), dscope); nothing)   # :105-109
(self::StampPass)(L::SimpleInductor, A, B; dscope = CedarSim.defaultscope(L)) = (record!(self.tbl, CH_DEV.L, (A, B), (L.inductance,), dscope); nothing)      # :128-132
function (self::StampPass)(VS::VoltageSource, A, B; dscope = CedarSim.defaultscope(VS))     # :288-300
    s = source!(self.tbl, VS.dc, VS.tran, VS.ac)
    record!(self.tbl, CH_DEV.V, (A, B), (), dscope; ipar = (s, Int32(0))); nothing
end
function (self::StampPass)(IS::CurrentSource, A, B; dscope = CedarSim.defaultscope(IS))     # :327-339
    s = source!(self.tbl, IS.dc, IS.tran, IS.ac)
    record!(self.tbl, CH_DEV.I, (A, B), (), dscope; ipar = (s, Int32(0))); nothing
end
function (self::StampPass)(S::vcvs, A, B; dscope = CedarSim.defaultscope(S))                 # :347-351 — a constant voltage
    s = source!(self.tbl, S.voltage, S.voltage, 0.0)
    record!(self.tbl, CH_DEV.V, (A, B), (), dscope; ipar = (s, Int32(0))); nothing
end
(self::StampPass)(S::vcvs, A, B, C, D; dscope = CedarSim.defaultscope(S)) = (record!(self.tbl, CH_DEV.VCVS, (A, B, C, D), (S.gain,), dscope); nothing)      # :352-356
function (self::StampPass)(S::vccs, A, B; dscope = CedarSim.defaultscope(S))                 # :364-368 — a constant current
    s = source!(self.tbl, S.current, S.current, 0.0)
    record!(self.tbl, CH_DEV.I, (A, B), (), dscope; ipar = (s, Int32(0))); nothing
end
(self::StampPass)(S::vccs, A, B, C, D; dscope = CedarSim.defaultscope(S)) = (record!(self.tbl, CH_DEV.VCCS, (A, B, C, D), (S.gain,), dscope); nothing)      # :369-373
(self::StampPass)(::Gnd, A; dscope = debug_scope[]) = (push!(self.ground, A.id); nothing)                                                                   # :305-313

# waveforms: captured, not evaluated
(self::StampPass)(::typeof(SpectreEnvironment.pwl), wave) = begin
    ts, ys = Float64.(wave[1:2:end]), Float64.(wave[2:2:end])
    push!(self.tbl.waves, Wave(CH_SRC.PWL, par8(()), ts, ys)); WaveRef(Int32(length(self.tbl.waves)))
end
(self::StampPass)(::typeof(SpectreEnvironment.pulse), v1, v2, td, tr, tf, pw = Inf, period = Inf, count = -1) = begin
    push!(self.tbl.waves, Wave(CH_SRC.PULSE, par8((v1, v2, td, tr, tf, pw, period)), Float64[], Float64[])); WaveRef(Int32(length(self.tbl.waves)))
end
(self::StampPass)(::typeof(SpectreEnvironment.spsin), vo, va, freq, td = 0, theta = 0, phase = 0, ncycles = Inf) = begin
    push!(self.tbl.waves, Wave(CH_SRC.SIN, par8((vo, va, freq, td, theta, phase, ncycles)), Float64[], Float64[])); WaveRef(Int32(length(self.tbl.waves)))
end

# VA-generated functors (src/vasim.jl:853-867): a struct <: VAModel whose fields are the module's parameters (DefaultOr).
# BSIM4 (level 14/54, src/spectre.jl:589-630) has the engine's hand-written functor; every other module must have been
# compiled into the library by cedarsim.jl_amd/va (ch_va_find), like the reference's precompiled model packages.
va_module_name(dev) = lowercase(String(nameof(typeof(dev))))
b4_index(name) = (i = findfirst(==(String(name)), B4_NAMES[]); i === nothing ? 0 : i)
const B4_NAMES = Ref{Vector{String}}(String[])
function b4_names()
    if isempty(B4_NAMES[])
        n = ccall((:ch_bsim4_npar, lib), Int32, ())
        B4_NAMES[] = [unsafe_string(ccall((:ch_bsim4_param_name, lib), Cstring, (Int32,), Int32(i - 1))) for i in 1:n]
    end
    B4_NAMES[]
end
function (self::StampPass)(dev::VAModel, nets...; dscope = CedarSim.GenScope(debug_scope[], nameof(typeof(dev))))
    T = typeof(dev)
    given(f) = !isdefault(getfield(dev, f))
    if startswith(va_module_name(dev), "bsim4")
        card = fill(NaN, length(b4_names()))
        inst = Dict{Symbol,Float64}()
        for f in fieldnames(T)
            given(f) || continue
            v = Float64(undefault(getfield(dev, f))); lf = Symbol(lowercase(String(f)))
            if lf in (:w, :l, :nf, :as, :ad, :ps, :pd); inst[lf] = v
            else i = b4_index(lowercase(String(f))); i > 0 && (card[i] = v) end
        end
        key = (T, card)
        m = findfirst(==(key), self.tbl.model_keys)
        m === nothing && (push!(self.tbl.models, card); push!(self.tbl.model_keys, key); m = length(self.tbl.models))
        record!(self.tbl, CH_DEV.MOS, nets, (get(inst, :w, NaN), get(inst, :l, NaN), get(inst, :nf, NaN), get(inst, :as, NaN), get(inst, :ad, NaN),
                                            get(inst, :ps, NaN), get(inst, :pd, NaN)), dscope; ipar = (Int32(m - 1), Int32(0)))
    else
        id = ccall((:ch_va_find, lib), Int32, (Cstring,), va_module_name(dev))
        id < 0 && error("Verilog-A module $(va_module_name(dev)) is not compiled into libcedarhip.so (cedarsim.jl_amd/va/build.py)")
        np = Ref{Int32}(0); nn = Ref{Int32}(0); npt = Ref{Int32}(0)
        ccall((:ch_va_module_info, lib), Int32, (Int32, Ref{Int32}, Ref{Int32}, Ref{Int32}), id, npt, nn, np)
        ofs = length(self.tbl.va_par)
        vals, flags = zeros(np[]), zeros(np[])
        for k in 1:np[]                                    # declaration order of the module; defaults come from the functor's fields
            f = Symbol(unsafe_string(ccall((:ch_va_param_name, lib), Cstring, (Int32, Int32), id, Int32(k - 1))))
            hasfield(T, f) || continue
            vals[k] = Float64(undefault(getfield(dev, f))); flags[k] = given(f) ? 1.0 : 0.0
        end
        append!(self.tbl.va_par, vals); append!(self.tbl.va_par, flags)
        # internal nets of the module are extra circuit nodes, allocated here after the ports
        inets = [FakeNet(intern!(self.tbl, nothing), nothing, nets[1].multiplier) for _ in (length(nets) + 1):nn[]]
        record!(self.tbl, CH_DEV.VA, (nets..., inets...), (), dscope; ipar = (id, Int32(ofs)))
    end
    nothing
end

"Run the circuit closure of `sim` once under the overlay; returns the filled pass (table + ground nets)."
function stamp_extract(sim::AbstractSim)
    pass = StampPass()
    b4_names()
    with(spec => SimSpec(time = 0.0), sim_mode => :tran, debug_scope => DScope()) do
        pass(getfield(sim, :circuit))
    end
    pass
end

"Flat arrays of the description; `keep` owns them for the lifetime of the ccall."
function make_desc(pass::StampPass, sp::SimSpec = SimSpec(); slots = NTuple{3,Int32}[])
    tbl = pass.tbl
    nn = length(tbl.net_names)
    gnd = Set(pass.ground)
    remap = zeros(Int32, nn); k = Int32(0)
    for i in 1:nn; remap[i] = (Int32(i) in gnd) ? Int32(0) : (k += Int32(1)); end      # nets tied by Gnd() become node 0
    nd = length(tbl.kind)
    dev_node = Int32[(n = tbl.node[d][j]; n == 0 ? Int32(0) : remap[n]) for j in 1:NNODE, d in 1:nd][:]
    dev_ipar = Int32[tbl.ipar[d][j] for j in 1:NIPAR, d in 1:nd][:]
    dev_par = Float64[tbl.par[d][j] for j in 1:NPAR, d in 1:nd][:]
    ns = length(tbl.src_dc)
    src_kind = Int32[tbl.waves[tbl.src_wave[s]].kind for s in 1:ns]
    src_par = Float64[tbl.waves[tbl.src_wave[s]].par[j] for j in 1:SRC_NPAR, s in 1:ns][:]
    pwl_ofs = Int32[0]; pwl_t = Float64[]; pwl_y = Float64[]
    for s in 1:ns; w = tbl.waves[tbl.src_wave[s]]; append!(pwl_t, w.ts); append!(pwl_y, w.ys); push!(pwl_ofs, Int32(length(pwl_t))); end
    model_par = isempty(tbl.models) ? Float64[] : reduce(vcat, tbl.models)
    obs_kind = zeros(Int32, Int(k)); obs_index = Int32.(1:k)                         # every node voltage is an observable
    slot_kind = Int32[s[1] for s in slots]; slot_a = Int32[s[2] for s in slots]; slot_b = Int32[s[3] for s in slots]
    keep = (tbl.kind, dev_node, dev_ipar, dev_par, tbl.mult, src_kind, tbl.src_dc, src_par, pwl_ofs, pwl_t, pwl_y, model_par,
            slot_kind, slot_a, slot_b, obs_kind, obs_index, tbl.src_ac, tbl.va_par)
    desc = ChDesc(k, nd, pointer(tbl.kind), pointer(dev_node), pointer(dev_ipar), pointer(dev_par), pointer(tbl.mult),
                  ns, pointer(src_kind), pointer(tbl.src_dc), pointer(src_par), pointer(pwl_ofs), pointer(pwl_t), pointer(pwl_y),
                  length(tbl.models), pointer(model_par), undefault(sp.temp), undefault(sp.gmin), undefault(sp.scale),
                  length(slots), pointer(slot_kind), pointer(slot_a), pointer(slot_b), length(obs_kind), pointer(obs_kind), pointer(obs_index),
                  pointer(tbl.src_ac), length(tbl.va_par), isempty(tbl.va_par) ? Ptr{Float64}(C_NULL) : pointer(tbl.va_par))
    names = Dict{Any,Int}(tbl.net_names[i] => Int(remap[i]) for i in 1:nn if remap[i] > 0)   # scope/name -> observable row
    desc, keep, names
end

# ---- context and error mapping -------------------------------------------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function context(dev = 0)
    if CTX[] == C_NULL
        buf = zeros(UInt8, 512)
        h = ccall((:ch_create, lib), Ptr{Cvoid}, (Cint, Ptr{UInt8}, Csize_t), dev, buf, 512)
        h == C_NULL && error("cedarhip: " * unsafe_string(pointer(buf)))   # no CPU fallback
        CTX[] = h
    end
    CTX[]
end
last_error() = unsafe_string(ccall((:ch_last_error, lib), Cstring, (Ptr{Cvoid},), context()))
# CH_ERR_* -> SciML retcodes (SURVEY 8(b); src/dcop.jl:141-145 maps solver failures to InitialFailure)
retcode(rc) = rc == 0 ? ReturnCode.Success : rc == -3 ? ReturnCode.InitialFailure : rc == -2 ? ReturnCode.InitialFailure :
              rc == -4 ? ReturnCode.DtLessThanMin : rc == -7 ? ReturnCode.MaxIters : ReturnCode.Failure
function check_build(circ)
    circ == C_NULL && throw(CedarSim.CedarError(last_error()))       # src/util.jl:14-21
    circ
end

# ---- solution object -----------------------------------------------------------------------------------------------------
struct CedarHIPSolution{P}
    prob::P
    t::Vector{Float64}
    u::Array{Float64,3}            # [n_samples, n_times, n_obs] (column-major view of the library's [n_obs][n_times][n_samples])
    names::Dict{Any,Int}           # net scope / name -> observable row
    retcode::ReturnCode.T
    stats::ChStats
end
obs_row(sol::CedarHIPSolution, ref) = get(sol.names, ref) do
    # sys.node_q style references carry a scope: try the scope itself, then its last name component
    haskey(sol.names, getfield(ref, :name)) ? sol.names[getfield(ref, :name)] : throw(KeyError(ref))
end
Base.getindex(sol::CedarHIPSolution, ref) = sol.u[1, :, obs_row(sol, ref)]
Base.getindex(sol::CedarHIPSolution, ref, sample::Integer) = sol.u[sample, :, obs_row(sol, ref)]
function (sol::CedarHIPSolution)(t::Real; idxs = nothing)        # sol(t, idxs = [sys.node_q]) (test/gf180_dff.jl:29-33): piecewise linear
    i = clamp(searchsortedlast(sol.t, t), 1, length(sol.t) - 1)
    θ = (t - sol.t[i]) / (sol.t[i + 1] - sol.t[i])
    rows = idxs === nothing ? collect(1:size(sol.u, 3)) : [obs_row(sol, r) for r in idxs]
    [(1 - θ) * sol.u[1, i, r] + θ * sol.u[1, i + 1, r] for r in rows]
end

# ---- solve ---------------------------------------------------------------------------------------------------------------
struct CedarHIPAlg <: SciMLBase.AbstractDAEAlgorithm end

sim_spec(sim) = hasproperty(sim, :spec) ? getfield(sim, :spec) : SimSpec()
dcopts(alg; tran_mode = false) = ChDcOpts(alg.abstol, Int32(200), Int32(10), UInt64(10), Int32(tran_mode), 2.0, Ptr{Float64}(C_NULL))   # src/dcop.jl:28,53

function build(sim; slots = NTuple{3,Int32}[])
    pass = stamp_extract(sim)
    desc, keep, names = make_desc(pass, sim_spec(sim); slots)
    circ = GC.@preserve keep check_build(ccall((:ch_circuit_build, lib), Ptr{Cvoid}, (Ptr{Cvoid}, Ref{ChDesc}), context(), desc))
    circ, Int(desc.n_obs), names
end

function run_tran(circ, n_obs, n_samples, tspan, abstol, reltol, initializealg, saveat)
    sv = Float64.(collect(saveat))
    opts = ChTranOpts(abstol, reltol, Int32(5), 0.0, 0.0, 0.0, Int32(0), Int32(10), Int32(length(sv)), isempty(sv) ? Ptr{Float64}(C_NULL) : pointer(sv),
                      dcopts(initializealg; tran_mode = initializealg isa CedarTranOp), Int32(0), Int32(0))
    res = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve sv ccall((:ch_tran, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Ref{ChTranOpts}, Ref{Ptr{Cvoid}}), circ, tspan[1], tspan[2], opts, res)
    res[] == C_NULL && throw(CedarSim.CedarError(last_error()))     # an exception barrier answered (CH_ERR_NOMEM / CH_ERR_INTERNAL)
    nt = ccall((:ch_result_n_times, lib), Int64, (Ptr{Cvoid},), res[])
    t = copy(unsafe_wrap(Array, ccall((:ch_result_times, lib), Ptr{Float64}, (Ptr{Cvoid},), res[]), nt))
    u = nt > 0 && n_obs > 0 ? copy(unsafe_wrap(Array, ccall((:ch_result_values, lib), Ptr{Float64}, (Ptr{Cvoid},), res[]), (n_samples, nt, n_obs))) :
                              zeros(n_samples, nt, n_obs)
    st = Ref(ChStats())
    ccall((:ch_result_stats, lib), Cint, (Ptr{Cvoid}, Ref{ChStats}), res[], st)
    ccall((:ch_result_free, lib), Cvoid, (Ptr{Cvoid},), res[])
    rc, t, u, st[]
end

function SciMLBase.__solve(prob::DAEProblem, ::CedarHIPAlg; abstol = 1e-6, reltol = 1e-3, initializealg = CedarDCOp(), saveat = Float64[], kwargs...)
    circ, n_obs, names = build(prob.p)
    try
        rc, t, u, st = run_tran(circ, n_obs, 1, prob.tspan, abstol, reltol, initializealg, saveat)
        CedarHIPSolution(prob, t, u, names, retcode(rc), st)
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), circ)
    end
end

"DC operating point (dc!, src/sweeps.jl:437-447): node voltages by net name."
function dc_hip(sim; abstol = 1e-10)
    circ, n_obs, names = build(sim)
    try
        info = Ref(ChInfo(ntuple(_ -> Int32(0), 12)..., 0, 0, Int32(0)))
        ccall((:ch_circuit_info, lib), Cint, (Ptr{Cvoid}, Ref{ChInfo}), circ, info)
        x = zeros(Float64, max(1, Int(info[].n_mna))); status = zeros(Int32, 1); st = Ref(ChStats())   # x_out is [n_samples][n_mna]
        rc = ccall((:ch_dc, lib), Cint, (Ptr{Cvoid}, Ref{ChDcOpts}, Ptr{Float64}, Ptr{Int32}, Ref{ChStats}), circ, dcopts(CedarDCOp(; abstol)), x, status, st)
        rc == 0 || @warn "DC operating point analysis failed" rc last_error()          # src/dcop.jl:141-145
        Dict(k => x[v] for (k, v) in names), retcode(rc)
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), circ)
    end
end

# ---- sweeps: every point a sample of ONE batched solve (replaces the remake loop of src/sweeps.jl:471-502) -----------------
# `slots` maps swept parameters to engine slots {kind, a, b} (include/cedarhip.h CH_SLOT_*); `values[slot][point]`.
function tran_sweep_hip(sim, tspan, slots::Vector{NTuple{3,Int32}}, values::Matrix{Float64}; abstol = 1e-6, reltol = 1e-3,
                        initializealg = CedarDCOp(), saveat = Float64[])
    circ, n_obs, names = build(sim; slots)
    try
        S = size(values, 2)
        ccall((:ch_set_samples, lib), Cint, (Ptr{Cvoid}, Int32), circ, Int32(S)) == 0 || throw(CedarSim.CedarError(last_error()))
        ids = Int32.(0:length(slots) - 1)
        vals = Float64[values[i, s] for s in 1:S, i in 1:length(slots)][:]     # slot-major: vals[slot_i * S + s]
        ccall((:ch_set_params, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Ptr{Float64}), circ, Int32(0), Int32(S), Int32(length(slots)), ids, vals) == 0 ||
            throw(CedarSim.CedarError(last_error()))
        rc, t, u, st = run_tran(circ, n_obs, S, tspan, abstol, reltol, initializealg, saveat)
        CedarHIPSolution(nothing, t, u, names, retcode(rc), st)
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), circ)
    end
end

# The reference's broadcast entry points (src/sweeps.jl:471-502) keep their shape: `tran!.(Ref(sys), Ref(tspan), sims)` lands here
# when the sweep varies engine-visible parameters only; anything else falls back to CedarSim's own loop.
tran_hip!(prob::DAEProblem; kwargs...) = solve(prob, CedarHIPAlg(); kwargs...)

# freqresp(ac, sym, ωs) (src/ac.jl:267-284) and PSD(noise, sym, ωs) (src/ac.jl:286-305) on the GPU
function freqresp_hip(circ_handle::Ptr{Cvoid}, n_mna::Integer, ωs::Vector{Float64}; abstol = 1e-10)
    f = ωs ./ 2π
    out = zeros(Float64, 2, n_mna, length(f))
    rc = ccall((:ch_ac, lib), Cint, (Ptr{Cvoid}, Ref{ChDcOpts}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), circ_handle, dcopts(CedarDCOp(; abstol)), length(f), f, out, C_NULL)
    rc == 0 || error(last_error())
    complex.(out[1, :, :], out[2, :, :])
end
function psd_hip(circ_handle::Ptr{Cvoid}, out_kind::Integer, out_index::Integer, ωs::Vector{Float64}; abstol = 1e-10)
    f = ωs ./ 2π
    out = zeros(Float64, length(f))
    rc = ccall((:ch_noise, lib), Cint, (Ptr{Cvoid}, Ref{ChDcOpts}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
               circ_handle, dcopts(CedarDCOp(; abstol)), out_kind, out_index, length(f), f, out, C_NULL)
    rc == 0 || error(last_error())
    out
end

end # module

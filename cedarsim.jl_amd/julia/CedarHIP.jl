# CedarHIP.jl — the reference-side binding of libcedarhip.so.
#
# STATUS: written against the reference's sources (file:line below) and desk-checked statement by statement against
# test/DFF/DFF_cap_all.cir (INTEGRATION.md section 3 walks the deck through every method this file defines); NEVER EXECUTED —
# there is no Julia in this pipeline and the reference needs a custom Julia build plus un-vendored packages (DESIGN.md 3).
# Every ccall below has the argument shapes that examples/c_abi_demo.c and the ctypes binding (cedarsim.jl_amd/engine.py,
# circuit.py) exercise on the GPU; the struct mirrors follow include/cedarhip.h field by field.
#
# What it does: keeps CedarSim's netlist-compiled circuit closure and the SciMLBase problem surface, and replaces what
# happens inside `solve` (src/sweeps.jl:456) / `dc!` / `tran!` (src/sweeps.jl:437-465) and the sweep broadcast (:471-502):
#   1. StampPass — a binding overlay in the style of AliasInterp (src/aliasextract.jl:10-39): the circuit closure is run ONCE
#      with fake nets.  Under a binding overlay every call `f(args...)` of overlaid code becomes `pass(f, args...)`, and a
#      keyword call `f(args...; kw...)` — lowered by Julia to `Core.kwcall(kw, f, args...)` — becomes
#      `pass(Core.kwcall, kw, f, args...)`.  The pass therefore defines, for every contact point, a positional method and a
#      `Core.kwcall` method:
#        * the CONSTRUCTORS `VoltageSource(; dc, tran, ac)` / `CurrentSource(; …)` (src/simpledevices.jl:279-284, 319-324) —
#          intercepted because `struct VoltageSource{T}; dc::T; tran::T; ac::Complex{T}` cannot hold a number next to a captured
#          waveform, and `promote(dc, tran, …)` must never see one;
#        * every device functor of src/simpledevices.jl:49-373 and every VA-generated functor (src/vasim.jl:853-867): they
#          record (kind, nets, fields, multiplier, scope) instead of emitting equations;
#        * `pwl` / `pulse` / `spsin` (src/spectre_env.jl:144-176): they return a WaveRef (a plain struct, NOT a Real);
#        * `Net`, `net_alias`, `with`, and the scope constructors `DScope` / `GenScope` (names are recorded as symbol paths).
#   2. make_desc — the records become the flat ch_desc of include/cedarhip.h.
#   3. ch_circuit_build / ch_set_params / ch_dc / ch_tran — the engine; results come back as a CedarHIPSolution whose
#      `sol[sys.node_q]`, `sol.t`, `sol.retcode`, `sol(t; idxs)` cover what the reference's tests use
#      (test/gf180_dff.jl:28-33, test/basic.jl:37-42).
#   4. `dc!(cs::CircuitSweep, CedarHIPAlg())` / `tran!(cs, tspan, CedarHIPAlg())`: the swept names of the iterator are mapped
#      to engine slots (CH_SLOT_*) by diffing the stamp tables of a handful of `ParamSim`s, all points become samples of ONE
#      batched solve (`ch_set_samples` + ONE `ch_set_params`).
module CedarHIP

using CedarSim, SciMLBase, DAECompiler
using CedarSim: AbstractNet, AbstractSim, DefaultSim, ParamSim, ParamLens, SimSpec, ParallelInstances, DScope, debug_scope, spec, sim_mode,
                SimpleResistor, SimpleCapacitor, SimpleInductor, VoltageSource, CurrentSource, vcvs, vccs, Gnd, Net, net_alias,
                undefault, isdefault, CedarDCOp, CedarTranOp, CircuitSweep, pwl_at_time
using CedarSim.VerilogAEnvironment: VAModel
using DAECompiler: IRODESystem
using DAECompiler.Intrinsics: AbstractScope, GenScope
import CedarSim.SpectreEnvironment
import CedarSim: dc!, tran!
using CassetteOverlay
using Base.ScopedValues: with

const lib = joinpath(@__DIR__, "..", "lib", "libcedarhip.so")
const CH_DEV = (R = Int32(1), C = Int32(2), L = Int32(3), V = Int32(4), I = Int32(5), VCVS = Int32(6), VCCS = Int32(7), MOS = Int32(8), VA = Int32(9))
const CH_SRC = (DC = Int32(0), PWL = Int32(1), PULSE = Int32(2), SIN = Int32(3))
const CH_SLOT = (DEV_PAR = Int32(1), MODEL_PAR = Int32(2), SRC_DC = Int32(3), SRC_PAR = Int32(4), TEMP = Int32(5), GMIN = Int32(6), DEV_MULT = Int32(7), VA_PAR = Int32(8))
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
    max_steps::Int32; newton_maxiters::Int32; n_saveat::Int32; saveat::Ptr{Float64}; dc::ChDcOpts; skip_dc::Int32; stepper::Int32; step_control::Int32
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

# ---- names: symbol paths instead of DAECompiler scopes ------------------------------------------------------------------
# The reference names nets and devices with `DScope(parent, name)` (src/simulate_ir.jl:51, :84-92; src/simpledevices.jl:31-47)
# and reads them back as `sys.node_q`, `sys.x1.r1.I` (a DAECompiler ScopeRef).  The overlay intercepts the scope CONSTRUCTORS
# and records a tuple of symbols; nothing here depends on the field layout of DAECompiler's types.
struct PathScope <: AbstractScope
    path::Tuple{Vararg{Symbol}}
end
"What `sys.node_q` / `sys.x1.vq.I` is for a circuit that never went through DAECompiler: `HipSystem()` has every property."
struct HipRef
    path::Tuple{Vararg{Symbol}}
end
struct HipSystem end
Base.getproperty(::HipSystem, s::Symbol) = HipRef((s,))
Base.getproperty(r::HipRef, s::Symbol) = s === :path ? getfield(r, :path) : HipRef((getfield(r, :path)..., s))
# A DAECompiler ScopeRef (what `sys.node_q` of a real IRODESystem returns) carries a Scope built by `DScope(parent, name)`:
# default-constructor field order (parent, name), the root `DScope()` has no name.  Walk it positionally; no field NAMES assumed.
function scope_path(sc::AbstractScope)
    sc isa PathScope && return sc.path
    nfields(sc) < 2 && return ()
    parent, name = getfield(sc, 1), getfield(sc, 2)
    (parent isa AbstractScope ? scope_path(parent) : ())..., Symbol(name)
end
function ref_path(ref)
    ref isa HipRef && return getfield(ref, :path)
    ref isa Symbol && return (ref,)
    ref isa AbstractString && return Tuple(Symbol.(split(ref, '.')))
    ref isa AbstractScope && return scope_path(ref)
    for i in 1:nfields(ref)                                   # ScopeRef(sys, scope): take the field that is a scope
        f = getfield(ref, i)
        f isa AbstractScope && return scope_path(f)
    end
    throw(KeyError(ref))
end

# ---- what the overlay records ------------------------------------------------------------------------------------------
struct FakeNet <: AbstractNet     # stands in for Net (src/simulate_ir.jl:28-54): an id instead of a DAE variable
    id::Int32
    name::Union{PathScope,Nothing}
    multiplier::Float64
end
struct WaveRef                    # a source waveform captured instead of evaluated (spectre_env.jl:144-176); deliberately NOT <: Real
    id::Int32
end
struct Wave
    kind::Int32; par::NTuple{8,Float64}; ts::Vector{Float64}; ys::Vector{Float64}
end
"What the intercepted constructors of VoltageSource / CurrentSource return: any mix of numbers and captured waveforms."
struct CapturedSource
    kind::Int32                   # CH_DEV.V or CH_DEV.I
    dc::Any                       # Float64, WaveRef (the :dcop value is then the waveform at $time = 0, spectre_env.jl:190-196) or nothing
    tran::Any                     # Float64 or WaveRef
    ac::ComplexF64
end
mutable struct StampTable
    net_ids::Dict{Any,Int32}; net_names::Vector{Any}
    aliases::Vector{Pair{Tuple{Vararg{Symbol}},Int32}}      # net_alias (src/spectre.jl:903-905): extra names of subcircuit ports
    kind::Vector{Int32}; node::Vector{NTuple{NNODE,Int32}}; ipar::Vector{NTuple{NIPAR,Int32}}; par::Vector{NTuple{NPAR,Float64}}
    mult::Vector{Float64}; scope::Vector{Tuple{Vararg{Symbol}}}
    waves::Vector{Wave}                       # waveform table (WaveRef.id indexes it)
    src_dc::Vector{Float64}; src_wave::Vector{Int32}; src_ac::Vector{Float64}   # one entry per V / I device
    models::Vector{Vector{Float64}}; model_keys::Vector{Any}                     # BSIM4 cards [CH_B4_NPAR], NaN = not given
    va_par::Vector{Float64}
    spec::SimSpec                             # the SimSpec in force where the devices are instantiated (.option gmin/temp/scale)
end
StampTable() = StampTable(Dict{Any,Int32}(), Any[], Pair{Tuple{Vararg{Symbol}},Int32}[], Int32[], NTuple{NNODE,Int32}[], NTuple{NIPAR,Int32}[],
                          NTuple{NPAR,Float64}[], Float64[], Tuple{Vararg{Symbol}}[], Wave[], Float64[], Int32[], Float64[], Vector{Float64}[], Any[],
                          Float64[], SimSpec())

function intern!(tbl::StampTable, name)
    key = name === nothing ? gensym(:net) : name.path
    get!(tbl.net_ids, key) do
        push!(tbl.net_names, key)
        Int32(length(tbl.net_names))          # node ids start at 1; nets tied by Gnd() are renumbered to 0 in make_desc
    end
end
nodes8(nets) = ntuple(k -> k <= length(nets) ? nets[k].id : Int32(0), NNODE)
par8(vals) = ntuple(k -> k <= length(vals) ? Float64(vals[k]) : NaN, NPAR)
scope_of(dscope) = dscope isa AbstractScope ? scope_path(dscope) : ()
function record!(tbl::StampTable, kind, nets, pars, dscope; ipar = (Int32(0), Int32(0)))
    push!(tbl.kind, kind); push!(tbl.node, nodes8(nets)); push!(tbl.ipar, ipar); push!(tbl.par, par8(pars))
    # ParallelInstances multiplies the nets' multipliers (simulate_ir.jl:56-75): a device's own m is that of its first net
    push!(tbl.mult, isempty(nets) ? 1.0 : nets[1].multiplier); push!(tbl.scope, scope_of(dscope))
    tbl.spec = spec[]                         # `.option` lines wrap the instances in `with(spec => …)` (src/spectre.jl:1529-1544, :1698-1700)
    length(tbl.kind)
end

"Value of a captured waveform at `\$time` = 0 — what `dc = something(dc, tran, 0.0)` (simpledevices.jl:280) evaluates to in :dcop mode."
function wave_at0(w::Wave)
    if w.kind == CH_SRC.PWL
        pwl_at_time(w.ts, w.ys, 0.0)                                         # spectre_env.jl:43-69
    elseif w.kind == CH_SRC.PULSE
        v1, v2, td, tr, tf, pw, per = w.par[1:7]
        pwl_at_time([td, td + tr, td + tr + pw, td + tr + pw + tf], [v1, v2, v2, v1], isfinite(per) ? rem(0.0, per) : 0.0)   # :153-166
    elseif w.kind == CH_SRC.SIN
        vo, va, freq, td, theta, phase, nc = w.par[1:7]
        (td < 0.0 < nc / freq) ? vo + va * exp(td * theta) * sind(-360 * freq * td + phase) : vo + va * sind(phase)         # :169-176
    else
        w.par[1]
    end
end
function source!(tbl::StampTable, s::CapturedSource)
    tran = s.tran
    if tran isa WaveRef
        push!(tbl.src_wave, tran.id)
    else                                      # a plain number: constant waveform
        push!(tbl.waves, Wave(CH_SRC.DC, par8((Float64(tran),)), Float64[], Float64[])); push!(tbl.src_wave, Int32(length(tbl.waves)))
    end
    push!(tbl.src_dc, s.dc isa WaveRef ? Float64(wave_at0(tbl.waves[s.dc.id])) : Float64(s.dc))
    push!(tbl.src_ac, abs(s.ac))
    Int32(length(tbl.src_dc) - 1)             # 0-based source index for dev_ipar[0]
end

# ---- the overlay pass (same shape as AliasInterp, src/aliasextract.jl:10-39; call site src/simulate_ir.jl:85-87) ---------
struct StampPass <: CassetteOverlay.AbstractBindingOverlay{nothing,nothing}
    tbl::StampTable
    ground::Vector{Int32}
end
StampPass() = StampPass(StampTable(), Int32[])
const KW = typeof(Core.kwcall)

# nets, aliases, scoped values, scopes
(self::StampPass)(::Type{Net}, name::Union{AbstractScope,Nothing} = nothing, multiplier::Float64 = 1.0) =
    FakeNet(intern!(self.tbl, name === nothing ? nothing : PathScope(scope_path(name))), name === nothing ? nothing : PathScope(scope_path(name)), multiplier)
(self::StampPass)(::Type{Net}, net::FakeNet, multiplier::Float64) = FakeNet(net.id, net.name, net.multiplier * multiplier)   # ParallelInstances, simulate_ir.jl:70-73
# (Net(::Symbol) and Net(::String), simulate_ir.jl:51-52, are overlaid code: they end in the first method with a scope)
(self::StampPass)(::typeof(net_alias), net::FakeNet, name) = (push!(self.tbl.aliases, (scope_path(debug_scope[])..., Symbol(name)) => net.id); nothing)
(self::StampPass)(::typeof(with), f, pairs...) = with(pairs...) do; self(f); end          # aliasextract.jl:30-34
(self::StampPass)(::Type{DScope}) = PathScope(())
(self::StampPass)(::Type{DScope}, parent::AbstractScope, name) = PathScope((scope_path(parent)..., Symbol(name)))
(self::StampPass)(::Type{GenScope}, parent::AbstractScope, name) = PathScope((scope_path(parent)..., Symbol(name)))
# kcl!/branch!/equation!/variable never run: every functor below returns before reaching them.

# ---- source CONSTRUCTORS (simpledevices.jl:279-284, 319-324): `spicecall(vsource; dc=…, tran=pwl(…))` reaches
#      `model(; kwargs...)` = `Core.kwcall(kwargs, VoltageSource)` (src/spectre.jl:1179-1181) ----
function captured(kind, kw::NamedTuple)
    dc, tran, ac = get(kw, :dc, nothing), get(kw, :tran, nothing), get(kw, :ac, 0.0 + 0.0im)
    if kind == CH_DEV.V
        dc = something(dc, tran, 0.0)                     # :280
    else
        dc === nothing && (dc = tran)                     # :320  something(dc, Some(tran))
    end
    tran = something(tran, dc)                            # :281 / :321
    num(x) = x isa WaveRef ? x : Float64(undefault_any(x))
    CapturedSource(kind, num(dc), num(tran), ComplexF64(ac))
end
undefault_any(x) = x isa CedarSim.DefaultOr ? undefault(x) : x
(self::StampPass)(::KW, kw::NamedTuple, ::Type{VoltageSource}) = captured(CH_DEV.V, kw)
(self::StampPass)(::KW, kw::NamedTuple, ::Type{CurrentSource}) = captured(CH_DEV.I, kw)
(self::StampPass)(::Type{VoltageSource}) = captured(CH_DEV.V, (;))
(self::StampPass)(::Type{CurrentSource}) = captured(CH_DEV.I, (;))
(self::StampPass)(::Type{VoltageSource}, dc, tran, ac) = CapturedSource(CH_DEV.V, dc, tran, ComplexF64(ac))   # the positional form (:275-278)
(self::StampPass)(::Type{CurrentSource}, dc, tran, ac) = CapturedSource(CH_DEV.I, dc, tran, ComplexF64(ac))

# ---- device functors: `Named(dev, name)(nets...)` calls `dev(nets...; dscope=DScope(debug_scope[], name))` (simulate_ir.jl:84-86),
#      i.e. Core.kwcall((dscope=…,), dev, nets...); a direct call `dev(A, B)` is the positional form ----
dscope_of(kw::NamedTuple, default) = get(kw, :dscope, default)
here(sym) = PathScope((scope_path(debug_scope[])..., sym))     # stands in for defaultscope(dev) = GenScope(debug_scope[], :R) (simpledevices.jl:62-63)

stamp!(self, R::SimpleResistor, nets, dscope) = begin                                                    # simpledevices.jl:65-77
    res = isdefault(R.r) ? R.rsh * (R.l - R.short) / (R.w - R.narrow) : undefault(R.r)
    record!(self.tbl, CH_DEV.R, nets, (res,), dscope); nothing
end
stamp!(self, C::SimpleCapacitor, nets, dscope) = (record!(self.tbl, CH_DEV.C, nets, (C.capacitance,), dscope); nothing)   # :105-109
stamp!(self, L::SimpleInductor, nets, dscope) = (record!(self.tbl, CH_DEV.L, nets, (L.inductance,), dscope); nothing)     # :128-132
stamp!(self, S::CapturedSource, nets, dscope) = begin                                                    # :288-300, :327-339
    s = source!(self.tbl, S)
    record!(self.tbl, S.kind, nets, (), dscope; ipar = (s, Int32(0))); nothing
end
stamp!(self, S::vcvs, nets, dscope) = begin                                                              # :347-356
    if length(nets) == 2                                                                                 # a constant voltage
        s = source!(self.tbl, CapturedSource(CH_DEV.V, S.voltage, S.voltage, 0.0im))
        record!(self.tbl, CH_DEV.V, nets, (), dscope; ipar = (s, Int32(0)))
    else
        record!(self.tbl, CH_DEV.VCVS, nets, (S.gain,), dscope)
    end
    nothing
end
stamp!(self, S::vccs, nets, dscope) = begin                                                              # :364-373
    if length(nets) == 2                                                                                 # a constant current
        s = source!(self.tbl, CapturedSource(CH_DEV.I, S.current, S.current, 0.0im))
        record!(self.tbl, CH_DEV.I, nets, (), dscope; ipar = (s, Int32(0)))
    else
        record!(self.tbl, CH_DEV.VCCS, nets, (S.gain,), dscope)
    end
    nothing
end
const LinearDev = Union{SimpleResistor,SimpleCapacitor,SimpleInductor,CapturedSource,vcvs,vccs}
dev_symbol(d) = d isa SimpleResistor ? :R : d isa SimpleCapacitor ? :C : d isa SimpleInductor ? :L : d isa CapturedSource ? (d.kind == CH_DEV.V ? :V : :I) :
                d isa vcvs ? :E : d isa vccs ? :G : :X
(self::StampPass)(dev::LinearDev, nets::FakeNet...) = stamp!(self, dev, nets, here(dev_symbol(dev)))
(self::StampPass)(::KW, kw::NamedTuple, dev::LinearDev, nets::FakeNet...) = stamp!(self, dev, nets, dscope_of(kw, here(dev_symbol(dev))))
(self::StampPass)(::Gnd, A::FakeNet) = (push!(self.ground, A.id); nothing)                               # :305-313
(self::StampPass)(::KW, kw::NamedTuple, ::Gnd, A::FakeNet) = (push!(self.ground, A.id); nothing)

# waveforms: captured, not evaluated.  `pwl(@SVector [t1, y1, t2, y2, …])` (src/spectre.jl:1036-1040)
(self::StampPass)(::typeof(SpectreEnvironment.pwl), wave) = begin
    ts, ys = Float64.(collect(wave[1:2:end])), Float64.(collect(wave[2:2:end]))
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
function stamp!(self, dev::VAModel, nets, dscope)
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
        m = findfirst(k -> k[1] === T && isequal(k[2], card), self.tbl.model_keys)      # isequal: NaN entries compare equal
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
(self::StampPass)(dev::VAModel, nets::FakeNet...) = stamp!(self, dev, nets, here(nameof(typeof(dev))))
(self::StampPass)(::KW, kw::NamedTuple, dev::VAModel, nets::FakeNet...) = stamp!(self, dev, nets, dscope_of(kw, here(nameof(typeof(dev)))))

"Run the circuit closure of `sim` once under the overlay; returns the filled pass (table + ground nets).  Mirrors the two call
operators of src/circuitodesystem.jl:20-25 (DefaultSim: `circuit()`) and :92-97 (ParamSim: `circuit(ParamLens(params))` under
`SimSpec(; sim.spec...)`)."
function stamp_extract(sim::AbstractSim)
    pass = StampPass()
    b4_names()
    circuit = getfield(sim, :circuit)
    if sim isa ParamSim
        sp = SimSpec(; time = 0.0, getfield(sim, :spec)...)
        lens = ParamLens(getfield(sim, :params))
        with(spec => sp, sim_mode => getfield(sim, :mode), debug_scope => PathScope(())) do
            pass.tbl.spec = sp
            pass(circuit, lens)
        end
    else
        with(spec => SimSpec(time = 0.0), sim_mode => :tran, debug_scope => PathScope(())) do
            pass.tbl.spec = spec[]
            pass(circuit)
        end
    end
    pass
end
stamp_extract(circuit) = stamp_extract(DefaultSim(circuit))      # dc!(circ) / tran!(circ, tspan) wrap the same way (src/sweeps.jl:437-442, 450-455)

is_branch_kind(k) = k == CH_DEV.V || k == CH_DEV.L || k == CH_DEV.VCVS

"Flat arrays of the description; `keep` owns them for the lifetime of the ccall.  `names` maps symbol paths to observable rows:
every net (`(:node_q,)`, `(:x1, :node_a)` through net_alias), and `(scope..., :I)` for the branch current of every V / L / VCVS
(`sys.vq.I`: branch!() names the current `scope(:I)`, src/simulate_ir.jl:112-120)."
function make_desc(pass::StampPass; slots = NTuple{3,Int32}[])
    tbl = pass.tbl
    sp = tbl.spec
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
    # observables: every node voltage, then the branch current of every branch device
    branch_devs = [d for d in 1:nd if is_branch_kind(tbl.kind[d])]
    obs_kind = vcat(zeros(Int32, Int(k)), ones(Int32, length(branch_devs)))
    obs_index = vcat(Int32.(1:k), Int32.(branch_devs .- 1))                           # node ids are 1-based, device indices 0-based (cedarhip.h)
    slot_kind = Int32[s[1] for s in slots]; slot_a = Int32[s[2] for s in slots]; slot_b = Int32[s[3] for s in slots]
    keep = (tbl.kind, dev_node, dev_ipar, dev_par, tbl.mult, src_kind, tbl.src_dc, src_par, pwl_ofs, pwl_t, pwl_y, model_par,
            slot_kind, slot_a, slot_b, obs_kind, obs_index, tbl.src_ac, tbl.va_par)
    desc = ChDesc(k, nd, pointer(tbl.kind), pointer(dev_node), pointer(dev_ipar), pointer(dev_par), pointer(tbl.mult),
                  ns, pointer(src_kind), pointer(tbl.src_dc), pointer(src_par), pointer(pwl_ofs), pointer(pwl_t), pointer(pwl_y),
                  length(tbl.models), pointer(model_par), undefault(sp.temp), undefault(sp.gmin), undefault(sp.scale),
                  length(slots), pointer(slot_kind), pointer(slot_a), pointer(slot_b), length(obs_kind), pointer(obs_kind), pointer(obs_index),
                  pointer(tbl.src_ac), length(tbl.va_par), isempty(tbl.va_par) ? Ptr{Float64}(C_NULL) : pointer(tbl.va_par))
    names = Dict{Tuple{Vararg{Symbol}},Int}()
    for i in 1:nn; (remap[i] > 0 && tbl.net_names[i] isa Tuple) && (names[tbl.net_names[i]] = Int(remap[i])); end
    for (path, id) in tbl.aliases; remap[id] > 0 && (names[path] = Int(remap[id])); end
    for (r, d) in enumerate(branch_devs); names[(tbl.scope[d]..., :I)] = Int(k) + r; end
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
    pts::Vector{Int32}             # ch_result_dense_points: rows the dense-output polynomial of the step ending at row i runs through
    names::Dict{Tuple{Vararg{Symbol}},Int}   # symbol path -> observable row
    retcode::ReturnCode.T
    stats::ChStats
    sample::Int                    # which sample `sol[ref]` / `sol(t)` read (one per sweep point)
end
obs_row(sol::CedarHIPSolution, ref) = (p = ref_path(ref); haskey(sol.names, p) ? sol.names[p] : throw(KeyError(ref)))
Base.getindex(sol::CedarHIPSolution, ref) = sol.u[sol.sample, :, obs_row(sol, ref)]
"`sol(t; idxs = [sys.node_q])` (test/gf180_dff.jl:29-33): the engine's own dense output.  For t in (t[i-1], t[i]] the BDF step
that ended at row i defines a polynomial through its `pts[i]` newest rows — the value a `saveat` grid would have returned at t
(include/cedarhip.h, ch_result_dense_points).  Rows without one (results on a `saveat` grid) are joined linearly."
function (sol::CedarHIPSolution)(t::Real; idxs = nothing)
    rows = idxs === nothing ? collect(1:size(sol.u, 3)) : [obs_row(sol, r) for r in idxs]
    tt, s = sol.t, sol.sample
    t <= tt[1] && return [sol.u[s, 1, r] for r in rows]
    t >= tt[end] && return [sol.u[s, end, r] for r in rows]
    i = searchsortedfirst(tt, t)                                    # tt[i-1] < t <= tt[i]
    m = min(Int(sol.pts[i]), i)
    win = (i - m + 1):i
    if m < 2 || !allunique(tt[win])
        θ = tt[i] > tt[i - 1] ? (t - tt[i - 1]) / (tt[i] - tt[i - 1]) : 1.0
        return [(1 - θ) * sol.u[s, i - 1, r] + θ * sol.u[s, i, r] for r in rows]
    end
    w = [prod((t - tt[b]) / (tt[a] - tt[b]) for b in win if b != a; init = 1.0) for a in win]   # Lagrange weights
    [sum(w[q] * sol.u[s, win[q], r] for q in 1:m) for r in rows]
end
(sol::CedarHIPSolution)(t::Real, ::Type{Val{0}}; idxs = nothing, kwargs...) = sol(t; idxs)        # SciML's positional derivative-order form

# ---- solve ---------------------------------------------------------------------------------------------------------------
struct CedarHIPAlg <: SciMLBase.AbstractDAEAlgorithm end

dcopts(alg; tran_mode = false) = ChDcOpts(alg.abstol, Int32(200), Int32(10), UInt64(10), Int32(tran_mode), 2.0, Ptr{Float64}(C_NULL))   # src/dcop.jl:28,53

function build(sim; slots = NTuple{3,Int32}[], pass = stamp_extract(sim))
    desc, keep, names = make_desc(pass; slots)
    circ = GC.@preserve keep check_build(ccall((:ch_circuit_build, lib), Ptr{Cvoid}, (Ptr{Cvoid}, Ref{ChDesc}), context(), desc))
    circ, Int(desc.n_obs), names
end

# shared_steps = true: CH_STEPS_SHARED — one step sequence and error norm over the whole system even on a saveat grid, i.e. what the
# ONE IDA() integrator of `tran!` (src/sweeps.jl:456) does; the default lets independent blocks / samples take their own steps there
function run_tran(circ, n_obs, n_samples, tspan, abstol, reltol, initializealg, saveat; shared_steps::Bool = false)
    sv = Float64.(collect(saveat))
    opts = ChTranOpts(abstol, reltol, Int32(5), 0.0, 0.0, 0.0, Int32(0), Int32(10), Int32(length(sv)), isempty(sv) ? Ptr{Float64}(C_NULL) : pointer(sv),
                      dcopts(initializealg; tran_mode = initializealg isa CedarTranOp), Int32(0), Int32(0), Int32(shared_steps ? 1 : 0))
    res = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve sv ccall((:ch_tran, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Ref{ChTranOpts}, Ref{Ptr{Cvoid}}), circ, tspan[1], tspan[2], opts, res)
    res[] == C_NULL && throw(CedarSim.CedarError(last_error()))     # an exception barrier answered (CH_ERR_NOMEM / CH_ERR_INTERNAL)
    nt = ccall((:ch_result_n_times, lib), Int64, (Ptr{Cvoid},), res[])
    t = copy(unsafe_wrap(Array, ccall((:ch_result_times, lib), Ptr{Float64}, (Ptr{Cvoid},), res[]), nt))
    u = nt > 0 && n_obs > 0 ? copy(unsafe_wrap(Array, ccall((:ch_result_values, lib), Ptr{Float64}, (Ptr{Cvoid},), res[]), (n_samples, nt, n_obs))) :
                              zeros(n_samples, nt, n_obs)
    pp = ccall((:ch_result_dense_points, lib), Ptr{Int32}, (Ptr{Cvoid},), res[])
    pts = (pp == C_NULL || nt == 0) ? zeros(Int32, nt) : copy(unsafe_wrap(Array, pp, nt))
    st = Ref(ChStats())
    ccall((:ch_result_stats, lib), Cint, (Ptr{Cvoid}, Ref{ChStats}), res[], st)
    ccall((:ch_result_free, lib), Cvoid, (Ptr{Cvoid},), res[])
    rc, t, u, pts, st[]
end

# shared_steps = true asks for what one IDA() over the whole system does on a saveat grid too (one step sequence, one error norm);
# the default lets structurally independent blocks of the circuit step on their own there (a stricter, per-block norm)
function SciMLBase.__solve(prob::DAEProblem, ::CedarHIPAlg; abstol = 1e-6, reltol = 1e-3, initializealg = CedarDCOp(), saveat = Float64[], shared_steps = false, kwargs...)
    circ, n_obs, names = build(prob.p)
    try
        rc, t, u, pts, st = run_tran(circ, n_obs, 1, prob.tspan, abstol, reltol, initializealg, saveat; shared_steps)
        CedarHIPSolution(prob, t, u, pts, names, retcode(rc), st, 1)
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), circ)
    end
end
"tran!(circ, tspan, CedarHIPAlg(); abstol, reltol): the reference's `tran!(circ, tspan)` (src/sweeps.jl:450-456) without DAECompiler —
the circuit closure is stamped and solved; index the result with `HipSystem()`: `sol[HipSystem().node_q]`."
function tran!(circ, tspan, ::CedarHIPAlg; abstol = 1e-6, reltol = 1e-3, initializealg = CedarDCOp(), saveat = Float64[], shared_steps = false)
    sim = circ isa AbstractSim ? circ : DefaultSim(circ)
    c, n_obs, names = build(sim)
    try
        rc, t, u, pts, st = run_tran(c, n_obs, 1, tspan, abstol, reltol, initializealg, saveat; shared_steps)
        CedarHIPSolution(nothing, t, u, pts, names, retcode(rc), st, 1)
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), c)
    end
end
tran!(prob::DAEProblem, alg::CedarHIPAlg; kwargs...) = solve(prob, alg; kwargs...)     # tran!(prob) = solve(prob, IDA()) in the reference (:457)

function run_dc(circ, S, abstol)
    info = Ref(ChInfo(ntuple(_ -> Int32(0), 12)..., 0, 0, Int32(0)))
    ccall((:ch_circuit_info, lib), Cint, (Ptr{Cvoid}, Ref{ChInfo}), circ, info)
    nm = max(1, Int(info[].n_mna)); nn = Int(info[].n_nodes)
    x = zeros(Float64, nm, S); status = zeros(Int32, S); st = Ref(ChStats())       # x_out is [n_samples][n_mna]: column s of a Julia matrix
    rc = ccall((:ch_dc, lib), Cint, (Ptr{Cvoid}, Ref{ChDcOpts}, Ptr{Float64}, Ptr{Int32}, Ref{ChStats}), circ, dcopts(CedarDCOp(; abstol)), x, status, st)
    rc == 0 || @warn "DC operating point analysis failed. Further failures may follow." rc last_error()          # src/dcop.jl:141-145
    rc, x, status, st[], nn, info[]
end
"Branch-current rows of x_mna: the k-th branch device (V, L, E in device order) sits at n_nodes + k (cedarhip.h conventions)."
function dc_solution(pass, names, x, s, nn, rc, st)
    tbl = pass.tbl
    kb = 0; u = zeros(1, 1, maximum(values(names); init = 0))
    for (path, row) in names
        row <= nn && (u[1, 1, row] = x[row, s])
    end
    for d in 1:length(tbl.kind)
        is_branch_kind(tbl.kind[d]) || continue
        kb += 1
        row = get(names, (tbl.scope[d]..., :I), 0)
        row > 0 && (u[1, 1, row] = x[nn + kb, s])
    end
    CedarHIPSolution(nothing, [0.0], u, Int32[0], names, retcode(rc), st, 1)
end
dc!(sys::IRODESystem, alg::CedarHIPAlg; kwargs...) = dc!(DAECompiler.arg1_from_sys(sys), alg; kwargs...)   # dc!(sys) form, src/sweeps.jl:443-446
"dc!(circ, CedarHIPAlg()) — dc!(circ) of src/sweeps.jl:437-447: DC operating point, indexable like a solution."
function dc!(circ, ::CedarHIPAlg; abstol = 1e-10)
    sim = circ isa AbstractSim ? circ : DefaultSim(circ)
    pass = stamp_extract(sim)
    c, n_obs, names = build(sim; pass)
    try
        rc, x, status, st, nn, _ = run_dc(c, 1, abstol)
        dc_solution(pass, names, x, 1, nn, rc, st)
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), c)
    end
end

# ---- sweeps: every point a sample of ONE batched solve (replaces the remake loop of src/sweeps.jl:471-502) -----------------
# The swept names reach the circuit through ParamSim's lens (src/circuitodesystem.jl:66-97; `find_param_ranges`, src/sweeps.jl:507-546,
# lists them).  Which table entries a name moves is LEARNED, not assumed: the stamp table of the base point is diffed against the
# tables of a few single-variable variations — one per distinct value when a variable has at most four, otherwise two that fix an
# identity / proportional / affine map plus a third that checks it — and the assembled per-point table is validated against full
# extractions of the far corner, of seeded random points and of the points that hold the extremes of every fitted variable (an entry
# that answers to two variables, or is clipped beyond the fitted values, is invisible from single-axis variations).  Any failed check falls back to one extraction per point.  Same algorithm as cedarsim.jl_amd/api.py CircuitSweep._batch,
# which the CPU tests cover (tests/test_netlist_and_sweeps.py).
"Every sweepable entry of a stamp table as one vector, with the engine slot (kind, a, b) of each position (cedarhip.h CH_SLOT_*)."
function flat_table(pass::StampPass)
    tbl = pass.tbl
    vals = Float64[]; keys = NTuple{3,Int32}[]
    for d in 1:length(tbl.kind), j in 1:NPAR; push!(vals, tbl.par[d][j]); push!(keys, (CH_SLOT.DEV_PAR, Int32(d - 1), Int32(j - 1))); end
    for d in 1:length(tbl.kind); push!(vals, tbl.mult[d]); push!(keys, (CH_SLOT.DEV_MULT, Int32(d - 1), Int32(0))); end
    for s in 1:length(tbl.src_dc)
        push!(vals, tbl.src_dc[s]); push!(keys, (CH_SLOT.SRC_DC, Int32(s - 1), Int32(0)))
        w = tbl.waves[tbl.src_wave[s]]
        for j in 1:SRC_NPAR; push!(vals, w.par[j]); push!(keys, (CH_SLOT.SRC_PAR, Int32(s - 1), Int32(j - 1))); end
    end
    for m in 1:length(tbl.models), j in 1:length(tbl.models[m]); push!(vals, tbl.models[m][j]); push!(keys, (CH_SLOT.MODEL_PAR, Int32(m - 1), Int32(j - 1))); end
    for i in 1:length(tbl.va_par); push!(vals, tbl.va_par[i]); push!(keys, (CH_SLOT.VA_PAR, Int32(i - 1), Int32(0))); end
    push!(vals, undefault(tbl.spec.temp)); push!(keys, (CH_SLOT.TEMP, Int32(0), Int32(0)))
    push!(vals, undefault(tbl.spec.gmin)); push!(keys, (CH_SLOT.GMIN, Int32(0), Int32(0)))
    vals, keys
end
topology(pass::StampPass) = (pass.tbl.kind, pass.tbl.node, pass.tbl.ipar, length(pass.tbl.models), length(pass.tbl.va_par))
samev(a, b) = (a == b) | (isnan(a) & isnan(b))
closev(a, b) = samev(a, b) || abs(a - b) <= 1e-13 * max(abs(a), abs(b))

"Per-point flat tables [n_points, n_entries] for the sweep, the base pass and the slot keys; `builds` counts stamp extractions."
function sweep_table(circuit, iterator)
    # a sweep yields ((selector, value), …) per point (src/sweeps.jl:226-233, 261-338); SerialSweep pads with `nothing`
    points = [Dict{Symbol,Any}(Symbol(k) => v for (k, v) in pt if v !== nothing) for pt in iterator]
    sim_of(pt) = ParamSim(circuit; pt...)                                                   # _iterate_alter, src/sweeps.jl:423-434
    base = stamp_extract(sim_of(points[1]))
    v0, skeys = flat_table(base)
    builds = Ref(1)
    function flat_of(pt)
        p = stamp_extract(sim_of(pt)); builds[] += 1
        topology(p) == topology(base) || throw(CedarSim.CedarError("sweep points must not change the circuit topology"))
        first(flat_table(p))
    end
    names = sort(collect(union((Set(keys(p)) for p in points)...)))
    n = length(points)
    table = nothing
    if n > 1 && all(p -> Set(keys(p)) == Set(names), points)
        distinct = Dict(k => unique(p[k] for p in points) for k in names)
        ncheck = min(4, n - 1)
        fitted = [k for k in names if length(distinct[k]) > 4 && all(v -> v isa Real, distinct[k])]    # maps that are fitted, not looked up
        cost = 1 + sum(min(length(distinct[k]) - 1, 2) for k in names) + ncheck + 2 * length(fitted)
        cost < n && (table = learn_table(points, names, distinct, v0, flat_of))
        if table !== nothing
            away(r) = count(k -> points[r][k] != points[1][k], names)
            far = argmax([away(r) for r in 1:n])
            picks = vcat(far, n, [2 + (7919 * q) % (n - 1) for q in 1:max(0, ncheck - 2)])             # far corner, last point, seeded others
            # for every fitted variable the points that hold its smallest and largest value, with the most OTHER variables away from the
            # base point: a clipped entry (affine on the three fitted values, wrong beyond the kink) or one gated by another variable
            for k in fitted, ext in (minimum(distinct[k]), maximum(distinct[k]))
                cand = [r for r in 1:n if points[r][k] == ext]
                push!(picks, cand[argmax([(away(r), r) for r in cand])])
            end
            for r in unique(picks)
                r == 1 && continue
                all(closev.(flat_of(points[r]), table[r, :])) || (table = nothing; break)   # e.g. an entry that depends on two swept variables
            end
        end
    end
    if table === nothing
        table = Matrix{Float64}(undef, n, length(v0)); table[1, :] = v0
        for r in 2:n; table[r, :] = flat_of(points[r]); end
    end
    table, base, skeys, builds[]
end
function learn_table(points, names, distinct, v0, flat_of)
    n = length(points)
    owner = zeros(Int, length(v0))
    table = repeat(reshape(v0, 1, :), n, 1)
    for (ki, k) in enumerate(names)
        x0 = points[1][k]
        others = [v for v in distinct[k] if v != x0]
        isempty(others) && continue
        cols = Dict{Any,Vector{Float64}}(x0 => v0)
        function lookup()
            moved = falses(length(v0))
            for val in others
                haskey(cols, val) || (cols[val] = flat_of(merge(points[1], Dict(k => val))))
                moved .|= .!samev.(cols[val], v0)
            end
            any(moved .& (owner .> 0) .& (owner .!= ki)) && return false
            owner[moved] .= ki
            for r in 1:n; table[r, moved] = cols[points[r][k]][moved]; end
            true
        end
        if length(others) <= 3 || !all(v -> v isa Real, distinct[k])
            lookup() || return nothing
            continue
        end
        x1 = others[argmax([abs(v - x0) for v in others])]
        rest = [v for v in others if v != x1]
        x2 = rest[argmin([abs(v - 0.5 * (x0 + x1)) for v in rest])]
        v1 = flat_of(merge(points[1], Dict(k => x1))); v2 = flat_of(merge(points[1], Dict(k => x2)))
        cols[x1] = v1; cols[x2] = v2
        moved = .!samev.(v1, v0) .| .!samev.(v2, v0)
        any(moved .& (owner .> 0)) && return nothing
        idx = findall(moved)
        isempty(idx) && continue
        ok = true
        col = Matrix{Float64}(undef, n, length(idx))
        xs = Float64[points[r][k] for r in 1:n]
        for (q, e) in enumerate(idx)
            a0, a1, a2 = v0[e], v1[e], v2[e]
            if a0 == x0 && a1 == x1 && a2 == x2                         # identity: the entry IS the swept value
                col[:, q] = xs
            elseif x1 != 0 && (a1 / x1) * x0 == a0 && (a1 / x1) * x2 == a2   # proportional: the builder's own product, bit for bit
                col[:, q] = (a1 / x1) .* xs
            else
                slope = (a1 - a0) / (x1 - x0)
                closev(a0 + slope * (x2 - x0), a2) || (ok = false; break)   # not affine in the swept variable
                col[:, q] = a0 .+ slope .* (xs .- x0)
            end
        end
        if !ok
            (2 * length(distinct[k]) < n && lookup()) || return nothing
            continue
        end
        owner[idx] .= ki
        table[:, idx] = col
    end
    table
end

"Circuit with every sweep point as a sample: build once, ch_set_samples, ONE ch_set_params.  Returns (handle, pass, n_obs, names, S, builds)."
function build_sweep(circuit, iterator)
    table, base, skeys, builds = sweep_table(circuit, iterator)
    S = size(table, 1)
    changed = [e for e in 1:size(table, 2) if any(r -> !samev(table[r, e], table[1, e]), 2:S)]
    # a constant source whose dc is swept: SRC_DC already updates the transient value (cedarhip.h CH_SLOT_SRC_DC)
    drop(e) = skeys[e][1] == CH_SLOT.SRC_PAR && skeys[e][3] == 0 && base.tbl.waves[base.tbl.src_wave[skeys[e][2] + 1]].kind == CH_SRC.DC &&
              any(f -> skeys[f] == (CH_SLOT.SRC_DC, skeys[e][2], Int32(0)), changed)
    changed = [e for e in changed if !drop(e)]
    slots = NTuple{3,Int32}[skeys[e] for e in changed]
    circ, n_obs, names = build(nothing; slots, pass = base)
    if S > 1 || !isempty(slots)
        ccall((:ch_set_samples, lib), Cint, (Ptr{Cvoid}, Int32), circ, Int32(S)) == 0 || throw(CedarSim.CedarError(last_error()))
        if !isempty(slots)
            ids = Int32.(0:length(slots) - 1)
            vals = Float64[table[s, e] for s in 1:S, e in changed][:]         # slot-major: vals[slot_i * S + s]
            ccall((:ch_set_params, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Ptr{Float64}), circ, Int32(0), Int32(S), Int32(length(slots)), ids, vals) == 0 ||
                throw(CedarSim.CedarError(last_error()))
        end
    end
    circ, base, n_obs, names, S, builds
end

"tran!(cs::CircuitSweep, tspan, CedarHIPAlg(); …): the broadcast `tran!.(cs.sys, tspan, cs)` of src/sweeps.jl:486-502 as ONE batched solve.
Returns an array of solutions shaped like the sweep (`size(cs)`, src/sweeps.jl:414-417); index them with `HipSystem()` or `cs.sys`."
tran!(cs::CircuitSweep, tspan, alg::CedarHIPAlg; kwargs...) = tran_sweep(cs.circuit, cs.iterator, tspan; kwargs...)
"The same without a CircuitSweep object (whose constructor compiles an IRODESystem, src/sweeps.jl:414-417): circuit + sweep iterator."
function tran_sweep(circuit, iterator, tspan; abstol = 1e-6, reltol = 1e-3, initializealg = CedarDCOp(), saveat = Float64[])
    circ, base, n_obs, names, S, _ = build_sweep(circuit, iterator)
    try
        rc, t, u, pts, st = run_tran(circ, n_obs, S, tspan, abstol, reltol, initializealg, saveat)
        reshape([CedarHIPSolution(nothing, t, u, pts, names, retcode(rc), st, s) for s in 1:S], size(iterator))
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), circ)
    end
end
"dc!(cs::CircuitSweep, CedarHIPAlg()): `dc!.(cs.sys, cs)` of src/sweeps.jl:448, 471-484 as ONE batched Newton solve."
dc!(cs::CircuitSweep, alg::CedarHIPAlg; kwargs...) = dc_sweep(cs.circuit, cs.iterator; kwargs...)
function dc_sweep(circuit, iterator; abstol = 1e-10)
    circ, base, n_obs, names, S, _ = build_sweep(circuit, iterator)
    try
        rc, x, status, st, nn, _ = run_dc(circ, S, abstol)
        reshape([dc_solution(base, names, x, s, nn, Int(status[s]), st) for s in 1:S], size(iterator))
    finally
        ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), circ)
    end
end

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

# CedarHIP.jl — Julia-side shim for libcedarhip.so (WRITTEN, NOT RUN: no Julia in this pipeline).
#
# Keeps CedarSim's netlist-compiled circuit closure and the SciMLBase problem surface intact and
# replaces what happens inside `solve`.  See INTEGRATION.md for the mapping of call sites to exports.
module CedarHIP

using CedarSim, SciMLBase
using CedarSim: DefaultSim, ParamSim, SimSpec, ParallelInstances, Named
using CassetteOverlay, Base.Experimental: @MethodTable, @overlay

const lib = joinpath(@__DIR__, "..", "lib", "libcedarhip.so")
const CH_DEV = (R=1, C=2, L=3, V=4, I=5, VCVS=6, VCCS=7, MOS=8, VA=9)   # CH_DEV_NNODE = 8 node slots per device

# ---- struct mirrors (field order == include/cedarhip.h) ----
struct ChDesc
    n_nodes::Int32; n_dev::Int32
    dev_kind::Ptr{Int32}; dev_node::Ptr{Int32}; dev_ipar::Ptr{Int32}; dev_par::Ptr{Float64}; dev_mult::Ptr{Float64}
    n_src::Int32
    src_kind::Ptr{Int32}; src_dc::Ptr{Float64}; src_par::Ptr{Float64}; src_pwl_ofs::Ptr{Int32}; pwl_t::Ptr{Float64}; pwl_y::Ptr{Float64}
    n_model::Int32; model_par::Ptr{Float64}
    temp::Float64; gmin::Float64; scale::Float64
    n_slot::Int32; slot_kind::Ptr{Int32}; slot_a::Ptr{Int32}; slot_b::Ptr{Int32}
    n_obs::Int32; obs_kind::Ptr{Int32}; obs_index::Ptr{Int32}
    src_ac::Ptr{Float64}   # |ac| per source or C_NULL
    n_va_par::Int64; va_par::Ptr{Float64}   # parameter blocks of compiled Verilog-A instances or C_NULL
end
struct ChDcOpts
    abstol::Float64; maxiters::Int32; n_restarts::Int32; seed::UInt64; tran_mode::Int32; dv_max::Float64; x0::Ptr{Float64}
end
struct ChTranOpts
    abstol::Float64; reltol::Float64; max_order::Int32; dtmin::Float64; dtmax::Float64; dt0::Float64
    max_steps::Int32; newton_maxiters::Int32; n_saveat::Int32; saveat::Ptr{Float64}; dc::ChDcOpts; skip_dc::Int32
end

# ---- StampExtract: record (device type, fields, net ids, multiplier, scope) from the closure ----
# Same technique as AliasInterp (src/aliasextract.jl:10-39): re-run the circuit with fake nets.
mutable struct StampTable
    nets::Dict{Symbol,Int32}
    kind::Vector{Int32}; node::Vector{NTuple{8,Int32}}; par::Vector{NTuple{8,Float64}}; mult::Vector{Float64}
    names::Vector{Symbol}
    sources::Vector{Any}; models::Vector{Any}
end
struct FakeNet; id::Int32; multiplier::Float64; end
@MethodTable STAMP_MT
# @overlay STAMP_MT CedarSim.net(name) = FakeNet(intern!(TABLE[], name), 1.0)
# @overlay STAMP_MT (R::CedarSim.SimpleResistor)(A, B; dscope) = record!(TABLE[], CH_DEV.R, (A, B), (resistance(R),), dscope)
# … one overlay per device functor of src/simpledevices.jl and per VA-generated functor (BSIM4 → CH_DEV.MOS,
#   instance fields via modelparams(), src/spectre.jl:290-295) …

function stamp_extract(sim)
    tbl = StampTable(Dict{Symbol,Int32}(), Int32[], NTuple{8,Int32}[], NTuple{8,Float64}[], Float64[], Symbol[], Any[], Any[])
    # with(TABLE => tbl) do; StampPass()(sim.circuit); end
    tbl
end

# ---- context, problem, solve ----
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

struct CedarHIPAlg <: SciMLBase.AbstractDAEAlgorithm end

retcode(rc) = rc == 0 ? ReturnCode.Success : rc == -3 ? ReturnCode.InitialFailure :
              rc == -4 ? ReturnCode.DtLessThanMin : rc == -7 ? ReturnCode.MaxIters : ReturnCode.Failure

function SciMLBase.__solve(prob::DAEProblem, ::CedarHIPAlg; abstol = 1e-6, reltol = 1e-3,
                           initializealg = CedarDCOp(), saveat = Float64[], kwargs...)
    tbl = stamp_extract(prob.p)
    desc, keep = make_desc(tbl, prob.p)          # flat arrays kept alive in `keep`
    circ = GC.@preserve keep ccall((:ch_circuit_build, lib), Ptr{Cvoid}, (Ptr{Cvoid}, Ref{ChDesc}), context(), desc)
    circ == C_NULL && error(unsafe_string(ccall((:ch_last_error, lib), Cstring, (Ptr{Cvoid},), context())))
    dc = ChDcOpts(initializealg.abstol, 200, 10, 10, initializealg isa CedarTranOp, 2.0, C_NULL)
    opts = ChTranOpts(abstol, reltol, 5, 0.0, 0.0, 0.0, 0, 10, length(saveat), pointer(saveat), dc, 0)
    res = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve saveat ccall((:ch_tran, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Ref{ChTranOpts}, Ref{Ptr{Cvoid}}),
                                   circ, prob.tspan[1], prob.tspan[2], opts, res)
    nt = ccall((:ch_result_n_times, lib), Int64, (Ptr{Cvoid},), res[])
    t = copy(unsafe_wrap(Array, ccall((:ch_result_times, lib), Ptr{Float64}, (Ptr{Cvoid},), res[]), nt))
    v = copy(unsafe_wrap(Array, ccall((:ch_result_values, lib), Ptr{Float64}, (Ptr{Cvoid},), res[]), (1, nt, Int(desc.n_obs))))
    ccall((:ch_result_free, lib), Cvoid, (Ptr{Cvoid},), res[])
    ccall((:ch_circuit_free, lib), Cvoid, (Ptr{Cvoid},), circ)
    CedarHIPSolution(prob, t, v, tbl, retcode(rc))   # getindex(sol, sys.node_q) → column by observable name
end

# freqresp(ac, sym, ωs) (src/ac.jl:267-284) on the GPU: DC point + linearisation + the whole sweep in one call.
# Returns the MNA phasors [n_mna, n_freq]; the caller picks the row of `sym` through the observable name map.
function freqresp_hip(circ_handle::Ptr{Cvoid}, n_mna::Integer, ωs::Vector{Float64}; abstol = 1e-10)
    dc = ChDcOpts(abstol, 200, 10, 10, false, 2.0, C_NULL)
    f = ωs ./ 2π
    out = zeros(Float64, 2, n_mna, length(f))
    rc = ccall((:ch_ac, lib), Cint, (Ptr{Cvoid}, Ref{ChDcOpts}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
               circ_handle, dc, length(f), f, out, C_NULL)
    rc == 0 || error(unsafe_string(ccall((:ch_last_error, lib), Cstring, (Ptr{Cvoid},), context())))
    complex.(out[1, :, :], out[2, :, :])
end

# PSD(noise, sym, ωs) (src/ac.jl:286-305): out_kind 0 = node voltage (node id), 1 = branch current (device index)
function psd_hip(circ_handle::Ptr{Cvoid}, out_kind::Integer, out_index::Integer, ωs::Vector{Float64}; abstol = 1e-10)
    dc = ChDcOpts(abstol, 200, 10, 10, false, 2.0, C_NULL)
    f = ωs ./ 2π
    out = zeros(Float64, length(f))
    rc = ccall((:ch_noise, lib), Cint, (Ptr{Cvoid}, Ref{ChDcOpts}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
               circ_handle, dc, out_kind, out_index, length(f), f, out, C_NULL)
    rc == 0 || error(unsafe_string(ccall((:ch_last_error, lib), Cstring, (Ptr{Cvoid},), context())))
    out
end

# dc!/tran! keep their signatures (src/sweeps.jl:437-465)
tran_hip!(prob::DAEProblem; kwargs...) = solve(prob, CedarHIPAlg(); kwargs...)

end # module

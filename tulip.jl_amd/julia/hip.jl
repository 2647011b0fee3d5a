# hip.jl -- `HIPNormalEquations <: AbstractKKTSolver{Float64}`: the MI355X backend behind Tulip's
# own KKT interface.  Goes to `src/KKT/HIP/hip.jl` and is `include`d next to the other backends
# (/root/reference/src/KKT/KKT.jl:126-131).  Mirrors the CHOLMOD normal-equations backend
# (/root/reference/src/KKT/Cholmod/cholmod.jl:18-65, spd.jl:1-70); src/IPM is untouched.
#
# Select it exactly like the CHOLMOD backend (cholmod.jl:34-44):
#     set_parameter(model, "KKT_Backend", Tulip.KKT.TlpHIP.Backend())
#     set_parameter(model, "KKT_System",  Tulip.KKT.K1())
#
# NOTE: never executed in this project (no Julia in the images); reviewed against the interface.
module TlpHIP

using LinearAlgebra
using SparseArrays

using ..KKT: AbstractKKTBackend, AbstractKKTSolver
using ..KKT: AbstractKKTSystem, K1, K2
import ..KKT: setup, update!, solve!, backend, linear_system

# libtlpk.jl is included from src/LinearAlgebra/LinearAlgebra.jl, i.e. INSIDE `module TLPLinearAlgebra`
# (/root/reference/src/Tulip.jl:21 includes that file; `...` is `Tulip` seen from `Tulip.KKT.TlpHIP`)
using ...TLPLinearAlgebra.LibTLPK

"""
    Backend(; device=0, row_block=nothing, streams=0, ngpus=1, devices=nothing, refine=0)

HIP (gfx950) backend.  `row_block` is the optional block-angular structure hook (length m; block id ≥ 0, or
-1 for a linking row).  `ngpus > 1` (block-angular LPs, system `K1`): this ONE Julia process shards the
diagonal blocks over `ngpus` devices of the node (`devices`: HIP ordinals, default `0:ngpus-1`); the
linking-block reductions happen inside the library.  `streams`: concurrent stream groups (0 = auto).
`refine`: iterative-refinement steps per `solve!` (0 = none, as `spd.jl:68`; `K1` on one GPU).
"""
struct Backend <: AbstractKKTBackend
    device::Int
    row_block::Union{Nothing,Vector{Int}}
    streams::Int
    ngpus::Int
    devices::Union{Nothing,Vector{Int32}}
    refine::Int
end
Backend(; device::Int=0, row_block=nothing, streams::Int=0, ngpus::Int=1, devices=nothing, refine::Int=0) =
    Backend(device, row_block, streams, ngpus, devices === nothing ? nothing : Vector{Int32}(devices), refine)

"""
    HIPNormalEquations

KKT solver whose numeric factorisation and solves run on the GPU: the normal equations (`K1`, Cholesky of
`A·D·Aᵀ + Rd`, the counterpart of `Cholmod/spd.jl`) or the augmented system (`K2`, signed Cholesky `L·S·Lᵀ`
of the quasi-definite matrix, the counterpart of `Cholmod/sqd.jl` / `LDLFactorizations/ldlfact.jl`).
Supported arithmetic: `Float64`.
"""
mutable struct HIPNormalEquations <: AbstractKKTSolver{Float64}
    m::Int
    n::Int
    A::SparseMatrixCSC{Float64,Int}   # kept by reference like cholmod.jl:50; never mutated
    handle::Ptr{Cvoid}

    function HIPNormalEquations(m, n, A, handle)
        kkt = new(m, n, A, handle)
        finalizer(k -> (k.handle == C_NULL || (LibTLPK.destroy(k.handle); k.handle = C_NULL)), kkt)
        return kkt
    end
end

backend(::HIPNormalEquations) = LibTLPK.backend_name()        # "HIP (gfx950)"
linear_system(kkt::HIPNormalEquations) = LibTLPK.linear_system(kkt.handle)   # "Normal equations (K1)" | "Augmented system (K2)"

# error-code -> exception mapping (SURVEY.md section 5, "Failure detection")
function _check(rc, handle, what)
    rc == LibTLPK.TLPK_OK && return nothing
    rc == LibTLPK.TLPK_NOT_POSDEF && throw(PosDefException(0))            # spd.jl:47
    rc == LibTLPK.TLPK_BADARG && throw(DimensionMismatch("$what: " * LibTLPK.last_error(handle)))
    (rc == LibTLPK.TLPK_OOM || rc == LibTLPK.TLPK_TOO_LARGE) && throw(OutOfMemoryError())
    error("$what: " * LibTLPK.strerror(rc) * " " * LibTLPK.last_error(handle))
end

# Convert to sparse matrix if other type is used (cholmod.jl:65)
setup(A, system::Union{K1,K2}, backend::Backend) = setup(convert(SparseMatrixCSC{Float64,Int}, A), system, backend)

_system_code(::K1) = LibTLPK.TLPK_SYSTEM_K1
_system_code(::K2) = LibTLPK.TLPK_SYSTEM_K2

function setup(A::SparseMatrixCSC{Float64,Int}, system::Union{K1,K2}, b::Backend)
    m, n = size(A)
    rc, h = LibTLPK.create(m, n, A.colptr, A.rowval, A.nzval; device=b.device, row_block=b.row_block,
                           system=_system_code(system), streams=b.streams, ngpus=b.ngpus, devices=b.devices, refine=b.refine)
    rc == LibTLPK.TLPK_OK || (h == C_NULL || LibTLPK.destroy(h); _check(rc, C_NULL, "KKT.setup"))
    return HIPNormalEquations(m, n, A, h)
end

function update!(kkt::HIPNormalEquations, θ::Vector{Float64}, regP::Vector{Float64}, regD::Vector{Float64})
    m, n = kkt.m, kkt.n
    # Sanity checks, same messages as spd.jl:26-34
    length(θ) == n || throw(DimensionMismatch("length(θ)=$(length(θ)) but KKT solver has n=$n."))
    length(regP) == n || throw(DimensionMismatch("length(regP)=$(length(regP)) but KKT solver has n=$n"))
    length(regD) == m || throw(DimensionMismatch("length(regD)=$(length(regD)) but KKT solver has m=$m"))
    # the library copies θ, regP, regD (spd.jl:36-38) and retains no pointer
    _check(LibTLPK.update(kkt.handle, θ, regP, regD), kkt.handle, "KKT.update!")
    return nothing
end
# generic vectors (views, ranges) are materialised first
update!(kkt::HIPNormalEquations, θ, regP, regD) =
    update!(kkt, Vector{Float64}(θ), Vector{Float64}(regP), Vector{Float64}(regD))

function solve!(dx::Vector{Float64}, dy::Vector{Float64}, kkt::HIPNormalEquations,
                ξp::Vector{Float64}, ξd::Vector{Float64})
    m, n = kkt.m, kkt.n
    length(dx) == n && length(ξd) == n || throw(DimensionMismatch("dx/ξd must have length n=$n"))
    length(dy) == m && length(ξp) == m || throw(DimensionMismatch("dy/ξp must have length m=$m"))
    # ξp may alias problem data (HSD/step.jl:63 passes dat.b): inputs are const in the C ABI
    _check(LibTLPK.solve(kkt.handle, dx, dy, ξp, ξd), kkt.handle, "KKT.solve!")
    return nothing
end

end  # module

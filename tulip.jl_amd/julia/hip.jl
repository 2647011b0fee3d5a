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
    Backend(; device=0, row_block=nothing, streams=0, ngpus=1, devices=nothing, refine=0, max_link_rows=0)

HIP (gfx950) backend.  `row_block` is the optional block-angular structure hook:
  * `:auto` -- the library finds the structure of the matrix `KKT.setup` receives (`tlpk_detect_blocks`).  This is the
    form that works with Tulip's default `Presolve_Level`: presolve removes rows / columns and renumbers them before
    the KKT backend is set up (`/root/reference/src/model.jl:88-131`), so the user never sees the row numbering;
  * a `Vector{Int}` of length m (block id ≥ 0, or -1 for a linking row), indexed by the rows of the matrix
    `KKT.setup` receives -- only meaningful with `Presolve_Level = 0`;
  * a `BlockAngularMatrix` built through `MatrixFactory` (below) carries its own map; `setup` picks it up.
`ngpus > 1` (block-angular LPs, systems `K1` and `K2`): this ONE Julia process shards the
diagonal blocks over `ngpus` devices of the node (`devices`: HIP ordinals, default `0:ngpus-1`); the
linking-block reductions happen inside the library.  `streams`: concurrent stream groups (0 = auto).
`refine`: iterative-refinement steps per `solve!` (0 = none, as `spd.jl:68`; `K1`, one GPU or `ngpus > 1`).
"""
struct Backend <: AbstractKKTBackend
    device::Int
    row_block::Union{Nothing,Symbol,Vector{Int}}
    streams::Int
    ngpus::Int
    devices::Union{Nothing,Vector{Int32}}
    refine::Int
    max_link_rows::Int
end
function Backend(; device::Int=0, row_block=nothing, streams::Int=0, ngpus::Int=1, devices=nothing, refine::Int=0, max_link_rows::Int=0)
    row_block isa Symbol && row_block !== :auto && throw(ArgumentError("row_block: a vector of block ids, nothing, or :auto"))
    return Backend(device, row_block, streams, ngpus, devices === nothing ? nothing : Vector{Int32}(devices), refine, max_link_rows)
end

"""
    BlockAngularMatrix(A, row_block)

Tulip's structured-matrix hook (`/root/reference/src/parameters.jl:11` `MatrixFactory`,
`/root/reference/src/IPM/ipmdata.jl:166` `construct_matrix(mfact.T, m, n, aI, aJ, aV, mfact.options...)`): a sparse matrix
that knows its block-angular row partition.  `row_block == nothing`: detected when the matrix is built, i.e. on the
presolved, rescaled matrix of the reduced problem.

    set_parameter(model, "MatrixFactory", Tulip.Factory(Tulip.KKT.TlpHIP.BlockAngularMatrix))                  # detect
    set_parameter(model, "MatrixFactory", Tulip.Factory(Tulip.KKT.TlpHIP.BlockAngularMatrix; row_block=rb))    # Presolve_Level = 0
"""
struct BlockAngularMatrix <: AbstractMatrix{Float64}
    A::SparseMatrixCSC{Float64,Int}
    row_block::Vector{Int}         # block id ≥ 0, -1 = linking row; all zeros = no block structure found
    n_blocks::Int
end
Base.size(B::BlockAngularMatrix) = size(B.A)
Base.getindex(B::BlockAngularMatrix, i::Int, j::Int) = B.A[i, j]
SparseArrays.nnz(B::BlockAngularMatrix) = nnz(B.A)
Base.:*(B::BlockAngularMatrix, x::AbstractVector) = B.A * x
LinearAlgebra.mul!(y::AbstractVector, B::BlockAngularMatrix, x::AbstractVector, α::Number, β::Number) = mul!(y, B.A, x, α, β)
LinearAlgebra.mul!(y::AbstractVector, Bt::Adjoint{Float64,BlockAngularMatrix}, x::AbstractVector, α::Number, β::Number) =
    mul!(y, parent(Bt).A', x, α, β)
Base.convert(::Type{SparseMatrixCSC{Float64,Int}}, B::BlockAngularMatrix) = B.A

# the method ipmdata.jl:166 dispatches to (LinearAlgebra.jl:16-32 defines the Matrix and SparseMatrixCSC ones)
import ...TLPLinearAlgebra: construct_matrix
# ipmdata.jl:166 splats `mfact.options` (the keyword arguments given to `Factory`, a `Pairs`) POSITIONALLY: they arrive
# as trailing `Pair`s (`:row_block => rb`); keyword form accepted too
function construct_matrix(::Type{BlockAngularMatrix}, m::Int, n::Int, aI::Vector{Int}, aJ::Vector{Int}, aV::Vector{Float64},
                          opts::Pair...; kwargs...)
    o = Dict{Symbol,Any}(opts..., kwargs...)
    row_block = get(o, :row_block, nothing)
    max_link_rows = get(o, :max_link_rows, 0)
    A = sparse(aI, aJ, aV, m, n)
    if row_block === nothing
        rb, nb, _ = LibTLPK.detect_blocks(m, n, A.colptr, A.rowval; max_link_rows=max_link_rows)
        return BlockAngularMatrix(A, rb, nb)
    end
    length(row_block) == m || throw(DimensionMismatch("length(row_block)=$(length(row_block)) but the matrix has m=$m rows (an explicit map needs Presolve_Level = 0)"))
    return BlockAngularMatrix(A, row_block, maximum(row_block; init=-1) + 1)
end

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
    # a failed create returns no handle: its message is the thread's last create error
    msg = handle == C_NULL ? LibTLPK.last_create_error() : LibTLPK.last_error(handle)
    rc == LibTLPK.TLPK_BADARG && throw(DimensionMismatch("$what: " * msg))
    if rc == LibTLPK.TLPK_OOM || rc == LibTLPK.TLPK_TOO_LARGE
        isempty(msg) || @warn "$what: $msg"        # e.g. "factor needs 860 GB ...; column j of A has c entries ...: KKT_System = K2 ..."
        throw(OutOfMemoryError())                   # -> Trm_MemoryLimit (HSD.jl:327-329)
    end
    error("$what: " * LibTLPK.strerror(rc) * " " * msg)
end

# Convert to sparse matrix if other type is used (cholmod.jl:65)
setup(A, system::Union{K1,K2}, backend::Backend) = setup(convert(SparseMatrixCSC{Float64,Int}, A), system, backend)

_system_code(::K1) = LibTLPK.TLPK_SYSTEM_K1
_system_code(::K2) = LibTLPK.TLPK_SYSTEM_K2

# a BlockAngularMatrix brings its own partition (unless the backend names one explicitly)
function setup(B::BlockAngularMatrix, system::Union{K1,K2}, b::Backend)
    rb = b.row_block isa Vector{Int} ? b.row_block : (B.n_blocks >= 2 ? B.row_block : nothing)
    return setup(B.A, system, Backend(b.device, rb, b.streams, b.ngpus, b.devices, b.refine, b.max_link_rows))
end

function setup(A::SparseMatrixCSC{Float64,Int}, system::Union{K1,K2}, b::Backend)
    m, n = size(A)
    rc, h = LibTLPK.create(m, n, A.colptr, A.rowval, A.nzval; device=b.device,
                           row_block=(b.row_block isa Vector{Int} ? b.row_block : nothing), detect_blocks=(b.row_block === :auto),
                           max_link_rows=b.max_link_rows,
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

# libtlpk.jl -- ccall shim over the C ABI of libtlpk.so (include/tlpk.h).
#
# Where it goes in Tulip: `src/LinearAlgebra/libtlpk.jl`, included from
# `src/LinearAlgebra/LinearAlgebra.jl` INSIDE `module TLPLinearAlgebra` (reference file:
# /root/reference/src/LinearAlgebra/LinearAlgebra.jl:1-33), i.e. it becomes `Tulip.TLPLinearAlgebra.LibTLPK`;
# src/KKT/HIP/hip.jl reaches it with `using ...TLPLinearAlgebra.LibTLPK`.
# Nothing here touches src/IPM.  No CUDA.jl / AMDGPU.jl: plain `ccall` on a C-ABI shared library.
#
# NOTE: Julia is not available in the build or GPU images of this project, so this file has been
# reviewed by eye against include/tlpk.h but never executed.  It is deliberately mechanical.
module LibTLPK

using Libdl

const libtlpk = Ref{String}(get(ENV, "TULIP_LIBTLPK", "libtlpk.so"))

# The library runs up to 4 HIP streams concurrently; the ROCm runtime multiplexes the process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (4 by default), read at the first HIP call of the process.  A tuning knob of the
# HOST process (the library never touches the environment): set it here unless the user already did.
function __init__()
    haskey(ENV, "GPU_MAX_HW_QUEUES") || (ENV["GPU_MAX_HW_QUEUES"] = "8")
    return nothing
end

const TLPK_SYSTEM_K1 = Int32(0)
const TLPK_SYSTEM_K2 = Int32(1)

# return codes (include/tlpk.h)
const TLPK_OK = Cint(0)
const TLPK_NOT_POSDEF = Cint(1)
const TLPK_BADARG = Cint(2)
const TLPK_OOM = Cint(3)
const TLPK_HIPERR = Cint(4)
const TLPK_NO_DEVICE = Cint(5)
const TLPK_TOO_LARGE = Cint(6)
const TLPK_NOT_FACTORED = Cint(7)

# mirror of `tlpk_options` (field order and types must match include/tlpk.h)
Base.@kwdef mutable struct Options
    struct_size::Int32 = 0
    device::Int32 = 0
    ordering::Int32 = 0          # TLPK_ORDER_AMD
    relax::Int32 = 1
    profile::Int32 = 0
    rank::Int32 = 0
    nranks::Int32 = 1
    streams::Int32 = 0
    user_perm::Ptr{Int64} = C_NULL
    row_block::Ptr{Int64} = C_NULL
    mem_budget_bytes::Int64 = 0
    system::Int32 = 0            # TLPK_SYSTEM_K1 | TLPK_SYSTEM_K2
    refine_steps::Int32 = 0
    detect_blocks::Int32 = 0     # 1: the library finds the block-angular structure of the matrix it is given
    keep_on_too_large::Int32 = 0   # 1: tlpk_create returns a live analyse-only handle with TLPK_TOO_LARGE
    max_link_rows::Int64 = 0
end

strerror(code::Integer) = unsafe_string(ccall((:tlpk_strerror, libtlpk[]), Cstring, (Cint,), code))
last_error(h::Ptr{Cvoid}) = unsafe_string(ccall((:tlpk_last_error, libtlpk[]), Cstring, (Ptr{Cvoid},), h))
# a failed tlpk_create / tlpk_create_multi returns no handle: the diagnostic of the calling thread's last failed create
last_create_error() = unsafe_string(ccall((:tlpk_last_create_error, libtlpk[]), Cstring, ()))
backend_name() = unsafe_string(ccall((:tlpk_backend_name, libtlpk[]), Cstring, ()))
system_name() = unsafe_string(ccall((:tlpk_system_name, libtlpk[]), Cstring, ()))
linear_system(h::Ptr{Cvoid}) = unsafe_string(ccall((:tlpk_linear_system, libtlpk[]), Cstring, (Ptr{Cvoid},), h))

"""
    create(A; device, row_block) -> Ptr{Cvoid}

`tlpk_create`: host analyse (ordering, elimination tree, supernodes) + upload.  `A` is passed
with its 1-based `colptr`/`rowval` (index_base = 1); the library copies everything.
"""
function create(m::Int, n::Int, colptr::Vector{Int}, rowval::Vector{Int}, nzval::Vector{Float64};
                device::Integer=0, row_block::Union{Nothing,Vector{Int}}=nothing, system::Int32=TLPK_SYSTEM_K1,
                streams::Integer=0, ngpus::Integer=1, devices::Union{Nothing,Vector{Int32}}=nothing, refine::Integer=0,
                detect_blocks::Bool=false, max_link_rows::Integer=0)
    opt = Options()
    opt.struct_size = Int32(sizeof(Options))
    opt.device = Int32(device)
    opt.system = system
    opt.streams = Int32(streams)
    opt.refine_steps = Int32(refine)
    opt.detect_blocks = Int32(detect_blocks && row_block === nothing)
    opt.max_link_rows = Int64(max_link_rows)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rb = row_block === nothing ? Int[] : row_block
    dv = devices === nothing ? Int32[] : devices
    rc = GC.@preserve colptr rowval nzval rb dv opt begin
        row_block === nothing || (opt.row_block = pointer(rb))
        if ngpus > 1
            # one Julia process, several GPUs: block-angular LPs only (tlpk_create_multi, include/tlpk.h)
            ccall((:tlpk_create_multi, libtlpk[]), Cint,
                  (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint, Ref{Options}, Cint, Ptr{Int32}),
                  h, m, n, colptr, rowval, nzval, 1, opt, ngpus, devices === nothing ? Ptr{Int32}(C_NULL) : pointer(dv))
        else
            ccall((:tlpk_create, libtlpk[]), Cint,
                  (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint, Ref{Options}),
                  h, m, n, colptr, rowval, nzval, 1, opt)
        end
    end
    return rc, h[]
end

destroy(h::Ptr{Cvoid}) = ccall((:tlpk_destroy, libtlpk[]), Cvoid, (Ptr{Cvoid},), h)

"""
    detect_blocks(m, n, colptr, rowval; max_link_rows=0) -> (row_block::Vector{Int}, n_blocks, n_link)

`tlpk_detect_blocks`: block id (0-based, -1 = linking row) of every row of a 1-based CSC matrix; `n_blocks == 1` means
no block-angular structure was found.
"""
function detect_blocks(m::Int, n::Int, colptr::Vector{Int}, rowval::Vector{Int}; max_link_rows::Integer=0)
    rb = zeros(Int, max(m, 1)); nb = Ref{Int64}(1); nl = Ref{Int64}(0)
    rc = GC.@preserve colptr rowval rb ccall((:tlpk_detect_blocks, libtlpk[]), Cint,
        (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Cint, Int64, Ptr{Int64}, Ref{Int64}, Ref{Int64}),
        m, n, colptr, rowval, 1, max_link_rows, rb, nb, nl)
    rc == TLPK_OK || error("tlpk_detect_blocks: " * strerror(rc))
    return resize!(rb, m), Int(nb[]), Int(nl[])
end

update(h::Ptr{Cvoid}, θinv::Vector{Float64}, regP::Vector{Float64}, regD::Vector{Float64}) =
    GC.@preserve θinv regP regD ccall((:tlpk_update, libtlpk[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, θinv, regP, regD)

solve(h::Ptr{Cvoid}, dx::Vector{Float64}, dy::Vector{Float64}, ξp::Vector{Float64}, ξd::Vector{Float64}) =
    GC.@preserve dx dy ξp ξd ccall((:tlpk_solve, libtlpk[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, dx, dy, ξp, ξd)

end  # module

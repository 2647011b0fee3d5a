# BASELINE.md B4: the REAL reference on the host cores -- Tulip's own CHOLMOD backend (KKT_Backend = TlpCholmod.Backend(), KKT_System = K1,
# /root/reference/src/KKT/Cholmod/spd.jl:5-70) timed on the matrices bench.py dumps.  bench.py runs this only if a `julia` binary with Tulip.jl
# installed is on the PATH; neither exists in the build image or on the GPU boxes (no network), so this script has never been executed here --
# it is the probe SURVEY.md 8(d) asks for, kept short enough to be checked by reading.
#   julia tools/reference_cpu_bench.jl <dir> <solves>
# <dir> holds colptr.i64 / rowval.i64 (1-based) / nzval.f64 / dims.i64 (m, n, nnz) / theta.f64 regP.f64 regD.f64 xip.f64 xid.f64.
using SparseArrays, LinearAlgebra, Printf
import Tulip
const KKT = Tulip.KKT

function readvec(T, path, n)
    v = Vector{T}(undef, n)
    open(io -> read!(io, v), path)
    return v
end

function main(dir, nsolves)
    m, n, nz = readvec(Int64, joinpath(dir, "dims.i64"), 3)
    A = SparseMatrixCSC(m, n, readvec(Int64, joinpath(dir, "colptr.i64"), n + 1), readvec(Int64, joinpath(dir, "rowval.i64"), nz),
                        readvec(Float64, joinpath(dir, "nzval.f64"), nz))
    θ, regP, regD = (readvec(Float64, joinpath(dir, f), k) for (f, k) in (("theta.f64", n), ("regP.f64", n), ("regD.f64", m)))
    ξp, ξd = readvec(Float64, joinpath(dir, "xip.f64"), m), readvec(Float64, joinpath(dir, "xid.f64"), n)
    t_setup = @elapsed kkt = KKT.setup(A, KKT.K1(), KKT.TlpCholmod.Backend())
    KKT.update!(kkt, θ, regP, regD)                                  # untimed: first numeric factorisation allocates
    t_update = @elapsed KKT.update!(kkt, θ, regP, regD)
    dx, dy = zeros(n), zeros(m)
    t_solve = @elapsed for _ in 1:nsolves
        KKT.solve!(dx, dy, kkt, ξp, ξd)
    end
    r1 = norm(A * dx .+ regD .* dy .- ξp, Inf); r2 = norm(-(θ .+ regP) .* dx .+ A' * dy .- ξd, Inf)
    @printf("{\"kind\": \"reference\", \"backend\": \"%s\", \"threads\": %d, \"seconds_setup\": %.4f, \"seconds_update\": %.4f, \"seconds_solves\": %.4f, \"ms_per_step\": %.3f, \"residual_inf\": [%.3e, %.3e]}\n",
            KKT.backend(kkt), BLAS.get_num_threads(), t_setup, t_update, t_solve, 1e3 * (t_update + t_solve), r1, r2)
end

main(ARGS[1], parse(Int, ARGS[2]))

# Acceptance checks of the MI355X drop-in, restated from what the reference's tests ASSERT
# (/root/reference/test/runtests.jl:42-63,71-82 and test/partialdot.jl:12-20) without its benchmarking,
# profiling and thread-pinning dependencies.  Usage (the DArray part binds worker i to GPU i-1 over RCCL when every worker has a GPU
# of its own, and lets the workers share the GPUs through the callback transport otherwise, e.g. 2 workers on a one-GPU box):
#   cd distributedhouseholderqr.jl_amd/julia && julia --project=@. test/runtests.jl [nworkers]
# NOT executed in the build image (no Julia); the same checks run through the C ABI in tests/test_gpu_parity.py
# (test_reference_acceptance_*) and tests/test_gpu_complex.py (test_reference_distributed_acceptance).
using Test, Random, LinearAlgebra, Distributed

const nw = isempty(ARGS) ? 0 : parse(Int, ARGS[1])
Random.seed!(0)

using DistributedHouseholderQR
const DHQR = DistributedHouseholderQR

if nw > 0
  addprocs(nw; exeflags=["--project=@."])
  @everywhere using Distributed, DistributedArrays, SharedArrays, LinearAlgebra
  @everywhere using DistributedHouseholderQR
end

normal_eq_residual(A, b, x) = norm(A' * A * x .- A' * b)

@testset "partialdot" begin
  for N in 1:20, T in (Float64, ComplexF64)
    a = rand(T, N); b = rand(T, N)
    for i in 1:N
      @test DHQR.partialdot(a, b, i:N, T) ≈ dot(a[i:end], b[i:end])
    end
  end
end

@testset "qr! and \\ against LinearAlgebra" begin
  for (m, n) in ((110, 100), (220, 200), (440, 400), (880, 800), (1100, 1000), (2200, 2000), (4400, 4000)),
      T in (Float64, ComplexF64)
    A = rand(T, m, n); b = rand(T, m)
    xl = LinearAlgebra.qr!(copy(A), NoPivot()) \ copy(b)
    bound = 8 * normal_eq_residual(A, b, xl)

    H = DHQR.qr!(copy(A))
    @test H isa DHQR.DistributedHouseholderQRStruct
    @test H.α isa Vector{T}
    x = H \ copy(b)
    @test normal_eq_residual(A, b, x) < bound

    if nw > 0
      Ad = DArray(ij -> A[ij[1], ij[2]], size(A), workers(), (1, nworkers()))
      Hd = DHQR.qr!(Ad)
      @test Hd.α isa SharedArray{T}
      @test Vector(Hd.α) ≈ H.α
      xd = Hd \ copy(b)
      @test normal_eq_residual(A, b, xd) < bound
    end
  end
end

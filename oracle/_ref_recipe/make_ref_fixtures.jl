# oracle/_ref_recipe/make_ref_fixtures.jl -- TEST INFRASTRUCTURE (not executed in this image: no Julia toolchain).
#
# The day a Julia toolchain exists next to /root/reference this pins the oracle against THE REFERENCE ITSELF:
#
#   julia --project=/root/reference oracle/_ref_recipe/make_ref_fixtures.jl /root/reference tests/golden/ref
#
# runs the real DistributedHouseholderQR.qr! / `\` (src/DistributedHouseholderQR.jl:311-321) on inputs from the
# portable generator shared with oracle/dhqr_oracle.c (u01(seed, i + j*m): splitmix64 finaliser, 53-bit mantissa)
# and writes, per case, raw little-endian Float64 column-major files
#     <out>/<tag>_A.bin  <tag>_H.bin  <tag>_alpha.bin  <tag>_b.bin  <tag>_x.bin   +   <out>/manifest.txt
# tests/test_oracle.py::test_oracle_against_reference_fixtures picks them up when present and compares the
# oracle's restatement element by element (H, alpha, x); until then the element-wise parity stays "unpinned".
using LinearAlgebra

refroot, outdir = ARGS[1], ARGS[2]
include(joinpath(refroot, "src", "DistributedHouseholderQR.jl"))
const DHQR = DistributedHouseholderQR
mkpath(outdir)

mix64(z::UInt64) = begin
  z = (z ⊻ (z >> 30)) * 0xBF58476D1CE4E5B9
  z = (z ⊻ (z >> 27)) * 0x94D049BB133111EB
  z ⊻ (z >> 31)
end
u01(seed::UInt64, idx::UInt64) = Float64(mix64(seed + (idx + 0x1) * 0x9E3779B97F4A7C15) >> 11) * (1.0 / 9007199254740992.0)
randmat(m, n, seed) = [u01(UInt64(seed), UInt64((i - 1) + (j - 1) * m)) for i in 1:m, j in 1:n]
randvec(m, seed) = [u01(UInt64(seed), UInt64(i - 1)) for i in 1:m]

open(joinpath(outdir, "manifest.txt"), "w") do man
  # the reference's own shapes (test/runtests.jl:42: m = 1.1 n) plus three small ones with every entry checked
  for (m, n, seed) in [(33, 17, 1), (64, 64, 2), (129, 129, 3), (110, 100, 0), (220, 200, 0), (440, 400, 0), (1100, 1000, 0)]
    tag = "ref_$(m)x$(n)_seed$(seed)"
    A = randmat(m, n, seed)
    b = randvec(m, seed + 1)
    H = DHQR.qr!(copy(A))          # src:311-315
    x = H \ b                      # src:317-321
    for (name, arr) in (("A", A), ("H", H.A), ("alpha", H.α), ("b", b), ("x", x))
      write(joinpath(outdir, "$(tag)_$(name).bin"), Float64.(vec(Array(arr))))
    end
    println(man, "$tag $m $n $seed")
  end
end

"""Prints the PrimeTab<R> literal tables of csrc/kernels_reg.hpp (cos / sin of 2 pi m / R at 21 digits; needs mpmath)."""
import mpmath
mpmath.mp.dps = 40
for R in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
    H = (R - 1) // 2
    c = ", ".join(mpmath.nstr(mpmath.cos(2 * mpmath.pi * m / R), 21, strip_zeros=False) for m in range(1, H + 1))
    s = ", ".join(mpmath.nstr(mpmath.sin(2 * mpmath.pi * m / R), 21, strip_zeros=False) for m in range(1, H + 1))
    print("template <> struct PrimeTab<%d> {" % R)
    print("    static constexpr double c[%d] = {%s};" % (H, c))
    print("    static constexpr double s[%d] = {%s};" % (H, s))
    print("};")

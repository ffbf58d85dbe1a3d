"""Derives and checks the constants of gelu_erf (rassengine_amd/csrc/encoder_gemm.hip):
   GELU(x) = max(x, 0) - h * erfc(h sqrt2),  h = |x| / 2,  erfc(h sqrt2) ~= exp2(Q(min(h, 5 / sqrt2))),
Q = the degree-7 least-squares (Chebyshev-node) fit of log2(erfc(z)) on z in [0, 5], re-expressed in h.  Prints the
coefficients and the maximum |error| of the fp32 evaluation against math.erf over [-12, 12]."""
import math
import numpy as np
from numpy.polynomial import chebyshev as C, Polynomial as P
from scipy.special import erfc
Z, DEG = 5.0, 7
z = np.cos(np.pi * (np.arange(8000) + 0.5) / 8000) * Z / 2 + Z / 2
c = C.chebfit(2 * z / Z - 1, np.log2(erfc(z)), DEG)
t = P([-1, 2 / Z])
q = sum(ci * t ** k for k, ci in enumerate(C.cheb2poly(c)))
coef = q(P([0, math.sqrt(2)])).coef
print("Q(h) = sum c_k h^k, c_0..c_7 =", ", ".join(f"{np.float32(v):.7e}" for v in coef))
x = np.linspace(-12, 12, 2_000_001).astype(np.float32)
h = (np.float32(0.5) * np.abs(x)).astype(np.float32)
hc = np.minimum(h, np.float32(Z / math.sqrt(2)))
acc = np.full_like(hc, np.float32(coef[7]))
for k in range(6, -1, -1):
    acc = (acc * hc + np.float32(coef[k])).astype(np.float32)
g = (np.maximum(x, np.float32(0)) - h * np.exp2(acc.astype(np.float64)).astype(np.float32)).astype(np.float32)
exact = 0.5 * x.astype(np.float64) * (1 + np.vectorize(math.erf)(x.astype(np.float64) / math.sqrt(2)))
err = np.abs(g - exact)
print(f"max |GELU error| {err.max():.3e} at x = {x[err.argmax()]:.4f}  (bf16 half-ulp of an output of 0.01: 2e-5)")

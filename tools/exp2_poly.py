"""Coefficients and error of a polynomial exp2 for the encoder-attention softmax (DESIGN.md §7 "Next"): the flash
kernel's inner loop is bound by quarter-rate v_exp_f32; a share of the exponentials can go to the packed-FMA pipe.
x <= 0 (score minus running maximum, in log2 units): x = n + f with n = floor(x), f in [0, 1);
2^f ~ c0 + f*(c1 + f*(c2 + f*c3)) (Horner: 3 FMAs, v_pk_fma_f32 does two values per instruction); the result is
scaled by 2^n by adding n << 23 to the bit pattern (flush to 0 below 2^-126).  P goes to fp16 for the PV MFMA
(relative step 4.9e-4), so a relative error of ~1e-4 is invisible.  Run: python tools/exp2_poly.py"""
import numpy as np

f = np.linspace(0.0, 1.0, 200001)
target = np.exp2(f)
# least squares on Chebyshev nodes, then a few Remez-style reweighting rounds on the relative error
for degree in (2, 3, 4):
    w = np.ones_like(f)
    for _ in range(30):
        V = np.vander(f, degree + 1, increasing=True)
        c, *_ = np.linalg.lstsq(V * (w / target)[:, None], w, rcond=None)
        rel = np.abs(V @ c / target - 1.0)
        w = w * (1.0 + 4.0 * rel / rel.max())
    c32 = c.astype(np.float32)
    x = np.float32(-np.random.default_rng(0).uniform(0, 24, 2_000_000))
    n = np.floor(x)
    fr = (x - n).astype(np.float32)
    p = np.zeros_like(fr) + c32[-1]
    for k in range(degree - 1, -1, -1):
        p = p * fr + c32[k]                                  # fp32 Horner, as the FMAs would do it
    bits = p.view(np.int32) + (n.astype(np.int32) << 23)
    approx = bits.view(np.float32)
    rel32 = np.abs(approx.astype(np.float64) / np.exp2(x.astype(np.float64)) - 1.0)
    print(f"degree {degree}: coefficients {[float(v) for v in c32]}  max rel err (fp32 Horner) {rel32.max():.2e}"
          f"  mean {rel32.mean():.2e}")

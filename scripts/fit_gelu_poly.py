"""Coefficients of the GELU epilogue polynomials in multimae_amd/csrc/common.h (gelu_poly2 / gelu_grad_poly2).

Both Phi(x) - 1/2 and gelu'(x) - 1/2 = Phi(x) + x phi(x) - 1/2 are odd: x * P(u), u = 2 x^2 / R^2 - 1 in [-1, 1], P of degree
n - 1 in u (Horner in the kernel).  Least squares on Chebyshev nodes of [0, R] in x, with the equality R * P(1) = 1/2, so that
an argument clamped to [-R, R] gives exactly +-1/2 and no select is needed beyond the clamp.  Prints the fp32 coefficients
and the largest absolute error of the fp32 evaluation over [-8, 8]."""
import numpy as np
from math import erf, exp, pi, sqrt

Phi = np.vectorize(lambda x: 0.5 * (1.0 + erf(x / sqrt(2.0))))
phi = lambda x: np.exp(-0.5 * x * x) / sqrt(2.0 * pi)


def fit(f, R, n, nodes=4000):
    k = np.arange(nodes)
    x = 0.5 * R * (1.0 - np.cos(pi * (k + 0.5) / nodes))  # Chebyshev nodes on [0, R]
    u = 2.0 * x * x / (R * R) - 1.0
    T = np.polynomial.chebyshev.chebvander(u, n - 1) * x[:, None]
    y = f(x)
    # equality constraint R * sum_k a_k T_k(1) = 1/2 (T_k(1) = 1): eliminate a_0
    A = T[:, 1:] - T[:, :1]
    b = y - T[:, 0] * (0.5 / R)
    a_rest = np.linalg.lstsq(A, b, rcond=None)[0]
    a = np.concatenate([[0.5 / R - a_rest.sum()], a_rest])
    c = np.polynomial.chebyshev.cheb2poly(a)  # monomial coefficients in u, lowest first
    return np.concatenate([c, np.zeros(n - len(c))]).astype(np.float32)


def evaluate(c, R, x):
    x = x.astype(np.float32)
    xc = np.clip(x, np.float32(-R), np.float32(R))
    u = xc * xc * np.float32(2.0 / (R * R)) - np.float32(1.0)
    acc = np.full_like(x, c[-1])
    for k in range(len(c) - 2, -1, -1):
        acc = acc * u + c[k]
    return xc * acc + np.float32(0.5)


if __name__ == "__main__":
    xs = np.linspace(-8, 8, 400001)
    for name, f, R, n in [("Phi(x) - 1/2 (gelu = x * (.. + 1/2))", lambda x: Phi(x) - 0.5, 4.0, 8),
                          ("gelu'(x) - 1/2", lambda x: Phi(x) + x * phi(x) - 0.5, 4.5, 10)]:
        c = fit(f, R, n)
        err = np.abs(evaluate(c, R, xs).astype(np.float64) - (f(xs) + 0.5))
        print(f"{name}: R = {R}, {n} coefficients, max |error| = {err.max():.3g} at x = {xs[err.argmax()]:.3f}; value at -8: {evaluate(c, R, np.array([-8.0]))[0]:.3g}")
        print("  {" + ", ".join(f"{float(v)!r}f" for v in c) + "}")
        # the same polynomial re-expanded in t = x^2 (double precision), evaluated in fp32 Horner form in t
        ct = np.polynomial.Polynomial(c.astype(np.float64))(np.polynomial.Polynomial([-1.0, 2.0 / (R * R)])).coef
        x32 = xs.astype(np.float32); xc = np.clip(x32, np.float32(-R), np.float32(R)); t = xc * xc
        acc = np.full_like(x32, np.float32(ct[-1]))
        for k in range(len(ct) - 2, -1, -1):
            acc = acc * t + np.float32(ct[k])
        err_t = np.abs((xc * acc + np.float32(0.5)).astype(np.float64) - (f(xs) + 0.5))
        print(f"  in x^2: max |error| = {err_t.max():.3g}")
        print("  {" + ", ".join(f"{float(np.float32(v))!r}f" for v in ct) + "}")

"""The reference's only data source (easy_boston_data.py:7-45): five "Boston-like" features with two correlated
blocks.  Host-side input generation (NumPy), not part of the device path; the draws reproduce the reference's
stream bit for bit (tests/test_benchmark_grid.py checks A and b against the golden fixture made by the reference)."""
import numpy as np


def generate_correlated_boston_like_data(m: int = 1000, seed: int = 42, noise_std: float = 2.0, rho1: float = 0.8,
                                         rho2: float = 0.9):
    """Returns (A, b, x_true): A is m x 5 float64, unstandardised (cond(A^T A) ~ 1e9, L ~ 9.4e7).

    Columns: rooms & crime ~ N([6, 0.2], 0.25*[[1, rho1], [rho1, 1]]), tax & age ~ N([300, 60],
    100*[[1, rho2], [rho2, 1]]), distance ~ N(4, 1); b = A @ [5, 0, -0.02, -0.05, 1.5] + N(0, noise_std).
    One PCG64 generator, drawn in that order (block 1, block 2, distance, noise)."""
    gen = np.random.default_rng(seed)
    unit = lambda rho: np.array([[1.0, rho], [rho, 1.0]])               # noqa: E731
    rooms_crime = gen.multivariate_normal(mean=[6, 0.2], cov=0.25 * unit(rho1), size=m)
    tax_age = gen.multivariate_normal(mean=[300, 60], cov=100 * unit(rho2), size=m)
    distance = gen.normal(4, 1.0, size=(m, 1))
    A = np.hstack([rooms_crime, tax_age, distance])
    x_true = np.array([5.0, 0.0, -0.02, -0.05, 1.5])
    b = A @ x_true + gen.normal(0, noise_std, size=m)
    return A, b, x_true

"""MI355X drop-in for the reference's ``objective_functions.py`` (one residual pass over A, kernel K5)."""
from . import _core


def compute_objective(x, A, b, reg_type, alpha1, alpha2):
    """f(x) = ½‖Ax−b‖² (+ ½α₂‖x‖² for ridge/elasticnet) (+ α₁‖x‖₁ for lasso/elasticnet).
    objective_functions.py:3-30; ValueError for any other reg_type (:28)."""
    if reg_type not in ("lasso", "ridge", "elasticnet"):
        raise ValueError(f"Unsupported reg_type='{reg_type}'")
    prob = _core.as_problem(A, b)
    xt = _core.to_device_vec(x, prob.device)
    rr, x2, x1 = prob.residual_objective(xt)
    g = 0.5 * rr
    if reg_type in ("ridge", "elasticnet"):
        g += 0.5 * alpha2 * x2
    h = alpha1 * x1 if reg_type in ("lasso", "elasticnet") else 0.0
    return g + h

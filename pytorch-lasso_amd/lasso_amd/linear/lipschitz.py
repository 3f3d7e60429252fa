"""lambda_max(W^T W) (reference ista.py:8-14) -- placeholder until the native
estimator lands."""


def lipschitz_constant(weight):
    raise NotImplementedError("lasso_amd: lr='auto' needs the native Lipschitz kernel (pending)")

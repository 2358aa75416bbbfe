from bayesianinferencedl_amd.bayesian_inference.gaussian_field import make_cov_chol  # noqa: F401

"""Drop-in import path of the reference (`from bayesian_inference.gaussian_field import make_cov_chol`): thin re-exports of
bayesianinferencedl_amd.bayesian_inference (repo root on sys.path)."""

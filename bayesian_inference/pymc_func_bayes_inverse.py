from bayesianinferencedl_amd.bayesian_inference.pymc_func_bayes_inverse import (  # noqa: F401
    SqError, SqErrorOpFOM, SqErrorOpROM, SqErrorOpROMML, make_op)

from bayesianinferencedl_amd.rom.model_constr_adaptive_sampling import *  # noqa: F401,F403
from bayesianinferencedl_amd.rom.model_constr_adaptive_sampling import sample, enrich, generate_five_param_basis  # noqa: F401

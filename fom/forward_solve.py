from bayesianinferencedl_amd.fom.forward_solve import *  # noqa: F401,F403
from bayesianinferencedl_amd.fom.forward_solve import Fin  # noqa: F401

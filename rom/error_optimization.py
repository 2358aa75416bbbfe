from bayesianinferencedl_amd.rom.error_optimization import *  # noqa: F401,F403
from bayesianinferencedl_amd.rom.error_optimization import optimize_five_param, optimize_nine_param, rom_error_batch  # noqa: F401

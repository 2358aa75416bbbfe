from bayesianinferencedl_amd.fom.thermal_fin import get_space  # noqa: F401

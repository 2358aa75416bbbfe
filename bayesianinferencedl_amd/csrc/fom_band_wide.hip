// The frontal band sweep's widest windows (lattice m = 24: NSF = 8, NSP = 26, ten extras; m = 28: 9, 30, twelve) as their own translation unit: the same
// templates as fom_band.hip, only the instantiations differ (launch_fom_band_wide), so that the two compile side by side.
#define FINROM_BAND_TU_WIDE 1
#include "fom_band.hip"

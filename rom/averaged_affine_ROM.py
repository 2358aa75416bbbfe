from bayesianinferencedl_amd.rom.averaged_affine_ROM import AffineROMFin  # noqa: F401

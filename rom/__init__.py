"""Drop-in import path of the reference (`from rom.averaged_affine_ROM import AffineROMFin`): thin re-exports of
bayesianinferencedl_amd.rom (repo root on sys.path)."""

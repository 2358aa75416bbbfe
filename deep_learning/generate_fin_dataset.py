from bayesianinferencedl_amd.deep_learning.generate_fin_dataset import gen_affine_avg_rom_dataset  # noqa: F401

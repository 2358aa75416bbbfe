"""Drop-in import path of the reference (`from deep_learning.generate_fin_dataset import gen_affine_avg_rom_dataset`): thin re-exports of
bayesianinferencedl_amd.deep_learning (repo root on sys.path)."""

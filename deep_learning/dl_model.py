from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel, load_dataset_avg_rom, res_bn_fc_model  # noqa: F401

from bayesianinferencedl_amd.deep_learning.dl_model import ResBnFcModel, res_bn_fc_model  # noqa: F401

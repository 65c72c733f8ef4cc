from hqtransformer_amd.config import get_base_config, merge, load_config  # noqa: F401

from hqtransformer_amd.models import ImageGPT2  # noqa: F401

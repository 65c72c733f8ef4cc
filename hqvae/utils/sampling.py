from hqtransformer_amd.sampling import sampling_ihqgpt, sampling_hqtransformer, rearrange_codes, rearrange_codes3  # noqa: F401

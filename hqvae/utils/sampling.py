from hqtransformer_amd.sampling import sampling_ihqgpt, rearrange_codes  # noqa: F401

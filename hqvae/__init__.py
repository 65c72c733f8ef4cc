"""Import-compatibility shim: the reference's drivers (`measure_throughput`, `sampling_hqmodel.py`, the demo
notebook) import `hqvae.models.ImageGPT2`, `hqvae.utils.config2.get_base_config`,
`hqvae.utils.sampling.sampling_ihqgpt` and `hqvae.utils.utils.set_seed`; with this directory ahead of the
reference on PYTHONPATH those names resolve to the MI355X-native path."""

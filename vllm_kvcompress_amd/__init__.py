"""MI355X-native KV-Compress eviction + compaction hot path.

Drop-in for the reference fork's op surface for this path only:

* ``vllm_kvcompress_amd._custom_ops``           <-> ``vllm._custom_ops`` (KV-Compress section)
* ``vllm_kvcompress_amd.kvcompress.metrics``    <-> ``vllm.kvcompress.metrics``
* ``vllm_kvcompress_amd.torch_ops.register()``  registers ``torch.ops._C_kvc_ops.*`` /
  ``torch.ops._C_cache_ops.kvcompress_reshape_and_cache`` so the fork's own wrappers
  resolve to the HIP kernels unchanged.

See DESIGN.md / INTEGRATION.md at the repository root.
"""
from ._lib import LIB_PATH, MAX_INT, block_layout, load, set_block_layout  # noqa: F401

__all__ = ["LIB_PATH", "MAX_INT", "load", "block_layout", "set_block_layout"]

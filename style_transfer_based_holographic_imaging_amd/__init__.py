"""MI355X-native holographic field-retrieval hot path (VGG-AdaIN-decoder + angular spectrum).

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C-ABI), ``_lib`` (ctypes
binding of ``include/holo_hip.h``), ``ops`` (tensor-level wrappers), ``engine`` (layer plans),
``shard`` (batch sharding + RCCL gather) and ``synth`` (seeded weights / inputs).
"""
__all__ = ["synth"]

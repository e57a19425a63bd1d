"""`tfra_amd.dynamic_embedding` — mirrors `tensorflow_recommenders_addons.dynamic_embedding`
(`/root/reference/tensorflow_recommenders_addons/dynamic_embedding/__init__.py`) for the hot path."""
from . import device_ops
from . import optimizer as optimizers
from .optimizer import (CapturedTrainStep, DynamicEmbeddingOptimizer, MultiTablePrefetchStep, OverlapAssignStep,
                        PrefetchAssignStep, PrefetchStep, assign_step_driver_for, assign_step_for)
from .restrict_policies import FrequencyRestrictPolicy, RestrictPolicy, TimestampRestrictPolicy
from .table_ops import (SparsePlan, CuckooHashTable, HkvEvictStrategy, HkvHashTable, KHkvHashTableInitCapacity,
                        KHkvHashTableMaxCapacity, KHkvHashTableMaxHbmForValuesByBytes)
from .variable import (CuckooHashTableConfig, CuckooHashTableCreator, HkvHashTableConfig, HkvHashTableCreator,
                       KVCreator, TrainableWrapper, Variable, default_partition_fn, embedding_lookup,
                       embedding_lookup_sparse, embedding_lookup_unique, get_variable,
                       safe_embedding_lookup_sparse)

__all__ = [n for n in dir() if not n.startswith("_")]

"""Shared-pool serving layer: many client sessions, one launch per tick - on one GPU (SharedStreamPool) or
one pool per GPU behind one front (ShardedStreamPool, SURVEY §8 e)."""

from .shared_pool import PooledSession, SharedStreamPool
from .sharded_pool import ShardedStreamPool

__all__ = ["SharedStreamPool", "ShardedStreamPool", "PooledSession"]

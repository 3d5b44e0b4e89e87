"""Shared-pool serving layer: many client sessions, one launch per tick."""

from .shared_pool import PooledSession, SharedStreamPool

__all__ = ["SharedStreamPool", "PooledSession"]

"""Scalar log the hot path writes and icrl() scrapes (ref: stable_baselines3/common/logger.py:441-560 — only the
``record`` / ``Logger.CURRENT.name_to_value`` surface the ICRL loop touches, icrl/icrl.py:212).  ``Logger.CURRENT`` is per
host thread: several runs may share one process (icrl_amd/seed_batch.py)."""
import threading

_tls = threading.local()


class _Current(type):
    @property
    def CURRENT(cls):
        cur = getattr(_tls, "logger", None)
        if cur is None:
            cur = _tls.logger = cls()
        return cur

    @CURRENT.setter
    def CURRENT(cls, value):
        _tls.logger = value


class Logger(metaclass=_Current):
    def __init__(self):
        self.name_to_value = {}

    def record(self, key, value, exclude=None):
        self.name_to_value[key] = value

    def dump(self, step=0):
        self.name_to_value = {}


def record(key, value, exclude=None):
    Logger.CURRENT.record(key, value, exclude)


def dump(step=0):
    """ref: logger.py:331-335, 492-504 — the reference writes the diagnostics of the iteration and clears them."""
    Logger.CURRENT.dump(step)


def configure():
    Logger.CURRENT = Logger()

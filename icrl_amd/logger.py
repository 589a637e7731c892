"""Scalar log the hot path writes and icrl() scrapes (ref: stable_baselines3/common/logger.py:441-560 — only the
``record`` / ``Logger.CURRENT.name_to_value`` surface the ICRL loop touches, icrl/icrl.py:212)."""


class Logger:
    CURRENT = None

    def __init__(self):
        self.name_to_value = {}

    def record(self, key, value, exclude=None):
        self.name_to_value[key] = value

    def dump(self, step=0):
        self.name_to_value = {}


Logger.CURRENT = Logger()


def record(key, value, exclude=None):
    Logger.CURRENT.record(key, value, exclude)


def configure():
    Logger.CURRENT = Logger()

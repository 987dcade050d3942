"""Minimal observability for the MCMC path: console/file logging and a no-op scalar writer.

The reference's TensorBoard / nibabel / tvtk / seaborn logger (logger/, 588 lines) is out of scope (SURVEY.md section 2,
row 13); the trainer only needs `add_scalar` / `set_step` and a python logger.
"""
import logging


def setup_logging(save_dir=None, level=logging.INFO):
    handlers = [logging.StreamHandler()]
    if save_dir is not None:
        handlers.append(logging.FileHandler(str(save_dir) + '/info.log'))
    logging.basicConfig(level=level, format='%(message)s', handlers=handlers, force=True)


class ScalarWriter:
    """stand-in for TensorboardWriter (logger/visualization.py:12-55): remembers the last value of every scalar"""

    def __init__(self):
        self.step = 0
        self.scalars = {}

    def set_step(self, step):
        self.step = step

    def add_scalar(self, key, value):
        self.scalars[key] = (self.step, value)

    def write_hparams(self, *_):
        pass

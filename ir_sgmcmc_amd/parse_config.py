"""ConfigParser: the reference's name-based object factory (parse_config.py:20-350) over this package's modules.

`init_obj(name, module)` = `getattr(module, cfg[name]['type'])(**cfg[name]['args'])` (parse_config.py:251-266), with the
same key schema as configs/*/config.json, so an unmodified reference config resolves to the HIP-backed classes.
Run-directory creation, tensorboard and the metric-name bookkeeping are reduced to what the MCMC path needs.
"""
import logging
from collections import OrderedDict
from pathlib import Path

import numpy as np

from .data_loader import data_loaders as module_data
from .logger import setup_logging
from .model import distributions as model_distr
from .model import loss as model_loss
from .optimizers import Adam
from .utils import registration, transformation
from .utils.util import read_json, write_json


class ConfigParser:
    def __init__(self, config, modification=None, timestamp=None, make_dirs=True):
        self._config = _update_config(config, modification)
        self.log_levels = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}
        verbosity = self['trainer']['verbosity']
        assert verbosity in self.log_levels, f'verbosity option {verbosity} is invalid. Valid options are {self.log_levels.keys()}.'
        self._logger = logging.getLogger('default')
        self._logger.setLevel(self.log_levels[verbosity])
        self.structures_dict = {'left_thalamus': 10, 'left_caudate': 11, 'left_putamen': 12, 'left_pallidum': 13,
                                'brain_stem': 16, 'left_hippocampus': 17, 'left_amygdala': 18, 'left_accumbens': 26,
                                'right_thalamus': 49, 'right_caudate': 50, 'right_putamen': 51, 'right_pallidum': 52,
                                'right_hippocampus': 53, 'right_amygdala': 54, 'right_accumbens': 58}
        self._dir = Path(self.config['trainer']['save_dir']) / self.config['name'] / (timestamp or '')
        self._log_dir = self._dir / 'log'
        if make_dirs:
            self._log_dir.mkdir(parents=True, exist_ok=True)
            (self._dir / 'samples' / 'MCMC').mkdir(parents=True, exist_ok=True)
            setup_logging(self._log_dir, self.log_levels[verbosity])
            write_json(self.config, self._dir / 'config.json')

    @classmethod
    def from_args(cls, args, options='', timestamp=None):
        if not isinstance(args, tuple):
            args = args.parse_args()
        return cls(read_json(Path(args.config)), timestamp=timestamp)

    @classmethod
    def from_dict(cls, config, **kw):
        return cls(OrderedDict(config), **kw)

    # ---- factories (parse_config.py:100-249)
    def init_data_loader(self):
        self['data_loader']['args']['save_dirs'] = self.save_dirs
        t_args = self['transformation_module'].get('args', {})
        if 'cps' in t_args:
            self['data_loader']['args']['cps'] = t_args['cps']
        return self.init_obj('data_loader', module_data)

    def init_losses(self):
        data_loss = self.init_obj('data_loss', model_loss)
        losses = {'data': {'loss': data_loss}, 'reg': {}, 'entropy': None}
        if 'data_loss_scale_prior' in self.config:
            losses['data']['scale_prior'] = self.init_obj('data_loss_scale_prior', model_distr)
            losses['data']['proportion_prior'] = self.init_obj('data_loss_proportion_prior', model_distr)
        if 'entropy_loss' in self.config:
            losses['entropy'] = self.init_obj('entropy_loss', model_loss)
        self['reg_loss']['args']['dims'] = self['data_loader']['args']['dims']
        reg_loss = self.init_obj('reg_loss', model_loss)
        losses['reg']['loss'] = reg_loss
        if reg_loss.learnable:
            dof = np.prod(self['data_loader']['args']['dims']) * 3.0
            if type(reg_loss).__name__ == 'RegLoss_LogNormal':
                self['reg_loss_loc_prior']['args']['dof'] = dof
                losses['reg']['loc_prior'] = self.init_obj('reg_loss_loc_prior', model_distr)
                losses['reg']['scale_prior'] = self.init_obj('reg_loss_scale_prior', model_distr)
            elif type(reg_loss).__name__ == 'RegLoss_L2':
                shape = 0.5 * dof
                self['reg_loss_w_reg_prior']['args']['shape'] = shape
                self['reg_loss_w_reg_prior']['args']['rate'] = 1.0 / shape
                losses['reg']['w_reg_prior'] = self.init_obj('reg_loss_w_reg_prior', model_distr)
        return losses

    def init_metrics(self):
        """names of the scalars the MCMC loop tracks (parse_config.py:150-208, MCMC subset)"""
        K = self['data_loss']['args'].get('no_components', 1)
        C = self['trainer']['no_chains']
        m = [f'MCMC/GMM/scale_{i}' for i in range(K)] + [f'MCMC/GMM/proportion_{i}' for i in range(K)] + ['MCMC/avg_loss']
        for i in range(C):
            m += [f'MCMC/chain_{i}/{t}' for t in ('data_term', 'reg_term', 'VD/alpha', 'reg/energy', 'no_non_diffeomorphic_voxels')]
            m += [f'MCMC/chain_{i}/DSC/{s}' for s in self.structures_dict]
        return m

    def init_transformation_and_registration_modules(self):
        self['transformation_module'].setdefault('args', {})['dims'] = self['data_loader']['args']['dims']
        return self.init_obj('transformation_module', transformation), self.init_obj('registration_module', registration)

    def init_optimizer_q_v(self, var_params_q_v):
        a = self['optimizer_q_v']['args']   # parse_config.py:226-232
        return Adam([{'params': [var_params_q_v['mu']], 'lr': a['lr_mu']}, {'params': [var_params_q_v['log_var']], 'lr': a['lr_log_var']},
                     {'params': [var_params_q_v['u']], 'lr': a['lr_u']}], lr_decay=a['lr_decay'])

    def init_optimizer_GMM(self, data_loss):
        if self['optimizer_GMM']['type'] != 'Adam':
            raise RuntimeError('only the Adam optimiser is supported for the GMM')
        a = self['optimizer_GMM']['args']
        return Adam([{'params': [data_loss.log_std], 'lr': a['lr_log_std']}, {'params': [data_loss.logits], 'lr': a['lr_logits']}],
                    lr_decay=a['lr_decay'])

    def init_optimizer_reg(self, reg_loss):
        if self['optimizer_reg']['type'] != 'Adam':
            raise RuntimeError('only the Adam optimiser is supported for the regularisation')
        a = self['optimizer_reg']['args']
        if type(reg_loss).__name__ == 'RegLoss_LogNormal':
            return Adam([{'params': [reg_loss.loc], 'lr': a['lr_loc']}, {'params': [reg_loss.log_scale], 'lr': a['lr_log_scale']}],
                        lr_decay=a['lr_decay'])
        return Adam(reg_loss.parameters(), lr=a['lr_log_w_reg'], lr_decay=a['lr_decay'])

    def init_obj(self, name, module, *args, **kwargs):
        module_name = self[name]['type']
        module_args = dict(self[name]['args']) if 'args' in dict(self[name]) else dict()
        module_args.update(kwargs)
        return getattr(module, module_name)(*args, **module_args)

    def __getitem__(self, name):
        return self.config[name]

    @property
    def logger(self):
        return self._logger

    @property
    def config(self):
        return self._config

    @property
    def dir(self):
        return self._dir

    @property
    def log_dir(self):
        return self._log_dir

    @property
    def save_dirs(self):
        return {'dir': self._dir, 'tensors': self._dir / 'tensors', 'samples': self._dir / 'samples', 'images': self._dir / 'images',
                'fields': self._dir / 'fields', 'grids': self._dir / 'grids', 'norms': self._dir / 'norms',
                'checkpoints': self._dir / 'checkpoints'}


def _update_config(config, modification):
    if modification:
        for k, v in modification.items():
            if v is not None:
                node = config
                keys = k.split(';')
                for key in keys[:-1]:
                    node = node[key]
                node[keys[-1]] = v
    return config

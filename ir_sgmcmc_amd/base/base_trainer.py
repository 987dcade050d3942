"""BaseTrainer (reference base/base_trainer.py:7-61): reads the trainer section of the config and holds the modules."""
from abc import abstractmethod

from ..logger import ScalarWriter
from ..utils.util import MetricTracker


class BaseTrainer:
    def __init__(self, config, data_loader, losses, transformation_module, registration_module, metrics, device='cuda:0'):
        self.config = config
        self.logger = config.logger
        self.device = device  # the reference hard-codes 'cuda:0' (base/base_trainer.py:16)

        self.data_loader = data_loader
        self.structures_dict = getattr(config, 'structures_dict', {})
        self.save_dirs = getattr(data_loader, 'save_dirs', None)

        self.losses = {'data': dict(losses['data']), 'reg': dict(losses['reg']), 'entropy': losses.get('entropy')}
        self.transformation_module = transformation_module
        self.registration_module = registration_module
        self.diff_op = self.losses['reg']['loss'].diff_op

        cfg_trainer = config['trainer']
        self.VI = cfg_trainer['VI']
        self.start_iter_VI, self.no_iters_VI = 1, int(cfg_trainer['no_iters_VI'])
        self.no_samples_VI_test = int(cfg_trainer['no_samples_VI_test'])
        self.log_period_VI = cfg_trainer['log_period_VI']

        self.MCMC = cfg_trainer['MCMC']
        self.MCMC_init = cfg_trainer['MCMC_init']  # one of 'VI', 'identity', 'noise'
        self.no_chains = int(cfg_trainer['no_chains'])
        self.no_samples_MCMC = int(cfg_trainer['no_samples_MCMC'])
        self.no_iters_burn_in = int(cfg_trainer['no_iters_burn_in'])
        self.log_period_MCMC = cfg_trainer['log_period_MCMC']

        self.writer = ScalarWriter()
        self.metrics = MetricTracker(*[m for m in metrics], writer=self.writer)

    @abstractmethod
    def _run_model(self):
        raise NotImplementedError

    def run(self):
        self._run_model()

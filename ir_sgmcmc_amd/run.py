"""python -m ir_sgmcmc_amd.run -c config.json   (reference run.py:8-38)"""
import argparse
from datetime import datetime

from .parse_config import ConfigParser
from .trainer import Trainer


def run(config):
    data_loader = config.init_data_loader()
    losses = config.init_losses()
    transformation_module, registration_module = config.init_transformation_and_registration_modules()
    metrics = config.init_metrics()
    trainer = Trainer(config, data_loader, losses, transformation_module, registration_module, metrics)
    trainer.run()
    return trainer


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='MCMC')
    parser.add_argument('-c', '--config', default=None, type=str, help='config file path (default: None)')
    config = ConfigParser.from_args(parser, timestamp=datetime.now().strftime(r'%m%d_%H%M%S'))
    run(config)

"""drop-in entry point: python run.py -c config.json  (same CLI as the reference's run.py)"""
import runpy

if __name__ == '__main__':
    runpy.run_module('ir_sgmcmc_amd.run', run_name='__main__')

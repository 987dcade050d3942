"""MI355X-native SG-MCMC registration inner loop (drop-in for dgrzech/ir-sgmcmc's `_SGLD_transition` path).

Importing the package never touches the GPU; the HIP library (`csrc/libirsgmcmc.so`) is loaded on
first use of a device op and its absence is a hard error (there is no CPU fallback in the product).
"""
__version__ = '0.1.0'

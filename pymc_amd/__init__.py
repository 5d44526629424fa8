"""pymc_amd: MI355X-native NUTS/HMC sampling engine behind PyMC's step-method surface."""

__version__ = "0.1.0"

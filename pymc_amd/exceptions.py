"""Exception types of the hot path, same names as the reference's.

`SamplingError` (pymc/exceptions.py; raised at pymc/step_methods/hmc/base_hmc.py:205-224),
`IntegrationError` (pymc/step_methods/hmc/integration.py:37-38).
"""


class SamplingError(RuntimeError):
    pass


class IntegrationError(RuntimeError):
    pass

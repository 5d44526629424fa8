"""Host-only step methods for the layout tests of `CompoundStep` sampling (tests/test_compound_layouts.py): the protocol of the
device steps (`vars`, `step`, `setup_chain`, `stop_tuning`, `reset_tuning`, `sampling_state`, `stats_dtypes_shapes`) without a GPU.
Importable by name so that worker processes (`mp_ctx="spawn"`) can unpickle them.  TEST INFRASTRUCTURE."""
import numpy as np


class _Base:
    stats_dtypes_shapes: dict = {}
    tune = True

    def setup_chain(self, rng, tune, draws):
        self.rng = rng

    def stop_tuning(self):
        self.tune = False

    def reset_tuning(self):
        self.count = 0

    @property
    def sampling_state(self):
        return {"count": 0}

    @sampling_state.setter
    def sampling_state(self, st):
        self.count = st["count"]

    def close(self):
        pass


class StubContinuous(_Base):
    """A random walk on `x` whose step depends on the discrete variable another method updates."""

    name = "stub_continuous"
    stats_dtypes_shapes = {"tune": (bool, []), "energy": (np.float64, []), "model_logp": (np.float64, [])}

    def __init__(self):
        self.vars = ["x"]
        self.rng = np.random.default_rng(0)
        self.count = 0
        self.iter_count = 0

    def step(self, point):
        self.count += 1
        x = np.asarray(point["x"], dtype="float64") + (0.5 if self.tune else 0.1) * self.rng.normal(size=3) + 0.01 * float(np.sum(point["c"]))
        new = dict(point)
        new["x"] = x
        return new, [{"tune": self.tune, "energy": 0.5 * float(x @ x), "model_logp": -0.5 * float(x @ x)}]


class StubDiscrete(_Base):
    name = "stub_discrete"

    def __init__(self):
        self.vars = ["c"]
        self.rng = np.random.default_rng(0)
        self.count = 0

    def step(self, point):
        c = (np.asarray(point["c"]) + self.rng.integers(0, 3, size=5) + int(np.sign(point["x"][0]))) % 3
        new = dict(point)
        new["c"] = c.astype(np.asarray(point["c"]).dtype)
        return new, [{}]

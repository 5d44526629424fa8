"""One drawn model of tests/test_gpu_fuzz.py sampled by the device and by the oracle's sampler, transition by transition (which statistic
leaves the oracle's first, and by how much the floating-point ones differed before that).  usage: python tools/fuzz_case_trace.py <case> | rows:<case> | mix:<case>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf  # noqa: E402
from oracle import ref_models, ref_sampler  # noqa: E402
from pymc_amd.sampling import sample  # noqa: E402

arg = sys.argv[1] if len(sys.argv) > 1 else "81"
if arg.startswith("rows:"):        # a drawn model around the logit rows (tests/test_gpu_rows_fuzz.py); NUTS_* variables select the pass
    import test_gpu_rows_fuzz as tr  # noqa: E402

    spec, _shape, env, desc = tr.rows_fuzz_model(int(arg[5:]))
    os.environ.update(env)
    tune, draws, seed = 12, 5, 7
elif arg.startswith("mix:"):       # a drawn model around the mixture node (tests/test_gpu_mixture_fuzz.py)
    import test_gpu_mixture_fuzz as tm  # noqa: E402

    spec, desc = tm.mixture_fuzz_model(int(arg[4:]))
    tune, draws, seed = 12, 4, 5
else:
    spec, desc = tf.fuzz_model(int(arg))
    tune, draws, seed = 10, 3, 5
print(desc)
res = sample(draws=draws, tune=tune, chains=1, model=spec, init="adapt_diag", random_seed=seed, device=0)
from pymc_amd.sampling import initial_point  # noqa: E402

_pt = initial_point(spec)
_, ref = ref_sampler.sample_reference(ref_models.SpecLogpGrad(spec), [np.concatenate([np.ravel(_pt[v.value_name]) for v in spec.vars])], draws=draws, tune=tune, random_seed=seed, init="adapt_diag")
got = res["warmup_stats"][0] + res["stats"][0]
res["step"].close()
for i, (a, b) in enumerate(zip(got, ref[0])):
    print(i, "dev", int(a["tree_size"]), int(a["depth"]), bool(a["diverging"]), f"{float(a['step_size']):.15e} {float(a['energy']):.15e} {float(a['mean_tree_accept']):.12f} {float(a['energy_error']):.6e}")
    print(i, "ref", int(b["tree_size"]), int(b["depth"]), bool(b["diverging"]), f"{float(b['step_size']):.15e} {float(b['energy']):.15e} {float(b['mean_tree_accept']):.12f} {float(b['energy_error']):.6e}")

import os, sys, time, numpy as np, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "mpc-interface_amd")); sys.path.insert(0, ROOT)
from mpcasm import problems
api = problems.load_api("mpc_interface")
conf = problems.BipedConfig(step_samples=8)
form = problems.biped(api, conf)
n = conf.step_samples
rng = np.random.default_rng(0)
def tick(k):
    phi = k % n
    times = np.array([(i + 1) * n - 1 - phi for i in range(conf.num_steps)])
    form.update(step_times=times, step_count=k // n)
    given = rng.normal(0, 0.1, [form.given_len, 1])
    return form.generate_all_qp_matrices(given)
for k in range(2 * n):
    tick(k)
pr = cProfile.Profile(); pr.enable()
for k in range(256):
    tick(k)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)

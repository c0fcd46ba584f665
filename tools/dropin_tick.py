import os, sys, time, numpy as np
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
t0 = time.perf_counter()
K = 64
for k in range(K):
    tick(k)
print("drop-in tick (update + generate_all_qp_matrices, B=1): %.2f ms" % ((time.perf_counter() - t0) / K * 1e3))

"""Where do the 256-tile and 128-tile paths of one dense-input step differ?  (diagnostic, GPU)
A: everything on the 256-tile kernels; B: everything on the 128-tile kernels; C: plan built for the 256-tile kernels (same K slices),
launches routed to the 128-tile kernels."""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from dae_rnn_news_recommendation_amd import _lib as L
from dae_rnn_news_recommendation_amd.engine import Engine
lib = L.load()
rng = np.random.default_rng(5)
N, F, H, B = 1000, 25000, 1000, 896
x = (rng.random((N, F)) < 0.01).astype(np.float32) * rng.random((N, F)).astype(np.float32)
lab = rng.integers(0, 4, N).astype(np.int32)
W0 = rng.uniform(-0.02, 0.02, (F, H)).astype(np.float32)


def run(plan_mode, launch_mode, triplet="batch_all"):
    lib.dae_set_glds(plan_mode)
    eng = Engine(F, H, B, dtype="bf16", opt="gradient_descent", learning_rate=0.05, triplet=triplet, loss_func="mean_squared",
                 dec_act="none", enc_act="sigmoid")
    eng.upload_dense(x); eng.set_params(W0)
    lib.dae_set_glds(launch_mode)
    stats = torch.zeros(8, device="cuda")
    idx = np.arange(B) % N
    eng.train_step(torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(lab[idx]).cuda(), stats, phase=0,
                   corr_mode=L.CORR_PHILOX_MASK, seed=3, rng_stream=0, corr_frac=0.3)
    torch.cuda.synchronize()
    g = eng.grads()
    lib.dae_set_glds(-7)
    return stats.cpu().numpy(), g, eng.info()["encode_splits"]


def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def report(name, ga, gb):
    d = (ga[0].astype(np.float64) - gb[0].astype(np.float64))
    e_rows = (d ** 2).sum(1); e_cols = (d ** 2).sum(0)
    tot = e_rows.sum()
    print(name, "rel dW %.3e dbh %.3e dbv %.3e" % (rel(ga[0], gb[0]), rel(ga[1], gb[1]), rel(ga[2], gb[2])))
    if tot > 0:
        top = np.argsort(e_cols)[::-1][:8]
        print("   top columns", [(int(c), "%.2e" % (e_cols[c] / tot)) for c in top])
        topr = np.argsort(e_rows)[::-1][:5]
        print("   top rows", [(int(r), "%.2e" % (e_rows[r] / tot)) for r in topr])
        print("   entries with |d| > 1e-3 max|g|: %d of %d" % ((np.abs(d) > 1e-3 * np.abs(gb[0]).max()).sum(), d.size))


for trip in ("batch_all", "none"):
    print("triplet", trip)
    sA, gA, kA = run(-7, -7, trip)
    sB, gB, kB = run(-6, -6, trip)
    sC, gC, kC = run(-7, -6, trip)
    print("splits", kA, kB, kC)
    print("stats A", sA[:5], "\nstats B", sB[:5], "\nstats C", sC[:5])
    report("A vs B", gA, gB)
    report("A vs C", gA, gC)
    report("C vs B", gC, gB)


print("determinism: the test's sequence, two steps each")


def run2(mode):
    lib.dae_set_glds(mode)
    eng = Engine(F, H, B, dtype="bf16", opt="gradient_descent", learning_rate=0.05, triplet="batch_all", loss_func="mean_squared",
                 dec_act="none", enc_act="sigmoid")
    eng.upload_dense(x); eng.set_params(W0)
    stats = torch.zeros((2, 8), device="cuda")
    g = []
    for s_ in range(2):
        idx = np.arange(s_ * 100, s_ * 100 + B) % N
        eng.train_step(torch.from_numpy(idx.astype(np.int32)).cuda(), torch.from_numpy(lab[idx]).cuda(), stats[s_], phase=0,
                       corr_mode=L.CORR_PHILOX_MASK, seed=3, rng_stream=s_, corr_frac=0.3)
        g.append(np.array(eng.grads()[0], copy=True))
    torch.cuda.synchronize()
    lib.dae_set_glds(-7)
    return g


runs = {k: run2(m) for k, m in (("A1", -7), ("A2", -7), ("B1", -6), ("B2", -6), ("A3", -7), ("B3", -6))}
for a, b in (("A1", "A2"), ("A1", "A3"), ("B1", "B2"), ("B1", "B3"), ("A1", "B1"), ("A3", "B3")):
    print(a, b, "step1 %.3e step2 %.3e" % (rel(runs[a][0], runs[b][0]), rel(runs[a][1], runs[b][1])))

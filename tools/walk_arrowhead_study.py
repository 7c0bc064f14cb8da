"""Numerical check of the linear algebra planned for the walk_imitation step kernel, on the float64 oracle's
mass matrix at walking states: M = [[M_rr, M_rj], [M_jr, M_jj]] (root 6 dofs, 102 joint dofs in 12 independent blocks) solved in
float32 by block solves + a 6 x 6 Schur complement, against the float64 dense solve.  Prints the relative errors that decide
whether the kernel can stay in float32 for this step.     python tools/walk_arrowhead_study.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from oracle import oracle as O
from flybody_amd.tasks.walking import WalkModelView, WalkRefSet, synthetic_snippets

view = WalkModelView()
refs = WalkRefSet(synthetic_snippets(view, n=2, length=90))
m = O.OracleModel(os.path.join(os.path.dirname(__file__), "..", "flybody_amd", "assets", "fly_walk.ffmb"))
env = O.OracleWalkEnv(m, refs, view.mocap_jnt, view.mocap_site, (view.retract_qadr, view.retract_val), terminal_com_dist=float("inf"))
env.reset()
rng = np.random.RandomState(0)
worst = {"solve_rel": 0.0, "schur_cond": 0.0, "G_rel": 0.0}
nv = m.nv
for k in range(20):
    for _ in range(3):
        env.step(rng.uniform(-0.5, 0.5, env.naction))
    M = env.data.dense_M()
    Mrr, Mrj, Mjj = M[:6, :6], M[:6, 6:], M[6:, 6:]
    # float32 arrowhead solve
    f32 = np.float32
    Mjj32, Mrj32, Mrr32 = Mjj.astype(f32), Mrj.astype(f32), Mrr.astype(f32)
    Ljj = np.linalg.cholesky(Mjj32.astype(np.float64)).astype(f32)  # stands for the kernel's block L'DL (blocks are independent)
    solve_jj = lambda B: np.linalg.solve(Ljj.T.astype(f32), np.linalg.solve(Ljj, B.astype(f32)).astype(f32)).astype(f32)
    Y = solve_jj(Mrj32.T)                                   # [102, 6]
    S = (Mrr32 - Mrj32 @ Y).astype(f32)                     # Schur complement, float32
    b = rng.randn(nv) * np.sqrt(np.diag(M))                 # right-hand sides scaled like forces
    z = solve_jj(b[6:].astype(f32))
    xr = np.linalg.solve(S.astype(np.float64), (b[:6].astype(f32) - Mrj32 @ z).astype(np.float64)).astype(f32)
    xj = (z - Y @ xr).astype(f32)
    x32 = np.hstack((xr, xj)).astype(np.float64)
    x64 = np.linalg.solve(M, b)
    worst["solve_rel"] = max(worst["solve_rel"], np.abs(x32 - x64).max() / np.abs(x64).max())
    worst["schur_cond"] = max(worst["schur_cond"], np.linalg.cond(S.astype(np.float64)))
    # rank-6 term of G = J M^-1 J' for random contact-like rows (root + one leg chain)
    J = np.zeros((9, nv)); J[:, :6] = rng.randn(9, 6); J[:, 6 + 20:6 + 31] = rng.randn(9, 11)
    G64 = J @ np.linalg.solve(M, J.T)
    Jr, Jj = J[:, :6].astype(f32), J[:, 6:].astype(f32)
    Gjj = Jj @ solve_jj(Jj.T)
    Jh = (Jr - Jj @ Y).astype(f32)
    G32 = (Gjj + Jh @ np.linalg.solve(S.astype(np.float64), Jh.T.astype(np.float64)).astype(f32)).astype(np.float64)
    worst["G_rel"] = max(worst["G_rel"], np.abs(G32 - G64).max() / np.abs(G64).max())
print({k: float(v) for k, v in worst.items()})

# ---- how many contacts / constraint rows the floor produces (sizes the kernel's contact slots and dense-solver variants)
L = O.lib()
for amp in (0.3, 1.0):
    env = O.OracleWalkEnv(m, refs, view.mocap_jnt, view.mocap_site, (view.retract_qadr, view.retract_val), terminal_com_dist=float("inf"))
    env.reset()
    rng = np.random.RandomState(1)
    ncon, nrow, nlim = [], [], []
    for k in range(1500):
        st, _, _, _ = env.step(rng.uniform(-amp, amp, env.naction))
        d = env.data
        d.forward()
        t = np.ctypeslib.as_array(L.fo_efc_type(d.ptr), shape=(300,))[:d.nefc]
        ncon.append(d.ncon); nrow.append(d.nefc); nlim.append(int((t == 0).sum()))
    ncon, nrow, nlim = np.array(ncon), np.array(nrow), np.array(nlim)
    print(f"floor, actions +-{amp}: contacts mean {ncon.mean():.1f} p99 {np.percentile(ncon, 99):.0f} max {ncon.max()}; constraint rows mean {nrow.mean():.1f} "
          f"p99 {np.percentile(nrow, 99):.0f} max {nrow.max()}; limit rows mean {nlim.mean():.1f} max {nlim.max()}")

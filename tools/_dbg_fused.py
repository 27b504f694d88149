import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from dust_amd import Context
from oracle import grid_4x4_map
def run(model,N,S,M,H,ticks=1,iters=1):
    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(3)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + rng.standard_normal((N, H, da))).astype(np.float32)
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    up = None if M == 1 else (("mass",) if da==2 else ("length","mass"))
    P = 0 if up is None else len(up)
    params = None if M == 1 else (1.0 + 0.1 * rng.standard_normal((iters, M, P))).astype(np.float32)
    grid = grid_4x4_map() if model=="particle" else None
    out=[]
    for unfused in (False, True):
        c = Context(model=model, N=N, S=S, M=M, H=H, kernel="K1", lr=0.5, sigma_a=1.0, sigma_p=1.0, uncertain_params=up, grid=grid, seed=11)
        c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
        c.profile(unfused)
        for _ in range(ticks):
            c.svmpc_optimize(state, iters, params=params)
        c.sync()
        out.append(dict(theta=c.get_theta(), score=c.get_score(), a_mat=c.get_a_mat(), costs=c.get_costs(), phi=c.get_phi()))
        c.close()
    for k in out[0]:
        d=np.abs(out[0][k]-out[1][k]); print(model,N,S,M,H,k, "maxdiff", float(d.max()), "n_diff", int((d>0).sum()), "of", d.size)
run("particle",256,64,4,20)
run("particle",256,64,1,20)
run("pendulum",256,64,4,20)
def cols(model,N,S,M,H):
    da=2; rng=np.random.default_rng(3)
    mu = rng.standard_normal((N, H, da)).astype(np.float32); th = (mu + rng.standard_normal((N, H, da))).astype(np.float32)
    state=np.array([-9.0,-9.0,0,0],np.float32); out=[]
    for unfused in (False,True):
        c = Context(model=model, N=N, S=S, M=M, H=H, kernel="K1", lr=0.5, sigma_a=1.0, sigma_p=1.0, grid=grid_4x4_map(), seed=11)
        c.set_theta(th); c.set_prior(mu); c.set_a_mat(th); c.profile(unfused)
        c.svmpc_optimize(state,1); c.sync(); out.append(c.get_score().reshape(N,-1)); c.close()
    d=np.abs(out[0]-out[1]); print("cols differing:", np.unique(np.nonzero(d)[1]), "rows", len(np.unique(np.nonzero(d)[0])))
    print(out[0][0,28:40]); print(out[1][0,28:40])
cols("particle",256,64,1,20)
cols("particle",256,64,1,16)
cols("particle",256,64,1,17)

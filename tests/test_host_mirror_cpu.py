"""Host-side logic of the reference-API mirror that needs no GPU: spaces, obstacle maps, cost recognition, kernel
selection, plant-side model steps (against the reference's own golden states)."""
import numpy as np
import pytest
import torch

from dust_amd.costs import PendulumQuadCos, recognise
from dust_amd.kernels import IMQ, RBF, RBFKernel, iid_mp, kernel_config
from dust_amd.models import Particle, PendulumModel
from dust_amd.utils.obstacle_map import generate_obstacle_map, get_obst_preset
from dust_amd.utils.spaces import Box

PARTICLE_ENV = dict(dt=0.015, control_type="acceleration", noise_std=[0.1, 0.1], init_state=[-9.0, -9.0, 0, 0],
                    target_state=[9.0, 9.0, 0, 0], can_crash=True, with_obstacle=True, deterministic=True,
                    cost_params=dict(w_qpos=0.5, w_qvel=0.25, w_ctrl=0.2, w_obs=1.0e6, w_qpos_T=1.0e3, w_qvel_T=0.1),
                    obst_preset="grid_4x4", obst_width=2.1, max_speed=5, max_accel=10, map_cell_size=0.1, map_size=[22, 22],
                    map_type="direct")


def test_box():
    b = Box(dim=2, low=-2.0, high=torch.tensor([1.0, 3.0]))
    assert b.dim == 2 and b.shape == torch.Size([2]) and b.low.tolist() == [-2.0, -2.0] and b.high.tolist() == [1.0, 3.0]


def test_obstacle_presets_match_reference(golden):
    g = golden("maps")
    for preset, w in (("grid_4x4", 2.1), ("grid_3x3", 2.0), ("staggered_3-2-3", 2.0), ("staggered_4-3-4-3-4", 1.5), ("grid_6x6", 1.2),
                      ("single_centred", 3.0)):
        m = generate_obstacle_map([22, 22], get_obst_preset(preset, w), 0.1, map_type="direct")
        ref = np.unpackbits(g["map_%s_w%s" % (preset, str(w).replace(".", "p"))])[: 220 * 220].reshape(220, 220)
        assert np.array_equal(ref, m.map.astype(np.uint8)), preset
    m = generate_obstacle_map([22, 22], get_obst_preset("grid_4x4", 2.1), 0.1, map_type="direct")
    assert np.array_equal(m.get_collisions(torch.tensor(g["points"])).numpy(), g["collisions"])
    with pytest.raises(IOError):
        get_obst_preset("nope")


def test_cost_recognition():
    def inst_cost(states, controls=None, n_pol=1, debug=None):  # demo/pendulum_example.py:21-24, verbatim semantics
        theta, theta_d = states.chunk(2, dim=1)
        return 50.0 * (theta.cos() - 1) ** 2 + 1.0 * theta_d ** 2

    def term_cost(states, n_pol=1, debug=None):
        return inst_cost(states).squeeze()

    assert recognise(PendulumModel(), inst_cost, term_cost) == dict(w_cos=50.0, w_vel=1.0)
    q = PendulumQuadCos(12.5, 0.25)
    assert recognise(PendulumModel(), q.inst_cost, q.term_cost) == dict(w_cos=12.5, w_vel=0.25)
    with pytest.raises(NotImplementedError):  # an opaque callable is rejected, never run on the CPU
        recognise(PendulumModel(), lambda s, c=None, **k: s.abs().sum(1, keepdim=True), term_cost)
    with pytest.raises(NotImplementedError):
        recognise(PendulumModel(), inst_cost, lambda s, **k: 2 * inst_cost(s).squeeze())
    p = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    r = recognise(p, p.default_inst_cost, p.default_term_cost)
    assert r["w_obs"] == 1e6 and r["w_term"][0] == 1e3 and r["target"] == (9.0, 9.0, 0.0, 0.0)
    with pytest.raises(NotImplementedError):
        recognise(p, inst_cost, term_cost)


def test_kernel_config():
    assert kernel_config(RBFKernel()) == dict(kernel="K1")
    assert kernel_config(None) == dict(kernel="K1")
    assert kernel_config(iid_mp(RBF(bandwidth=-1), ctrl_dim=2, indep_controls=True))["kernel"] == "K2"
    assert kernel_config(iid_mp(RBF(bandwidth=-1), ctrl_dim=2, indep_controls=False))["kernel"] == "K2shared"
    assert kernel_config(IMQ(0.7)) == dict(kernel="IMQ", imq_ell=0.7)
    # RBF(minimum_bw=) travels with the median trick as well as with a fixed bandwidth (base_kernels.py:44, 83-89)
    kc = kernel_config(iid_mp(RBF(bandwidth=-1, minimum_bw=1.4), ctrl_dim=1, indep_controls=True))
    assert kc["k2_minimum_bw"] == 1.4 and kc["k2_bandwidth"] == -1
    with pytest.raises(ValueError):
        kernel_config(RBF())


def test_plant_steps_match_reference_rollouts(golden):
    """model.step on the host (the plant of a closed-loop driver) against states the reference rolled out."""
    g = golden("pend_k1")
    m = PendulumModel()
    st, ac = g["states_iter0"][0][0, 0], g["actions"][0, 0][0]  # [N,H+1,2], [N,H,1]
    x = torch.tensor(st[:, 0])
    for t in range(ac.shape[1]):
        x = m.step(x, torch.tensor(ac[:, t]))
        assert np.allclose(x.numpy(), st[:, t + 1], rtol=1e-6, atol=1e-6)
    g = golden("part_k1_near_obst")
    p = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    st, ac, prm = g["states_iter0"][0][1, 3], g["actions"][0, 0][3], np.exp(g["params"][0, 0][1])
    x = torch.tensor(st[:, 0])
    for t in range(ac.shape[1]):
        x = p.step(x, torch.tensor(ac[:, t]), {"mass": torch.full((x.shape[0], 1), float(prm[0]))})
        assert np.allclose(x.numpy(), st[:, t + 1], rtol=1e-5, atol=1e-5)
    assert p.params_to_dict(torch.tensor([[1.0], [2.0]]))["mass"].shape == (2, 1)
    # velocity control (particle.py:41-48, 152-153, 165): a two-state model whose closing clamp lands on the positions
    g = golden("part_k1_velocity")
    env = dict(PARTICLE_ENV, control_type="velocity", init_state=[-4.0, -3.0], target_state=[4.0, 4.5])
    pv = Particle(**env, uncertain_params=["mass"], mass=torch.tensor(2.0))
    assert pv.observation_space.dim == 2 and pv.w_state.tolist() == [0.5, 0.5] and pv.w_term.tolist() == [1e3, 1e3]
    assert pv.action_space.high.tolist() == [5.0, 5.0]
    st, ac = g["states_iter0"][0][1, 3], g["actions"][0, 0][3]  # [N,H+1,2], [N,H,2]
    x = torch.tensor(st[:, 0])
    for t in range(ac.shape[1]):
        x = pv.step(x, torch.tensor(ac[:, t]))
        assert np.allclose(x.numpy(), st[:, t + 1], rtol=1e-6, atol=1e-6)
    assert float(np.abs(st).max()) <= 5.0
    # control-channel noise on the plant side (particle.py:145-148): the recorded draw of the reference's own plant step
    g = golden("part_k1_noisy")
    pn = Particle(**dict(PARTICLE_ENV, deterministic=False, noise_std=torch.tensor(g["dyn_std"])), uncertain_params=["mass"], mass=torch.tensor(2.0))
    x0, a0 = torch.tensor(g["state"][0, 0]).view(1, -1), torch.tensor(g["tick_a_seq"][0][0]).view(1, -1)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: torch.tensor(g["plant_noise"][0])
    try:
        nxt = pn.step(x0, a0)
    finally:
        torch.randn_like = orig
    assert np.allclose(nxt.numpy().reshape(-1), g["state"][1, 0], rtol=1e-6, atol=1e-6)
    with pytest.raises(IOError):
        Particle(**dict(PARTICLE_ENV, control_type="torque"))


def test_skid_steer_plant_step_matches_reference(golden):
    """SkidSteerRobot.step (skid_steer_robot.py:73-122) - plant-side only: default parameters, per-row parameter
    columns (params_dict), other bounds and time step; and no rollout kernel family exists for it."""
    from dust_amd.controllers import MultiDISCO
    from dust_amd.models import SkidSteerRobot

    g = golden("skid_steer")
    m = SkidSteerRobot(delta_t=0.1)
    assert m.observation_space.dim == 5 and m.action_space.dim == 2 and m.action_space.high.tolist() == [0.5, 0.5]
    st, ac = torch.tensor(g["states"]), torch.tensor(g["actions"])
    assert np.allclose(m.step(st, ac).numpy(), g["next"], rtol=1e-6, atol=1e-7)
    pd = {k: torch.tensor(g[k]) for k in ("x_icr", "wheel_radius", "axial_distance")}
    assert np.allclose(m.step(st, ac, pd).numpy(), g["next_params"], rtol=1e-6, atol=1e-7)
    m2 = SkidSteerRobot(delta_t=0.05, x_icr=0.1, wheel_radius=0.08, axial_distance=0.5, min_wheel_speed=-1.0, max_wheel_speed=0.8)
    assert np.allclose(m2.step(st, ac * 2).numpy(), g["next2"], rtol=1e-6, atol=1e-7)
    assert m.params_dict == {"x_icr": 0.2, "wheel_radius": 0.0625, "axial_distance": 0.475}

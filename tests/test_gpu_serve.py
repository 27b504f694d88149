"""Closed-loop serving (dust_svmpc_serve_start / _stop, dust_amd/csrc/tick2.hpp armed launches): the loop of
dust/utils/simulations.py:104-123 - optimize, forward, first action to the plant, the new state to the next tick - with the outputs
coming back through pinned host memory and the next tick launched AHEAD of its plant state.  The served loop must give the results of
the same calls without serving, bit for bit; a launched-ahead tick that is cancelled, or whose state never comes, leaves no trace."""
import time

import numpy as np
import pytest

from test_gpu_tick2 import _make, _state

pytestmark = pytest.mark.gpu


def _plant(model, st, a):
    """a host plant: any deterministic map (state, action) -> state will do for the comparison"""
    if model == "pendulum":
        thd = np.float32(np.clip(st[1] + 0.05 * (14.7 * np.sin(st[0]) + 3.0 * np.clip(a[0], -2, 2)), -8, 8))
        return np.array([st[0] + thd * 0.05, thd], np.float32)
    v = np.clip(st[2:] + 0.015 * np.clip(a / 2.0, -10, 10), -5, 5)
    return np.concatenate([st[:2] + 0.015 * st[2:], v]).astype(np.float32)


def _loop(c, model, iters, ticks, serve, wait_us=20000.0):
    st = _state(model)
    c.svmpc_tick(st, 1)  # the first tick of a context: afterwards the prior means alias the particles (svgd.py:87)
    if serve:
        c.serve_start(iters, wait_us)
    out = []
    for t in range(ticks):
        a_seq, pw = c.svmpc_tick(st, iters)
        out.append((a_seq.copy(), pw.copy(), st.copy()))
        st = _plant(model, st, a_seq[0])
    if serve:
        c.serve_stop()
    return out, c.get_theta(), c.get_a_mat(), c.get_costs(), c.tick_stats()


SHAPES = [("pendulum", 1024, 128, 30, 5, {}),                    # BASELINE configs[1]: the shape bench.py's closed loop runs
          ("pendulum", 64, 128, 15, 1, {}),                      # ONE iteration per tick (costs stored behind the go word)
          ("pendulum", 96, 64, 20, 3, dict(optimizer="Adam")),
          ("pendulum", 256, 128, 30, 2, dict(weighted_prior=True)),
          ("particle", 64, 64, 12, 2, {}),                       # four-entry state: both halves of the mailbox
          ("particle", 128, 64, 16, 1, dict(kernel="IMQ"))]


@pytest.mark.parametrize("model,N,S,H,iters,kw", SHAPES)
def test_served_loop_is_bit_identical_to_per_launch_ticks(model, N, S, H, iters, kw):
    """'results bit-identical to the per-launch tick2 path'.  40 closed-loop ticks with device (Philox) noise, the
    plant stepped on the host between ticks: every a_seq, every particle weight, the final particles / a_mat / costs."""
    a, _ = _make(model, N, S, H, **kw)  # (one context on the device at a time: a second tenant switches launching ahead off)
    ra, tha, ama, ca, sa = _loop(a, model, iters, 40, serve=False)
    a.close()
    b, _ = _make(model, N, S, H, **kw)
    rb, thb, amb, cb, sb = _loop(b, model, iters, 40, serve=True)
    b.close()
    for t, ((a0, p0, s0), (a1, p1, s1)) in enumerate(zip(ra, rb)):
        assert np.array_equal(s0, s1), t
        assert np.array_equal(a0, a1), (t, np.abs(a0 - a1).max())
        assert np.array_equal(p0, p1), t
    assert np.array_equal(tha, thb) and np.array_equal(ama, amb) and np.array_equal(ca, cb)
    assert sa["served"] == 0 and sa["replayed"] == 0
    # every served tick came back through the done word; all but the first had been launched ahead (+ the one cancelled by serve_stop)
    assert sb["served"] == 40 and sb["replayed"] == 0, sb
    assert sb["tick2"] == 41, sb


def _script(c, model, iters, serve):
    """12 closed-loop ticks with getters, a setter and a clone in between"""
    st = _state(model)
    c.svmpc_tick(st, 1)
    if serve:
        c.serve_start(iters, 20000.0)
    log = []
    for t in range(12):
        r = c.svmpc_tick(st, iters)
        log.append((r[0].copy(), r[1].copy()))
        if t % 3 == 0:  # getters between two served ticks (a launch is waiting for its state at this point)
            log.append((c.get_theta(), c.get_costs(), c.get_a_mat()))
        if t % 4 == 1:  # a setter
            c.set_theta((c.get_theta() * np.float32(0.5)).astype(np.float32))
        if t == 7:  # a clone taken in the middle continues identically (and the original keeps being served)
            c2 = c.clone()
            r2 = c2.svmpc_tick(_plant(model, st, r[0][0]), iters)
            log.append((r2[0].copy(), r2[1].copy()))
            c2.close()
        st = _plant(model, st, r[0][0])
    stats = c.tick_stats()
    if serve:
        c.serve_stop()
    log.append((c.get_theta(),))
    return log, stats


def test_other_calls_cancel_the_armed_launch_without_a_trace():
    """Every other entry point settles first: the launch that waits for its state is cancelled, nothing it did is visible (particles,
    a_mat, costs, stream position), and the loop goes on - served again - with the results of the unserved loop."""
    model, N, S, H, iters = "pendulum", 256, 128, 30, 2
    a, _ = _make(model, N, S, H)
    la, sa = _script(a, model, iters, False)
    a.close()
    b, _ = _make(model, N, S, H)
    lb, sb = _script(b, model, iters, True)
    b.close()
    assert len(la) == len(lb)
    for i, (x, y) in enumerate(zip(la, lb)):
        for u, v in zip(x, y):
            assert np.array_equal(u, v), i
    assert sb["served"] == 12 and sb["replayed"] == 0, sb
    assert sb["tick2"] > sa["tick2"] + 6, (sa, sb)  # launches ahead happened (and those cancelled by the other calls are counted too)


def test_state_that_never_comes_and_late_states():
    """The launched-ahead tick waits a bounded time: (a) a caller that stops calling leaves a launch that gives up by itself - the
    context is usable afterwards, nothing changed; (b) a loop slower than the bound gets its ticks late but right (the launch that
    gave up is replayed on the launch-per-iteration path: equal to 2e-3, not bitwise - another summation order), and after three
    misses in a row the context stops launching ahead."""
    model, N, S, H, iters = "pendulum", 128, 128, 20, 2
    st = _state(model)

    def run(serve):
        c, _ = _make(model, N, S, H)
        c.svmpc_tick(st, 1)
        if serve:
            c.serve_start(iters, 2000.0)  # 2 ms
        out = [c.svmpc_tick(st, iters)]
        if serve:
            time.sleep(0.05)  # the armed launch has given up by now
        out.append((c.get_theta(), c.get_a_mat()))
        return c, out

    a, oa = run(False)
    a.close()
    b, ob = run(True)
    assert np.array_equal(oa[0][0], ob[0][0]) and np.array_equal(oa[1][0], ob[1][0]) and np.array_equal(oa[1][1], ob[1][1])
    n0 = b.tick_stats()["tick2"]
    for t in range(8):
        a_seq, pw = b.svmpc_tick(st, iters)
        assert np.isfinite(a_seq).all() and abs(float(pw.sum()) - 1.0) < 1e-3
        time.sleep(0.01)  # a plant that takes 10 ms: slower than the bound
    sb = b.tick_stats()
    assert 1 <= sb["replayed"] <= 3, sb   # the states that came too late found launches that had given up: replayed, then no more arming
    n1 = sb["tick2"]
    for t in range(3):
        b.svmpc_tick(st, iters)
    assert b.tick_stats()["tick2"] == n1 + 3 and n1 - n0 < 16  # plain served ticks now: one launch per tick
    b.serve_stop()
    b.close()
    # the late ticks are RIGHT: the same loop without serving, compared tick by tick up to the first replayed tick's tolerance
    a, _ = _make(model, N, S, H)
    b, _ = _make(model, N, S, H)
    a.svmpc_tick(st, 1)
    b.svmpc_tick(st, 1)
    a.close()
    b.serve_start(iters, 1000.0)
    b.svmpc_tick(st, iters)
    time.sleep(0.02)
    th_before = b.get_theta()          # (cancels / settles: nothing pending)
    r_late = b.svmpc_tick(st, iters)   # not armed: a plain served tick
    b.serve_stop()
    c, _ = _make(model, N, S, H)
    c.svmpc_tick(st, 1)
    c.svmpc_tick(st, iters)
    assert np.array_equal(c.get_theta(), th_before)
    r_ref = c.svmpc_tick(st, iters)
    assert np.array_equal(r_ref[0], r_late[0])
    b.close()
    c.close()


def test_serving_next_to_the_dynamics_filter_and_a_second_context():
    """The dual loop (simulations.py:104-138): a filter update between two control ticks needs the device - it cancels the waiting
    launch instead of standing behind it; so does a context created meanwhile.  Results stay those of the unserved loop."""
    from dust_amd import MpfContext

    model, N, S, H, iters = "pendulum", 128, 128, 20, 2
    rng = np.random.default_rng(3)
    x0 = rng.uniform(0.6, 1.3, (32, 2)).astype(np.float32)

    def run(serve):
        c, _ = _make(model, N, S, H)
        st = _state(model)
        c.svmpc_tick(st, 1)
        mpf = MpfContext(x0, st, model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-3, init_bw=0.1)
        if serve:
            c.serve_start(iters, 40000.0)
        out = []
        t0 = time.perf_counter()
        for t in range(6):
            r = c.svmpc_tick(st, iters)
            nxt = _plant(model, st, r[0][0])
            gn = mpf.optimize(r[0][0], nxt, 0.1, 5)
            out.append((r[0].copy(), r[1].copy(), gn.copy()))
            st = nxt
        el = time.perf_counter() - t0
        if serve:
            c3, _ = _make(model, 64, 64, 10)  # a second context on the device while a launch is armed
            c3.svmpc_tick(_state(model), 1)
            r = c.svmpc_tick(st, iters)
            out.append((r[0].copy(), r[1].copy(), np.zeros(1)))
            c3.close()
            c.serve_stop()
        else:
            r = c.svmpc_tick(st, iters)
            out.append((r[0].copy(), r[1].copy(), np.zeros(1)))
        stats = c.tick_stats()
        c.close()
        mpf.close()
        return out, el, stats

    oa, _, _ = run(False)
    ob, el, sb = run(True)
    for t, (x, y) in enumerate(zip(oa, ob)):
        for u, v in zip(x, y):
            assert np.array_equal(u, v), t
    assert el < 0.15, "the filter must not wait for the armed launch's 40 ms bound (6 ticks took %.3f s)" % el
    assert sb["served"] == 7 and sb["replayed"] == 0, sb


def test_cancel_from_another_thread_during_a_served_loop():
    """dust_create (and the filter's calls) cancel the armed launch from WHATEVER thread they run on, while the owner thread
    may be spinning on its tick's done word.  The cancelled launch must stay on record until dust_sync has taken the device's abort
    count for it - otherwise the next launch aborts on a count the host never saw and the settle fails with 'n reported, n - 1 on
    record'.  A second thread creates and destroys contexts and runs the filter as fast as it can beside 120 served ticks."""
    import threading

    from dust_amd import MpfContext

    model, N, S, H, iters = "pendulum", 128, 128, 20, 2
    rng = np.random.default_rng(4)
    x0 = rng.uniform(0.6, 1.3, (32, 2)).astype(np.float32)

    def run(serve, disturb):
        c, _ = _make(model, N, S, H)
        st = _state(model)
        c.svmpc_tick(st, 1)
        stop = threading.Event()
        errs = []

        def other():
            try:
                mpf = MpfContext(x0, _state(model), model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-3, init_bw=0.1)
                k = 0
                while not stop.is_set():
                    if k % 2:
                        d, _ = _make(model, 32, 64, 8)
                        d.close()
                    else:
                        mpf.optimize(np.zeros(1, np.float32), _state(model), 0.1, 2)
                    k += 1
                mpf.close()
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = threading.Thread(target=other)
        if serve:
            c.serve_start(iters, 5000.0)
        if disturb:
            th.start()
        out = []
        try:
            for t in range(120):
                r = c.svmpc_tick(st, iters)
                out.append((r[0].copy(), r[1].copy()))
                st = _plant(model, st, r[0][0])
        finally:
            stop.set()
            if disturb:
                th.join()
        if serve:
            c.serve_stop()
        theta = c.get_theta()
        stats = c.tick_stats()
        c.close()
        assert not errs, errs
        return out, theta, stats

    oa, ta, _ = run(False, False)
    ob, tb, sb = run(True, True)
    # a tick that did not start next to the other thread's kernels is replayed on the launch-per-iteration path (other summation order)
    tol = 0.0 if sb["replayed"] == 0 else 2e-3
    for t, (x, y) in enumerate(zip(oa, ob)):
        assert np.allclose(x[0], y[0], rtol=tol, atol=tol), (t, sb)
    assert np.allclose(ta, tb, rtol=tol, atol=tol * 10), sb
    assert sb["served"] + sb["replayed"] >= 100, sb


def test_serve_start_rejects_what_it_cannot_serve():
    from dust_amd import Context
    from dust_amd._lib import DustError

    c, _ = _make("pendulum", 64, 64, 12, M=2)  # sampled dynamics: per-tick parameter uploads
    with pytest.raises(DustError):
        c.serve_start(2)
    c.close()
    c, _ = _make("pendulum", 64, 64, 12, kernel="K2")
    with pytest.raises(DustError):
        c.serve_start(2)
    c.close()
    c, _ = _make("pendulum", 64, 64, 12)
    with pytest.raises(DustError):
        c.serve_start(2, wait_us=1e6)
    c.serve_start(2, wait_us=0.0)  # outputs through pinned memory only
    st = _state("pendulum")
    c.svmpc_tick(st, 1)
    for _ in range(3):
        a_seq, pw = c.svmpc_tick(st, 2)
        assert np.isfinite(a_seq).all() and abs(float(pw.sum()) - 1) < 1e-3
    assert c.tick_stats()["served"] == 3 and c.tick_stats()["tick2"] == 3
    c.close()

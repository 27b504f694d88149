"""The sharded cfg4 tick's three exchanges timed ALONE (dust_comm_probe), G ranks as G PROCESSES sharing the one GPU of the box, in both
forms: the collective library's all-gather (here tests/fake_rccl - a synchronous stand-in, NOT RCCL: its figure only shows the harness
works) and the direct peer stores of dust_amd/csrc/peer_gather.hpp.  On one GPU a "peer" is the same device: the figure is the protocol's
own cost - two launches per exchange, the token round trip of the particle buffer, the arrival words - without the xGMI transfer
(655 KB per peer per exchange at G = 8: ~13 us on a 50 GB/s link, all seven links in parallel), and with G processes' kernels
time-sharing one GPU: an upper bound of the fixed part.  Also: whole sharded ticks per second in both forms (G ranks on ONE GPU: no
speed-up expected; the ranks must agree bit for bit).
    python tools/peer_probe.py [G,G,..]          (parent)        python tools/peer_probe.py --rank r G dir   (child)"""
import os, subprocess, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def child(rank, world, d):
    import bench
    from sharded_worker import FileDist
    from dust_amd.parallel import ShardedSVMPC
    c4 = bench.CFG4
    mu, theta = bench.synth(c4["N"], c4["H"], 2, spread=1.0)
    common = dict(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                  uncertain_params=("mass",), grid=bench.particle_grid(), seed=1234)
    params = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
    st = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    fd = FileDist(d, rank, world)
    sh = ShardedSVMPC(common, rank, world, fd, c_side=True)
    sh.set_state(theta, mu)
    out = {}
    for form in ("library", "peer"):
        if form == "peer":
            sh.ctx.comm_peer_gather(True)
        for _ in range(10):
            sh.tick(st, c4["n_iters"], params=params)
        sh.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            sh.tick(st, c4["n_iters"], params=params)
        sh.sync()
        tick_us = (time.perf_counter() - t0) / 20 * 1e6
        comm = sh.ctx.comm_probe(c4["n_iters"], 30)
        th = sh.ctx.get_theta()
        sums = [None] * world
        fd.all_gather_object(sums, zlib.crc32(th.tobytes()))
        out[form] = (tick_us, comm, all(x == sums[0] for x in sums))
    if rank == 0:
        for form, (t, cm, ok) in out.items():
            print("G=%d %-8s gathers: %7.1f us per tick for the 3 exchanges alone (comm_probe); sharded tick %8.1f us; ranks agree: %s" % (world, form, cm, t, ok), flush=True)
    sh.ctx.close()


if len(sys.argv) > 1 and sys.argv[1] == "--rank":
    child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
else:
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
    if not os.path.exists(fake) or os.path.getmtime(fake) < os.path.getmtime(src):
        subprocess.run(["g++", "-shared", "-fPIC", "-O1", src, "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", fake], check=True)
    for G in ([int(g) for g in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4, 8]):
        with tempfile.TemporaryDirectory() as d:
            env = dict(os.environ, DUST_RCCL_LIB=fake, DUST_NO_FUSE="1")
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(G), d], env=env, stdout=subprocess.PIPE,
                                   stderr=subprocess.STDOUT, text=True) for r in range(G)]
            for r, p in enumerate(ps):
                o, _ = p.communicate(timeout=600)
                if r == 0 or p.returncode:
                    print(o[-1500:] if p.returncode else "\n".join(l for l in o.splitlines() if l.startswith("G=")), flush=True)
    # the protocol's fixed cost on ONE device, one process (dust_debug_peer_selftest: the "peers" are scratch buffers, a one-wave kernel
    # plays their words): the store + wait kernels of the three exchanges of a cfg4 tick at 2 / 4 / 8 ranks, the two kernels of which each
    # consists included - the figure to add the link time to
    import ctypes as C
    import bench
    from dust_amd import Context, _lib as L
    lib = L.load()
    lib.dust_debug_peer_selftest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.dust_debug_peer_selftest.restype = C.c_int
    c4 = bench.CFG4
    for G in (2, 4, 8):
        ctx = Context(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, sigma_a=1.0, sigma_p=1.0,
                      uncertain_params=("mass",), grid=bench.particle_grid(), shard_offset=0, shard_size=c4["N"] // G)
        us = C.c_double(0.0)
        L.check(lib.dust_debug_peer_selftest(ctx._h, G, c4["n_iters"], 200, C.byref(us)))
        piece = c4["N"] // G * c4["H"] * 2 * 4
        print("one device, rank 0 of %d: %.1f us per tick for the three exchanges' kernels (piece %d KB; + link time 2 x %.1f us at 50 GB/s per link)"
              % (G, us.value, piece // 1024, piece / 50e9 * 1e6), flush=True)
        ctx.close()

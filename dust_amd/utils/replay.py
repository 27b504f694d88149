"""Replay of recorded random draws (parity tests against reference-produced fixtures).

The reference draws its policy noise and dynamics samples from torch's CPU generator inside `SVMPC.optimize` /
`MultiDISCO.forward`; the build draws policy noise on the device (Philox) and dynamics samples from the filter's device GMM.
Neither stream can reproduce the other, so closed-loop fixtures (tests/golden/make_golden_driver.py) record every draw and the
drivers are re-run under `with replay.feed(eps=[...], params=[...])`: while a feed is active, `SVMPC.optimize` consumes the next
recorded noise array ([S,N,H,da] per SVGD step) and the controller the next recorded dynamics samples ([M,P]) instead of drawing."""
_feed = {"eps": None, "params": None}


class feed:
    def __init__(self, eps=None, params=None):
        self._new = {"eps": None if eps is None else iter(list(eps)), "params": None if params is None else iter(list(params))}

    def __enter__(self):
        self._old = dict(_feed)
        _feed.update(self._new)
        return self

    def __exit__(self, *exc):
        _feed.update(self._old)
        return False


def _next(kind):
    it = _feed[kind]
    if it is None:
        return None
    try:
        return next(it)
    except StopIteration:
        raise RuntimeError("replay feed for %r is exhausted" % kind)


def next_eps():
    return _next("eps")


def next_params():
    return _next("params")

"""Content stamps of the kernel sources a committed PMC summary was measured on (`roofline.traffic` is read from a file under
profiles/ - PMC passes cannot run inside bench.py's process - and went stale silently when the kernel changed).  The measuring script
writes stamp(family) into the summary; bench.py recomputes it and reports the traffic as null when the sources have changed since."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = {
    "tick2": ["tick2.hpp", "tick2_args.hpp", "tick2_reduce.hpp", "handoff.hpp", "common.hpp"],
    "states": ["rollout_states.hpp", "rollout.hpp", "common.hpp"],
}


def stamp(family):
    h = hashlib.sha256()
    for name in FAMILIES[family]:
        with open(os.path.join(ROOT, "dust_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    for k in FAMILIES:
        print(k, stamp(k))

import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import kprof
which = sys.argv[1]
if which == "ext": kprof.run("external eps  ", ext=True)
if which == "h1": kprof.run("H=1           ", H=1)
if which == "base": kprof.run("base", )
print("done", which, flush=True)

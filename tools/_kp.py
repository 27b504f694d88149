import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import kprof
which = sys.argv[1]
if which == "ext": kprof.run("external eps  ", ext=True)
if which == "base": kprof.run("base", )

"""a reference inputs file run to its max_step: python tools/probes/run_full_input.py <inputs file> [print every]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs, advance as adv
text = open(sys.argv[1]).read()
every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = time.perf_counter()
worst = [0, 0]
def rep(G):
    m, h = adv.last_solver_stats("mac"), adv.last_solver_stats("hg")
    worst[0], worst[1] = max(worst[0], m[0]), max(worst[1], h[0])
    if G.istep % every == 0:
        print("step %3d t %.5f dt %.3e boxes %s MAC %d HG %d regrids %d  %.1f s" % (G.istep, G.time, G.dt, [len(b) for b in G.boxes] if hasattr(G, "nlev") else len(G.boxes), m[0], h[0], getattr(G, "nregrids", 0),
              time.perf_counter() - t0), flush=True)
nl, G = inputs.run(text, report=rep, outdir="/tmp")
print("done: %d steps, t = %.5f, most iterations MAC %d HG %d, %.1f s" % (G.istep, G.time, worst[0], worst[1], time.perf_counter() - t0))
G.close()

"""inputs_RayleighTaylor_3d on ONE level cut into boxes (periodic x, y; the dead band of velpred is per box): does the singular MAC system stay solvable to 1e-10?"""
import sys, re, time
sys.path.insert(0, ".")
from varden_amd import inputs, advance as adv
text = open("tests/golden/inputs/inputs_RayleighTaylor_3d").read()
text = re.sub(r"max_levs\s*=\s*\d+", "max_levs = 1", text); text = re.sub(r"plot_int\s*=\s*\d+", "plot_int = 0", text); text = re.sub(r"chk_int\s*=\s*\d+", "chk_int = 0", text)
n, mgs = sys.argv[1], sys.argv[2]
for a in "xyz": text = re.sub(r"n_cell%s\s*=\s*\d+" % a, "n_cell%s = %s" % (a, n), text)
text = text.replace("&PROBIN", "&PROBIN\n max_grid_size = %s\n abort_on_max_iter = 0" % mgs)
worst = [0, 0]; t0 = time.perf_counter()
def rep(G):
    m, h = adv.last_solver_stats("mac"), adv.last_solver_stats("hg")
    worst[0], worst[1] = max(worst[0], m[0]), max(worst[1], h[0])
    if m[0] > 30: print("  step %d: MAC %s" % (G.istep, m), flush=True)
try:
    nl, G = inputs.run(text, nsteps=int(sys.argv[3]), report=rep, outdir="/tmp")
    print("n %s max_grid_size %s: %d steps, %d boxes, most V-cycles MAC %d HG %d, %.1f s" % (n, mgs, G.istep, len(G.boxes), worst[0], worst[1], time.perf_counter() - t0))
except Exception as e:
    print("n %s max_grid_size %s: FAILED %s" % (n, mgs, str(e)[-150:]))

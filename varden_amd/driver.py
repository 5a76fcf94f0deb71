"""Thin single-level driver around the hot path: what reference src/varden.f90 does around
``advance_timestep`` (initial projection 126-138, ghost fills 165-178 / 291-300, estdt 186-199 /
302-318, initial pressure iterations 460-490, the step loop 237-345), with every array in HBM.

The driver is NOT the product of this round (SURVEY.md section 8 marks it "next"); it exists so that the
hot path can be exercised and timed exactly the way the reference calls it.
"""
import numpy as np

from . import advance as adv
from . import boxlib as bl
from .capi import default_params


def initdata_numpy(n, dx, prob_type=1, ng=3, nscal=2, lo=(0, 0, 0), centre=(0.5, 0.5, 0.5), dm=3):
    """reference src/initdata.f90:212-259 (prob_type 1 bubble / 2 advected blob), numpy restatement used
    for synthetic bench input; fills the box [lo, lo+n) (with ng ghost layers left at the background value).
    dm = 2: initdata_2d (initdata.f90:127-171, densfact = 2), arrays of shape (nx+2ng, ny+2ng, 1, nc)."""
    if dm == 2:
        u = np.zeros((n[0] + 2 * ng, n[1] + 2 * ng, 1, 2), order="F")
        s = np.zeros((n[0] + 2 * ng, n[1] + 2 * ng, 1, nscal), order="F")
        s[..., 0] = 1.0
        if prob_type == 2:
            u[..., 0] = 1.0
        x = dx[0] * (lo[0] + np.arange(n[0]) + 0.5)
        y = dx[1] * (lo[1] + np.arange(n[1]) + 0.5)
        X, Y = np.meshgrid(x, y, indexing="ij")
        dist = np.sqrt((X - centre[0]) ** 2 + (Y - centre[1]) ** 2)
        if prob_type == 3:                                # Rayleigh-Taylor interface, initdata.f90:175-185, 195-200 (tracer = 0); prob_lo = 0, prob_hi(1) = 2 * centre(1)
            Lx = 2.0 * centre[0]
            h = 0.02 * np.sin(4.0 * np.pi * X * Lx) + 0.01 * np.sin(8.0 * np.pi * X * Lx)
            s[ng:-ng, ng:-ng, 0, 0] = 1.0 + 0.5 + 0.5 * np.tanh((Y - 0.5 - h) / 0.01)
            if nscal > 1:
                s[..., 1] = 0.0
            return u, s
        r = 1.0 + 0.5 * (2.0 - 1.0) * (1.0 - np.tanh(30.0 * (dist - 0.1)))
        s[ng:-ng, ng:-ng, 0, 0] = r
        if nscal > 1:
            s[ng:-ng, ng:-ng, 0, 1] = r
        return u, s
    u = np.zeros(tuple(x + 2 * ng for x in n) + (3,), order="F")
    s = np.zeros(tuple(x + 2 * ng for x in n) + (nscal,), order="F")
    s[..., 0] = 1.0
    if prob_type == 2:
        u[..., 0] = 1.0
    x = dx[0] * (lo[0] + np.arange(n[0]) + 0.5)
    y = dx[1] * (lo[1] + np.arange(n[1]) + 0.5)
    z = dx[2] * (lo[2] + np.arange(n[2]) + 0.5)
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    g = ng
    if prob_type == 4:                                    # vortex tube, initdata.f90:276-306.  The reference measures x, y, z from the
        # BOX's low corner (float(k - lo(3))): with the one 32^3 box of inputs_vortextube_3d that is the domain's; kept as written
        xb = dx[0] * (np.arange(n[0]) + 0.5) - 0.5
        yb = dx[1] * (np.arange(n[1]) + 0.5) - 0.5
        zb = dx[2] * (np.arange(n[2]) + 0.5) - 0.5
        Xb, Yb, Zb = np.meshgrid(xb, yb, zb, indexing="ij")
        r_yz = np.sqrt(Yb * Yb + Zb * Zb)
        u[g:-g, g:-g, g:-g, 0] = np.tanh((0.15 - r_yz) / 0.0333)
        u[g:-g, g:-g, g:-g, 2] = 0.05 * np.exp(-15.0 * (Xb * Xb + Yb * Yb))
        s[g:-g, g:-g, g:-g, 0] = 1.0
        if nscal > 1:
            s[g:-g, g:-g, g:-g, 1] = np.exp(-500.0 * (0.15 - r_yz) ** 2)
        return u, s
    if prob_type == 3:                                    # Rayleigh-Taylor interface, initdata.f90:195-200, 261-274 (tracer = 0)
        h = lambda t: 0.02 * np.sin(4.0 * np.pi * t) + 0.01 * np.sin(8.0 * np.pi * t)      # noqa: E731
        s[g:-g, g:-g, g:-g, 0] = 1.0 + 0.5 + 0.5 * np.tanh((Z - 0.5 - h(X) - h(Y)) / 0.01)
        return u, s
    dist = np.sqrt((X - centre[0]) ** 2 + (Y - centre[1]) ** 2 + (Z - centre[2]) ** 2)
    r = 1.0 + 0.5 * (10.0 - 1.0) * (1.0 - np.tanh(30.0 * (dist - 0.1)))
    s[g:-g, g:-g, g:-g, 0] = r
    if nscal > 1:
        s[g:-g, g:-g, g:-g, 1] = r
    return u, s


def _limit_dt(sim, dt, first=False):
    """fixed_dt and stop_time of src/varden.f90:196-199 (first step) and :318-326 (later steps)"""
    if sim.fixed_dt > 0.0:
        dt = sim.fixed_dt
    if sim.stop_time >= 0.0 and sim.time + dt > sim.stop_time:
        dt = min(dt, sim.stop_time - sim.time) if first else sim.stop_time - sim.time
    return dt


class Varden:
    def __init__(self, n, phys_bc, params=None, prob_type=1, grav=-9.8, prob_hi=(1.0, 1.0, 1.0), init_shrink=1.0,
                 init_iter=4, do_initial_projection=1, u0=None, s0=None, device=0, decomp=(1, 1, 1), rank=0, nranks=1,
                 comm_id=None, restart=None, restart_step=0, fixed_dt=-1.0, stop_time=-1.0, swap_state=False, grav_dir=None, extruded2d=False):
        """decomp = (bx, by, bz): the domain is cut into bx*by*bz equal boxes (max_grid_size of the reference,
        src/_parameters:27), dealt round-robin to the ranks (one rank per GPU).  comm_id: the 128-byte RCCL unique
        id broadcast by the caller when nranks > 1.  restart: a checkpoint read by plotfile.read_checkfile -- the state comes from it
        and the start-up sequence (initial projection, pressure iterations) is skipped, src/varden.f90:94-97, 119, 180, 227."""
        self.prm = params or default_params()
        self.fixed_dt, self.stop_time = float(fixed_dt), float(stop_time)
        self.swap_state = bool(swap_state)     # uold <- unew by exchanging the handles instead of copying (see step)
        dm = int(self.prm.dm)
        self.n = tuple(int(x) for x in (n if hasattr(n, "__len__") else (n,) * dm))
        if dm == 2:
            self.n = self.n[:2] + (1,)
            decomp = (decomp[0], decomp[1], 1)
        self.prm.prob_type = prob_type
        self.rank, self.nranks = rank, nranks
        bl.initialize(self.prm, rank, nranks, device)
        if extruded2d:                                    # a z-uniform, z-periodic copy of a 2-D problem: velpred_2d's hi-x OUTLET rule (include/varden_amd.h)
            bl.set_extruded_2d(True)
        if nranks > 1:
            bl.comm_init(comm_id)
        self.phys = [[int(phys_bc[d][s]) for s in range(2)] if d < dm else [bl.INTERIOR, bl.INTERIOR] for d in range(3)]
        pmask = tuple(1 if self.phys[d][0] == bl.PERIODIC else 0 for d in range(3))
        lo, hi = (0, 0, 0), tuple(x - 1 for x in self.n)
        bs = tuple(self.n[d] // decomp[d] for d in range(3))
        assert all(bs[d] * decomp[d] == self.n[d] for d in range(3)), "decomp must divide n"
        self.boxes = []
        for kz in range(decomp[2]):
            for ky in range(decomp[1]):
                for kx in range(decomp[0]):
                    blo = (kx * bs[0], ky * bs[1], kz * bs[2])
                    self.boxes.append((blo, tuple(blo[d] + bs[d] - 1 for d in range(3))))
        self.owner = [i % nranks for i in range(len(self.boxes))]
        self.local = [i for i in range(len(self.boxes)) if self.owner[i] == rank]
        self.mla = bl.MLLayout([(lo, hi)], [self.boxes], owner=[self.owner], pmask=pmask)
        self.bct = bl.BCTower(self.mla, self.phys)
        self.dx = [[prob_hi[d] / self.n[d] for d in range(dm)]]
        ns = self.prm.nscal
        self.dm, self.nscal, self.press_comp = dm, ns, dm + ns + 1
        mk = lambda nc, ng, nodal=None: [bl.MultiFab(self.mla, 0, nc, ng, nodal)]   # noqa: E731
        self.uold, self.sold, self.unew, self.snew = mk(dm, 3), mk(ns, 3), mk(dm, 3), mk(ns, 3)
        self.gp, self.p = mk(dm, 1), mk(1, 1, (1, 1, 1))
        self.ext_vel_force, self.ext_scal_force = mk(dm, 1), mk(ns, 1)
        self.ext_vel_force[0].setval(grav, dm - 1 if grav_dir is None else int(grav_dir), 1, all=True)   # varden.f90:428-429 (grav_dir: the extruded 2-D problems, gravity along y)
        if restart is not None:
            from . import plotfile
            plotfile.load_restart(self, restart)
            self.istep = int(restart_step)
            self.fill_state_ghosts()
            self.unew[0].copy_c(0, self.uold[0], 0, dm, 3)
            self.snew[0].copy_c(0, self.sold[0], 0, ns, 3)
            return
        for li, gi in enumerate(self.local):              # each rank initialises / uploads only the boxes it owns
            blo, bhi = self.boxes[gi]
            if u0 is None:                                # blob centred in the domain (= (0.5,0.5,0.5) on the unit cube)
                ub, sb = initdata_numpy(bs, self.dx[0], prob_type, 3, ns, lo=blo, centre=tuple(0.5 * prob_hi[d] for d in range(3)), dm=dm)
            else:                                         # caller-supplied global arrays (carry 3 ghost layers)
                sl = tuple(slice(blo[d], bhi[d] + 1 + 6) if d < dm else slice(None) for d in range(3))
                ub, sb = np.array(u0[sl], order="F"), np.array(s0[sl], order="F")
            self.uold[0].from_numpy(ub, li)
            self.sold[0].from_numpy(sb, li)
        self.time, self.dt, self.istep = 0.0, 0.0, 0
        self.fill_state_ghosts()                                                   # initdata.f90:52-56
        if do_initial_projection:                                                  # varden.f90:126-138
            rhohalf = mk(1, 1)
            rhohalf[0].setval(1.0, all=True)
            adv.hgproject(bl.INITIAL_PROJECTION, self.mla, self.uold, self.uold, rhohalf, self.p, self.gp, self.dx, 1.0,
                          self.bct, self.press_comp)
            self.initial_projection_stat = adv.last_solver_stats("hg")
            rhohalf[0].destroy()
        self.p[0].setval(0.0, all=True)
        self.gp[0].setval(0.0, all=True)
        self.fill_state_ghosts()                                                   # varden.f90:165-172
        self.unew[0].copy_c(0, self.uold[0], 0, dm, 3)                             # varden.f90:175-176
        self.snew[0].copy_c(0, self.sold[0], 0, ns, 3)
        self.dt = _limit_dt(self, self.estdt(1.0e20) * init_shrink, first=True)    # varden.f90:186-199
        for it in range(init_iter):                                                # varden.f90:460-490
            self.advance(bl.PRESSURE_ITERS, it + 1)

    def fill_state_ghosts(self):
        for mf in (self.uold[0], self.sold[0], self.gp[0]):
            mf.fill_boundary()
        self.uold[0].physbc(0, 0, self.dm, self.bct)
        self.sold[0].physbc(0, self.dm, self.nscal, self.bct)

    def estdt(self, dtold):
        return adv.estdt(1, self.uold[0], self.sold[0], self.gp[0], self.ext_vel_force[0], self.dx[0], dtold)

    def advance(self, proj_type=bl.REGULAR_TIMESTEP, istep=0):
        adv.advance_timestep(istep, self.mla, self.sold, self.uold, self.snew, self.unew, self.gp, self.p,
                             self.ext_vel_force, self.ext_scal_force, self.bct, self.dt, self.time, self.dx,
                             self.press_comp, proj_type)

    def step(self):
        """one pass of the time-loop body, varden.f90:291-328"""
        self.istep += 1
        self.fill_state_ghosts()
        if self.istep > 1:
            self.dt = _limit_dt(self, self.estdt(self.dt))
        self.advance(bl.REGULAR_TIMESTEP, self.istep)
        if self.swap_state:
            # varden.f90:323-326 copies the valid cells; exchanging the handles gives the same next step because every ghost cell of
            # uold / sold is refilled before it is read (fill_state_ghosts above) and advance_timestep overwrites unew / snew. After such a
            # step self.unew / self.snew hold the PREVIOUS state.
            self.uold[0], self.unew[0] = self.unew[0], self.uold[0]
            self.sold[0], self.snew[0] = self.snew[0], self.sold[0]
        else:
            self.uold[0].copy_c(0, self.unew[0], 0, self.dm, 0)
            self.sold[0].copy_c(0, self.snew[0], 0, self.nscal, 0)
        self.time += self.dt

    def gather_valid(self, mf):
        """valid cells of the LOCAL boxes assembled into a global array (NaN where other ranks own the data)"""
        out = np.full(self.n + (mf.nc,), np.nan, order="F")
        g = mf.ng
        gz = g if self.dm == 3 else 0
        for li, gi in enumerate(self.local):
            blo, bhi = self.boxes[gi]
            a = mf.to_numpy(li)
            v = a[g:a.shape[0] - g, g:a.shape[1] - g, gz:a.shape[2] - gz] if g else a
            out[tuple(slice(blo[d], bhi[d] + 1) for d in range(3))] = v[:bhi[0] - blo[0] + 1, :bhi[1] - blo[1] + 1, :bhi[2] - blo[2] + 1]
        return out

    def close(self):
        for lst in (self.uold, self.sold, self.unew, self.snew, self.gp, self.p, self.ext_vel_force, self.ext_scal_force):
            lst[0].destroy()
        self.bct.destroy()
        self.mla.destroy()
        if self.nranks > 1:
            bl.comm_finalize()


def distribute(boxes, nranks):
    """boxes -> ranks by cell count: largest box first onto the least loaded rank (the knapsack of BASELINE.json configs[4]);
    deterministic, the same on every rank"""
    if nranks == 1:
        return [0] * len(boxes)
    cells = [int(np.prod([b[1][d] - b[0][d] + 1 for d in range(3)])) for b in boxes]
    load, owner = [0] * nranks, [0] * len(boxes)
    for i in sorted(range(len(boxes)), key=lambda i: (-cells[i], i)):
        r = min(range(nranks), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += cells[i]
    return owner


def prm_cluster(prm, name):
    """cluster_min_eff / cluster_min_width / cluster_blocking_factor (src/_parameters:37-39); not part of vdn_params"""
    return {"min_eff": 0.9, "min_width": 4, "blocking": 4}[name]


def _ncs(nc):
    """cells of level 0 per direction: an int (a cube) or three ints"""
    return tuple(int(x) for x in nc) if hasattr(nc, "__len__") else (int(nc),) * 3


def _level_domain(ncs, n):
    return ((0, 0, 0), tuple((c << n) - 1 for c in ncs))


def extruded_initdata(prob_type, nscal, prob_hi2=(1.0, 1.0)):
    """init_fn of a 2-D problem run as its z-uniform 3-D copy (VardenAMR(extrude2d = nz)): initdata_2d (src/initdata.f90:127-171) on the box's (x, y) cells, the same
    values on every plane; w = 0"""
    def fn(level, blo, nb, dx):
        u2, s2 = initdata_numpy((nb[0], nb[1]), dx, prob_type, 3, nscal, lo=blo, centre=(0.5 * prob_hi2[0], 0.5 * prob_hi2[1], 0.0), dm=2)
        u = np.zeros((nb[0] + 6, nb[1] + 6, nb[2] + 6, 3), order="F")
        s = np.zeros((nb[0] + 6, nb[1] + 6, nb[2] + 6, nscal), order="F")
        u[..., :2] = u2[:, :, 0, None, :]
        s[...] = s2[:, :, 0, None, :]
        return u, s
    return fn


class VardenAMR:
    """multi-level hierarchy on fixed grids (the reference's fixed_grids mode, src/initialize.f90:93-150): level 0 = the domain
    [0,nc)^3 in one box or in `base_boxes` (equal boxes, what max_grid_size makes of it), level 1 = the given fine boxes (fine index
    space, refinement ratio 2); `finer_levels`: box lists of the levels 2.. (any properly nested unions).  The loop body of src/varden.f90: ml_restrict_and_fill ghost fills,
    dt = min over levels of estdt, advance_timestep, new -> old copies."""

    def __init__(self, nc, fine_boxes, phys_bc, params=None, prob_type=1, grav=-9.8, init_shrink=0.1, device=0, finer_levels=(),
                 regrid_int=-1, max_levs=None, max_grid_size=256, init_iter=0, do_initial_projection=0,
                 rank=0, nranks=1, comm_id=None, base_boxes=None, init_fn=None, restart=None, restart_step=0,
                 fixed_dt=-1.0, stop_time=-1.0, amr_buf_width=-1, swap_state=False, extrude2d=None, extrude_zbc=None):
        """nc: cells of level 0 per direction (an int: a cube; dx = dy = dz = 1 / nc[0]).
        extrude2d = nz: a 2-D problem of nc[0] x nc[1] cells (phys_bc: its two directions) run as its z-uniform copy -- nz cells of level 0 along a periodic z, gravity along y,
        initdata_2d on every plane, velpred_2d's outlet rule (include/varden_amd.h: vdn_set_extruded_2d).  With w = 0 and nothing varying along z the 3-D scheme is the 2-D one
        (tests/test_dim2_gpu.py::test_extruded_copy_reproduces_the_2d_path: 1e-12 on every boundary pair of the 2-D inputs), so the hierarchies of the reference's four 2-D inputs
        -- tagging, regridding, composite solves -- run on the 3-D machinery; plane k = 0 is the 2-D answer (slice2d).
        init_fn(level, box_lo, box_shape, dx) -> (u, s) with 3 ghost layers replaces the analytic initial data of prob_type.
        several ranks (one per GPU): the boxes of every level are dealt to the ranks by cell count (`distribute`), `base_boxes` cuts
        level 0 into several boxes, comm_id is the RCCL unique id broadcast by the caller; regridding is single-rank in this round"""
        self.prm = params or default_params()
        self.grav, self.regrid_int, self.max_grid_size = grav, regrid_int, max_grid_size
        self.amr_buf_width = max(amr_buf_width, regrid_int, 1)    # the tag buffer of initialize.f90:248 AND regrid.f90:149 (probin.template:147-154)
        self.fixed_dt, self.stop_time = float(fixed_dt), float(stop_time)
        self.swap_state = bool(swap_state)     # uold <- unew by exchanging the handles instead of copying (see step)
        self.prm.prob_type = prob_type
        self.rank, self.nranks = rank, nranks
        bl.initialize(self.prm, rank, nranks, device)
        self.extrude2d = extrude2d
        self.grav_dir = 2
        if extrude2d:
            bl.set_extruded_2d(True)
            nc2 = _ncs(nc)
            nc = (nc2[0], nc2[1], int(extrude2d))
            phys_bc = [list(phys_bc[0]), list(phys_bc[1]), list(extrude_zbc) if extrude_zbc else [bl.PERIODIC, bl.PERIODIC]]      # (slip walls along z carry the same z-uniform solution: the oracle's form, tests/test_dim2_gpu.py)
            self.grav_dir = 1
            if init_fn is None:
                init_fn = extruded_initdata(prob_type, self.prm.nscal, (1.0, nc2[1] / float(nc2[0])))
        if nranks > 1:
            bl.comm_init(comm_id)
        self.nc = nc
        self.ncs = _ncs(nc)
        self.init_fn = init_fn
        self.phys = [[int(phys_bc[d][s]) for s in range(2)] for d in range(3)]
        lev_boxes = [fine_boxes] + list(finer_levels)
        self.nlev = NL = 1 + len(lev_boxes)
        self.max_levs = max_levs or NL
        self.nregrids = 0
        pd = [_level_domain(self.ncs, n) for n in range(NL)]
        base = [pd[0]] if base_boxes is None else [(tuple(b[0]), tuple(b[1])) for b in base_boxes]
        self.boxes = [base] + [[(tuple(b[0]), tuple(b[1])) for b in lb] for lb in lev_boxes]
        self.owner = [distribute(lb, nranks) for lb in self.boxes]
        self.local = [[i for i, o in enumerate(ow) if o == rank] for ow in self.owner]
        self.pmask = tuple(1 if self.phys[d][0] == bl.PERIODIC else 0 for d in range(3))
        self.mla = bl.MLLayout(pd, self.boxes, owner=self.owner, rr=[(2, 2, 2)] * (NL - 1), pmask=self.pmask)
        self.bct = bl.BCTower(self.mla, self.phys)
        self.dx = [[1.0 / (self.ncs[0] << n)] * 3 for n in range(NL)]
        dm, ns = 3, self.prm.nscal
        self.dm, self.nscal, self.press_comp = dm, ns, dm + ns + 1
        mk = lambda nc_, ng, nodal=None: [bl.MultiFab(self.mla, n, nc_, ng, nodal) for n in range(NL)]   # noqa: E731
        self.uold, self.sold, self.unew, self.snew = mk(dm, 3), mk(ns, 3), mk(dm, 3), mk(ns, 3)
        self.gp, self.p = mk(dm, 1), mk(1, 1, (1, 1, 1))
        self.ext_vel_force, self.ext_scal_force = mk(dm, 1), mk(ns, 1)
        for n in range(self.nlev):
            self.ext_vel_force[n].setval(grav, self.grav_dir, 1, all=True)
        if restart is not None:                          # initialize_from_restart, src/initialize.f90:22-88
            from . import plotfile
            plotfile.load_restart(self, restart)
            self.istep = int(restart_step)
            self.fill_state_ghosts()
            for n in range(self.nlev):
                self.unew[n].copy_c(0, self.uold[n], 0, dm, 3)
                self.snew[n].copy_c(0, self.sold[n], 0, ns, 3)
            return
        for n in range(self.nlev):
            for li, gi in enumerate(self.local[n]):         # each rank initialises the boxes it owns
                blo, bhi = self.boxes[n][gi]
                nb = tuple(bhi[d] - blo[d] + 1 for d in range(3))
                ub, sb = initdata_numpy(nb, self.dx[n], prob_type, 3, ns, lo=blo) if init_fn is None else init_fn(n, blo, nb, self.dx[n])
                self.uold[n].from_numpy(ub, li)
                self.sold[n].from_numpy(sb, li)
        self.time, self.istep = 0.0, 0
        self.fill_state_ghosts()
        if do_initial_projection:                                                      # varden.f90:126-138
            rhohalf = mk(1, 1)
            for m in rhohalf:
                m.setval(1.0, all=True)
            adv.hgproject(bl.INITIAL_PROJECTION, self.mla, self.uold, self.uold, rhohalf, self.p, self.gp, self.dx, 1.0, self.bct, self.press_comp)
            self.initial_projection_stat = adv.last_solver_stats("hg")
            for n in range(self.nlev):
                rhohalf[n].destroy()
                self.p[n].setval(0.0, all=True)
                self.gp[n].setval(0.0, all=True)
            self.fill_state_ghosts()                                                   # varden.f90:165-172
        for n in range(self.nlev):
            self.unew[n].copy_c(0, self.uold[n], 0, dm, 3)
            self.snew[n].copy_c(0, self.sold[n], 0, ns, 3)
        self.dt = _limit_dt(self, self.estdt(1.0e20) * init_shrink, first=True)
        for it in range(init_iter):                                                    # varden.f90:460-490
            adv.advance_timestep(it + 1, self.mla, self.sold, self.uold, self.snew, self.unew, self.gp, self.p,
                                 self.ext_vel_force, self.ext_scal_force, self.bct, self.dt, self.time, self.dx, self.press_comp, bl.PRESSURE_ITERS)

    @staticmethod
    def tagged_grids(nc, phys_bc, params=None, prob_type=1, max_levs=2, buf_wid=2, max_grid_size=256, device=0, rank=0, nranks=1, comm_id=None, base_boxes=None, extrude2d=None):
        """the grids the reference's initialize_with_adaptive_grids builds (src/initialize.f90:152-342): level by level, initial data on
        the level -> tag_boxes -> make_new_grids, until nothing is tagged or max_levs is reached.  Returns the box lists of the levels
        1.. (each in its own index space).  Nesting: a new level keeps 2 cells of its parent level around itself."""
        prm = params or default_params()
        prm.prob_type = prob_type
        bl.initialize(prm, rank, nranks, device)
        if nranks > 1:
            bl.comm_init(comm_id)
        ns = prm.nscal
        levels = []
        init_fn = None
        if extrude2d:                                      # (the 2-D problem as its z-uniform copy: see __init__)
            nc2 = _ncs(nc)
            nc = (nc2[0], nc2[1], int(extrude2d))
            phys_bc = [list(phys_bc[0]), list(phys_bc[1]), [bl.PERIODIC, bl.PERIODIC]]
            init_fn = extruded_initdata(prob_type, ns, (1.0, nc2[1] / float(nc2[0])))
        ncs = _ncs(nc)
        pd = [_level_domain(ncs, 0)]
        boxes = [[pd[0]] if base_boxes is None else [(tuple(b[0]), tuple(b[1])) for b in base_boxes]]
        for lev in range(1, max_levs):
            owner = [distribute(lb, nranks) for lb in boxes]
            mla = bl.MLLayout(pd, boxes, owner=owner, rr=[(2, 2, 2)] * (lev - 1), pmask=tuple(1 if int(phys_bc[d][0]) == bl.PERIODIC else 0 for d in range(3)))
            sold = bl.MultiFab(mla, lev - 1, ns, 3)
            dx = [1.0 / (ncs[0] << (lev - 1))] * 3
            for li, gi in enumerate([i for i, o in enumerate(owner[lev - 1]) if o == rank]):
                blo, bhi = boxes[lev - 1][gi]
                nb = tuple(bhi[d] - blo[d] + 1 for d in range(3))
                _, sb = initdata_numpy(nb, dx, prob_type, 3, ns, lo=blo) if init_fn is None else init_fn(lev - 1, blo, nb, dx)
                sold.from_numpy(sb, li)
            new, _ = adv.make_new_grids(sold, lev, buf_wid=buf_wid, nest=0 if lev == 1 else 2, min_eff=prm_cluster(prm, "min_eff"),
                                        min_width=prm_cluster(prm, "min_width"), blocking=prm_cluster(prm, "blocking"), max_grid_size=max_grid_size)
            sold.destroy(); mla.destroy()
            if not new:
                break
            levels.append(new)
            pd.append(_level_domain(ncs, lev))
            boxes.append(new)
        return levels

    def fill_state_ghosts(self):
        adv.ml_restrict_and_fill(self.uold, 0, 0, self.dm, self.bct)
        adv.ml_restrict_and_fill(self.sold, 0, self.dm, self.nscal, self.bct)
        adv.ml_restrict_and_fill(self.gp, 0, self.press_comp, self.dm, self.bct, same_boundary=True)   # extrap_comp (0-based press_comp + 1)

    def estdt(self, dtold):
        return min(adv.estdt(n + 1, self.uold[n], self.sold[n], self.gp[n], self.ext_vel_force[n], self.dx[n], dtold) for n in range(self.nlev))

    def step(self):
        self.istep += 1
        if self.max_levs > 1 and self.regrid_int > 0 and (self.istep - 1) % self.regrid_int == 0:   # varden.f90:256-264 (also at istep = 1)
            self.regrid()
        self.fill_state_ghosts()
        if self.istep > 1:
            self.dt = _limit_dt(self, self.estdt(self.dt))
        adv.advance_timestep(self.istep, self.mla, self.sold, self.uold, self.snew, self.unew, self.gp, self.p,
                             self.ext_vel_force, self.ext_scal_force, self.bct, self.dt, self.time, self.dx, self.press_comp, bl.REGULAR_TIMESTEP)
        for n in range(self.nlev):
            if self.swap_state:                      # see Varden.step
                self.uold[n], self.unew[n] = self.unew[n], self.uold[n]
                self.sold[n], self.snew[n] = self.snew[n], self.sold[n]
            else:
                self.uold[n].copy_c(0, self.unew[n], 0, self.dm, 0)
                self.sold[n].copy_c(0, self.snew[n], 0, self.nscal, 0)
        self.time += self.dt

    # ---- regridding (src/regrid.f90:17-263) ------------------------------------------------------------------------------------
    def _alloc_state(self, boxes):
        """layout, bc tower and the four carried state multifabs (uold, sold, gp, p) on the given box lists"""
        NL = len(boxes)
        pd = [_level_domain(self.ncs, n) for n in range(NL)]
        owner = [distribute(lb, self.nranks) for lb in boxes]
        mla = bl.MLLayout(pd, boxes, owner=owner, rr=[(2, 2, 2)] * (NL - 1), pmask=self.pmask)
        bct = bl.BCTower(mla, self.phys)
        mk = lambda nc_, ng, nodal=None: [bl.MultiFab(mla, n, nc_, ng, nodal) for n in range(NL)]   # noqa: E731
        st = dict(mla=mla, bct=bct, boxes=boxes, owner=owner, uold=mk(self.dm, 3), sold=mk(self.nscal, 3), gp=mk(self.dm, 1), p=mk(1, 1, (1, 1, 1)))
        for m in st["p"]:
            m.setval(0.0, all=True)                                               # regrid.f90:298
        return st

    @staticmethod
    def _free_state(st):
        for k in ("uold", "sold", "gp", "p"):
            for m in st[k]:
                m.destroy()
        st["bct"].destroy()
        st["mla"].destroy()

    def _fill_levels(self, st, nl):
        """ghost cells of the levels 0 .. nl-1 (what tagging with ghost cells and fillpatch read)"""
        adv.ml_restrict_and_fill(st["uold"][:nl], 0, 0, self.dm, st["bct"])
        adv.ml_restrict_and_fill(st["sold"][:nl], 0, self.dm, self.nscal, st["bct"])
        adv.ml_restrict_and_fill(st["gp"][:nl], 0, self.press_comp, self.dm, st["bct"], same_boundary=True)

    def regrid(self, buf_wid=None):
        """new grids from the current state, level by level: tag_boxes + make_new_grids on the (already regridded) level below, then
        build_and_fill_data (regrid.f90:269-339): interpolate from the coarser level, copy the old data of the level over it"""
        buf = self.amr_buf_width if buf_wid is None else buf_wid
        old = dict(mla=self.mla, bct=self.bct, uold=self.uold, sold=self.sold, gp=self.gp, p=self.p)
        old_nlev = self.nlev
        comps = (("uold", self.dm), ("sold", self.nscal), ("gp", self.dm), ("p", 1))
        cur = self._alloc_state([self.boxes[0]])
        for k, nc_ in comps:
            adv.copy_layouts(cur[k][0], 0, old[k][0], 0, nc_)
        lev = 1
        while lev < self.max_levs:
            self._fill_levels(cur, lev)
            new, _ = adv.make_new_grids(cur["sold"][lev - 1], lev, buf_wid=buf, nest=0 if lev == 1 else 2, min_eff=prm_cluster(self.prm, "min_eff"),
                                        min_width=prm_cluster(self.prm, "min_width"), blocking=prm_cluster(self.prm, "blocking"), max_grid_size=self.max_grid_size)
            if not new:
                break
            nxt = self._alloc_state(cur["boxes"] + [new])
            for n in range(lev):
                for k, nc_ in comps:
                    adv.copy_layouts(nxt[k][n], 0, cur[k][n], 0, nc_)
            self._fill_levels(nxt, lev)
            adv.fillpatch(nxt["uold"][lev], nxt["uold"][lev - 1], 0, self.dm)
            adv.fillpatch(nxt["sold"][lev], nxt["sold"][lev - 1], 0, self.nscal)
            adv.fillpatch(nxt["gp"][lev], nxt["gp"][lev - 1], 0, self.dm)
            adv.ml_nodal_prolongation(nxt["p"][lev], nxt["p"][lev - 1])
            if old_nlev > lev:
                for k, nc_ in comps:
                    adv.copy_layouts(nxt[k][lev], 0, old[k][lev], 0, nc_)
            self._free_state(cur)
            cur = nxt
            lev += 1
        # the temporaries of the old hierarchy go, the new one takes over
        for lst in (self.unew, self.snew, self.ext_vel_force, self.ext_scal_force):
            for m in lst:
                m.destroy()
        self._free_state(old)
        self.mla, self.bct, self.boxes, self.owner = cur["mla"], cur["bct"], cur["boxes"], cur["owner"]
        self.local = [[i for i, o in enumerate(ow) if o == self.rank] for ow in self.owner]
        self.uold, self.sold, self.gp, self.p = cur["uold"], cur["sold"], cur["gp"], cur["p"]
        self.nlev = NL = len(self.boxes)
        self.dx = [[1.0 / (self.ncs[0] << n)] * 3 for n in range(NL)]
        mk = lambda nc_, ng: [bl.MultiFab(self.mla, n, nc_, ng) for n in range(NL)]   # noqa: E731
        self.unew, self.snew = mk(self.dm, 3), mk(self.nscal, 3)
        self.ext_vel_force, self.ext_scal_force = mk(self.dm, 1), mk(self.nscal, 1)
        for n in range(NL):
            self.ext_vel_force[n].setval(self.grav, self.grav_dir, 1, all=True)
        self.fill_state_ghosts()                                                  # regrid.f90:252-254
        for n in range(NL):
            self.unew[n].copy_c(0, self.uold[n], 0, self.dm, 3)
            self.snew[n].copy_c(0, self.sold[n], 0, self.nscal, 3)
            self.p[n].fill_boundary()
        self.nregrids += 1

    def slice2d(self, mfs):
        """plane k = 0 of a per-level list of cell multifabs as level-domain arrays (nx << n, ny << n, nc), NaN outside the level's boxes: the 2-D answer of an extruded run"""
        out = []
        for n, mf in enumerate(mfs):
            a = np.full(((self.ncs[0] << n), (self.ncs[1] << n), mf.nc), np.nan)
            g = mf.ng
            for li, gi in enumerate(self.local[n]):
                lo, hi = self.boxes[n][gi]
                if lo[2] != 0:
                    continue
                f = mf.to_numpy(li)
                a[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, :] = f[g:f.shape[0] - g, g:f.shape[1] - g, g, :]
            out.append(a)
        return out

    def close(self):
        for lst in (self.uold, self.sold, self.unew, self.snew, self.gp, self.p, self.ext_vel_force, self.ext_scal_force):
            for m in lst:
                m.destroy()
        self.bct.destroy()
        self.mla.destroy()
        if self.nranks > 1:
            bl.comm_finalize()

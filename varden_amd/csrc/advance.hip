// advance.hip -- advance_timestep orchestration and the C-ABI entry points of the hot path.
//
//   vdn_advance_timestep   reference src/advance_timestep.f90:26-170 with advance_premac.f90:17-59,
//                          scalar_advance.f90:17-171, velocity_advance.f90:17-140
//   vdn_estdt              reference src/estdt.f90:15-87
//   vdn_hgproject / vdn_macproject and the per-kernel test hooks
//
// Differences from the reference that are deliberate MI355X design: the ~25 per-step multifabs come
// from a persistent HBM arena (no allocation inside a step); the four parallel_barrier calls that only
// fence the reference's timers (advance_timestep.f90:103,111,127,136) become stream synchronisations
// used for the same per-phase timing print.
#include "vdn_dev.h"
#include <chrono>
#include <cmath>

static double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void sync() { HIPCHK(hipStreamSynchronize(ctx().stream)); }


// ml_restrict_and_fill: one level = fill_boundary + physbc; two levels = average down + coarse-fine ghost interpolation too
static void restrict_and_fill(int nlev, vdn_multifab **mf, int icomp, int bcomp, int nc, bool same_boundary, const vdn_bc_tower *bct) {
  if (nlev == 1) mf_restrict_and_fill(mf[0], icomp, bcomp, nc, same_boundary, bct);
  else ml_restrict_and_fill(nlev, mf, icomp, bcomp, nc, same_boundary, bct);
}

// VDN_PHASE_HASH=1 (testing build; the hunt for the run-to-run differences of profiles/r06_determinism.txt): a checksum of whole multifabs (ghost cells included) at the
// phase boundaries of a step, on stderr
static void dbg_phase_hash(const char *tag, int nlevs, vdn_multifab **mfs, int per_level = 1, bool whole = false) {
  static const bool on = vdn_env("VDN_PHASE_HASH") && atoi(vdn_env("VDN_PHASE_HASH")) != 0;
  if (!on) return;
  HIPCHK(hipStreamSynchronize(ctx().stream));
  unsigned long long h = 1469598103934665603ull;
  for (int n = 0; n < nlevs; n++) for (int q = 0; q < per_level; q++) {
    const vdn_multifab *m = mfs[per_level == 1 ? n : 3 * n + q];
    if (!m) continue;
    std::vector<unsigned long long> buf(m->bytes / 8);
    HIPCHK(hipMemcpy(buf.data(), m->base, m->bytes, hipMemcpyDeviceToHost));
    for (int b = 0; b < m->nfabs(); b++) {                 // the valid cells / faces / nodes only: ghost entries of a temporary may never have been written
      const FV &f = m->fabs[b]; vdn_box vb = m->vbox[b];
      if (whole) for (int d = 0; d < 3; d++) { vb.lo[d] -= m->ng; vb.hi[d] += m->ng; }
      const size_t off = (size_t)(f.p - m->base);
      unsigned long long hb = 1469598103934665603ull;
      for (int c = 0; c < m->nc; c++)
        for (int k = vb.lo[2]; k <= vb.hi[2] + m->nodal[2]; k++) for (int j = vb.lo[1]; j <= vb.hi[1] + m->nodal[1]; j++) for (int i = vb.lo[0]; i <= vb.hi[0] + m->nodal[0]; i++) {
          { const unsigned long long v_ = buf[off + (size_t)(i - f.a0) + (size_t)f.n0 * ((size_t)(j - f.a1) + (size_t)f.n1 * (size_t)(k - f.a2)) + (size_t)c * (size_t)f.sc]; hb ^= v_; hb *= 1099511628211ull; }
          const unsigned long long v = buf[off + (size_t)(i - f.a0) + (size_t)f.n0 * ((size_t)(j - f.a1) + (size_t)f.n1 * (size_t)(k - f.a2)) + (size_t)c * (size_t)f.sc];
          h ^= v; h *= 1099511628211ull;
        }
      if (whole) fprintf(stderr, "PHASEBOX %-20s lev %d box %3d (%d,%d,%d)-(%d,%d,%d) %016llx\n", tag, n, b, m->vbox[b].lo[0], m->vbox[b].lo[1], m->vbox[b].lo[2], m->vbox[b].hi[0], m->vbox[b].hi[1], m->vbox[b].hi[2], hb);
    }
  }
  fprintf(stderr, "PHASE %-22s %016llx\n", tag, h);
  if (per_level == 3) {                                     // face fields: a checksum per level, direction and box as well
    for (int n = 0; n < nlevs; n++) for (int q = 0; q < 3; q++) {
      const vdn_multifab *m = mfs[3 * n + q];
      std::vector<unsigned long long> buf(m->bytes / 8);
      HIPCHK(hipMemcpy(buf.data(), m->base, m->bytes, hipMemcpyDeviceToHost));
      for (int b = 0; b < m->nfabs(); b++) {
        const FV &f = m->fabs[b]; const vdn_box &vb = m->vbox[b];
        const size_t off = (size_t)(f.p - m->base);
        unsigned long long hb = 1469598103934665603ull; long nan = 0;
        for (int k = vb.lo[2]; k <= vb.hi[2] + m->nodal[2]; k++) for (int j = vb.lo[1]; j <= vb.hi[1] + m->nodal[1]; j++) for (int i = vb.lo[0]; i <= vb.hi[0] + m->nodal[0]; i++) {
          const unsigned long long v = buf[off + (size_t)(i - f.a0) + (size_t)f.n0 * ((size_t)(j - f.a1) + (size_t)f.n1 * (size_t)(k - f.a2))];
          hb ^= v; hb *= 1099511628211ull;
          double dv; memcpy(&dv, &v, 8); if (dv != dv) nan++;
        }
        fprintf(stderr, "PHASEBOX %-20s lev %d dir %d box %3d (%d,%d,%d)-(%d,%d,%d) %016llx nan %ld\n", tag, n, q, b, vb.lo[0], vb.lo[1], vb.lo[2], vb.hi[0], vb.hi[1], vb.hi[2], hb, nan);
      }
    }
  }
}
extern "C" int vdn_advance_timestep(int istep, vdn_layout *mla, vdn_multifab **sold, vdn_multifab **uold,
                                    vdn_multifab **snew, vdn_multifab **unew, vdn_multifab **gp, vdn_multifab **p,
                                    vdn_multifab **ext_vel_force, vdn_multifab **ext_scal_force,
                                    const vdn_bc_tower *bct, double dt, double time, const double *dx,
                                    int press_comp, int proj_type) {
  VDN_TRY
  (void)istep; (void)time;
  REQUIRE(ctx().inited, "vdn_init has not been called");
  REQUIRE(mla && mla->nlev >= 1 && mla->nlev <= VDN_MAXLEV, "advance_timestep: 1..%d levels are implemented (nlevel = %d)", VDN_MAXLEV, mla ? mla->nlev : -1);
  const vdn_params &P = ctx().prm;
  const int dm = P.dm, nscal = P.nscal, nlevs = mla->nlev;
  const bool viscous = P.visc_coef > 0.0, diffusive = P.diff_coef > 0.0;
  REQUIRE(nlevs == 1 || dm == 3, "advance_timestep: multi-level hierarchies are implemented for dm = 3");
  REQUIRE(press_comp == dm + nscal + 1, "press_comp must be dm+nscal+1 (got %d)", press_comp);
  for (int n = 0; n < nlevs; n++) {
    REQUIRE(uold[n]->ng >= 3 && sold[n]->ng >= 3 && unew[n]->ng >= 3 && snew[n]->ng >= 3, "state needs ng_cell = 3");
    REQUIRE(gp[n]->ng >= 1 && p[n]->ng >= 1 && ext_vel_force[n]->ng >= 1 && ext_scal_force[n]->ng >= 1, "gp/p/ext forces need ng = 1");
    REQUIRE(sold[n]->nc == nscal && nscal <= 3, "sold must have nscal (<= 3) components");
  }
  dbg_phase_hash("uold at entry", nlevs, uold); dbg_phase_hash("sold at entry", nlevs, sold); dbg_phase_hash("gp at entry", nlevs, gp);
  dbg_phase_hash("uold+ghosts at entry", nlevs, uold, 1, true); dbg_phase_hash("sold+ghosts at entry", nlevs, sold, 1, true); dbg_phase_hash("gp+ghosts at entry", nlevs, gp, 1, true);
  Prof prof_advance("advance");                                                        // bl_prof names of advance_timestep.f90:60,99,107,123,132
  arena_reset();
  arena_reserve_for(mla);
  const double t_begin = wall();
  #define DXL(n) (dx + (n) * dm)

  // advance_timestep.f90:65-80; umac is [lev*3 + d]
  vdn_multifab *mac_rhs[VDN_MAXLEV], *rhohalf[VDN_MAXLEV], *umac[3 * VDN_MAXLEV] = { nullptr }, *lapu[VDN_MAXLEV] = { nullptr };
  for (int n = 0; n < nlevs; n++) {
    mac_rhs[n] = mf_temp(mla, n, 1, 1, -1, true, 0.0);
    rhohalf[n] = mf_temp(mla, n, dm, 1, -1, false, 0.0);          // (dm components as the reference builds it, advance_timestep.f90:70; only the first is ever written or read:
    mf_setval(rhohalf[n], 0.0, 0, 1, true);                        //  its setval, :73, is applied to that one)
    for (int d = 0; d < dm; d++) umac[3 * n + d] = mf_temp(mla, n, 1, 1, d, true, 1.e20);
    // lapu (advance_timestep.f90:85-93); NULL stands for the all-zero field when visc_coef == 0
    if (viscous) {
      lapu[n] = mf_temp(mla, n, dm, 0, -1, true, 0.0);
      for (int c = 0; c < dm; c++) k_explicit_diffusive_term(lapu[n], uold[n], c, c, DXL(n), bct);
    }
  }
  if (viscous) for (int n = nlevs - 1; n >= 1; n--) ml_cc_restriction(lapu[n - 1], lapu[n], 0, dm);     // cc_applyop per level, then average down
  // velpred (advance_premac) and the velocity mkflux (velocity_advance) both start from the limited slopes of the same uold: the
  // reference computes them twice (velpred.f90:1985-1990, mkflux.f90:1207-1212); one level of one box keeps velpred's for mkflux
  ctx().drop_step_caches();
  if (nlevs == 1 && dm == 3 && god_per_box(uold[0]) && !vdn_env("VDN_NO_SLOPE_CACHE")) {
    const int nb = uold[0]->nfabs();
    for (int d = 0; d < 3; d++) ctx().slope_cache[d].assign(nb, nullptr);
    ctx().slope_src.assign(nb, nullptr); ctx().macmax_cache.assign(nb, nullptr); ctx().macmax_src.assign(nb, nullptr);
    for (int ib = 0; ib < nb; ib++) {
      const vdn_box &b = uold[0]->vbox[ib];
      const size_t fld = (size_t)(b.hi[0] - b.lo[0] + 3) * (b.hi[1] - b.lo[1] + 3) * (b.hi[2] - b.lo[2] + 3) * sizeof(double);
      for (int d = 0; d < 3; d++) ctx().slope_cache[d][ib] = (double *)arena_alloc(fld * 3);
      // likewise max |umac| (the dead band of the upwinding, mkflux.f90:1374-1401): scalar_advance and velocity_advance see the same MAC field
      ctx().macmax_cache[ib] = (double *)arena_alloc(256);
    }
  }

  // advance_premac.f90:44-51
  // The velocity forcing of advance_premac -- mkvelforce(ext, gp, sold, lapu, visc_fac = 1) + its ghost fill -- is computed a second time, from
  // the same operands, at the top of velocity_advance (velocity_advance.f90:63-66; gp changes only in hgproject, sold and ext not at all; lapu is
  // zeroed in between only for diffusion_type = 2, advance_timestep.f90:116-120).  Unless that is the case the first one is kept for the
  // velocity mkflux (0.32 ms of a 41 ms step at 256^3): the same values, as with the limited slopes above.
  static const bool force_reuse = !(vdn_env("VDN_NO_FORCE_REUSE") && atoi(vdn_env("VDN_NO_FORCE_REUSE")) != 0);
  const bool keep_vel_force = force_reuse && !(viscous && P.diffusion_type == 2);
  vdn_multifab *vel_force0[VDN_MAXLEV] = { nullptr };
  if (keep_vel_force) for (int n = 0; n < nlevs; n++) vel_force0[n] = mf_temp(mla, n, dm, 1, -1, false, 0.0);
  {
    Prof pr("advance_premac");
    size_t mark = arena_mark();
    vdn_multifab *vel_force[VDN_MAXLEV];
    for (int n = 0; n < nlevs; n++) {
      vel_force[n] = keep_vel_force ? vel_force0[n] : mf_temp(mla, n, dm, 1, -1, false, 0.0);
      k_mkvelforce(vel_force[n], ext_vel_force[n], sold[n], gp[n], lapu[n], 1.0);
    }
    restrict_and_fill(nlevs, vel_force, 0, bct->extrap_comp0(), dm, true, bct);         // mkforce.f90:75-76
    for (int n = 0; n < nlevs; n++) k_velpred(uold[n], umac + 3 * n, vel_force[n], DXL(n), dt, bct);
    // velpred.f90:102-122: ghost faces (coarse: same-level images; fine: from the coarse level, then same-level), edge restriction
    for (int d = 0; d < dm; d++) mf_fill_boundary(umac[d]);
    for (int n = 1; n < nlevs; n++) for (int d = 0; d < dm; d++) { ml_create_umac_grown(umac[3 * n + d], umac[3 * (n - 1) + d], d); mf_fill_boundary(umac[3 * n + d]); }
    for (int n = nlevs - 1; n >= 1; n--) for (int d = 0; d < dm; d++) ml_edge_restriction(umac[3 * (n - 1) + d], umac[3 * n + d], d);
    dbg_phase_hash("vel_force", nlevs, vel_force); dbg_phase_hash("umac after velpred", nlevs, umac, 3);
    if (!keep_vel_force) for (int n = nlevs - 1; n >= 0; n--) mf_temp_free(vel_force[n]);
    arena_release(mark);
  }

  // MAC projection (advance_timestep.f90:97-104)
  sync(); double t0 = wall();
  { Prof pr("MAC_Project");
  do_macproject(mla, umac, sold, mac_rhs, dx, bct, press_comp - 1);
  }
  sync(); ctx().step_sec[2] = wall() - t0;
  dbg_phase_hash("umac after MAC", nlevs, umac, 3);

  // scalar_advance.f90:54-118
  t0 = wall();
  {
    Prof pr("Scalar_update");
    size_t mark = arena_mark();
    int is_cons[VDN_MAXCOMP]; is_cons[0] = 1; for (int c = 1; c < nscal; c++) is_cons[c] = 0;
    vdn_multifab *scal_force[VDN_MAXLEV], *divu[VDN_MAXLEV], *sflux[3 * VDN_MAXLEV], *sedge[3 * VDN_MAXLEV], *laps[VDN_MAXLEV] = { nullptr };
    if (diffusive) {                                                                    // scalar_advance.f90:80-89 (cc_applyop per level, then average down)
      for (int n = 0; n < nlevs; n++) {
        laps[n] = mf_temp(mla, n, nscal, 0, -1, true, 0.0);
        for (int c = 1; c < nscal; c++) k_explicit_diffusive_term(laps[n], sold[n], c, dm + c, DXL(n), bct);
      }
      for (int n = nlevs - 1; n >= 1; n--) ml_cc_restriction(laps[n - 1], laps[n], 1, nscal - 1);
    }
    for (int n = 0; n < nlevs; n++) {
      scal_force[n] = mf_temp(mla, n, nscal, 1, -1, false, 0.0);
      divu[n] = mac_rhs[n];                            // scalar_advance.f90:63,102 passes a zero divu as mac_rhs: the step's mac_rhs is that zero field (same shape)
      // mkflux writes every edge state; the fluxes of non-conservative components are never written NOR read on one level (a hierarchy
      // restricts all components of the flux multifab, so there they are zeroed as the reference's setval does)
      for (int d = 0; d < dm; d++) { sflux[3 * n + d] = mf_temp(mla, n, nscal, 0, d, nlevs > 1, 0.0); sedge[3 * n + d] = mf_temp(mla, n, nscal, 0, d, false, 0.0); }
      k_mkscalforce(scal_force[n], ext_scal_force[n], laps[n], 1.0);
    }
    restrict_and_fill(nlevs, scal_force, 0, bct->extrap_comp0(), nscal, true, bct);     // mkforce.f90:283-284
    // one level, no diffusion: the update runs inside the mkflux march where that is the fused one (a level of one box): same forcing term,
    // the edge states and fluxes never reach memory (godunov.hip, UPD); on a hierarchy the fluxes are restricted in between (mkflux.f90:137-146)
    bool s_updated = false;
    for (int n = 0; n < nlevs; n++) {
      MkUpdate U; U.snew = snew[n]; U.fmode = 0;
      const bool try_upd = force_reuse && nlevs == 1 && dm == 3 && !diffusive;
      s_updated = k_mkflux(sold[n], sedge + 3 * n, sflux + 3 * n, umac + 3 * n, scal_force[n], divu[n], DXL(n), dt, bct, false, is_cons, try_upd ? &U : nullptr);
      if (diffusive || !force_reuse) k_mkscalforce(scal_force[n], ext_scal_force[n], laps[n], 0.0);     // without diffusion: ext_scal_force again, already there
    }
    // mkflux.f90:137-146: on a hierarchy the flux of every conservative component through a coarse face under a finer level is the mean of the four fine
    // fluxes -- the coarse cells NEXT TO the finer level are updated with the fine level's fluxes through the interface, which is what conserves the mass
    // of the composite grid (round 5: rounds 2-4 left the coarse level its own fluxes)
    for (int n = nlevs - 1; n >= 1; n--) for (int c = 0; c < nscal; c++) if (is_cons[c]) for (int d = 0; d < dm; d++) ml_edge_restriction(sflux[3 * (n - 1) + d], sflux[3 * n + d], d, c);
    if (diffusive || !force_reuse) restrict_and_fill(nlevs, scal_force, 0, bct->extrap_comp0(), nscal, true, bct);
    dbg_phase_hash("sedge after mkflux", nlevs, sedge, 3); dbg_phase_hash("sflux restricted", nlevs, sflux, 3); dbg_phase_hash("scal_force", nlevs, scal_force);
    if (!s_updated) for (int n = 0; n < nlevs; n++) k_update(sold[n], umac + 3 * n, sedge + 3 * n, sflux + 3 * n, scal_force[n], snew[n], DXL(n), dt, false, is_cons);
    dbg_phase_hash("snew before r_and_f", nlevs, snew);
    restrict_and_fill(nlevs, snew, 0, dm, nscal, false, bct);                           // update.f90:106
    if (diffusive) {                                                                    // scalar_advance.f90:144-162
      const double visc_mu = (P.diffusion_type == 1) ? 0.5 * dt * P.diff_coef : dt * P.diff_coef;
      for (int c = 1; c < nscal; c++) {
        if (nlevs == 1) do_diff_scalar_solve(mla, snew[0], laps[0], dx, visc_mu, bct, c, dm + c);
        else do_ml_diff_scalar_solve(mla, snew, laps, dx, visc_mu, bct, c, dm + c);
      }
    }
    arena_release(mark);
  }
  sync(); ctx().step_sec[0] = wall() - t0;
  dbg_phase_hash("snew after scalars", nlevs, snew);

  // make_at_halftime (advance_timestep.f90:114, make_at_halftime.f90:64-65)
  { Prof pr("make_at_halftime"); for (int n = 0; n < nlevs; n++) k_make_at_halftime(rhohalf[n], sold[n], snew[n], 0, 0); }
  restrict_and_fill(nlevs, rhohalf, 0, dm + 0, 1, false, bct);
  if (viscous && P.diffusion_type == 2) for (int n = 0; n < nlevs; n++) mf_setval(lapu[n], 0.0, 0, dm, true);   // advance_timestep.f90:116-120

  // velocity_advance.f90:48-93
  t0 = wall();
  {
    Prof pr("Velocity_update");
    size_t mark = arena_mark();
    int is_cons[3] = { 0, 0, 0 };
    vdn_multifab *vel_force[VDN_MAXLEV], *uflux[3 * VDN_MAXLEV], *uedge[3 * VDN_MAXLEV];
    for (int n = 0; n < nlevs; n++) {
      vel_force[n] = keep_vel_force ? vel_force0[n] : mf_temp(mla, n, dm, 1, -1, false, 0.0);
      for (int d = 0; d < dm; d++) { uflux[3 * n + d] = mf_temp(mla, n, dm, 0, d, false, 0.0); uedge[3 * n + d] = mf_temp(mla, n, dm, 0, d, false, 0.0); }   // uedge: written everywhere; uflux: never read (no conservative velocity component)
      if (!keep_vel_force) k_mkvelforce(vel_force[n], ext_vel_force[n], sold[n], gp[n], lapu[n], 1.0);
    }
    if (!keep_vel_force) restrict_and_fill(nlevs, vel_force, 0, bct->extrap_comp0(), dm, true, bct);
    // one level: the forcing of the update (mkvelforce with rhohalf, visc_fac = 0) is formed inside the update pass; on a hierarchy the
    // average-down of ml_restrict_and_fill sits between the two, so they stay apart
    const bool fuse_force = force_reuse && nlevs == 1 && dm == 3;
    bool u_updated = false;
    for (int n = 0; n < nlevs; n++) {
      MkUpdate U; U.snew = unew[n]; U.fmode = 1; U.ext = ext_vel_force[n]; U.gp = gp[n]; U.rho = rhohalf[n]; U.lapu0 = P.visc_coef * 0.0 * 0.0;
      const bool try_upd = fuse_force && !viscous && P.boussinesq == 0;      // (viscous: lapu enters the forcing term; boussinesq: the tracer scales ext)
      u_updated = k_mkflux(uold[n], uedge + 3 * n, uflux + 3 * n, umac + 3 * n, vel_force[n], mac_rhs[n], DXL(n), dt, bct, true, is_cons, try_upd ? &U : nullptr);
      if (!fuse_force) k_mkvelforce(vel_force[n], ext_vel_force[n], rhohalf[n], gp[n], lapu[n], 0.0);
    }
    dbg_phase_hash("uedge after mkflux", nlevs, uedge, 3); dbg_phase_hash("vel_force (update)", nlevs, vel_force);
    if (u_updated) { /* unew is written */ }
    else if (fuse_force) k_update_velforce(uold[0], umac, uedge, ext_vel_force[0], rhohalf[0], gp[0], lapu[0], 0.0, unew[0], DXL(0), dt);
    else {
      restrict_and_fill(nlevs, vel_force, 0, bct->extrap_comp0(), dm, true, bct);
      for (int n = 0; n < nlevs; n++) k_update(uold[n], umac + 3 * n, uedge + 3 * n, uflux + 3 * n, vel_force[n], unew[n], DXL(n), dt, true, is_cons);
    }
    dbg_phase_hash("unew before r_and_f", nlevs, unew);
    restrict_and_fill(nlevs, unew, 0, 0, dm, false, bct);                               // update.f90:104
    dbg_phase_hash("unew after update", nlevs, unew);
    if (viscous) {                                                                      // velocity_advance.f90:103-118
      const double visc_mu = (P.diffusion_type == 1) ? 0.5 * dt * P.visc_coef : dt * P.visc_coef;
      if (nlevs == 1) do_visc_solve(mla, unew[0], lapu[0], rhohalf[0], mac_rhs[0], dx, visc_mu, bct);
      else do_ml_visc_solve(mla, unew, lapu, rhohalf, mac_rhs, dx, visc_mu, bct);
    }
    arena_release(mark);
  }
  sync(); ctx().step_sec[1] = wall() - t0;
  dbg_phase_hash("unew after viscous", nlevs, unew);

  // hgproject (advance_timestep.f90:129-137)
  t0 = wall();
  { Prof pr("HG_Project");
  do_hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, bct, press_comp - 1); }
  sync(); ctx().step_sec[3] = wall() - t0;
  dbg_phase_hash("unew after HG", nlevs, unew); dbg_phase_hash("p after HG", nlevs, p); dbg_phase_hash("gp after HG", nlevs, gp);
  #undef DXL

  arena_reset();
  ctx().drop_step_caches();
  ctx().step_sec[4] = wall() - t_begin;
  if (P.verbose >= 1 && ctx().rank == 0) {                                              // advance_timestep.f90:159-166
    printf(" Timing summary:\n Scalar   update: %g seconds\n Velocity update: %g seconds\n  MAC Projection: %g seconds\n   HG Projection: %g seconds\n\n",
           ctx().step_sec[0], ctx().step_sec[1], ctx().step_sec[2], ctx().step_sec[3]);
  }
  VDN_CATCH
}

extern "C" int vdn_estdt(int lev, const vdn_multifab *u, const vdn_multifab *s, const vdn_multifab *gp,
                         const vdn_multifab *ext, const double *dx, double dtold, double *dt_out) {
  VDN_TRY
  (void)lev;
  double m[6];
  k_estdt_max(u, s, gp, ext, m);
  for (int q = 0; q < 6; q++) REQUIRE(m[q] < HUGE_VAL, "estdt: non-finite velocity or pressure-gradient term (the state has blown up)");
  // (multi-rank: all-reduce MAX of the six maxima is equivalent to the reference's MIN of dt_proc)
  const double eps = (double)1.0e-8f;                 // single-precision literal, estdt.f90:146
  double dt = 1.e20;
  if (m[0] > eps) dt = fmin(dt, dx[0] / m[0]);
  if (m[1] > eps) dt = fmin(dt, dx[1] / m[1]);
  const int dm = ctx().prm.dm;
  if (dm == 3 && m[2] > eps) dt = fmin(dt, dx[2] / m[2]);
  if (m[3] > eps) dt = fmin(dt, sqrt(2.0 * dx[0] / m[3]));
  if (m[4] > eps) dt = fmin(dt, sqrt(2.0 * dx[1] / m[4]));
  if (dm == 3 && m[5] > eps) dt = fmin(dt, sqrt(2.0 * dx[2] / m[5]));
  if (dt == 1.e20) { dt = fmin(dx[0], dx[1]); if (dm == 3) dt = fmin(dt, dx[2]); }     // estdt.f90:71-74
  dt = dt * ctx().prm.cflfac;
  if (dtold > 0.0) dt = fmin(dt, ctx().prm.max_dt_growth * dtold);
  *dt_out = dt;
  VDN_CATCH
}

extern "C" int vdn_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold,
                             vdn_multifab **rhohalf, vdn_multifab **p, vdn_multifab **gp,
                             const double *dx, double dt, const vdn_bc_tower *bct, int press_comp) {
  VDN_TRY
  REQUIRE(mla && mla->nlev <= VDN_MAXLEV, "hgproject: at most %d levels are implemented (nlevel = %d)", VDN_MAXLEV, mla ? mla->nlev : -1);
  arena_reset(); arena_reserve_for(mla);
  do_hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, bct, press_comp - 1);
  arena_reset();
  VDN_CATCH
}
extern "C" int vdn_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs,
                              const double *dx, const vdn_bc_tower *bct, int bc_comp) {
  VDN_TRY
  REQUIRE(mla && mla->nlev <= VDN_MAXLEV, "macproject: at most %d levels are implemented (nlevel = %d)", VDN_MAXLEV, mla ? mla->nlev : -1);
  arena_reset(); arena_reserve_for(mla);
  do_macproject(mla, umac, rho, mac_rhs, dx, bct, bc_comp - 1);
  arena_reset();
  VDN_CATCH
}

// ---- per-kernel hooks -------------------------------------------------------------------------------------
#define HOOK_BEGIN(mf) VDN_TRY REQUIRE(ctx().inited, "vdn_init has not been called"); arena_reset(); arena_reserve_for((mf)->la);
#define HOOK_END arena_reset(); VDN_CATCH

extern "C" int vdn_k_slope(const vdn_multifab *s, vdn_multifab *slope, int dir, int bccomp, const vdn_bc_tower *bct) {
  HOOK_BEGIN(s) k_slope(s, slope, dir, bccomp, bct); HOOK_END
}
extern "C" int vdn_k_velpred(const vdn_multifab *u, vdn_multifab **umac, const vdn_multifab *force, const double *dx, double dt,
                             const vdn_bc_tower *bct) {
  HOOK_BEGIN(u)
  k_velpred(u, umac, force, dx, dt, bct);
  for (int d = 0; d < ctx().prm.dm; d++) mf_fill_boundary(umac[d]);
  HOOK_END
}
extern "C" int vdn_k_mkflux(const vdn_multifab *s, vdn_multifab **sedge, vdn_multifab **flux, vdn_multifab **umac,
                            const vdn_multifab *force, const vdn_multifab *mac_rhs, const double *dx, double dt,
                            const vdn_bc_tower *bct, int is_vel, const int *is_cons) {
  HOOK_BEGIN(s) k_mkflux(s, sedge, flux, umac, force, mac_rhs, dx, dt, bct, is_vel != 0, is_cons); HOOK_END
}
extern "C" int vdn_k_update(const vdn_multifab *sold, vdn_multifab **umac, vdn_multifab **sedge, vdn_multifab **flux,
                            const vdn_multifab *force, vdn_multifab *snew, const double *dx, double dt,
                            int is_vel, const int *is_cons, const vdn_bc_tower *bct) {
  HOOK_BEGIN(sold)
  k_update(sold, umac, sedge, flux, force, snew, dx, dt, is_vel != 0, is_cons);
  mf_restrict_and_fill(snew, 0, is_vel ? 0 : bct->dm, snew->nc, false, bct);
  HOOK_END
}
extern "C" int vdn_k_mkvelforce(vdn_multifab *vf, const vdn_multifab *ext, const vdn_multifab *s, const vdn_multifab *gp,
                                const vdn_multifab *lapu, double visc_fac, const vdn_bc_tower *bct) {
  HOOK_BEGIN(vf)
  k_mkvelforce(vf, ext, s, gp, lapu, visc_fac);
  mf_restrict_and_fill(vf, 0, bct->extrap_comp0(), vf->nc, true, bct);
  HOOK_END
}
extern "C" int vdn_k_mkscalforce(vdn_multifab *sf, const vdn_multifab *ext, const vdn_multifab *laps, double diff_fac,
                                 const vdn_bc_tower *bct) {
  HOOK_BEGIN(sf)
  k_mkscalforce(sf, ext, laps, diff_fac);
  mf_restrict_and_fill(sf, 0, bct->extrap_comp0(), sf->nc, true, bct);
  HOOK_END
}
extern "C" int vdn_k_make_at_halftime(vdn_multifab *rhohalf, const vdn_multifab *sold, const vdn_multifab *snew,
                                      int in_comp, int out_comp, const vdn_bc_tower *bct) {
  HOOK_BEGIN(rhohalf)
  k_make_at_halftime(rhohalf, sold, snew, in_comp, out_comp);
  mf_restrict_and_fill(rhohalf, out_comp, bct->dm + in_comp, 1, false, bct);
  HOOK_END
}
static void bc_from_flat(const int *bc, int out[3][2]) { for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) out[d][s] = bc[d * 2 + s]; }
extern "C" int vdn_cc_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int *bc,
                            double rel_eps, double abs_eps, int max_iter, int *cycles, double *res0, double *res) {
  HOOK_BEGIN(rh)
  int b[3][2]; bc_from_flat(bc, b);
  // nested iteration (vdn_params.mac_fmg) when phi comes in zero, ghost cells included -- the check costs a reduction here; macproject knows
  int fmg = 0;
  if (ctx().prm.mac_fmg && max_iter >= 0 && ctx().prm.dm == 3) fmg = mf_norm_inf_grown(phi, 0, 1, 1) == 0.0 ? 1 : 0;
  int rc = cc_solve(rh, phi, beta, dx, b, rel_eps, abs_eps, max_iter, cycles, res0, res, nullptr, nullptr, nullptr, nullptr, fmg);
  arena_reset();
  if (rc != 0) vdn_fail("cc multigrid did not converge: %d cycles, residual %g (rhs %g)", *cycles, *res, *res0);
  HOOK_END
}
extern "C" int vdn_cc_smooth(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int *bc, int nsweeps) {
  HOOK_BEGIN(rh) int b[3][2]; bc_from_flat(bc, b); cc_smooth(rh, phi, beta, dx, b, nsweeps); HOOK_END
}
extern "C" int vdn_nd_solve(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *coeffs, const vdn_multifab *u,
                            const double *dx, const int *bc, double rel_eps, double abs_eps, int max_iter,
                            int *cycles, double *res0, double *res) {
  HOOK_BEGIN(rh)
  int b[3][2]; bc_from_flat(bc, b);
  int rc = nd_solve(rh, phi, coeffs, u, dx, b, rel_eps, abs_eps, max_iter, cycles, res0, res);
  arena_reset();
  if (rc != 0) vdn_fail("nodal multigrid did not converge: %d cycles, residual %g (rhs %g)", *cycles, *res, *res0);
  HOOK_END
}
extern "C" int vdn_bench_cc_smoother(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const vdn_multifab *rho, const double *dx, const int *bc,
                                     int nlaunch, double *avg_ms, long *cells) {
  HOOK_BEGIN(rh) int b[3][2]; bc_from_flat(bc, b); cc_bench_smoother(rh, phi, beta, rho, dx, b, nlaunch, avg_ms, cells); HOOK_END
}
extern "C" int vdn_bench_cc_smoother_in_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const vdn_multifab *rho, const double *dx, const int *bc,
                                              int nsweeps, int nlaunch, double *avg_ms, long *cells) {
  HOOK_BEGIN(rh) int b[3][2]; bc_from_flat(bc, b); REQUIRE(nsweeps >= 1 && rho, "vdn_bench_cc_smoother_in_solve: nsweeps >= 1 and rho"); cc_bench_smoother(rh, phi, beta, rho, dx, b, nlaunch, avg_ms, cells, nsweeps); HOOK_END
}

/*
 * mom6x.h -- C ABI of the MI355X-native MOM6 split-explicit dynamical core.
 *
 * This is the drop-in boundary: every entry point below
 * is what a Fortran `bind(C)` interface in MOM6 binds to replace one module
 * procedure of the reference.  Each declaration cites the reference interface
 * (file:line under /root/reference) that it stands in for.  No torch / C++ types
 * appear in any signature: plain pointers, ints and doubles only.
 *
 * Memory model
 * ------------
 * All field pointers handed to the `mom6x_*` compute entry points are DEVICE
 * pointers (HBM) to FP64 arrays in the "pitched tile layout" described by
 * mom6x_dims.  The Fortran host keeps its own (i,j,k) column-major arrays
 * (i fastest); mom6x_upload_* / mom6x_download_* convert between a Fortran array
 * with MOM6's symmetric-memory extents and the pitched device layout with one
 * strided DMA (hipMemcpy3DAsync), so i stays the coalesced index on the device
 * exactly as in the reference's `do i` inner loops.
 *
 * Pitched tile layout (one layout for h-, u-, v- and q-point arrays):
 *   local C indices: compute domain i = 0..ni-1, j = 0..nj-1 (MOM6 isc..iec,
 *   jsc..jec); data domain i = -halo..ni-1+halo; the symmetric-memory extra
 *   row/column of u-, v-, q-point arrays is I = -halo-1 (MOM6 IsdB = isd-1).
 *   A vertex/face index I (capital in MOM6) refers to the east/north face of
 *   cell i, as in MOM6.
 *   flat(i,j,k) = (i + ioff) + (j + joff)*pitch + k*slab,
 *   with ioff >= halo+1, joff = halo+1, pitch >= ioff+ni+halo (multiple of 16
 *   doubles so that i = 0 starts a 128-byte line), slab = pitch*(nj+2*halo+1).
 *   All four staggerings use the same flat(); 2-D arrays are one slab.
 */
#ifndef MOM6X_H
#define MOM6X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: mom6x_continuity_params.sum_order, the Leith members of mom6x_hor_visc_params, Recon_Scheme / boundary_extrap /
 * h_nonvanished of mom6x_eos_params, mom6x_device_count.
 * 3: mom6x_coriolis_params grew (CORIOLIS_SCHEME ARAKAWA_LAMB81 / AL_BLEND / ROBUST_ENSTRO: wt_lin_blend, F_eff_max_blend, PV_Adv_Scheme),
 * mom6x_eos_params.EOS_quadrature and the EOS forms beyond LINEAR / WRIGHT, mom6x_barotropic_params' wide-halo members
 * (use_wide_halos, BTHALO, min_stencil) (round 3).
 * 5: (round 5) mom6x_barotropic_params.nonlinear_continuity / nonlin_cont_update_period; btstep accepts BT_cont == NULL;
 *    mom6x_continuity_params.sum_order = MOM6X_SUM_TREE16_FMA.
 * 6: (round 6) mom6x_barotropic_params.bt_thick_scheme / maxvel (BT_THICK_SCHEME without a BT_cont_type; BOUND_BT_CORRECTION through
 *    eta_cor_bound), mom6x_set_dtbt_pbce_eta.
 * 4: (round 4) mom6x_dyn_split_RK2_restart_fills + MOM6X_RK2_HAVE_*, mom6x_rk2_diag_*; see the end of this comment's list in
 * DESIGN.md section 1.  Hosts compare mom6x_abi_version() with the value they were
 * built against (fortran/mom6x_c_api.F90 MOM6X_ABI_BUILT_FOR, mom6_amd/abi.py ABI_VERSION) and refuse to run on a mismatch. */
#define MOM6X_ABI_VERSION 6

/* ------------------------------------------------------------------------- */
/* Tile dimensions and layout (MOM_hor_index.F90:14-44 hor_index_type +
 * config_src/memory/dynamic_symmetric/MOM_memory.h extents).                */
typedef struct mom6x_dims {
  int ni, nj, nk;   /* compute-domain size of this tile and number of layers  */
  int halo;         /* NIHALO = NJHALO (MOM_domains.F90:221), >= 4 for PPM+BT */
  int ioff, joff;   /* memory column/row of local index i = 0 / j = 0         */
  int pitch;        /* doubles per memory row                                 */
  int slab;         /* doubles per 2-D plane = pitch * (nj + 2*halo + 1)      */
  /* Position of this tile in the global (ni_glob x nj_glob) domain and the
   * 2-D processor layout (MOM_domains LAYOUT = npx,npy).                     */
  int i_glob0, j_glob0;       /* global index of local (0,0)                  */
  int ni_glob, nj_glob;
  int reentrant_x, reentrant_y; /* REENTRANT_X / REENTRANT_Y                  */
} mom6x_dims;

/* Fill ioff/joff/pitch/slab from ni,nj,nk,halo.  Returns 0 on success.       */
int mom6x_dims_init(mom6x_dims *d, int ni, int nj, int nk, int halo);

/* ------------------------------------------------------------------------- */
/* Grid metrics: ocean_grid_type (src/core/MOM_grid.F90:30-200).  All planes
 * live in ONE contiguous block `metrics[MOM6X_G_COUNT][slab]`.               */
enum mom6x_metric {
  MOM6X_G_mask2dT = 0, MOM6X_G_mask2dCu, MOM6X_G_mask2dCv, MOM6X_G_mask2dBu,
  MOM6X_G_dxT, MOM6X_G_dyT, MOM6X_G_IdxT, MOM6X_G_IdyT,
  MOM6X_G_dxCu, MOM6X_G_dyCu, MOM6X_G_IdxCu, MOM6X_G_IdyCu,
  MOM6X_G_dxCv, MOM6X_G_dyCv, MOM6X_G_IdxCv, MOM6X_G_IdyCv,
  MOM6X_G_dxBu, MOM6X_G_dyBu, MOM6X_G_IdxBu, MOM6X_G_IdyBu,
  MOM6X_G_areaT, MOM6X_G_IareaT, MOM6X_G_areaBu, MOM6X_G_IareaBu,
  MOM6X_G_areaCu, MOM6X_G_areaCv, MOM6X_G_IareaCu, MOM6X_G_IareaCv,
  MOM6X_G_dy_Cu, MOM6X_G_dx_Cv,     /* open face widths                      */
  MOM6X_G_bathyT,                   /* depth of the bottom, positive down [Z] */
  MOM6X_G_CoriolisBu, MOM6X_G_Coriolis2Bu,
  MOM6X_G_COUNT
};

/* ------------------------------------------------------------------------- */
/* verticalGrid_type (src/core/MOM_verticalGrid.F90:25-90) -- scalars only;
 * Rlay/g_prime are passed as arrays of nk doubles where needed.              */
typedef struct mom6x_vgrid {
  double g_Earth;        /* G_EARTH [L2 Z-1 T-2]                              */
  double Rho0;           /* RHO_0                                             */
  double Angstrom_H;     /* ANGSTROM in thickness units                       */
  double H_subroundoff;  /* verticalGrid.F90:214-218                          */
  double dZ_subroundoff;
  double H_to_Z, Z_to_H, H_to_RZ, RZ_to_H;  /* all 1 / 1 / Rho0 / 1/Rho0 (Bouss) */
  int    Boussinesq;     /* only 1 is supported            */
} mom6x_vgrid;

/* ------------------------------------------------------------------------- */
/* continuity_PPM_CS (src/core/MOM_continuity_PPM.F90:35-68); defaults from
 * continuity_PPM_init :2674-2754.                                            */
typedef struct mom6x_continuity_params {
  int    upwind_1st;       /* UPWIND_1ST_CONTINUITY      (F)                  */
  int    monotonic;        /* MONOTONIC_CONTINUITY       (F)                  */
  int    simple_2nd;       /* SIMPLE_2ND_PPM_CONTINUITY  (F)                  */
  double tol_eta;          /* ETA_TOLERANCE  (0.5*NK*ANGSTROM)                */
  double tol_vel;          /* VELOCITY_TOLERANCE (3e8 m/s)                    */
  double CFL_limit_adjust; /* CONTINUITY_CFL_LIMIT (0.5)                      */
  int    aggress_adjust;   /* CONT_PPM_AGGRESS_ADJUST (F).  With this or vol_CFL set the thread-per-column kernels run, whose column sums are in the reference's order whatever sum_order says */
  int    vol_CFL;          /* CONT_PPM_VOLUME_BASED_CFL (default = aggress_adjust, MOM_continuity_PPM.F90:2728) */
  int    better_iter;      /* CONT_PPM_BETTER_ITER (T)                        */
  int    use_visc_rem_max; /* CONT_PPM_USE_VISC_REM_MAX (T)                   */
  int    marginal_faces;   /* CONT_PPM_MARGINAL_FACE_AREAS (T)                */
  int    sum_order;        /* order of the column (k) sums of zonal/meridional_mass_flux, *_flux_adjust and
                            * set_*_BT_cont -- no counterpart in the reference, which sums sequentially in k:
                            *   MOM6X_SUM_REFERENCE (0): the reference's order, results bit-identical to the Fortran loop nest;
                            *   MOM6X_SUM_TREE16    (1): sixteen partial sums over k = q, q+16, q+32, ... (q = 0..15, each
                            *     in increasing k) combined by a balanced binary tree in q order, and the duL / duR
                            *     recurrences of set_*_BT_cont (:1293-1316) taken as the min / max they compute in exact
                            *     arithmetic.  Results agree with the reference order to round-off (<= 1e-13 of each
                            *     field's range); this is the order of a 16-lane wavefront row and lets one wavefront own a
                            *     face column (continuity_wave.hip).  Requires nk <= 128.
                            *   MOM6X_SUM_TREE16_FMA (2): the sums of TREE16, and fused multiply-adds at FIXED sites (what a Fortran
                            *     compiler's -ffp-contract does where it pleases): a + CFL * (p + q * r) of *_flux_layer and
                            *     *_flux_thickness (:936-955, :1017-1030) as fma(CFL, fma(q, r, p), a); u + du * visc_rem as
                            *     fma(du, visc_rem, u); the masked neighbours and the edge values of PPM_reconstruction_x/y
                            *     (:2396-2405).  oracle/orc_continuity.c restates the same sites; against the reference order the
                            *     results agree to round-off (bound and 10-step drift: tests/test_sum_order_gpu.py).  The default of the
                            *     hosts since round 6 (mom6_amd/abi.py default_sum_order, MOM6X_CONTINUITY_SUMS of the Fortran shim).       */
} mom6x_continuity_params;
#define MOM6X_SUM_REFERENCE  0
#define MOM6X_SUM_TREE16     1
#define MOM6X_SUM_TREE16_FMA 2

/* BT_cont_type (src/core/MOM_variables.F90:315-350): 12 2-D planes + h_u,h_v.
 * Any pointer may be NULL only if the whole struct pointer is NULL.          */
typedef struct mom6x_BT_cont {
  double *FA_u_EE, *FA_u_E0, *FA_u_W0, *FA_u_WW, *uBT_WW, *uBT_EE;
  double *FA_v_NN, *FA_v_N0, *FA_v_S0, *FA_v_SS, *vBT_SS, *vBT_NN;
  double *h_u, *h_v;   /* 3-D; may be NULL (allocated(BT_cont%h_u) false)     */
} mom6x_BT_cont;

/* barotropic_CS parameters (src/core/MOM_barotropic.F90:108-330, read in
 * barotropic_init :5403-5713).  Only the default-path flags listed in
 * SURVEY.md 8(b.1) are implemented; others must keep their default.          */
typedef struct mom6x_barotropic_params {
  double bebt;                 /* BEBT (0.1)                                  */
  double dtbt;                 /* CS%dtbt [T]: the stable BT step (set_dtbt)  */
  double dt_bt_filter;         /* DT_BT_FILTER (-0.25)                        */
  int    BT_project_velocity;  /* BT_PROJECT_VELOCITY (F)                     */
  int    Sadourny;             /* SADOURNY (T)                                */
  int    strong_drag;          /* BT_STRONG_DRAG (F)                          */
  int    wt_uv_bug;            /* VISC_REM_BT_WEIGHT_BUG (T)                  */
  int    use_old_coriolis_bracket_bug; /* BT_USE_OLD_CORIOLIS_BRACKET_BUG (F) */
  int    visc_rem_u_uh0;       /* BT_USE_VISC_REM_U_UH0 (F)                   */
  int    clip_velocity;        /* CLIP_BT_VELOCITY (F)                        */
  double CFL_trunc;            /* CFL_TRUNCATE (0.5)                          */
  double vel_underflow;        /* VEL_UNDERFLOW (0)                           */
  double G_extra;              /* G_BT_EXTRA (0)                              */
  double BT_Coriolis_scale;    /* BT_CORIOLIS_SCALE (1)                       */
  double maxCFL_BT_cont;       /* MAXCFL_BT_CONT (0.25)                       */
  int    bound_BT_corr;        /* BOUND_BT_CORRECTION (F)                     */
  int    BT_cont_bounds;       /* BT_CONT_CORR_BOUNDS (T)                     */
  double dtbt_fraction;        /* |DTBT| when DTBT<0 (0.98)                   */
  double Z_ref;                /* G%Z_ref (0)                                 */
  /* The wide-halo cycle of btstep (MOM_barotropic.F90:766-794, :2505-2512): with BT_USE_WIDE_HALOS the solver marches in from its
   * wide halo and exchanges every (halo / stencil) sub-steps.  On the device the barotropic domain's halo IS the context's halo
   * (mom6x_dims.halo): a host that wants BTHALO > NIHALO creates the context with halo = max(NIHALO, BTHALO) (and may keep the
   * 3-D passes at NIHALO rows: mom6x_set_dyn_pass_width).  The answers do not depend on any of the three.                    */
  int    use_wide_halos;       /* BT_USE_WIDE_HALOS (T); F: an exchange every sub-step                                       */
  int    BTHALO;               /* BTHALO (0): refused if it exceeds the context's halo                                        */
  int    min_stencil;          /* BT_WIDE_HALO_MIN_STENCIL (0)                                                                */
  /* Without a BT_cont_type (btstep's BT_cont argument NULL; USE_BT_CONT_TYPE = False), ABI 5: the face areas of the linear barotropic
   * continuity equation from the bathymetry alone, or (NONLINEAR_BT_CONTINUITY) from bathymetry + eta, recomputed every
   * NONLIN_BT_CONT_UPDATE_PERIOD sub-steps with a stencil of 2 (MOM_barotropic.F90:767-768, :1131-1136, :2539-2543, :5146-5237).     */
  int    nonlinear_continuity;       /* NONLINEAR_BT_CONTINUITY (F); ignored when BT_cont is given                                  */
  int    nonlin_cont_update_period;  /* NONLIN_BT_CONT_UPDATE_PERIOD (1); 0: the face areas of the step's first eta throughout      */
  /* ABI 6.  BT_THICK_SCHEME (:5566-5591) for btcalc calls without h_u / h_v: MOM6X_BT_THICK_*.  FROM_BT_CONT (the reference's default)
   * needs the BT_cont_type: btcalc without h_u then falls back to HYBRID only where the reference passes may_use_default (:4419-4430,
   * barotropic_init :6127), and the step without a BT_cont_type refuses it as barotropic_init does (:5589-5591).                       */
  int    bt_thick_scheme;
  /* MAXVEL (3e8 m/s) [L T-1]: only in eta_cor_bound = IareaT 0.1 maxvel (Datu + Datu + Datv + Datv) (:6164-6173), the bound of
   * BOUND_BT_CORRECTION when the BT_cont fits are not used for it (no BT_cont_type, or BT_CONT_CORR_BOUNDS = False) :1582-1585.       */
  double maxvel;
} mom6x_barotropic_params;
#define MOM6X_BT_THICK_FROM_BT_CONT 0
#define MOM6X_BT_THICK_HYBRID       1
#define MOM6X_BT_THICK_HARMONIC     2
#define MOM6X_BT_THICK_ARITHMETIC   3

/* CoriolisAdv_CS (src/core/MOM_CoriolisAdv.F90:29-100; CoriolisAdv_init :1054).            */
enum mom6x_coriolis_scheme {       /* CORIOLIS_SCHEME, values as in MOM_CoriolisAdv.F90:82-90       */
  MOM6X_SADOURNY75_ENERGY = 1,     /* default                                                   */
  MOM6X_ARAKAWA_HSU90 = 2,
  MOM6X_ROBUST_ENSTRO = 3,         /* :687-714, :808-838; switches CORIOLIS_EN_DIS and BOUND_CORIOLIS off (:1118, :1158) */
  MOM6X_SADOURNY75_ENSTRO = 4,
  MOM6X_ARAKAWA_LAMB81 = 5,        /* :534-542 + the ep_u / ep_v terms :716-721, :840-845       */
  MOM6X_AL_BLEND = 6               /* ARAKAWA_LAMB_BLEND :543-588                               */
};
enum mom6x_ke_scheme { MOM6X_KE_ARAKAWA = 10, MOM6X_KE_SIMPLE_GUDONOV = 11, MOM6X_KE_GUDONOV = 12 };
enum mom6x_pv_adv_scheme { MOM6X_PV_ADV_CENTERED = 21, MOM6X_PV_ADV_UPWIND1 = 22 };   /* MOM_CoriolisAdv.F90:115-119 */
typedef struct mom6x_coriolis_params {
  int Coriolis_Scheme;   /* SADOURNY75_ENERGY                                                   */
  int KE_Scheme;         /* KE_ARAKAWA                                                          */
  int bound_Coriolis;    /* BOUND_CORIOLIS (F; tc1/p0: T)                                       */
  int no_slip;           /* NOSLIP (F)                                                          */
  int Coriolis_En_Dis;   /* CORIOLIS_EN_DIS (F): the energy-dissipating biased SADOURNY75_ENERGY scheme  */
  int PV_Adv_Scheme;     /* PV_ADV_SCHEME (PV_ADV_CENTERED); read by ROBUST_ENSTRO only; 0 = the default   */
  double F_eff_max_blend;/* CORIOLIS_BLEND_F_EFF_MAX (4.0), ARAKAWA_LAMB_BLEND only                      */
  double wt_lin_blend;   /* CORIOLIS_BLEND_WT_LIN (0.125), ARAKAWA_LAMB_BLEND only; clipped to [1e-16, 1] as :1139 does */
} mom6x_coriolis_params;

/* PressureForce_FV_CS (src/core/MOM_PressureForce_FV.F90:40-110; PressureForce_FV_init :2020). */
typedef struct mom6x_pgf_params {
  double rho_ref;        /* RHO_PGF_REF (= RHO_0)                                               */
  int    rho_ref_bug;    /* RHO_PGF_REF_BUG (T)                                                 */
  double Z_ref;          /* G%Z_ref (0)                                                         */
} mom6x_pgf_params;

/* vertvisc_CS (src/parameterizations/vertical/MOM_vert_friction.F90:39-180; vertvisc_init :3135).  The
 * branches of vertvisc_coef / find_coupling_coef that are on the device path: HARMONIC_VISC on or off,
 * BOTTOMDRAGLAW on (visc%Kv_bbl_u/v, visc%bbl_thick_u/v from set_viscous_BBL are inputs) or off
 * (KV_EXTRA_BBL), KV_ML_INVZ2, visc%Kv_shear.  Ice shelves, open boundaries, GL90, visc%Kv_shear_Bu and
 * the dynamic / law-of-the-wall mixed-layer viscosities (DYNAMIC_VISCOUS_ML, FIXED_DEPTH_LOTW_ML,
 * LOTW_VISCOUS_ML_FLOOR, a bulk mixed layer) are not.                                               */
typedef struct mom6x_vertvisc_params {
  double Kv;            /* KV           [H Z T-1]                                                 */
  double Kvml_invZ2;    /* KV_ML_INVZ2  (0)                                                       */
  double Hmix;          /* HMIX_FIXED   [Z]                                                       */
  double Hbbl;          /* HBBL         [Z]                                                       */
  double harm_BL_val;   /* HARMONIC_BL_SCALE (0)                                                  */
  double Kv_extra_bbl;  /* KV_EXTRA_BBL (0); only without BOTTOMDRAGLAW                           */
  int    harmonic_visc; /* HARMONIC_VISC (F)                                                      */
  int    bottomdraglaw; /* BOTTOMDRAGLAW (T)                                                      */
  int    answer_date;   /* VERT_FRICTION_ANSWER_DATE (99991231)                                   */
} mom6x_vertvisc_params;

/* hor_visc_CS (src/parameterizations/lateral/MOM_hor_visc.F90:36-259; hor_visc_init :2322).  On the device path:
 * LAPLACIAN and/or BIHARMONIC with constant / velocity-scale / time-scale background coefficients, SMAGORINSKY_KH,
 * SMAGORINSKY_AH (+ BOUND_CORIOLIS_BIHARM), BOUND_KH / BOUND_AH in both the "better" and the legacy form,
 * ADD_LES_VISCOSITY, USE_LAND_MASK_FOR_HVISC, NOSLIP (Laplacian only, as in the reference).  Not on it (rejected):
 * Leith / Leith+E, MEKE, GME, backscatter, anisotropic viscosity, KH_SIN_LAT, 2-D background files, RE_AH,
 * USE_CONT_THICKNESS, open boundaries, the FrictWork diagnostics.                                              */
typedef struct mom6x_hor_visc_params {
  int    Laplacian;        /* LAPLACIAN (F)                                                     */
  int    biharmonic;       /* BIHARMONIC (T)                                                    */
  double Kh;               /* KH (0)              [L2 T-1]                                      */
  double Kh_bg_min;        /* KH_BG_MIN (0)                                                     */
  double Kh_vel_scale;     /* KH_VEL_SCALE (0)    [L T-1]                                       */
  int    Smagorinsky_Kh;   /* SMAGORINSKY_KH (F)                                                */
  double Smag_Lap_const;   /* SMAG_LAP_CONST (0)                                                */
  int    bound_Kh;         /* BOUND_KH (T)                                                      */
  int    better_bound_Kh;  /* BETTER_BOUND_KH (= BOUND_KH)                                      */
  int    add_LES_viscosity;/* ADD_LES_VISCOSITY (F)                                             */
  double Ah;               /* AH (0)              [L4 T-1]                                      */
  double Ah_vel_scale;     /* AH_VEL_SCALE (0)                                                  */
  double Ah_time_scale;    /* AH_TIME_SCALE (0)                                                 */
  int    Smagorinsky_Ah;   /* SMAGORINSKY_AH (F)                                                */
  double Smag_bi_const;    /* SMAG_BI_CONST (0)                                                 */
  int    bound_Ah;         /* BOUND_AH (T)                                                      */
  int    better_bound_Ah;  /* BETTER_BOUND_AH (= BOUND_AH)                                      */
  int    bound_Coriolis;   /* BOUND_CORIOLIS_BIHARM (= BOUND_CORIOLIS, F); only with SMAGORINSKY_AH */
  double bound_Cor_vel;    /* BOUND_CORIOLIS_VEL (= MAXVEL = 3e8)                               */
  int    use_land_mask;    /* USE_LAND_MASK_FOR_HVISC (T)                                       */
  double bound_coef;       /* HORVISC_BOUND_COEF (0.8)                                          */
  int    no_slip;          /* NOSLIP (F)                                                        */
  int    backscatter_underbound; /* BACKSCATTER_UNDERBOUND (T)                                  */
  double dt;               /* DT: the time step the stability bounds are made for (:2714)       */
  /* Leith (1996) viscosities from the gradient of the vertical vorticity (:80-85, :987-1113);
   * USE_LEITHY (Leith+E) and USE_QG_LEITH_VISC are not carried                                  */
  int    Leith_Kh;         /* LEITH_KH (F); needs LAPLACIAN                                     */
  double Leith_Lap_const;  /* LEITH_LAP_CONST (0)                                               */
  int    Leith_Ah;         /* LEITH_AH (F); needs BIHARMONIC                                    */
  double Leith_bi_const;   /* LEITH_BI_CONST (0)                                                */
  int    modified_Leith;   /* MODIFIED_LEITH (F): add the gradient of the divergence            */
  int    use_beta_in_Leith;/* USE_BETA_IN_LEITH (= LEITH_KH): grad f (G%dF_dx, G%dF_dy of
                            * MOM_calculate_grad_Coriolis, MOM_shared_initialization.F90:91) joins grad(vorticity) */
} mom6x_hor_visc_params;

/* tv%eqn_of_state (EOS_type, src/equation_of_state/MOM_EOS.F90:99-150) and the switches of
 * PressureForce_FV_CS that only matter with an equation of state.  Analytic density integrals
 * (analytic_int_density_dz, MOM_EOS.F90:1384) exist for EOS_LINEAR and the WRIGHT family: LINEAR, WRIGHT (the default,
 * MOM_EOS_Wright.F90), WRIGHT_FULL (MOM_EOS_Wright_full.F90) and WRIGHT_REDUCED (MOM_EOS_Wright_red.F90) are carried, each
 * with the analytic integrals and, with EOS_QUADRATURE or a pressure reconstruction, the generic quadratures.  UNESCO
 * (MOM_EOS_UNESCO.F90) has no analytic integrals: it needs EOS_QUADRATURE or a pressure reconstruction, as in the reference
 * ("No analytic integration option is available with this EOS!"); likewise ROQUET_RHO (= NEMO, MOM_EOS_Roquet_rho.F90).
 * JACKETT_06 (MOM_EOS_Jackett06.F90) and ROQUET_SPV (MOM_EOS_Roquet_SpV.F90).  TEOS10 (the gsw library, whose sources are not in
 * the reference tree) is refused.   */
enum mom6x_eos_form { MOM6X_EOS_LINEAR = 1, MOM6X_EOS_WRIGHT = 2, MOM6X_EOS_WRIGHT_FULL = 3, MOM6X_EOS_WRIGHT_REDUCED = 4, MOM6X_EOS_UNESCO = 5,
                      MOM6X_EOS_ROQUET_RHO = 6, /* "ROQUET_RHO" = "NEMO": MOM_EOS_Roquet_rho.F90; quadratures only, like UNESCO */
                      MOM6X_EOS_JACKETT06 = 7,  /* "JACKETT_06": MOM_EOS_Jackett06.F90; quadratures only ("JACKETT_MCD" is UNESCO) */
                      MOM6X_EOS_ROQUET_SPV = 8  /* "ROQUET_SPV": MOM_EOS_Roquet_SpV.F90; quadratures only */ };
typedef struct mom6x_eos_params {
  int    form;            /* EQN_OF_STATE                                                       */
  double Rho_T0_S0;       /* RHO_T0_S0 (1000)  } EOS_LINEAR                                     */
  double dRho_dT;         /* DRHO_DT (-0.2)    }                                                */
  double dRho_dS;         /* DRHO_DS (0.8)     }                                                */
  double dRho_dp;         /* linear_EOS%dRho_dp (0)                                             */
  int    MassWghtInterp;  /* bit 0: MASS_WEIGHT_IN_PRESSURE_GRADIENT, bit 1: ..._TOP (F, F)     */
  int    use_SSH_in_Z0p;  /* SSH_IN_EOS_PRESSURE_FOR_PGF (F)                                    */
  /* With ALE (USE_REGRIDDING): RECONSTRUCT_FOR_PRESSURE + PRESSURE_RECONSTRUCTION_SCHEME (PressureForce_FV_init :2172-2184)  */
  int    Recon_Scheme;    /* 0: layer-mean T, S (analytic integrals); 1: PLM edge values (TS_PLM_edge_values, MOM_ALE.F90:1495)
                           *    and the 5-point quadrature of int_density_dz_generic_plm (MOM_density_integrals.F90:418);
                           *    2: PPM edge values (TS_PPM_edge_values, MOM_ALE.F90:1581: edge_values_implicit_h4 + PPM_reconstruction)
                           *    and int_density_dz_generic_ppm (:874); needs NK >= 4                    */
  int    boundary_extrap; /* BOUNDARY_EXTRAPOLATION_PRESSURE (T)                                   */
  int    MassWghtInterpVanOnly; /* MASS_WEIGHT_IN_PGF_VANISHED_ONLY (F)                            */
  double h_nonvanished;   /* RESET_INTXPA_H_NONVANISHED (1e-6 m) [H]: the thickness below which a side counts as vanished */
  int    EOS_quadrature;  /* EOS_QUADRATURE (F, MOM_EOS.F90:1654): with Recon_Scheme = 0 the layer integrals come from the 5-point
                           * quadratures of int_density_dz_generic_pcm (MOM_density_integrals.F90:108) instead of the analytic forms */
} mom6x_eos_params;

/* MOM_dyn_split_RK2_CS parameters (src/core/MOM_dynamics_split_RK2.F90:85-273, read in
 * initialize_dyn_split_RK2 :1427-1495).                                                     */
typedef struct mom6x_rk2_params {
  double be;                   /* BE (0.6)                                                    */
  double begw;                 /* BEGW (0)                                                    */
  int    split_bottom_stress;  /* SPLIT_BOTTOM_STRESS (F)                                     */
  int    BT_use_layer_fluxes;  /* BT_USE_LAYER_FLUXES (T) -- only T supported                 */
  int    store_CAu;            /* STORE_CORIOLIS_ACCEL (T) -- only T supported                */
  int    visc_rem_dt_bug;      /* VISC_REM_TIMESTEP_BUG (T with ENABLE_BUGS_BY_DEFAULT)       */
  int    remap_aux;            /* REMAP_AUXILIARY_VARS (F): mom6x_remap_dyn_split_RK2_aux_vars */
  int    no_BT_cont;           /* 1: USE_BT_CONT_TYPE = False (MOM_barotropic.F90 barotropic_init: CS%BT_cont stays unassociated) --
                                * RK2.F90:467-469, :627, :644-652, :662-668, :867: btcalc from h (BT_THICK_SCHEME = HYBRID), the
                                * continuity calls and both btstep calls without a BT_cont_type, set_dtbt with eta (ABI 5)    */
} mom6x_rk2_params;

/* ------------------------------------------------------------------------- */
/* Context: owns the device copy of the metrics, the parameter structs, scratch
 * HBM, two HIP streams (compute + halo) and the RCCL communicator handle.    */
typedef struct mom6x_ctx mom6x_ctx;

/* Status codes.  The reference calls MOM_error(FATAL,...) (never returns); the
 * C side returns nonzero and mom6x_last_error() holds the message the Fortran
 * shim forwards to MOM_error.                                                */
#define MOM6X_OK          0
#define MOM6X_EINVAL      1
#define MOM6X_EHIP        2
#define MOM6X_EUNSUPPORTED 3
#define MOM6X_ENUMERIC    4   /* device-side flag (a NaN reached the thicknesses in continuity; NaN / overflow in the reproducing sums), returned by mom6x_ctx_sync */

const char *mom6x_last_error(void);
int  mom6x_abi_version(void);
/* hipGetDeviceCount: lets a host with one process per GPU pick its device as (local rank) mod (count).  < 0 on error. */
int  mom6x_device_count(void);
/* sizeof() of the public structs (0 dims, 1 vgrid, 2 continuity_params, 3 BT_cont, 4 barotropic_params,
 * 5 coriolis_params, 6 pgf_params, 7 rk2_params, 8 rk2_hooks, 9 eos_params, 10 vertvisc_params, 11 hor_visc_params,
 * 12 remapping_params, 13 regrid_zstar_params): lets ctypes / ISO_C_BINDING mirrors be checked at start-up.  */
int  mom6x_struct_size(int which);

/* Create a context for one tile on HIP device `device`.  `metrics_host` is a
 * HOST block of MOM6X_G_COUNT pitched planes (copied to HBM once; replaces the
 * `G` argument of every reference routine).                                  */
int mom6x_ctx_create(mom6x_ctx **ctx, const mom6x_dims *dims, int device,
                     const double *metrics_host, const mom6x_vgrid *GV,
                     int first_direction);
int mom6x_ctx_destroy(mom6x_ctx *ctx);
/* The stream all compute entry points launch on (a hipStream_t).            */
void *mom6x_ctx_stream(mom6x_ctx *ctx);
int  mom6x_ctx_sync(mom6x_ctx *ctx);
const mom6x_dims *mom6x_ctx_dims(const mom6x_ctx *ctx);
const double *mom6x_ctx_metrics_dev(const mom6x_ctx *ctx);

/* Per-kernel timing (HIP events on the compute stream around every launch).  Mirrors the
 * reference's cpu_clock_begin/end brackets (e.g. MOM_dynamics_split_RK2.F90:502/538) at kernel
 * granularity.  mom6x_prof_report writes "name<TAB>count<TAB>total_ms" lines into buf.       */
int mom6x_prof_enable(mom6x_ctx *ctx, int on);
int mom6x_prof_reset(mom6x_ctx *ctx);
int mom6x_prof_filter(mom6x_ctx *ctx, const char *prefix);  /* time only kernels named prefix*; NULL = all */
int mom6x_prof_report(mom6x_ctx *ctx, char *buf, int buflen);

/* Device memory helpers for non-torch hosts (Fortran).                       */
int mom6x_dev_alloc(mom6x_ctx *ctx, double **p, size_t n_doubles);
int mom6x_dev_free(mom6x_ctx *ctx, double *p);
int mom6x_dev_copy(mom6x_ctx *ctx, double *dst, const double *src, size_t n_doubles);   /* device -> device, on the context's stream */
/* Fortran array <-> pitched device array.  `stagger`: 0 = h-point
 * (SZI_,SZJ_), 1 = u-point (SZIB_,SZJ_), 2 = v-point (SZI_,SZJB_), 3 = q-point
 * (SZIB_,SZJB_), with MOM6 symmetric-memory extents; nk = 1 for 2-D.         */
int mom6x_upload(mom6x_ctx *ctx, double *dev, const double *host_f, int stagger, int nk);
int mom6x_download(mom6x_ctx *ctx, double *host_f, const double *dev, int stagger, int nk);

/* ------------------------------------------------------------------------- */
/* MOM_continuity_PPM                                                          */

int mom6x_continuity_init(mom6x_ctx *ctx, const mom6x_continuity_params *p);
  /* continuity_PPM_init, MOM_continuity_PPM.F90:2674                          */

/* Newton statistics of the mass-flux kernel (sum_order TREE16 only): out3[0] = flux re-evaluations inside
 * zonal/meridional_flux_adjust and set_*_BT_cont (whole-column sweeps, per wavefront of four face columns), out3[1] = Newton solves
 * (per wavefront), out3[2] = solves repeated with the exact CFL limits, accumulated while collection is on.  mode: 1 = switch
 * collection on and reset the counters, 0 = switch it off, -1 = just read.  The collecting variant of the kernel is slower (the
 * counters cost it registers): a diagnostic, not for timed runs.  Synchronises the context's stream.  out3 nullable.   */
int mom6x_continuity_stats(mom6x_ctx *ctx, int mode, unsigned long long *out3);

/* continuity_PPM(u, v, hin, h, uh, vh, dt, G, GV, US, CS, OBC, pbv, uhbt, vhbt,
 *   visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor, dv_cor)
 * MOM_continuity_PPM.F90:86-194.  Optional Fortran arguments are nullable
 * pointers; presence changes behaviour exactly as in the reference
 * (:590-592, :637, :737, :756).  OBC must be unassociated and pbv all ones
 * (USE_POROUS_BARRIER=False).  h may alias hin.                              */
int mom6x_continuity_PPM(mom6x_ctx *ctx,
    const double *u, const double *v, const double *hin, double *h,
    double *uh, double *vh, double dt,
    const double *uhbt, const double *vhbt,
    const double *visc_rem_u, const double *visc_rem_v,
    double *u_cor, double *v_cor, const mom6x_BT_cont *BT_cont,
    double *du_cor, double *dv_cor);

/* ------------------------------------------------------------------------- */
/* MOM_barotropic                                                              */

int mom6x_barotropic_init(mom6x_ctx *ctx, const mom6x_barotropic_params *p);
  /* barotropic_init, MOM_barotropic.F90:5301 (static fields :5784-5896,
   * :6146-6163).                                                              */

/* btcalc(h, G, GV, CS, h_u, h_v, may_use_default, OBC)  :4360.  h_u/h_v are the
 * BT_cont%h_u/h_v face thicknesses (BT_THICK_SCHEME=FROM_BT_CONT, the default);
 * when NULL the scheme of mom6x_barotropic_params.bt_thick_scheme (ARITHMETIC :4448, HYBRID :4453,
 * HARMONIC :4476) and, for FROM_BT_CONT, the HYBRID default of `may_use_default`.  Writes the
 * context's frhatu/frhatv.  mom6x_btcalc_strict is the call without may_use_default: FROM_BT_CONT
 * without h_u / h_v is the reference's "Inconsistent settings" error (:4426-4429).               */
int mom6x_btcalc(mom6x_ctx *ctx, const double *h, const double *h_u, const double *h_v);
int mom6x_btcalc_strict(mom6x_ctx *ctx, const double *h, const double *h_u, const double *h_v);

/* bt_mass_source(h, eta, set_cor, G, GV, CS)  :5243                           */
int mom6x_bt_mass_source(mom6x_ctx *ctx, const double *h, const double *eta, int set_cor);

/* set_dtbt(G, GV, US, CS, pbce=, gtot_est=, SSH_add=)  :3509.  Result goes to
 * the context's dtbt (and *dtbt_out if non-NULL).                            */
int mom6x_set_dtbt(mom6x_ctx *ctx, const double *pbce, double gtot_est, double SSH_add,
                   double *dtbt_out);

/* set_dtbt(G, GV, US, CS, pbce, eta=, SSH_add=) without a BT_cont argument, MOM_barotropic.F90:3576-3582: face areas
 * from find_face_areas(eta=eta) :5171-5186 when NONLINEAR_BT_CONTINUITY is set and eta is not NULL, otherwise from
 * find_face_areas(add_max=SSH_add) :5208-5219 (the deeper of the two neighbouring columns).  mom6x_set_dtbt_pbce is the call
 * of MOM_dynamics_split_RK2.F90:667 behind a BT_cont_type (= eta NULL, SSH_add 0).                                     */
int mom6x_set_dtbt_pbce_eta(mom6x_ctx *ctx, const double *pbce, const double *eta, double SSH_add, double *dtbt_out);
int mom6x_set_dtbt_pbce(mom6x_ctx *ctx, const double *pbce, double *dtbt_out);

/* btstep(U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, forces, pbce, eta_PF_in,
 *   U_Cor, V_Cor, accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, G, GV, US,
 *   CS, visc_rem_u, visc_rem_v, SpV_avg, ADp, OBC, BT_cont, eta_PF_start, taux_bot,
 *   tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav)          MOM_barotropic.F90:455-459.
 * forces%taux/tauy are passed as 2-D planes; SpV_avg/ADp/OBC/eta_PF_start are
 * not supported (must be absent in the caller); taux_bot/tauy_bot, the uh0 quad
 * and etaav are nullable.  Mutates the context's ubtav, vbtav, eta_cor.      */
int mom6x_btstep(mom6x_ctx *ctx,
    const double *U_in, const double *V_in, const double *eta_in, double dt,
    const double *bc_accel_u, const double *bc_accel_v,
    const double *taux, const double *tauy, const double *pbce,
    const double *eta_PF_in, const double *U_Cor, const double *V_Cor,
    double *accel_layer_u, double *accel_layer_v, double *eta_out,
    double *uhbtav, double *vhbtav,
    const double *visc_rem_u, const double *visc_rem_v,
    const mom6x_BT_cont *BT_cont,
    const double *taux_bot, const double *tauy_bot,
    const double *uh0, const double *vh0, const double *u_uh0, const double *v_vh0,
    double *etaav);

/* The warnings btstep issues for an unphysical sea surface height ("btstep: eta has
 * dropped below bathyT", MOM_barotropic.F90:2738-2745; the reference prints the first two
 * per call and counts the rest).  The device counts them over all sub-steps since the last
 * reset and keeps the first: info[4] = eta [H], -bathyT [Z], i, j (tile indices, 0-based).
 * Synchronises the context's stream.  `info` is nullable.                          */
int mom6x_btstep_warnings(mom6x_ctx *ctx, int reset, long long *count, double *info);

/* Access to barotropic_CS state that MOM_restart registers by pointer
 * (register_barotropic_restarts :6253: ubtav, vbtav) and that tests inspect.
 * `which`: 0 ubtav, 1 vbtav, 2 eta_cor, 3 frhatu (3-D), 4 frhatv (3-D),
 * 5 IDatu, 6 IDatv.  Returns a device pointer owned by the context.          */
double *mom6x_barotropic_field(mom6x_ctx *ctx, int which);
/* CS%dtbt, the scalar restart variable "DTBT" (register_barotropic_restarts :6290): *get (nullable) receives it, *set
 * (nullable, > 0) replaces it -- a restarted run keeps the file's DTBT as barotropic_init :5962-5970 does.         */
int mom6x_barotropic_dtbt(mom6x_ctx *ctx, double *get, const double *set);

/* ------------------------------------------------------------------------- */
/* MOM_CoriolisAdv                                                             */
int mom6x_CoriolisAdv_init(mom6x_ctx *ctx, const mom6x_coriolis_params *p);
  /* CoriolisAdv_init, MOM_CoriolisAdv.F90:1054                                */
/* CorAdCalc(u, v, h, uh, vh, CAu, CAv, OBC, AD, G, GV, US, CS, pbv, Waves)   :125.
 * OBC unassociated, no Stokes drift, pbv == 1.  Input halos as documented at :229-233. */
int mom6x_CorAdCalc(mom6x_ctx *ctx, const double *u, const double *v, const double *h,
                    const double *uh, const double *vh, double *CAu, double *CAv);

/* ------------------------------------------------------------------------- */
/* MOM_PressureForce (dispatcher :41 -> PressureForce_FV_Bouss, FV.F90:947)    */
int mom6x_PressureForce_init(mom6x_ctx *ctx, const mom6x_pgf_params *p, const double *Rlay,
                             const double *g_prime);
  /* PressureForce_init :85; Rlay/g_prime are HOST arrays of nk doubles
   * (GV%Rlay, GV%g_prime of verticalGrid_type).                               */
/* PressureForce(h, tv, PFu, PFv, G, GV, US, CS, ALE_CSp, ADp, p_atm, pbce, eta).  Layered
 * (tv%eqn_of_state unassociated) Boussinesq path; p_atm absent; pbce/eta nullable.       */
int mom6x_PressureForce(mom6x_ctx *ctx, const double *h, double *PFu, double *PFv,
                        double *pbce, double *eta);
/* The thermo_var_ptrs argument `tv` of PressureForce (:947): tv%T, tv%S (device, 3-D h-point arrays that
 * stay owned by the caller) and tv%eqn_of_state.  Once set, mom6x_PressureForce -- and the PressureForce
 * calls inside mom6x_step_dyn_split_RK2 -- take the use_EOS branch (:1206, :1289-1309: int_density_dz,
 * and Set_pbce_Bouss :692-722).  T == NULL dissociates tv%eqn_of_state again (layered path).  Bulk mixed
 * layers (GV%nk_rho_varies > 0) and ALE reconstructions are not on this path.                        */
/* ALE_PLM_edge_values(CS, G, GV, h, Q, bdry_extrap, Q_t, Q_b), MOM_ALE.F90:1520-1577: the values of a PLM reconstruction of the
 * tracer Q at the top and the bottom of every layer (TS_PLM_edge_values :1495 calls it for S and T); boundary cells PCM unless
 * bdry_extrap.  Answer dates >= 20190101 (h_neglect = GV%H_subroundoff).                                                   */
int mom6x_ALE_PLM_edge_values(mom6x_ctx *ctx, const double *h, const double *Q, int bdry_extrap, double *Q_t, double *Q_b);
/* One field of TS_PPM_edge_values (MOM_ALE.F90:1581): edge_values_implicit_h4 + PPM_reconstruction (+ PPM_boundary_extrapolation),
 * the edge values PRESSURE_RECONSTRUCTION_SCHEME = 2 hands to int_density_dz_generic_ppm.  NK >= 4. */
int mom6x_ALE_PPM_edge_values(mom6x_ctx *ctx, const double *h, const double *Q, int bdry_extrap, double *Q_t, double *Q_b);
int mom6x_PressureForce_set_tv(mom6x_ctx *ctx, const double *T, const double *S, const mom6x_eos_params *eos);

/* ------------------------------------------------------------------------- */
/* MOM_hor_visc                                                    */
/* hor_visc_init :2322: the 2-D coefficient fields (background and maximum viscosities, Smagorinsky
 * constants, metric products) are computed on the device from the metric block.                 */
int mom6x_hor_visc_init(mom6x_ctx *ctx, const mom6x_hor_visc_params *p);
/* horizontal_viscosity(u, v, h, uh, vh, diffu, diffv, MEKE, VarMix, G, GV, US, CS, tv, dt, ...) :266.
 * uh, vh only feed the FrictWork diagnostics and are not arguments here.  After mom6x_hor_visc_init,
 * mom6x_step_dyn_split_RK2 calls this itself at :886 (and the new-run initialisation at :1601) unless a
 * horizontal_viscosity callback is given.                                                      */
int mom6x_horizontal_viscosity(mom6x_ctx *ctx, const double *u, const double *v, const double *h,
                               double *diffu, double *diffv);

/* ------------------------------------------------------------------------- */
/* MOM_remapping / the remapping half of MOM_ALE                   */
/* remapping_CS (src/ALE/MOM_remapping.F90:47-84) as set by initialize_remapping :1654 / remapping_set_param :122.
 * On the device path: the OM4-era reconstruction functions PCM, PLM, PPM_H4, PPM_IH4 (build_reconstructions_1d :410) with
 * REMAPPING_ANSWER_DATE >= 20190101, with or without boundary extrapolation, both sub-cell integrators
 * (remap_src_to_sub_grid_om4 :845 / remap_src_to_sub_grid :962) and remap_sub_to_tgt_grid_om4 :1103.
 * Not on it (rejected): PPM_CW, the hybgen and PQM schemes, the Recon1d class ("C_*") schemes,
 * the 2018 answers, PCM_cell masks, check_reconstruction / check_remapping.                        */
enum mom6x_remap_scheme { MOM6X_REMAP_PCM = 0, MOM6X_REMAP_PLM = 2, MOM6X_REMAP_PPM_H4 = 4, MOM6X_REMAP_PPM_IH4 = 5 };   /* :86-96 */
typedef struct mom6x_remapping_params {
  int    scheme;                    /* REMAPPING_SCHEME: PCM, PLM (the module default), PPM_H4, PPM_IH4   */
  int    boundary_extrapolation;    /* REMAP_BOUNDARY_EXTRAP (type default .true.; ALE_init passes F)     */
  int    force_bounds_in_subcell;   /* REMAP_BOUND_INTERMEDIATE_VALUES (F)                                */
  int    force_bounds_in_target;    /* (T)                                                                */
  int    om4_remap_via_sub_cells;   /* REMAPPING_USE_OM4_SUBCELLS (type default F; ALE_init default T)    */
  int    answer_date;               /* REMAPPING_ANSWER_DATE; must be >= 20190101                         */
  double h_neglect;                 /* GV%H_subroundoff (or GV%kg_m2_to_H*1e-30 ...)                      */
  double h_neglect_edge;            /* = h_neglect for answer dates >= 20190101                           */
} mom6x_remapping_params;

/* ALE_remap_scalar-style column remap of `nfields` T-point fields in place (MOM_ALE.F90:760 ALE_remap_tracers,
 * without the diagnostics): for every wet column remapping_core_h(CS, nk, h_old, field, nk, h_new, field).   */
int mom6x_ALE_remap_tracers(mom6x_ctx *ctx, const mom6x_remapping_params *p, const double *h_old, const double *h_new,
                            double *const *fields, int nfields);
/* ALE_remap_set_h_vel :882 (no partial cells, no OBC): h_u = 0.5*(h(i)+h(i+1)) at open faces, else untouched.  */
int mom6x_ALE_remap_set_h_vel(mom6x_ctx *ctx, const double *h_new, double *h_u, double *h_v);
/* ALE_remap_velocities :1089 (REMAP_VEL_CONSERVE_KE off, no near-bottom masking, no diagnostics).              */
int mom6x_ALE_remap_velocities(mom6x_ctx *ctx, const mom6x_remapping_params *p, const double *h_old_u, const double *h_old_v,
                               const double *h_new_u, const double *h_new_v, double *u, double *v);
/* The three calls MOM.F90 makes in a row (ALE_regridding_and_remapping: ALE_remap_set_h_vel of the old and of the new grid, then
 * ALE_remap_velocities) as one, from the CELLS' thicknesses: the same bits; with OM4's switch set h_u / h_v are never stored. */
int mom6x_ALE_remap_velocities_from_h(mom6x_ctx *ctx, const mom6x_remapping_params *p, const double *h_old, const double *h_new,
                                      double *u, double *v);
/* ... with the KE-conserving correction of its baroclinic part (REMAP_VEL_CONSERVE_KE = True and allow_preserve_variance,
 * MOM_ALE.F90:1166-1195, :1240-1270): what MOM.F90 asks for inside the time step.                                          */
int mom6x_ALE_remap_velocities_conserve_ke(mom6x_ctx *ctx, const mom6x_remapping_params *p, const double *h_old_u,
                                           const double *h_old_v, const double *h_new_u, const double *h_new_v,
                                           double *u, double *v);
/* ALE_regrid (MOM_ALE.F90:518) -> regridding_main (MOM_regridding.F90:862) for REGRIDDING_ZSTAR without ice shelves
 * and with CS%nk == GV%ke (Boussinesq): nom_depth_H :920-922, build_zstar_grid :1257 (build_zstar_column,
 * coord_zlike.F90:63; filtered_grid_motion :1105 incl. the old-grid weight and its depth-dependent transition),
 * calc_h_new_by_dz :1008.  h needs one valid halo point (the reference regrids isc-1..iec+1).
 * coordinateResolution: nk host values (the nominal layer thicknesses ALE_COORDINATE_CONFIG gives, in Z units). */
typedef struct mom6x_regrid_zstar_params {
  double min_thickness;                 /* MIN_THICKNESS (0.001 m) [H]                                        */
  double old_grid_weight;               /* from REGRID_TIME_SCALE: exp(-dt/timescale) or 0 (:96)              */
  double depth_of_time_filter_shallow;  /* REGRID_FILTER_SHALLOW_DEPTH (0) [H]                                */
  double depth_of_time_filter_deep;     /* REGRID_FILTER_DEEP_DEPTH (0) [H]                                   */
  double Z_ref;                         /* G%Z_ref (0)                                                        */
} mom6x_regrid_zstar_params;
int mom6x_ALE_regrid_zstar(mom6x_ctx *ctx, const mom6x_regrid_zstar_params *p, const double *coordinateResolution,
                           const double *h, double *h_new, double *dzRegrid);

/* The density-following coordinate generators of regridding_main (MOM_regridding.F90:862): REGRIDDING_RHO
 * (build_rho_grid :1472 -> build_rho_column coord_rho.F90:92) and REGRIDDING_HYCOM1 (build_grid_HyCOM1 :1638 ->
 * build_hycom1_column coord_hycom.F90:106; Bleck 2002), CS%nk == GV%ke, no ice shelf.  Both find the positions of the
 * target interface densities in the column's (potential) density profile with regrid_interp.F90's
 * build_and_interpolate_grid :331 (regridding_set_ppolys :80, interpolate_grid :295, the Newton iteration of
 * get_polynomial_coordinate :376), blend old and new positions with filtered_grid_motion :1105 and hand back
 * dzInterface and h_new = calc_h_new_by_dz :1008.  REGRIDDING_RHO wants the column statically stable first
 * (regridding_preadjust_reqs :966): mom6x_ALE_convective_adjustment = convective_adjustment :1905.            */
enum mom6x_interp_scheme {            /* INTERPOLATION_SCHEME (regrid_interp.F90:38-49); others are not carried */
  MOM6X_INTERP_P1M_H2 = 0,            /* the default */
  MOM6X_INTERP_PLM = 3, MOM6X_INTERP_PPM_H4 = 5
};
typedef struct mom6x_regrid_rho_params {
  mom6x_regrid_zstar_params f;       /* MIN_THICKNESS, the time filter of filtered_grid_motion, G%Z_ref           */
  int    interp_scheme;              /* INTERPOLATION_SCHEME (P1M_H2)                                             */
  int    boundary_extrapolation;     /* BOUNDARY_EXTRAPOLATION (F)                                                */
  double ref_pressure;               /* P_REF (2e7 Pa): CS%ref_pressure of REGRIDDING_RHO, tv%P_Ref of HYCOM1     */
  double compressibility_fraction;   /* REGRID_COMPRESSIBILITY_FRACTION (0): HYCOM1 only                          */
  int    integrate_downward_for_e;   /* CS%integrate_downward_for_e (T): REGRIDDING_RHO only                      */
} mom6x_regrid_rho_params;
/* target_density: nk+1 host values (the interface densities set_target_densities :2297 makes from the layer ones).  */
int mom6x_ALE_regrid_rho(mom6x_ctx *ctx, const mom6x_regrid_rho_params *p, const mom6x_eos_params *eos,
                         const double *target_density, const double *h, const double *T, const double *S,
                         double *h_new, double *dzRegrid);
/* coordinateResolution: nk host values [Z]; max_interface_depths (nk+1) and max_layer_thickness (nk): host, nullable
 * (MAXIMUM_INT_DEPTH_CONFIG / MAX_LAYER_THICKNESS_CONFIG).  HYCOM1's "only improves" option is not carried.       */
int mom6x_ALE_regrid_hycom1(mom6x_ctx *ctx, const mom6x_regrid_rho_params *p, const mom6x_eos_params *eos,
                            const double *coordinateResolution, const double *target_density,
                            const double *max_interface_depths, const double *max_layer_thickness,
                            const double *h, const double *T, const double *S, double *h_new, double *dzRegrid);
/* convective_adjustment :1905: adjacent layers swap (h, T, S) until the density at the surface pressure increases downward */
int mom6x_ALE_convective_adjustment(mom6x_ctx *ctx, const mom6x_eos_params *eos, double *h, double *T, double *S);

/* remapping_core_h :234 for `ncol` independent columns stored back to back (n0 | n1 values each): the entry the
 * reference's own unit tests (remapping_unit_tests :2072) exercise.  Device pointers.                          */
int mom6x_remapping_core_h(mom6x_ctx *ctx, const mom6x_remapping_params *p, int ncol, int n0, const double *h0,
                           const double *u0, int n1, const double *h1, double *u1);

/* ------------------------------------------------------------------------- */
/* MOM_vert_friction                                                           */
/* The coupling coefficients CS%a_u, CS%a_v [(nk+1) levels], CS%h_u, CS%h_v and the optional
 * visc%Ray_u/Ray_v are produced by vertvisc_coef (:1357, not ported: SURVEY 8f-1); the host
 * hands their DEVICE copies to the context before vertvisc / vertvisc_remnant.             */
int mom6x_vertvisc_set_coef(mom6x_ctx *ctx, const double *a_u, const double *a_v,
                            const double *h_u, const double *h_v,
                            const double *Ray_u, const double *Ray_v);
/* vertvisc(u, v, h, forces, visc, dt, OBC, ADp, CDp, G, GV, US, CS, taux_bot, tauy_bot)  :557 */
/* vertvisc_init :3135: keeps the parameters and allocates CS%a_u, CS%a_v [(nk+1) levels], CS%h_u, CS%h_v on
 * the device; the context's coefficient set (mom6x_vertvisc_set_coef) then points at them.           */
int mom6x_vertvisc_init(mom6x_ctx *ctx, const mom6x_vertvisc_params *p);
/* The members of vertvisc_type (MOM_variables.F90) that vertvisc_coef / vertvisc read: visc%Kv_bbl_u/v and
 * visc%bbl_thick_u/v (2-D; required with BOTTOMDRAGLAW), visc%Kv_shear (3-D, nk+1 interfaces at h points,
 * nullable), visc%Ray_u/v (3-D, nullable).  Device arrays owned by the caller (set_viscous_BBL and the
 * shear-mixing schemes stay on the host).                                                            */
int mom6x_vertvisc_set_visc(mom6x_ctx *ctx, const double *Kv_bbl_u, const double *Kv_bbl_v, const double *bbl_thick_u,
                            const double *bbl_thick_v, const double *Kv_shear, const double *Ray_u, const double *Ray_v);
/* vertvisc_coef(u, v, h, dz, forces, visc, tv, dt, G, GV, US, CS, OBC, VarMix) :1357, with dz = H_to_Z*h
 * (thickness_to_dz, MOM_interface_heights.F90:855, Boussinesq).  Fills CS%a_u, a_v, h_u, h_v.  After
 * mom6x_vertvisc_init, mom6x_step_dyn_split_RK2 calls this itself at :609, :738 and :1003 unless a
 * vertvisc_coef callback is given.                                                                   */
int mom6x_vertvisc_coef(mom6x_ctx *ctx, const double *u, const double *v, const double *h, double dt);
/* CS%a_u (0), CS%a_v (1) [nk+1 levels], CS%h_u (2), CS%h_v (3) of the device vertvisc_CS (diagnostics, restarts of Kv_u). */
double *mom6x_vertvisc_field(mom6x_ctx *ctx, int which);
int mom6x_vertvisc(mom6x_ctx *ctx, double *u, double *v, const double *taux, const double *tauy,
                   double dt, double *taux_bot, double *tauy_bot);
/* DIRECT_STRESS / HMIX_STRESS (vertvisc_init :3208, :3258-3268; vertvisc :707-720, :958-971): with Hmix_stress > 0 the wind
 * stress is distributed as a body force over the topmost Hmix_stress [H] of fluid instead of entering as the surface
 * boundary condition.  h is vertvisc's third argument (device pointer, kept); mom6x_step_dyn_split_RK2 uses its own h. */
int mom6x_vertvisc_set_direct_stress(mom6x_ctx *ctx, double Hmix_stress, const double *h);
/* vertvisc_remnant(visc, visc_rem_u, visc_rem_v, dt, G, GV, US, CS)  :1229                   */
int mom6x_vertvisc_remnant(mom6x_ctx *ctx, double *visc_rem_u, double *visc_rem_v, double dt);

/* ------------------------------------------------------------------------- */
/* MOM_dynamics_split_RK2                                                      */

/* Host callbacks for the callees of step_MOM_dyn_split_RK2 that are NOT on the ported hot path
 * (SURVEY.md 8f).  All array arguments are DEVICE pointers.  A NULL callback means "keep what the
 * context already holds" (coefficients / diffu,diffv frozen over the step).                    */
typedef struct mom6x_rk2_hooks {
  void *user;
  /* set_viscous_ML + thickness_to_dz + vertvisc_coef (RK2.F90:602-609 stage 0, :737-738 stage 1,
   * :1002-1003 stage 2): must leave updated coefficients behind via mom6x_vertvisc_set_coef.   */
  int (*vertvisc_coef)(void *user, int stage, const double *u, const double *v, const double *h, double dt);
  /* horizontal_viscosity (RK2.F90:886): fills diffu, diffv.                                     */
  int (*horizontal_viscosity)(void *user, const double *u_av, const double *v_av, const double *h_av,
                              const double *uh, const double *vh, double *diffu, double *diffv);
} mom6x_rk2_hooks;

/* initialize_dyn_split_RK2 (RK2.F90:1346): allocates MOM_dyn_split_RK2_CS on the device (incl. the
 * BT_cont_type).  continuity, barotropic, CoriolisAdv, PressureForce must be initialised first,
 * as in :1552-1596.                                                                           */
int mom6x_initialize_dyn_split_RK2(mom6x_ctx *ctx, const mom6x_rk2_params *p);
/* The new-run fills of initialize_dyn_split_RK2 :1577-1650 (eta from h; u_av,v_av = u,v; h_av and
 * CAu_pred from one continuity + CorAdCalc pass).  A restarted run uploads the fields instead
 * (mom6x_rk2_field) and calls mom6x_rk2_set_CAu_pred_stored.                                   */
int mom6x_dyn_split_RK2_new_run(mom6x_ctx *ctx, const double *u, const double *v, const double *h,
                                double *uh, double *vh, double dt);
int mom6x_rk2_set_CAu_pred_stored(mom6x_ctx *ctx, int stored);
/* The same routine for a RESTARTED run (initialize_dyn_split_RK2 :1577-1668), field by field as the reference decides
 * with query_initialized: the host uploads the variables the restart file held (mom6x_rk2_field) and names them in
 * `have`; every other one is formed as the reference forms it -- eta from h (:1578-1590), diffu, diffv by
 * horizontal_viscosity (:1599-1606), u_av, v_av = u, v (:1608-1614), and, when CAu / CAv are absent, h_av from one
 * continuity call (or the file's uh, vh, h2: HAVE_UH + HAVE_H2, an older restart format) and CAu_pred, CAv_pred by
 * CorAdCalc (:1620-1640) -- followed by the group pass of :1670-1679.  have = 0 is mom6x_dyn_split_RK2_new_run.      */
#define MOM6X_RK2_HAVE_ETA   1   /* "sfc"            */
#define MOM6X_RK2_HAVE_DIFFU 2   /* "diffu", "diffv" */
#define MOM6X_RK2_HAVE_U2    4   /* "u2", "v2"       */
#define MOM6X_RK2_HAVE_CAU   8   /* "CAu", "CAv"     */
#define MOM6X_RK2_HAVE_UH    16  /* "uh", "vh"       */
#define MOM6X_RK2_HAVE_H2    32  /* "h2"             */
int mom6x_dyn_split_RK2_restart_fills(mom6x_ctx *ctx, const double *u, const double *v, const double *h,
                                      double *uh, double *vh, double dt, int have);
/* remap_dyn_split_RK2_aux_vars (RK2.F90:1302): after an ALE regridding the auxiliary restart variables move to the
 * new grid too -- u_av, v_av and CAu_pred, CAv_pred (STORE_CORIOLIS_ACCEL), then diffu, diffv, each pair with
 * ALE_remap_velocities and the first two followed by their pass_vector.  Returns at once unless REMAP_AUXILIARY_VARS.
 * p stands for ALE_CSp%vel_remapCS; the face thicknesses are those of mom6x_ALE_remap_set_h_vel.                  */
int mom6x_remap_dyn_split_RK2_aux_vars(mom6x_ctx *ctx, const mom6x_remapping_params *p, const double *h_old_u,
                                       const double *h_old_v, const double *h_new_u, const double *h_new_v);
/* Device pointers to the CS arrays MOM_restart registers (register_restarts_dyn_split_RK2 :1210:
 * sfc=eta, u2=u_av, v2=v_av, CAu, CAv, diffu, diffv) and the others, by name index:
 * 0 CAu, 1 CAv, 2 CAu_pred, 3 CAv_pred, 4 PFu, 5 PFv, 6 diffu, 7 diffv, 8 visc_rem_u, 9 visc_rem_v,
 * 10 u_accel_bt, 11 v_accel_bt, 12 u_av, 13 v_av, 14 h_av, 15 pbce, 16 eta, 17 eta_PF, 18 uhbt,
 * 19 vhbt, 20 taux_bot, 21 tauy_bot, 22 BT_cont%h_u, 23 BT_cont%h_v.                            */
double *mom6x_rk2_field(mom6x_ctx *ctx, int which);

/* step_MOM_dyn_split_RK2(u_inst, v_inst, h, tv, visc, Time_local, dt, forces, p_surf_begin,
 *   p_surf_end, uh, vh, uhtr, vhtr, eta_av, G, GV, US, CS, calc_dtbt, VarMix, MEKE,
 *   thickness_diffuse_CSp, pbv, STOCH, Waves)                       RK2.F90:294-296.
 * forces%taux/tauy are planes; tv (layered: no T,S), p_surf_*, VarMix, MEKE, STOCH, Waves absent. */
int mom6x_step_dyn_split_RK2(mom6x_ctx *ctx, double *u_inst, double *v_inst, double *h,
                             double *uh, double *vh, double *uhtr, double *vhtr, double *eta_av,
                             const double *taux, const double *tauy, double dt, int calc_dtbt,
                             const mom6x_rk2_hooks *hooks);

/* ------------------------------------------------------------------------- */
/* MOM_tracer_advect / MOM_diabatic_aux / MOM_tracer_diabatic                    */

/* TRACER_ADVECTION_SCHEME values (MOM_tracer_advect_schemes.F90:11-13)          */
#define MOM6X_ADVECT_PLM   0
#define MOM6X_ADVECT_PPMH3 1
#define MOM6X_ADVECT_PPM   2
/* tracer_advect_init (MOM_tracer_advect.F90:1155): DT, TRACER_ADVECTION_SCHEME ('PLM' default),
 * USE_HUYNH_STENCIL_BUG (F).                                                                       */
int mom6x_tracer_advect_init(mom6x_ctx *ctx, double dt_dyn, int default_scheme, int useHuynhStencilBug);
/* advect_tracer(h_end, uhtr, vhtr, OBC, dt, G, GV, US, CS, Reg, x_first_in, vol_prev, max_iter_in,
 *   update_vol_prev, uhr_out, vhr_out)                                MOM_tracer_advect.F90:53-54.
 * The registry Reg is an array of ntr device pointers (Reg%Tr(m)%t) with per-tracer schemes
 * (Reg%Tr(m)%advect_scheme; < 0 = the default).  x_first_in: -1 absent; max_iter_in: 0 absent;
 * uhr_out/vhr_out/iters_out nullable.  OBC unassociated; the offline-transport arguments (vol_prev,
 * update_vol_prev) are not supported.                                                              */
int mom6x_advect_tracer(mom6x_ctx *ctx, const double *h_end, const double *uhtr, const double *vhtr, double dt,
                        double *const *tracers, const int *schemes, int ntr, int x_first_in, int max_iter_in,
                        double *uhr_out, double *vhr_out, int *iters_out);
/* triDiagTS(G, GV, is, ie, js, je, hold, ea, eb, T, S)            MOM_diabatic_aux.F90:394; S may be NULL.
 * is..je are LOCAL 0-based indices (MOM6 is-isc etc.).                                             */
int mom6x_triDiagTS(mom6x_ctx *ctx, int is, int ie, int js, int je, const double *hold, const double *ea,
                    const double *eb, double *T, double *S);
/* triDiagTS_Eulerian(G, GV, is, ie, js, je, hold, ent, T, S)      :444; ent has nk+1 interfaces.     */
int mom6x_triDiagTS_Eulerian(mom6x_ctx *ctx, int is, int ie, int js, int je, const double *hold, const double *ent,
                             double *T, double *S);
/* tracer_vertdiff(h_old, ea, eb, dt, tr, G, GV, sfc_flux, btm_flux, btm_reservoir, sink_rate, convert_flux_in)
 * MOM_tracer_diabatic.F90:25 -- the branch without sink_rate (:181-214; btm_reservoir is only read with sink_rate).   */
int mom6x_tracer_vertdiff(mom6x_ctx *ctx, const double *h_old, const double *ea, const double *eb, double dt,
                          double *tr, const double *sfc_flux, const double *btm_flux, int convert_flux);
/* tracer_vertdiff_Eulerian(h_old, ent, dt, tr, G, GV, ...)         :224 (no-sink branch :382-414).     */
int mom6x_tracer_vertdiff_Eulerian(mom6x_ctx *ctx, const double *h_old, const double *ent, double dt, double *tr,
                                   const double *sfc_flux, const double *btm_flux, int convert_flux);
/* The same two routines WITH sink_rate (:123-179 / :315-380): the tracer sinks through the interfaces by up to sink_rate * dt,
 * limited so that characteristics do not cross within the step; with btm_reservoir (nullable = not present) the sinking is not
 * limited and what leaves the bottom layer is added to btm_reservoir [CU R Z].                                           */
int mom6x_tracer_vertdiff_sink(mom6x_ctx *ctx, const double *h_old, const double *ea, const double *eb, double dt,
                               double *tr, const double *sfc_flux, const double *btm_flux, double *btm_reservoir,
                               double sink_rate, int convert_flux);
int mom6x_tracer_vertdiff_Eulerian_sink(mom6x_ctx *ctx, const double *h_old, const double *ent, double dt, double *tr,
                                        const double *sfc_flux, const double *btm_flux, double *btm_reservoir,
                                        double sink_rate, int convert_flux);
/* diabatic (MOM_diabatic_driver.F90:277) is a host-side dispatcher over mixing physics that is out of
 * scope; its only device-relevant behaviour is the early return for GV%ke == 1 (:330) -- the solvers
 * above are what it (and the tracer packages, e.g. DOME_tracer.F90:338) call.                       */
int mom6x_diabatic_is_trivial(const mom6x_ctx *ctx);

/* ------------------------------------------------------------------------- */
/* MOM_domains: 2-D tile decomposition and halo updates over RCCL / xGMI         */

/* Index range (inclusive, local indices) of the region of a `stagger` field that is SENT to (send=1)
 * or RECEIVED from (send=0) the neighbour in direction dir = 0..7 (W,E,S,N,SW,SE,NW,NE).  Host-only,
 * no GPU needed: this is the exchange plan, tested on CPU with gloo.                              */
int mom6x_halo_region(const mom6x_dims *d, int stagger, int dir, int send, int *i0, int *i1, int *j0, int *j1);
/* Rank (px + npx*py) of the neighbour of tile (px,py) in direction dir, or -1 at a closed boundary. */
int mom6x_halo_neighbor(int npx, int npy, int px, int py, int dir, int reentrant_x, int reentrant_y);
/* A transport of the host's own instead of the process's RCCL: nine functions with RCCL's meaning (ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclAllReduce,
 * ncclGetErrorString), opaque handles as void*, return 0 for success.  For a host that wants its halo traffic in one
 * library (GPU-aware MPI: MOM6's own FMS domains) -- and how the layout tests run several tiles as threads of one
 * process on one GPU (tests/transport/).  Communicators made after the call use it; NULL returns to RCCL.           */
enum { MOM6X_T_INT32 = 0, MOM6X_T_INT64 = 1, MOM6X_T_FLOAT64 = 2 };
enum { MOM6X_OP_SUM = 0, MOM6X_OP_MIN = 1, MOM6X_OP_MAX = 2 };
typedef struct mom6x_transport {
  int (*get_unique_id)(char *id128);
  int (*comm_init_rank)(void **comm, int nranks, const char *id128, int rank);
  int (*comm_destroy)(void *comm);
  int (*send)(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream);
  int (*recv)(void *buf, size_t count, int dtype, int peer, void *comm, void *stream);
  int (*group_start)(void);
  int (*group_end)(void);
  int (*all_reduce)(const void *sendbuf, void *recvbuf, size_t count, int dtype, int op, void *comm, void *stream);
  const char *(*error_string)(int rc);       /* nullable */
} mom6x_transport;
int mom6x_comm_set_transport(const mom6x_transport *t);
/* ncclGetUniqueId on the calling rank: 128 bytes the host broadcasts (MPI_Bcast / torch.distributed). */
int mom6x_comm_unique_id(char *id128);
/* Attach LAYOUT = npx,npy (MOM_domains.F90:155) with this tile at (px,py) and create the RCCL
 * communicator (clone of MOM_domains_init / create_group_pass's message plan).  After this call every
 * halo update inside the mom6x_* routines is a packed ncclSend/ncclRecv group exchange.            */
int mom6x_comm_init(mom6x_ctx *ctx, int npx, int npy, int px, int py, const char *id128, int force_nccl_self);
int mom6x_comm_rank(const mom6x_ctx *ctx);
/* do_group_pass of n fields (pass_var / pass_vector, MOM_domain_infra.F90:171-560, :1141).          */
int mom6x_pass_fields(mom6x_ctx *ctx, double *const *fields, const int *staggers, const int *nks, int n);
/* The halo width of the 3-D fields in the RK2 step's own group passes (the reference's create_group_pass calls of
 * MOM_dynamics_split_RK2.F90 run on G%Domain, i.e. with NIHALO).  For a context whose halo was widened for the barotropic solver
 * -- BT_USE_WIDE_HALOS with BTHALO > NIHALO, MOM_barotropic.F90:5446-5461: create the context with halo = max(NIHALO, BTHALO);
 * btstep then exchanges every halo / stencil sub-steps (:2505-2512) -- this keeps the 3-D messages at NIHALO rows.
 * width = 0: the context's halo (the default); otherwise 4 <= width <= the context's halo.  2-D fields (eta) always travel at
 * the context's width: btstep reads them over its wide halo.                                                          */
int mom6x_set_dyn_pass_width(mom6x_ctx *ctx, int width);
/* btstep's own group pass of eta, ubt, vbt (pass_eta_ubt, MOM_barotropic.F90:2505-2512) started on the second stream -- packed on the
 * compute stream, messages and unpack behind it -- while the compute stream runs the half of the next sub-step that reads the tile's
 * own points only, completed before the other half (start_group_pass / complete_group_pass around interior work).  Off by default
 * (see halo.hip: it loses on the one-GPU model); the answers do not depend on it.                                       */
int mom6x_comm_overlap_btstep(mom6x_ctx *ctx, int on);
/* Packed group exchanges (one message per neighbour each) this tile has made since the last reset; reset != 0 clears the count. */
long long mom6x_comm_exchange_count(mom6x_ctx *ctx, int reset);
/* ... and the bytes this tile sent in them (all neighbours together): what the per-pass halo widths of the RK2 step
 * (create_group_pass(..., halo=), RK2.F90:484-495) save over NIHALO rows of every field.                             */
long long mom6x_comm_exchange_bytes(mom6x_ctx *ctx, int reset);

/* ------------------------------------------------------------------------- */
/* The order-invariant sums and checksums of the reference's regression artefacts (ocean.stats, the debugging
 * checksum lines, the restart files' `checksum` attributes), evaluated on the device-resident fields.  Index
 * ranges are local compute indices, inclusive (h-point computational domain: 0..ni-1, 0..nj-1).  Across tiles
 * the integers are summed over RCCL unless only_on_PE is set.                                              */

/* reproducing_sum_3d (MOM_coms.F90:349): array = nk pitched planes.  sum: the result; sums (nullable, [nk]): the
 * sums by layer -- their presence changes how `sum` is formed (:470-477 vs :541-542), exactly as in the reference;
 * EFP_sum (nullable, [6]) and EFP_lay_sums (nullable, [6*nk]): the extended-fixed-point integers (EFP_type%v);
 * err (nullable): the reference's error code instead of a failure (+1 conversion overflow, +2 overflow, +2 NaN). */
int mom6x_reproducing_sum_3d(mom6x_ctx *ctx, const double *array, int nk, int is, int ie, int js, int je, double unscale,
                             int only_on_PE, double *sum, double *sums, int64_t *EFP_sum, int64_t *EFP_lay_sums, int *err);
/* reproducing_sum_2d (:235, reproducing = .true.) of one pitched plane; err: +2 overflow, +4 NaN (:205-209).   */
int mom6x_reproducing_sum_2d(mom6x_ctx *ctx, const double *array, int is, int ie, int js, int je, double unscale,
                             int only_on_PE, double *sum, int64_t *EFP_sum, int *err);

/* chksum_{h,u,v,B}_{2d,3d} (MOM_checksums.F90:387/:1413 h, :1005/:1782 u, :1209/:1986 v, :688/:1586 B; the pair
 * routines hchksum_pair / uvchksum / Bchksum_pair are two of these calls).  rank = 2 or 3 as the reference's 2-d and
 * 3-d routines differ for B points; stagger as in mom6x_upload; scale: nullable (absent).  The numbers of the two
 * message lines are returned; the host prints them (chk_sum_msg :2563-2640).                                   */
enum mom6x_chksum_kind { MOM6X_CHK_NONE = 0,      /* " c="                          (chk_sum_msg1)     */
                         MOM6X_CHK_CORNERS = 1,   /* " c=" "sw=" "se=" "nw=" "ne="  (chk_sum_msg5)     */
                         MOM6X_CHK_NSEW = 2,      /* " c=" "N=" "S=" "E=" "W="      (chk_sum_msg_NSEW) */
                         MOM6X_CHK_W = 3,         /* " c=" "W="                     (chk_sum_msg_W)    */
                         MOM6X_CHK_S = 4 };       /* " c=" "S="                     (chk_sum_msg_S)    */
typedef struct mom6x_chksum_result {
  double mean, amin, amax;   /* subStats: reproducing mean over the h-point domain, min and max (as printed: 0. + x) */
  int    bc0;                /* bit count of the unshifted computational domain, mod 1 000 000 000                    */
  int    bc[4];              /* the shifted bit counts in the order of the message                                    */
  int    nbc;                /* how many of bc[] are set: 0, 1 or 4                                                  */
  int    bc_kind;            /* enum mom6x_chksum_kind                                                               */
} mom6x_chksum_result;
int mom6x_chksum(mom6x_ctx *ctx, const double *array, int nk, int rank, int stagger, int haloshift, int symmetric,
                 int omit_corners, const double *scale, mom6x_chksum_result *out);

/* field_checksum_real_3d (MOM_checksums.F90:2480) -> field_chksum -> FMS mpp_chksum: the wrapping 64-bit integer sum
 * of the bit patterns of unscale*field over (is..ie, js..je, all k), the value MOM_restart.F90:1741-1749 stores as
 * the `checksum` attribute (written with Z16, MOM_io_infra.F90:1986) of every restart variable.                    */
int mom6x_field_chksum(mom6x_ctx *ctx, const double *array, int nk, int is, int ie, int js, int je, double unscale,
                       int64_t *chksum);

/* write_energy (MOM_sum_output.F90:321): the globally summed diagnostics behind ocean.stats (and ocean.stats.nc).
 * The members of Sum_output_CS that the sums read; the write schedule (ENERGYSAVEDAYS), the values of the previous
 * call (mass_prev_EFP ...), the accumulated surface inputs and the files stay with the host, which formats the line
 * from the numbers returned here (:871-905).                                                                     */
typedef struct mom6x_sum_output_params {
  int    do_APE_calc;       /* CALCULATE_APE (T)                                                       */
  int    use_temperature;   /* ENABLE_THERMODYNAMICS: salt and heat content from tv%T, tv%S            */
  double dt_in_T;           /* DT: the baroclinic time step of the CFL numbers                         */
  double D_list_min_inc;    /* DEPTH_LIST_MIN_INC (1e-10 m)                                            */
  double Z_ref;             /* G%Z_ref                                                                 */
  double C_p;               /* tv%C_p                                                                  */
} mom6x_sum_output_params;
/* MOM_sum_output_init :147 + depth_list_setup :1161 (READ_DEPTH_LIST = False: create_depth_list :1203 from the
 * context's bathyT, areaT, mask2dT; with several tiles the global list is summed over RCCL).  g_prime: GV%g_prime(1:nk). */
int mom6x_sum_output_init(mom6x_ctx *ctx, const mom6x_sum_output_params *p, const double *g_prime);
/* The Depth_List (DL%depth, DL%area, DL%vol_below); the arrays (nullable) must hold listsize entries -- call once
 * with null arrays to learn listsize.                                                                           */
int mom6x_depth_list(const mom6x_ctx *ctx, int *listsize, double *depth, double *area, double *vol_below);
typedef struct mom6x_energy_sums {
  double  mass_tot, KE_tot, PE_tot;   /* :509, :674, :660                                               */
  double  max_CFL[2];                 /* max_CFL_trans, max_CFL_lin :701-727                            */
  int64_t mass_EFP[6];                /* mass_EFP, for mass_chg_EFP = mass_EFP - CS%mass_prev_EFP :749  */
  int64_t salt_EFP[6], heat_EFP[6];   /* :683-686 (zero without ENABLE_THERMODYNAMICS)                  */
} mom6x_energy_sums;
/* u, v, h (and tv%T, tv%S, nullable without ENABLE_THERMODYNAMICS) on the device; mass_lay[nk], KE[nk], PE[nk+1],
 * Z_0APE[nk+1] on the host: the vectors of ocean.stats.nc (Mass_lay, KE, APE, H0).                          */
int mom6x_write_energy(mom6x_ctx *ctx, const double *u, const double *v, const double *h, const double *T, const double *S,
                       mom6x_energy_sums *out, double *mass_lay, double *KE, double *PE, double *Z_0APE);

#ifdef __cplusplus
}
#endif
#endif /* MOM6X_H */

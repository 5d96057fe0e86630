// eos_dev.h -- the equation-of-state pieces more than one translation unit needs (device code).
// EOS_LINEAR (MOM_EOS_linear.F90) and the Wright (1997) family: WRIGHT (MOM_EOS_Wright.F90, the default), WRIGHT_FULL
// (MOM_EOS_Wright_full.F90), WRIGHT_REDUCED (MOM_EOS_Wright_red.F90).
#pragma once
#include "mom6x_dev.h"

namespace {
// Wright (1997) coefficient sets: the reduced-range fit (-2 < theta < 30 degC, 28 < S < 38, p < 5e7 Pa) of EQN_OF_STATE = "WRIGHT"
// (MOM_EOS_Wright.F90:23-37) and "WRIGHT_REDUCED" (MOM_EOS_Wright_red.F90:18-35), and the full-range fit of "WRIGHT_FULL"
// (MOM_EOS_Wright_full.F90:18-35).  "WRIGHT" keeps the original parenthesisation of its expressions; the other two share the
// corrected one, so one set of templates serves all three.
template <int FORM> struct WC {
  static constexpr double a0 = 7.057924e-4, a1 = 3.480336e-7, a2 = -1.112733e-7;
  static constexpr double b0 = 5.790749e8, b1 = 3.516535e6, b2 = -4.002714e4, b3 = 2.084372e2, b4 = 5.944068e5, b5 = -9.643486e3;
  static constexpr double c0 = 1.704853e5, c1 = 7.904722e2, c2 = -7.984422, c3 = 5.140652e-2, c4 = -2.302158e2, c5 = -3.079464;
};
template <> struct WC<MOM6X_EOS_WRIGHT_FULL> {
  static constexpr double a0 = 7.133718e-4, a1 = 2.724670e-7, a2 = -1.646582e-7;
  static constexpr double b0 = 5.613770e8, b1 = 3.600337e6, b2 = -3.727194e4, b3 = 1.660557e2, b4 = 6.844158e5, b5 = -8.389457e3;
  static constexpr double c0 = 1.609893e5, c1 = 8.427815e2, c2 = -6.931554, c3 = 3.869318e-2, c4 = -1.664201e2, c5 = -2.765195;
};

// al0, p0, lambda of the Wright EOS: MOM_EOS_Wright.F90:116-118 (FORM = WRIGHT) | MOM_EOS_Wright_full.F90:84-86 / _red.F90:84-86
template <int FORM>
__device__ __forceinline__ void wright_coefs(double T, double S, double &al0, double &p0, double &lambda) {
  typedef WC<FORM> W;
  if (FORM == MOM6X_EOS_WRIGHT) {
    al0 = (W::a0 + W::a1 * T) + W::a2 * S;
    p0 = (W::b0 + W::b4 * S) + T * (W::b1 + T * ((W::b2 + W::b3 * T)) + W::b5 * S);
    lambda = (W::c0 + W::c4 * S) + T * (W::c1 + T * ((W::c2 + W::c3 * T)) + W::c5 * S);
  } else {
    al0 = W::a0 + (W::a1 * T + W::a2 * S);
    p0 = W::b0 + (W::b4 * S + T * (W::b1 + (T * (W::b2 + W::b3 * T) + W::b5 * S)));
    lambda = W::c0 + (W::c4 * S + T * (W::c1 + (T * (W::c2 + W::c3 * T) + W::c5 * S)));
  }
}

// calculate_density(T, S, pressure, rho, EOS): density_elem of MOM_EOS_linear.F90:73-84 / MOM_EOS_Wright*.F90
template <int FORM>
__device__ __forceinline__ double wright_density(double T, double S, double p) {
  double al0, p0, lambda;
  wright_coefs<FORM>(T, S, al0, p0, lambda);
  return (p + p0) / (lambda + al0 * (p + p0));
}
// EQN_OF_STATE = "UNESCO": the UNESCO (1981) equation of state as refit by Jackett and McDougall (1995), MOM_EOS_UNESCO.F90.
// rho0 = R00 + sig0 at one atmosphere (Rab: the S^a T^b term, 6 for the power 1.5), the secant bulk modulus ks (Sabc: S^a T^b p^c),
// pressure in bar.  No analytic layer integrals exist for it: the pressure force takes the quadratures (EOS_QUADRATURE or a
// pressure reconstruction).
namespace unesco {
constexpr double R00 = 999.842594, R01 = 6.793952e-2, R02 = -9.095290e-3, R03 = 1.001685e-4, R04 = -1.120083e-6, R05 = 6.536332e-9;
constexpr double R10 = 0.824493, R11 = -4.0899e-3, R12 = 7.6438e-5, R13 = -8.2467e-7, R14 = 5.3875e-9;
constexpr double R60 = -5.72466e-3, R61 = 1.0227e-4, R62 = -1.6546e-6, R20 = 4.8314e-4;
constexpr double S000 = 1.965933e4, S010 = 1.444304e2, S020 = -1.706103, S030 = 9.648704e-3, S040 = -4.190253e-5;
constexpr double S100 = 52.84855, S110 = -3.101089e-1, S120 = 6.283263e-3, S130 = -5.084188e-5;
constexpr double S600 = 3.886640e-1, S610 = 9.085835e-3, S620 = -4.619924e-4;
constexpr double S001 = 3.186519, S011 = 2.212276e-2, S021 = -2.984642e-4, S031 = 1.956415e-6;
constexpr double S101 = 6.704388e-3, S111 = -1.847318e-4, S121 = 2.059331e-7, S601 = 1.480266e-4;
constexpr double S002 = 2.102898e-4, S012 = -1.202016e-5, S022 = 1.394680e-7, S102 = -2.040237e-6, S112 = 6.128773e-8, S122 = 6.207323e-10;

__device__ __forceinline__ double sig0_of(double t1, double s1, double s12) {          // :114-116
  return (t1 * (R01 + t1 * (R02 + t1 * (R03 + t1 * (R04 + t1 * R05)))) +
          s1 * ((R10 + t1 * (R11 + t1 * (R12 + t1 * (R13 + t1 * R14)))) + (s12 * (R60 + t1 * (R61 + t1 * R62)) + s1 * R20)));
}
__device__ __forceinline__ double ks_of(double t1, double s1, double s12, double p1) {  // :120-124
  return (S000 + (t1 * (S010 + t1 * (S020 + t1 * (S030 + t1 * S040))) +
                  s1 * ((S100 + t1 * (S110 + t1 * (S120 + t1 * S130))) + s12 * (S600 + t1 * (S610 + t1 * S620))))) +
         p1 * ((S001 + (t1 * (S011 + t1 * (S021 + t1 * S031)) + s1 * ((S101 + t1 * (S111 + t1 * S121)) + s12 * S601))) +
               p1 * (S002 + (t1 * (S012 + t1 * S022) + s1 * (S102 + t1 * (S112 + t1 * S122)))));
}
__device__ __forceinline__ double density(double T, double S, double pressure) {          // density_elem_UNESCO :95-128
  const double p1 = pressure * 1.0e-5, t1 = T, s1 = dmax(S, 0.0), s12 = sqrt(s1);
  const double rho0 = R00 + sig0_of(t1, s1, s12), ks = ks_of(t1, s1, s12, p1);
  return rho0 * ks / (ks - p1);
}
__device__ __forceinline__ double density_anomaly(double T, double S, double pressure, double rho_ref) {   // :133-167
  const double p1 = pressure * 1.0e-5, t1 = T, s1 = dmax(S, 0.0), s12 = sqrt(s1);
  const double sig0 = sig0_of(t1, s1, s12), ks = ks_of(t1, s1, s12, p1);
  return ((R00 - rho_ref) * ks + (sig0 * ks + p1 * rho_ref)) / (ks - p1);
}
__device__ __forceinline__ void density_derivs(double T, double S, double pressure, double &drho_dT, double &drho_dS) {   // :244-297
  const double p1 = pressure * 1.0e-5, t1 = T, s1 = dmax(S, 0.0), s12 = sqrt(s1);
  const double rho0 = R00 + sig0_of(t1, s1, s12);
  const double drho0_dT = R01 + (t1 * (2.0 * R02 + t1 * (3.0 * R03 + t1 * (4.0 * R04 + t1 * (5.0 * R05)))) +
                                 s1 * (R11 + (t1 * (2.0 * R12 + t1 * (3.0 * R13 + t1 * (4.0 * R14))) + s12 * (R61 + t1 * (2.0 * R62)))));
  const double drho0_dS = R10 + (t1 * (R11 + t1 * (R12 + t1 * (R13 + t1 * R14))) + (1.5 * (s12 * (R60 + t1 * (R61 + t1 * R62))) + s1 * (2.0 * R20)));
  const double ks = ks_of(t1, s1, s12, p1);
  const double dks_dT = (S010 + (t1 * (2.0 * S020 + t1 * (3.0 * S030 + t1 * (4.0 * S040))) +
                                 s1 * ((S110 + t1 * (2.0 * S120 + t1 * (3.0 * S130))) + s12 * (S610 + t1 * (2.0 * S620))))) +
                        p1 * (((S011 + t1 * (2.0 * S021 + t1 * (3.0 * S031))) + s1 * (S111 + t1 * (2.0 * S121))) +
                              p1 * (S012 + t1 * (2.0 * S022) + s1 * (S112 + t1 * (2.0 * S122))));
  const double dks_dS = (S100 + (t1 * (S110 + t1 * (S120 + t1 * S130)) + 1.5 * (s12 * (S600 + t1 * (S610 + t1 * S620))))) +
                        p1 * ((S101 + t1 * (S111 + t1 * S121) + s12 * (1.5 * S601)) + p1 * (S102 + t1 * (S112 + t1 * S122)));
  const double I_denom = 1.0 / (ks - p1);
  drho_dT = (ks * drho0_dT - dks_dT * ((rho0 * p1) * I_denom)) * I_denom;
  drho_dS = (ks * drho0_dS - dks_dS * ((rho0 * p1) * I_denom)) * I_denom;
}
}  // namespace unesco

namespace roquet {
#define RQ_FN __device__ __forceinline__

/* EQN_OF_STATE = "ROQUET_RHO" (alias "NEMO"): the polynomial of Roquet et al. (2015), MOM_EOS_Roquet_rho.F90.  EOSabc: the zs^a T^b p^c
 * term (zs = sqrt((S + 32) * 0.875 / 35.16504)), R0c: the reference profile's p^(c+1) term, ALP / BET: the T / zs derivatives'
 * coefficients -- the reference forms them as named constants from the published numbers (:12-155), and so do these macros, with
 * integer powers by repeated squaring as the compiler folds them. */
#define RQ_POW2(x) ((x) * (x))
#define RQ_POW3(x) ((x) * ((x) * (x)))
#define RQ_POW4(x) (((x) * (x)) * ((x) * (x)))
#define RQ_POW5(x) ((x) * (((x) * (x)) * ((x) * (x))))
#define RQ_POW6(x) (((x) * (x)) * (((x) * (x)) * ((x) * (x))))
#define RQ_Pa2kb (1.e-8)
#define RQ_rdeltaS (32.)
#define RQ_r1_S0 (0.875/35.16504)
#define RQ_I_Ts (0.025)
#define RQ_R00 (4.6494977072e+01*RQ_Pa2kb)
#define RQ_R01 (-5.2099962525*RQ_POW2(RQ_Pa2kb))
#define RQ_R02 (2.2601900708e-01*RQ_POW3(RQ_Pa2kb))
#define RQ_R03 (6.4326772569e-02*RQ_POW4(RQ_Pa2kb))
#define RQ_R04 (1.5616995503e-02*RQ_POW5(RQ_Pa2kb))
#define RQ_R05 (-1.7243708991e-03*RQ_POW6(RQ_Pa2kb))
#define RQ_EOS000 (8.0189615746e+02)
#define RQ_EOS100 (8.6672408165e+02)
#define RQ_EOS200 (-1.7864682637e+03)
#define RQ_EOS300 (2.0375295546e+03)
#define RQ_EOS400 (-1.2849161071e+03)
#define RQ_EOS500 (4.3227585684e+02)
#define RQ_EOS600 (-6.0579916612e+01)
#define RQ_EOS010 (2.6010145068e+01*RQ_I_Ts)
#define RQ_EOS110 (-6.5281885265e+01*RQ_I_Ts)
#define RQ_EOS210 (8.1770425108e+01*RQ_I_Ts)
#define RQ_EOS310 (-5.6888046321e+01*RQ_I_Ts)
#define RQ_EOS410 (1.7681814114e+01*RQ_I_Ts)
#define RQ_EOS510 (-1.9193502195*RQ_I_Ts)
#define RQ_EOS020 (-3.7074170417e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS120 (6.1548258127e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS220 (-6.0362551501e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS320 (2.9130021253e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS420 (-5.4723692739*RQ_POW2(RQ_I_Ts))
#define RQ_EOS030 (2.1661789529e+01*RQ_POW3(RQ_I_Ts))
#define RQ_EOS130 (-3.3449108469e+01*RQ_POW3(RQ_I_Ts))
#define RQ_EOS230 (1.9717078466e+01*RQ_POW3(RQ_I_Ts))
#define RQ_EOS330 (-3.1742946532*RQ_POW3(RQ_I_Ts))
#define RQ_EOS040 (-8.3627885467*RQ_POW4(RQ_I_Ts))
#define RQ_EOS140 (1.1311538584e+01*RQ_POW4(RQ_I_Ts))
#define RQ_EOS240 (-5.3563304045*RQ_POW4(RQ_I_Ts))
#define RQ_EOS050 (5.4048723791e-01*RQ_POW5(RQ_I_Ts))
#define RQ_EOS150 (4.8169980163e-01*RQ_POW5(RQ_I_Ts))
#define RQ_EOS060 (-1.9083568888e-01*RQ_POW6(RQ_I_Ts))
#define RQ_EOS001 (1.9681925209e+01*RQ_Pa2kb)
#define RQ_EOS101 (-4.2549998214e+01*RQ_Pa2kb)
#define RQ_EOS201 (5.0774768218e+01*RQ_Pa2kb)
#define RQ_EOS301 (-3.0938076334e+01*RQ_Pa2kb)
#define RQ_EOS401 (6.6051753097*RQ_Pa2kb)
#define RQ_EOS011 (-1.3336301113e+01*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS111 (-4.4870114575*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS211 (5.0042598061*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS311 (-6.5399043664e-01*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS021 (6.7080479603*(RQ_POW2(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS121 (3.5063081279*(RQ_POW2(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS221 (-1.8795372996*(RQ_POW2(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS031 (-2.4649669534*(RQ_POW3(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS131 (-5.5077101279e-01*(RQ_POW3(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS041 (5.5927935970e-01*(RQ_POW4(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS002 (2.0660924175*RQ_POW2(RQ_Pa2kb))
#define RQ_EOS102 (-4.9527603989*RQ_POW2(RQ_Pa2kb))
#define RQ_EOS202 (2.5019633244*RQ_POW2(RQ_Pa2kb))
#define RQ_EOS012 (2.0564311499*(RQ_I_Ts*RQ_POW2(RQ_Pa2kb)))
#define RQ_EOS112 (-2.1311365518e-01*(RQ_I_Ts*RQ_POW2(RQ_Pa2kb)))
#define RQ_EOS022 (-1.2419983026*(RQ_POW2(RQ_I_Ts)*RQ_POW2(RQ_Pa2kb)))
#define RQ_EOS003 (-2.3342758797e-02*RQ_POW3(RQ_Pa2kb))
#define RQ_EOS103 (-1.8507636718e-02*RQ_POW3(RQ_Pa2kb))
#define RQ_EOS013 (3.7969820455e-01*(RQ_I_Ts*RQ_POW3(RQ_Pa2kb)))
#define RQ_ALP000 (RQ_EOS010)
#define RQ_ALP100 (RQ_EOS110)
#define RQ_ALP200 (RQ_EOS210)
#define RQ_ALP300 (RQ_EOS310)
#define RQ_ALP400 (RQ_EOS410)
#define RQ_ALP500 (RQ_EOS510)
#define RQ_ALP010 (2.*RQ_EOS020)
#define RQ_ALP110 (2.*RQ_EOS120)
#define RQ_ALP210 (2.*RQ_EOS220)
#define RQ_ALP310 (2.*RQ_EOS320)
#define RQ_ALP410 (2.*RQ_EOS420)
#define RQ_ALP020 (3.*RQ_EOS030)
#define RQ_ALP120 (3.*RQ_EOS130)
#define RQ_ALP220 (3.*RQ_EOS230)
#define RQ_ALP320 (3.*RQ_EOS330)
#define RQ_ALP030 (4.*RQ_EOS040)
#define RQ_ALP130 (4.*RQ_EOS140)
#define RQ_ALP230 (4.*RQ_EOS240)
#define RQ_ALP040 (5.*RQ_EOS050)
#define RQ_ALP140 (5.*RQ_EOS150)
#define RQ_ALP050 (6.*RQ_EOS060)
#define RQ_ALP001 (RQ_EOS011)
#define RQ_ALP101 (RQ_EOS111)
#define RQ_ALP201 (RQ_EOS211)
#define RQ_ALP301 (RQ_EOS311)
#define RQ_ALP011 (2.*RQ_EOS021)
#define RQ_ALP111 (2.*RQ_EOS121)
#define RQ_ALP211 (2.*RQ_EOS221)
#define RQ_ALP021 (3.*RQ_EOS031)
#define RQ_ALP121 (3.*RQ_EOS131)
#define RQ_ALP031 (4.*RQ_EOS041)
#define RQ_ALP002 (RQ_EOS012)
#define RQ_ALP102 (RQ_EOS112)
#define RQ_ALP012 (2.*RQ_EOS022)
#define RQ_ALP003 (RQ_EOS013)
#define RQ_BET000 (0.5*RQ_EOS100*RQ_r1_S0)
#define RQ_BET100 (RQ_EOS200*RQ_r1_S0)
#define RQ_BET200 (1.5*RQ_EOS300*RQ_r1_S0)
#define RQ_BET300 (2.0*RQ_EOS400*RQ_r1_S0)
#define RQ_BET400 (2.5*RQ_EOS500*RQ_r1_S0)
#define RQ_BET500 (3.0*RQ_EOS600*RQ_r1_S0)
#define RQ_BET010 (0.5*RQ_EOS110*RQ_r1_S0)
#define RQ_BET110 (RQ_EOS210*RQ_r1_S0)
#define RQ_BET210 (1.5*RQ_EOS310*RQ_r1_S0)
#define RQ_BET310 (2.0*RQ_EOS410*RQ_r1_S0)
#define RQ_BET410 (2.5*RQ_EOS510*RQ_r1_S0)
#define RQ_BET020 (0.5*RQ_EOS120*RQ_r1_S0)
#define RQ_BET120 (RQ_EOS220*RQ_r1_S0)
#define RQ_BET220 (1.5*RQ_EOS320*RQ_r1_S0)
#define RQ_BET320 (2.0*RQ_EOS420*RQ_r1_S0)
#define RQ_BET030 (0.5*RQ_EOS130*RQ_r1_S0)
#define RQ_BET130 (RQ_EOS230*RQ_r1_S0)
#define RQ_BET230 (1.5*RQ_EOS330*RQ_r1_S0)
#define RQ_BET040 (0.5*RQ_EOS140*RQ_r1_S0)
#define RQ_BET140 (RQ_EOS240*RQ_r1_S0)
#define RQ_BET050 (0.5*RQ_EOS150*RQ_r1_S0)
#define RQ_BET001 (0.5*RQ_EOS101*RQ_r1_S0)
#define RQ_BET101 (RQ_EOS201*RQ_r1_S0)
#define RQ_BET201 (1.5*RQ_EOS301*RQ_r1_S0)
#define RQ_BET301 (2.0*RQ_EOS401*RQ_r1_S0)
#define RQ_BET011 (0.5*RQ_EOS111*RQ_r1_S0)
#define RQ_BET111 (RQ_EOS211*RQ_r1_S0)
#define RQ_BET211 (1.5*RQ_EOS311*RQ_r1_S0)
#define RQ_BET021 (0.5*RQ_EOS121*RQ_r1_S0)
#define RQ_BET121 (RQ_EOS221*RQ_r1_S0)
#define RQ_BET031 (0.5*RQ_EOS131*RQ_r1_S0)
#define RQ_BET002 (0.5*RQ_EOS102*RQ_r1_S0)
#define RQ_BET102 (RQ_EOS202*RQ_r1_S0)
#define RQ_BET012 (0.5*RQ_EOS112*RQ_r1_S0)
#define RQ_BET003 (0.5*RQ_EOS103*RQ_r1_S0)

RQ_FN void roquet_parts(double T, double S, double pressure, double *zs_out, double *rhoTS0, double *rhoTS1, double *rhoTS2,
                        double *rhoTS3, double *rho0S0, double *rho00p) {   /* :216-241 */
  const double zt = T, zs = sqrt(fabs(S + RQ_rdeltaS) * RQ_r1_S0), zp = pressure;
  *rhoTS3 = RQ_EOS003 + (zs * RQ_EOS103 + zt * RQ_EOS013);
  *rhoTS2 = RQ_EOS002 + (zs * (RQ_EOS102 + zs * RQ_EOS202) + zt * (RQ_EOS012 + (zs * RQ_EOS112 + zt * RQ_EOS022)));
  *rhoTS1 = RQ_EOS001 + (zs * (RQ_EOS101 + zs * (RQ_EOS201 + zs * (RQ_EOS301 + zs * RQ_EOS401))) +
                         zt * (RQ_EOS011 + (zs * (RQ_EOS111 + zs * (RQ_EOS211 + zs * RQ_EOS311)) +
                                            zt * (RQ_EOS021 + (zs * (RQ_EOS121 + zs * RQ_EOS221) +
                                                               zt * (RQ_EOS031 + (zs * RQ_EOS131 + zt * RQ_EOS041)))))));
  *rhoTS0 = zt * (RQ_EOS010 +
                  (zs * (RQ_EOS110 + zs * (RQ_EOS210 + zs * (RQ_EOS310 + zs * (RQ_EOS410 + zs * RQ_EOS510)))) +
                   zt * (RQ_EOS020 + (zs * (RQ_EOS120 + zs * (RQ_EOS220 + zs * (RQ_EOS320 + zs * RQ_EOS420))) +
                                      zt * (RQ_EOS030 + (zs * (RQ_EOS130 + zs * (RQ_EOS230 + zs * RQ_EOS330)) +
                                                         zt * (RQ_EOS040 + (zs * (RQ_EOS140 + zs * RQ_EOS240) +
                                                                            zt * (RQ_EOS050 + (zs * RQ_EOS150 + zt * RQ_EOS060))))))))));
  *rho0S0 = RQ_EOS000 + zs * (RQ_EOS100 + zs * (RQ_EOS200 + zs * (RQ_EOS300 + zs * (RQ_EOS400 + zs * (RQ_EOS500 + zs * RQ_EOS600)))));
  *rho00p = zp * (RQ_R00 + zp * (RQ_R01 + zp * (RQ_R02 + zp * (RQ_R03 + zp * (RQ_R04 + zp * RQ_R05)))));
  *zs_out = zs;
}
RQ_FN double roquet_density(double T, double S, double pressure) {   /* density_elem_Roquet_rho :192-243 */
  double zs, r0, r1, r2, r3, s0, p0;
  roquet_parts(T, S, pressure, &zs, &r0, &r1, &r2, &r3, &s0, &p0);
  const double zp = pressure;
  const double rhoTS = (r0 + s0) + zp * (r1 + zp * (r2 + zp * r3));
  return rhoTS + p0;
}
RQ_FN double roquet_density_anomaly(double T, double S, double pressure, double rho_ref) {   /* :248-306 */
  double zs, r0, r1, r2, r3, s0, p0;
  roquet_parts(T, S, pressure, &zs, &r0, &r1, &r2, &r3, &s0, &p0);
  const double zp = pressure;
  s0 = s0 - rho_ref;
  const double rhoTS = (r0 + s0) + zp * (r1 + zp * (r2 + zp * r3));
  return rhoTS + p0;
}
RQ_FN void roquet_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {   /* :340-411 */
  const double zt = T, zs = sqrt(fabs(S + RQ_rdeltaS) * RQ_r1_S0), zp = pressure;
  const double dRdzt3 = RQ_ALP003;
  const double dRdzt2 = RQ_ALP002 + (zs * RQ_ALP102 + zt * RQ_ALP012);
  const double dRdzt1 = RQ_ALP001 + (zs * (RQ_ALP101 + zs * (RQ_ALP201 + zs * RQ_ALP301)) +
                                     zt * (RQ_ALP011 + (zs * (RQ_ALP111 + zs * RQ_ALP211) + zt * (RQ_ALP021 + (zs * RQ_ALP121 + zt * RQ_ALP031)))));
  const double dRdzt0 = RQ_ALP000 + (zs * (RQ_ALP100 + zs * (RQ_ALP200 + zs * (RQ_ALP300 + zs * (RQ_ALP400 + zs * RQ_ALP500)))) +
                                     zt * (RQ_ALP010 + (zs * (RQ_ALP110 + zs * (RQ_ALP210 + zs * (RQ_ALP310 + zs * RQ_ALP410))) +
                                                        zt * (RQ_ALP020 + (zs * (RQ_ALP120 + zs * (RQ_ALP220 + zs * RQ_ALP320)) +
                                                                           zt * (RQ_ALP030 + (zt * (RQ_ALP040 + (zs * RQ_ALP140 + zt * RQ_ALP050)) +
                                                                                              zs * (RQ_ALP130 + zs * RQ_ALP230))))))));
  *drho_dT = dRdzt0 + zp * (dRdzt1 + zp * (dRdzt2 + zp * dRdzt3));
  const double dRdzs3 = RQ_BET003;
  const double dRdzs2 = RQ_BET002 + (zs * RQ_BET102 + zt * RQ_BET012);
  const double dRdzs1 = RQ_BET001 + (zs * (RQ_BET101 + zs * (RQ_BET201 + zs * RQ_BET301)) +
                                     zt * (RQ_BET011 + (zs * (RQ_BET111 + zs * RQ_BET211) + zt * (RQ_BET021 + (zs * RQ_BET121 + zt * RQ_BET031)))));
  const double dRdzs0 = RQ_BET000 + (zs * (RQ_BET100 + zs * (RQ_BET200 + zs * (RQ_BET300 + zs * (RQ_BET400 + zs * RQ_BET500)))) +
                                     zt * (RQ_BET010 + (zs * (RQ_BET110 + zs * (RQ_BET210 + zs * (RQ_BET310 + zs * RQ_BET410))) +
                                                        zt * (RQ_BET020 + (zs * (RQ_BET120 + zs * (RQ_BET220 + zs * RQ_BET320)) +
                                                                           zt * (RQ_BET030 + (zt * (RQ_BET040 + (zs * RQ_BET140 + zt * RQ_BET050)) +
                                                                                              zs * (RQ_BET130 + zs * RQ_BET230))))))));
  *drho_dS = (dRdzs0 + zp * (dRdzs1 + zp * (dRdzs2 + zp * dRdzs3))) / zs;
}

/* EQN_OF_STATE = "ROQUET_SPV": the specific-volume polynomial of Roquet et al. (2015), MOM_EOS_Roquet_SpV.F90 -- the same structure
 * as ROQUET_RHO with SPVabc / V0c / ALP / BET (zs = sqrt((S + 24) * 0.875 / 35.16504)); density = 1 / spec_vol. */
#define RS_Pa2kb (1.e-8)
#define RS_rdeltaS (24.)
#define RS_r1_S0 (0.875/35.16504)
#define RS_I_Ts (0.025)
#define RS_V00 (-4.4015007269e-05*RS_Pa2kb)
#define RS_V01 (6.9232335784e-06*RQ_POW2(RS_Pa2kb))
#define RS_V02 (-7.5004675975e-07*RQ_POW3(RS_Pa2kb))
#define RS_V03 (1.7009109288e-08*RQ_POW4(RS_Pa2kb))
#define RS_V04 (-1.6884162004e-08*RQ_POW5(RS_Pa2kb))
#define RS_V05 (1.9613503930e-09*RQ_POW6(RS_Pa2kb))
#define RS_SPV000 (1.0772899069e-03)
#define RS_SPV100 (-3.1263658781e-04)
#define RS_SPV200 (6.7615860683e-04)
#define RS_SPV300 (-8.6127884515e-04)
#define RS_SPV400 (5.9010812596e-04)
#define RS_SPV500 (-2.1503943538e-04)
#define RS_SPV600 (3.2678954455e-05)
#define RS_SPV010 (-1.4949652640e-05*RS_I_Ts)
#define RS_SPV110 (3.1866349188e-05*RS_I_Ts)
#define RS_SPV210 (-3.8070687610e-05*RS_I_Ts)
#define RS_SPV310 (2.9818473563e-05*RS_I_Ts)
#define RS_SPV410 (-1.0011321965e-05*RS_I_Ts)
#define RS_SPV510 (1.0751931163e-06*RS_I_Ts)
#define RS_SPV020 (2.7546851539e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV120 (-3.6597334199e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV220 (3.4489154625e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV320 (-1.7663254122e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV420 (3.5965131935e-06*RQ_POW2(RS_I_Ts))
#define RS_SPV030 (-1.6506828994e-05*RQ_POW3(RS_I_Ts))
#define RS_SPV130 (2.4412359055e-05*RQ_POW3(RS_I_Ts))
#define RS_SPV230 (-1.4606740723e-05*RQ_POW3(RS_I_Ts))
#define RS_SPV330 (2.3293406656e-06*RQ_POW3(RS_I_Ts))
#define RS_SPV040 (6.7896174634e-06*RQ_POW4(RS_I_Ts))
#define RS_SPV140 (-8.7951832993e-06*RQ_POW4(RS_I_Ts))
#define RS_SPV240 (4.4249040774e-06*RQ_POW4(RS_I_Ts))
#define RS_SPV050 (-7.2535743349e-07*RQ_POW5(RS_I_Ts))
#define RS_SPV150 (-3.4680559205e-07*RQ_POW5(RS_I_Ts))
#define RS_SPV060 (1.9041365570e-07*RQ_POW6(RS_I_Ts))
#define RS_SPV001 (-1.6889436589e-05*RS_Pa2kb)
#define RS_SPV101 (2.1106556158e-05*RS_Pa2kb)
#define RS_SPV201 (-2.1322804368e-05*RS_Pa2kb)
#define RS_SPV301 (1.7347655458e-05*RS_Pa2kb)
#define RS_SPV401 (-4.3209400767e-06*RS_Pa2kb)
#define RS_SPV011 (1.5355844621e-05*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV111 (2.0914122241e-06*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV211 (-5.7751479725e-06*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV311 (1.0767234341e-06*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV021 (-9.6659393016e-06*(RQ_POW2(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV121 (-7.0686982208e-07*(RQ_POW2(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV221 (1.4488066593e-06*(RQ_POW2(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV031 (3.1134283336e-06*(RQ_POW3(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV131 (7.9562529879e-08*(RQ_POW3(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV041 (-5.6590253863e-07*(RQ_POW4(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV002 (1.0500241168e-06*RQ_POW2(RS_Pa2kb))
#define RS_SPV102 (1.9600661704e-06*RQ_POW2(RS_Pa2kb))
#define RS_SPV202 (-2.1666693382e-06*RQ_POW2(RS_Pa2kb))
#define RS_SPV012 (-3.8541359685e-06*(RS_I_Ts*RQ_POW2(RS_Pa2kb)))
#define RS_SPV112 (1.0157632247e-06*(RS_I_Ts*RQ_POW2(RS_Pa2kb)))
#define RS_SPV022 (1.7178343158e-06*(RQ_POW2(RS_I_Ts)*RQ_POW2(RS_Pa2kb)))
#define RS_SPV003 (-4.1503454190e-07*RQ_POW3(RS_Pa2kb))
#define RS_SPV103 (3.5627020989e-07*RQ_POW3(RS_Pa2kb))
#define RS_SPV013 (-1.1293871415e-07*(RS_I_Ts*RQ_POW3(RS_Pa2kb)))
#define RS_ALP000 (RS_SPV010)
#define RS_ALP100 (RS_SPV110)
#define RS_ALP200 (RS_SPV210)
#define RS_ALP300 (RS_SPV310)
#define RS_ALP400 (RS_SPV410)
#define RS_ALP500 (RS_SPV510)
#define RS_ALP010 (2.*RS_SPV020)
#define RS_ALP110 (2.*RS_SPV120)
#define RS_ALP210 (2.*RS_SPV220)
#define RS_ALP310 (2.*RS_SPV320)
#define RS_ALP410 (2.*RS_SPV420)
#define RS_ALP020 (3.*RS_SPV030)
#define RS_ALP120 (3.*RS_SPV130)
#define RS_ALP220 (3.*RS_SPV230)
#define RS_ALP320 (3.*RS_SPV330)
#define RS_ALP030 (4.*RS_SPV040)
#define RS_ALP130 (4.*RS_SPV140)
#define RS_ALP230 (4.*RS_SPV240)
#define RS_ALP040 (5.*RS_SPV050)
#define RS_ALP140 (5.*RS_SPV150)
#define RS_ALP050 (6.*RS_SPV060)
#define RS_ALP001 (RS_SPV011)
#define RS_ALP101 (RS_SPV111)
#define RS_ALP201 (RS_SPV211)
#define RS_ALP301 (RS_SPV311)
#define RS_ALP011 (2.*RS_SPV021)
#define RS_ALP111 (2.*RS_SPV121)
#define RS_ALP211 (2.*RS_SPV221)
#define RS_ALP021 (3.*RS_SPV031)
#define RS_ALP121 (3.*RS_SPV131)
#define RS_ALP031 (4.*RS_SPV041)
#define RS_ALP002 (RS_SPV012)
#define RS_ALP102 (RS_SPV112)
#define RS_ALP012 (2.*RS_SPV022)
#define RS_ALP003 (RS_SPV013)
#define RS_BET000 (0.5*RS_SPV100*RS_r1_S0)
#define RS_BET100 (RS_SPV200*RS_r1_S0)
#define RS_BET200 (1.5*RS_SPV300*RS_r1_S0)
#define RS_BET300 (2.0*RS_SPV400*RS_r1_S0)
#define RS_BET400 (2.5*RS_SPV500*RS_r1_S0)
#define RS_BET500 (3.0*RS_SPV600*RS_r1_S0)
#define RS_BET010 (0.5*RS_SPV110*RS_r1_S0)
#define RS_BET110 (RS_SPV210*RS_r1_S0)
#define RS_BET210 (1.5*RS_SPV310*RS_r1_S0)
#define RS_BET310 (2.0*RS_SPV410*RS_r1_S0)
#define RS_BET410 (2.5*RS_SPV510*RS_r1_S0)
#define RS_BET020 (0.5*RS_SPV120*RS_r1_S0)
#define RS_BET120 (RS_SPV220*RS_r1_S0)
#define RS_BET220 (1.5*RS_SPV320*RS_r1_S0)
#define RS_BET320 (2.0*RS_SPV420*RS_r1_S0)
#define RS_BET030 (0.5*RS_SPV130*RS_r1_S0)
#define RS_BET130 (RS_SPV230*RS_r1_S0)
#define RS_BET230 (1.5*RS_SPV330*RS_r1_S0)
#define RS_BET040 (0.5*RS_SPV140*RS_r1_S0)
#define RS_BET140 (RS_SPV240*RS_r1_S0)
#define RS_BET050 (0.5*RS_SPV150*RS_r1_S0)
#define RS_BET001 (0.5*RS_SPV101*RS_r1_S0)
#define RS_BET101 (RS_SPV201*RS_r1_S0)
#define RS_BET201 (1.5*RS_SPV301*RS_r1_S0)
#define RS_BET301 (2.0*RS_SPV401*RS_r1_S0)
#define RS_BET011 (0.5*RS_SPV111*RS_r1_S0)
#define RS_BET111 (RS_SPV211*RS_r1_S0)
#define RS_BET211 (1.5*RS_SPV311*RS_r1_S0)
#define RS_BET021 (0.5*RS_SPV121*RS_r1_S0)
#define RS_BET121 (RS_SPV221*RS_r1_S0)
#define RS_BET031 (0.5*RS_SPV131*RS_r1_S0)
#define RS_BET002 (0.5*RS_SPV102*RS_r1_S0)
#define RS_BET102 (RS_SPV202*RS_r1_S0)
#define RS_BET012 (0.5*RS_SPV112*RS_r1_S0)
#define RS_BET003 (0.5*RS_SPV103*RS_r1_S0)
RQ_FN void roquet_spv_parts(double T, double S, double pressure, double *zs_out, double *svTS0, double *svTS1, double *svTS2,
                        double *svTS3, double *sv0S0, double *sv00p) {   /* :216-241 */
  const double zt = T, zs = sqrt(fabs(S + RS_rdeltaS) * RS_r1_S0), zp = pressure;
  *svTS3 = RS_SPV003 + (zs * RS_SPV103 + zt * RS_SPV013);
  *svTS2 = RS_SPV002 + (zs * (RS_SPV102 + zs * RS_SPV202) + zt * (RS_SPV012 + (zs * RS_SPV112 + zt * RS_SPV022)));
  *svTS1 = RS_SPV001 + (zs * (RS_SPV101 + zs * (RS_SPV201 + zs * (RS_SPV301 + zs * RS_SPV401))) +
                         zt * (RS_SPV011 + (zs * (RS_SPV111 + zs * (RS_SPV211 + zs * RS_SPV311)) +
                                            zt * (RS_SPV021 + (zs * (RS_SPV121 + zs * RS_SPV221) +
                                                               zt * (RS_SPV031 + (zs * RS_SPV131 + zt * RS_SPV041)))))));
  *svTS0 = zt * (RS_SPV010 +
                  (zs * (RS_SPV110 + zs * (RS_SPV210 + zs * (RS_SPV310 + zs * (RS_SPV410 + zs * RS_SPV510)))) +
                   zt * (RS_SPV020 + (zs * (RS_SPV120 + zs * (RS_SPV220 + zs * (RS_SPV320 + zs * RS_SPV420))) +
                                      zt * (RS_SPV030 + (zs * (RS_SPV130 + zs * (RS_SPV230 + zs * RS_SPV330)) +
                                                         zt * (RS_SPV040 + (zs * (RS_SPV140 + zs * RS_SPV240) +
                                                                            zt * (RS_SPV050 + (zs * RS_SPV150 + zt * RS_SPV060))))))))));
  *sv0S0 = RS_SPV000 + zs * (RS_SPV100 + zs * (RS_SPV200 + zs * (RS_SPV300 + zs * (RS_SPV400 + zs * (RS_SPV500 + zs * RS_SPV600)))));
  *sv00p = zp * (RS_V00 + zp * (RS_V01 + zp * (RS_V02 + zp * (RS_V03 + zp * (RS_V04 + zp * RS_V05)))));
  *zs_out = zs;
}
RQ_FN double roquet_spv(double T, double S, double pressure, double spv_ref, int anomaly) {   /* spec_vol_elem :191-248, _anomaly :253-315 */
  double zs, r0, r1, r2, r3, s0, p0;
  roquet_spv_parts(T, S, pressure, &zs, &r0, &r1, &r2, &r3, &s0, &p0);
  const double zp = pressure;
  if (anomaly) s0 = s0 - spv_ref;
  const double SV_TS = (r0 + s0) + zp * (r1 + zp * (r2 + zp * r3));
  return SV_TS + p0;
}
RQ_FN double roquet_spv_density(double T, double S, double pressure) {   /* density_elem_Roquet_SpV :318-330 */
  return 1.0 / roquet_spv(T, S, pressure, 0.0, 0);
}
RQ_FN double roquet_spv_density_anomaly(double T, double S, double pressure, double rho_ref) {   /* :335-348 */
  const double spv = roquet_spv(T, S, pressure, 1.0 / rho_ref, 1);
  return -((rho_ref * rho_ref) * spv / (rho_ref * spv + 1.0));
}
RQ_FN void roquet_spv_derivs(double T, double S, double pressure, double *dSV_dT, double *dSV_dS) {   /* calculate_specvol_derivs_elem_Roquet_SpV :352-423 */
  const double zt = T, zs = sqrt(fabs(S + RS_rdeltaS) * RS_r1_S0), zp = pressure;
  const double dRdzt3 = RS_ALP003;
  const double dRdzt2 = RS_ALP002 + (zs * RS_ALP102 + zt * RS_ALP012);
  const double dRdzt1 = RS_ALP001 + (zs * (RS_ALP101 + zs * (RS_ALP201 + zs * RS_ALP301)) +
                                     zt * (RS_ALP011 + (zs * (RS_ALP111 + zs * RS_ALP211) + zt * (RS_ALP021 + (zs * RS_ALP121 + zt * RS_ALP031)))));
  const double dRdzt0 = RS_ALP000 + (zs * (RS_ALP100 + zs * (RS_ALP200 + zs * (RS_ALP300 + zs * (RS_ALP400 + zs * RS_ALP500)))) +
                                     zt * (RS_ALP010 + (zs * (RS_ALP110 + zs * (RS_ALP210 + zs * (RS_ALP310 + zs * RS_ALP410))) +
                                                        zt * (RS_ALP020 + (zs * (RS_ALP120 + zs * (RS_ALP220 + zs * RS_ALP320)) +
                                                                           zt * (RS_ALP030 + (zt * (RS_ALP040 + (zs * RS_ALP140 + zt * RS_ALP050)) +
                                                                                              zs * (RS_ALP130 + zs * RS_ALP230))))))));
  *dSV_dT = dRdzt0 + zp * (dRdzt1 + zp * (dRdzt2 + zp * dRdzt3));
  const double dRdzs3 = RS_BET003;
  const double dRdzs2 = RS_BET002 + (zs * RS_BET102 + zt * RS_BET012);
  const double dRdzs1 = RS_BET001 + (zs * (RS_BET101 + zs * (RS_BET201 + zs * RS_BET301)) +
                                     zt * (RS_BET011 + (zs * (RS_BET111 + zs * RS_BET211) + zt * (RS_BET021 + (zs * RS_BET121 + zt * RS_BET031)))));
  const double dRdzs0 = RS_BET000 + (zs * (RS_BET100 + zs * (RS_BET200 + zs * (RS_BET300 + zs * (RS_BET400 + zs * RS_BET500)))) +
                                     zt * (RS_BET010 + (zs * (RS_BET110 + zs * (RS_BET210 + zs * (RS_BET310 + zs * RS_BET410))) +
                                                        zt * (RS_BET020 + (zs * (RS_BET120 + zs * (RS_BET220 + zs * RS_BET320)) +
                                                                           zt * (RS_BET030 + (zt * (RS_BET040 + (zs * RS_BET140 + zt * RS_BET050)) +
                                                                                              zs * (RS_BET130 + zs * RS_BET230))))))));
  *dSV_dS = (dRdzs0 + zp * (dRdzs1 + zp * (dRdzs2 + zp * dRdzs3))) / zs;
}
RQ_FN void roquet_spv_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {   /* :427-454 */
  double dSV_dT, dSV_dS;
  roquet_spv_derivs(T, S, pressure, &dSV_dT, &dSV_dS);
  const double rho = 1.0 / roquet_spv(T, S, pressure, 0.0, 0);
  *drho_dT = -dSV_dT * (rho * rho);
  *drho_dS = -dSV_dS * (rho * rho);
}
#undef RQ_FN
}  // namespace roquet

namespace jackett {
#define JK_FN __device__ __forceinline__
#define JK_MAX(a, b) dmax(a, b)

/* EQN_OF_STATE = "JACKETT_06": the 25-term rational function of Jackett et al. (2006), MOM_EOS_Jackett06.F90.  RNabc / RDabc: the
 * S^a T^b p^c term of the numerator / denominator (6 = power 1.5). */
#define JK_RN000 9.9984085444849347e+02
#define JK_RN001 1.1798263740430364e-06
#define JK_RN002 -2.5862187075154352e-16
#define JK_RN010 7.3471625860981584e+00
#define JK_RN020 -5.3211231792841769e-02
#define JK_RN021 9.8920219266399117e-12
#define JK_RN022 -3.2921414007960662e-20
#define JK_RN030 3.6492439109814549e-04
#define JK_RN100 2.5880571023991390e+00
#define JK_RN101 4.6996642771754730e-10
#define JK_RN110 -6.7168282786692355e-03
#define JK_RN200 1.9203202055760151e-03
#define JK_RD001 6.7103246285651894e-10
#define JK_RD010 7.2815210113327091e-03
#define JK_RD013 -9.1534417604289062e-30
#define JK_RD020 -4.4787265461983921e-05
#define JK_RD030 3.3851002965802430e-07
#define JK_RD032 -2.4461698007024582e-25
#define JK_RD040 1.3651202389758572e-10
#define JK_RD100 1.7632126669040377e-03
#define JK_RD110 -8.8066583251206474e-06
#define JK_RD130 -1.8832689434804897e-10
#define JK_RD600 5.7463776745432097e-06
#define JK_RD620 1.4716275472242334e-09
JK_FN void jackett_num_den(double T, double S, double pressure, double *num_STP, double *den) {   /* :94-102 */
  const double S1_2 = sqrt(JK_MAX(0.0, S)), T2 = T * T;
  *num_STP = (T * (JK_RN010 + T * (JK_RN020 + T * JK_RN030)) + S * (JK_RN100 + (T * JK_RN110 + S * JK_RN200))) +
             pressure * (JK_RN001 + ((T2 * JK_RN021 + S * JK_RN101) + pressure * (JK_RN002 + T2 * JK_RN022)));
  *den = 1.0 + ((T * (JK_RD010 + T * (JK_RD020 + T * (JK_RD030 + T * JK_RD040))) +
                 S * (JK_RD100 + (T * (JK_RD110 + T2 * JK_RD130) + S1_2 * (JK_RD600 + T2 * JK_RD620)))) +
                pressure * (JK_RD001 + pressure * T * (T2 * JK_RD032 + pressure * JK_RD013)));
}
JK_FN double jackett_density(double T, double S, double pressure) {   /* density_elem_Jackett06 :77-106 */
  double num_STP, den;
  jackett_num_den(T, S, pressure, &num_STP, &den);
  const double I_den = 1.0 / den;
  return (JK_RN000 + num_STP) * I_den;
}
JK_FN double jackett_density_anomaly(double T, double S, double pressure, double rho_ref) {   /* :111-144 */
  double num_STP, den;
  jackett_num_den(T, S, pressure, &num_STP, &den);
  const double I_den = 1.0 / den;
  const double rho0 = JK_RN000 - rho_ref * den;
  return (rho0 + num_STP) * I_den;
}
JK_FN void jackett_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {   /* :216-262 */
  const double S1_2 = sqrt(JK_MAX(0.0, S)), T2 = T * T;
  const double num = JK_RN000 + ((T * (JK_RN010 + T * (JK_RN020 + T * JK_RN030)) + S * (JK_RN100 + (T * JK_RN110 + S * JK_RN200))) +
                                 pressure * (JK_RN001 + ((T2 * JK_RN021 + S * JK_RN101) + pressure * (JK_RN002 + T2 * JK_RN022))));
  const double den = 1.0 + ((T * (JK_RD010 + T * (JK_RD020 + T * (JK_RD030 + T * JK_RD040))) +
                             S * (JK_RD100 + (T * (JK_RD110 + T2 * JK_RD130) + S1_2 * (JK_RD600 + T2 * JK_RD620)))) +
                            pressure * (JK_RD001 + pressure * T * (T2 * JK_RD032 + pressure * JK_RD013)));
  const double dnum_dT = ((JK_RN010 + T * (2. * JK_RN020 + T * (3. * JK_RN030))) + S * JK_RN110) +
                         pressure * T * (2. * JK_RN021 + pressure * (2. * JK_RN022));
  const double dnum_dS = (JK_RN100 + (T * JK_RN110 + S * (2. * JK_RN200))) + pressure * JK_RN101;
  const double dden_dT = ((JK_RD010 + T * ((2. * JK_RD020) + T * ((3. * JK_RD030) + T * (4. * JK_RD040)))) +
                          S * ((JK_RD110 + T2 * (3. * JK_RD130)) + S1_2 * T * (2. * JK_RD620))) +
                         (pressure * pressure) * (T2 * 3. * JK_RD032 + pressure * JK_RD013);
  const double dden_dS = JK_RD100 + (T * (JK_RD110 + T2 * JK_RD130) + S1_2 * (1.5 * JK_RD600 + T2 * (1.5 * JK_RD620)));
  const double I_denom2 = 1.0 / (den * den);
  *drho_dT = (dnum_dT * den - num * dden_dT) * I_denom2;
  *drho_dS = (dnum_dS * den - num * dden_dS) * I_denom2;
}
#undef JK_FN
}  // namespace jackett

__device__ __forceinline__ double eos_density(int form, double Rho_T0_S0, double dRho_dT, double dRho_dS, double dRho_dp, double T,
                                              double S, double p) {
  if (form == MOM6X_EOS_LINEAR) return Rho_T0_S0 + dRho_dT * T + dRho_dS * S + dRho_dp * p;
  if (form == MOM6X_EOS_UNESCO) return unesco::density(T, S, p);
  if (form == MOM6X_EOS_ROQUET_RHO) return roquet::roquet_density(T, S, p);
  if (form == MOM6X_EOS_JACKETT06) return jackett::jackett_density(T, S, p);
  if (form == MOM6X_EOS_ROQUET_SPV) return roquet::roquet_spv_density(T, S, p);
  if (form == MOM6X_EOS_WRIGHT_FULL) return wright_density<MOM6X_EOS_WRIGHT_FULL>(T, S, p);
  if (form == MOM6X_EOS_WRIGHT_REDUCED) return wright_density<MOM6X_EOS_WRIGHT_REDUCED>(T, S, p);
  return wright_density<MOM6X_EOS_WRIGHT>(T, S, p);
}
}  // namespace

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

__device__ __forceinline__ double eos_density(int form, double Rho_T0_S0, double dRho_dT, double dRho_dS, double dRho_dp, double T,
                                              double S, double p) {
  if (form == MOM6X_EOS_LINEAR) return Rho_T0_S0 + dRho_dT * T + dRho_dS * S + dRho_dp * p;
  if (form == MOM6X_EOS_UNESCO) return unesco::density(T, S, p);
  if (form == MOM6X_EOS_WRIGHT_FULL) return wright_density<MOM6X_EOS_WRIGHT_FULL>(T, S, p);
  if (form == MOM6X_EOS_WRIGHT_REDUCED) return wright_density<MOM6X_EOS_WRIGHT_REDUCED>(T, S, p);
  return wright_density<MOM6X_EOS_WRIGHT>(T, S, p);
}
}  // namespace

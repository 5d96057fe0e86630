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
__device__ __forceinline__ double eos_density(int form, double Rho_T0_S0, double dRho_dT, double dRho_dS, double dRho_dp, double T,
                                              double S, double p) {
  if (form == MOM6X_EOS_LINEAR) return Rho_T0_S0 + dRho_dT * T + dRho_dS * S + dRho_dp * p;
  if (form == MOM6X_EOS_WRIGHT_FULL) return wright_density<MOM6X_EOS_WRIGHT_FULL>(T, S, p);
  if (form == MOM6X_EOS_WRIGHT_REDUCED) return wright_density<MOM6X_EOS_WRIGHT_REDUCED>(T, S, p);
  return wright_density<MOM6X_EOS_WRIGHT>(T, S, p);
}
}  // namespace

// eos_dev.h -- the equation-of-state pieces more than one translation unit needs (device code).
// EOS_LINEAR (MOM_EOS_linear.F90) and the WRIGHT fit (MOM_EOS_Wright.F90:28-50: Wright 1997, the reduced-range
// coefficients of the default "WRIGHT").
#pragma once
#include "mom6x_dev.h"

namespace {
constexpr double W_a0 = 7.057924e-4, W_a1 = 3.480336e-7, W_a2 = -1.112733e-7;
constexpr double W_b0 = 5.790749e8, W_b1 = 3.516535e6, W_b2 = -4.002714e4, W_b3 = 2.084372e2, W_b4 = 5.944068e5, W_b5 = -9.643486e3;
constexpr double W_c0 = 1.704853e5, W_c1 = 7.904722e2, W_c2 = -7.984422, W_c3 = 5.140652e-2, W_c4 = -2.302158e2, W_c5 = -3.079464;

__device__ __forceinline__ void wright_coefs(double T, double S, double &al0, double &p0, double &lambda) {
  al0 = (W_a0 + W_a1 * T) + W_a2 * S;
  p0 = (W_b0 + W_b4 * S) + T * (W_b1 + T * ((W_b2 + W_b3 * T)) + W_b5 * S);
  lambda = (W_c0 + W_c4 * S) + T * (W_c1 + T * ((W_c2 + W_c3 * T)) + W_c5 * S);
}

// calculate_density(T, S, pressure, rho, EOS): density_elem of MOM_EOS_linear.F90:73-84 / MOM_EOS_Wright.F90:108-122
__device__ __forceinline__ double eos_density(int form, double Rho_T0_S0, double dRho_dT, double dRho_dS, double dRho_dp, double T,
                                              double S, double p) {
  if (form == MOM6X_EOS_LINEAR) return Rho_T0_S0 + dRho_dT * T + dRho_dS * S + dRho_dp * p;
  double al0, p0, lambda;
  wright_coefs(T, S, al0, p0, lambda);
  return (p + p0) / (lambda + al0 * (p + p0));
}
}  // namespace

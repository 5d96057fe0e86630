#!/bin/bash
# timing-only A/B of compile-time variants of continuity_wave.hip (no parity run): bash scripts/r04_ab_flags.sh "<flags A>" "<flags B>" ...
for fl in "$@"; do
  echo "=== variant [$fl]"
  touch mom6_amd/csrc/continuity_wave.hip
  MOM6X_CFLAGS="$fl" python -m mom6_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  PROF_MODES=plain,adjust,bt_cont timeout 100 python scripts/prof_continuity.py 2>&1 | grep '^lds' | sed 's/k_convergence<DIR>=[0-9.]* //; s/lds //; s/k_mass_flux_wave//g' | tr '\n' '|'; echo
done
touch mom6_amd/csrc/continuity_wave.hip

touch mom6_amd/csrc/continuity_wave.hip mom6_amd/csrc/continuity_lds.hip
MOM6X_CFLAGS=-DMOM6X_MFL_TIMING python -m mom6_amd.build > /dev/null 2>&1 || echo BUILD FAILED
PROF_MODES=adjust,bt_cont timeout 200 python scripts/prof_continuity.py 2>&1 | grep "phases\|^lds\|flux re"
touch mom6_amd/csrc/continuity_wave.hip mom6_amd/csrc/continuity_lds.hip

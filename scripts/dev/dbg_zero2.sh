touch mom6_amd/csrc/continuity_wave.hip
MOM6X_CFLAGS=-DMOM6X_DBG_ZERO python -m mom6_amd.build 2>&1 | grep -i "error\|DBG" | head -5
strings mom6_amd/lib/libmom6x.so | grep -c DBGZ
python scripts/dev/dbg_zero.py > /tmp/dz.log 2>&1
grep -c DBGZ /tmp/dz.log; grep "DBGZ" /tmp/dz.log | sort | uniq -c | sort -rn | head -12
grep -v "DBGZ\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" /tmp/dz.log | tail -12
touch mom6_amd/csrc/continuity_wave.hip

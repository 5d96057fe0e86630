!/// tests/fortran_stubs/MOM_memory.h -- the array-extent macros the shim modules use, for a build with dynamic,
!/// symmetric memory (what config_src/memory/dynamic_symmetric selects in a MOM6 tree).  Written for the stub build of
!/// tests/test_fortran_shims_cpu.py; not reference text.
#define SZI_(G)   G%isd:G%ied
#define SZJ_(G)   G%jsd:G%jed
#define SZIB_(G)  G%IsdB:G%IedB
#define SZJB_(G)  G%JsdB:G%JedB
#define SZK_(G)   G%ke

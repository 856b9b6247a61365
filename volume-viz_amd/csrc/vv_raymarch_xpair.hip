// vv_raymarch_xpair.hip -- the ray-march kernels instantiated on the x-pair copy (the z-pair copy with x and z
// in each other's roles: records {v(x,y,z), v(x+1,y,z)}, z fastest), used for side views of the volumes whose
// front views take the z-pair copy: two gathers per sample instead of four.
#define VV_ZPAIR 1
#define VV_XPAIR 1
#include "vv_raymarch.hip"

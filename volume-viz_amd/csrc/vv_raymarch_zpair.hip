// vv_raymarch_zpair.hip -- the ray-march kernels instantiated on the z-pair copy of an f32 volume
// (VolumeView::zpair: records {v(x,y,z), v(x,y,z+1)}), used for views along the memory axis:
// two 16-byte gathers per sample instead of four 8-byte gathers.
#define VV_ZPAIR 1
#include "vv_raymarch.hip"

// vv_raymarch_brick.hip -- the ray-march kernels instantiated on the bricked copy of the volume
// (4x4x4-voxel bricks with an x halo, VolumeView::bricks), used for views that are not aligned
// with the memory axis, where the linear layout costs one cache line per lane and gather.
#define VV_BRICKED 1
#include "vv_raymarch.hip"

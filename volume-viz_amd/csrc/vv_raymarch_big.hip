// vv_raymarch_big.hip -- the ray-march kernels instantiated for volumes above 4 GiB
// (64-bit slice base per sample; see tex3d_raw in vv_device.h).
#define VV_BIG_VOLUME 1
#include "vv_raymarch.hip"

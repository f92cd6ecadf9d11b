#define MRA_TU_DIM 1
#define MRA_TU_NAME launch_cascade_d1
#include "mra_launch_prior.inc"

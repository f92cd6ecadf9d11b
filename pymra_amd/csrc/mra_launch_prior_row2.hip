#define MRA_TU_DIM 2
#define MRA_TU_NAME launch_cascade_d2
#include "mra_launch_prior.inc"

#define MRA_TU_DIM 2
#define MRA_TU_KNOT
#define MRA_TU_NAME launch_knot_chain_d2
#include "mra_launch_prior.inc"

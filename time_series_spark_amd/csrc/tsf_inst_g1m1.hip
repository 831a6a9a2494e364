// kernel instantiations: growth=1 (0 linear, 1 logistic), column mode=1 (0 additive, 1 multiplicative, 2 mixed)
#define TSF_G 1
#define TSF_M 1
#define TSF_LAUNCH_NAME launch_g1m1
#define TSF_NEWTON_LAUNCH_NAME launch_newton_g1m1
#define TSF_MAP_LAUNCH_NAME launch_map_g1m1
#include "tsf_inst.inc"

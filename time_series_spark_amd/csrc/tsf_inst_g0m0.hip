// kernel instantiations: growth=0 (0 linear, 1 logistic), column mode=0 (0 additive, 1 multiplicative, 2 mixed)
#define TSF_G 0
#define TSF_M 0
#define TSF_LAUNCH_NAME launch_g0m0
#define TSF_NEWTON_LAUNCH_NAME launch_newton_g0m0
#define TSF_MAP_LAUNCH_NAME launch_map_g0m0
#include "tsf_inst.inc"

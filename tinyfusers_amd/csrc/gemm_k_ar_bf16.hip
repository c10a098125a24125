// bfloat16 instances of gemm_k_ar.hip, in a translation unit of their own (the fp16 unit compiles exactly as without them): TF_TU_BF, common.h
#define TF_TU_BF 1
#include "gemm_k_ar.hip"

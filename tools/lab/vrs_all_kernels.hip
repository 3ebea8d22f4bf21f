// the library's kernel translation units as ONE unit, for lab harnesses that launch kernels directly
#include "vrs_contract.hip"
#include "vrs_one_call.hip"
#include "vrs_msd_hybrid.hip"
